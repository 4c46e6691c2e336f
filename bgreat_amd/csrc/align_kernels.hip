// align_kernels.hip -- gfx950 kernels of the BGREAT read-mapping path.  One 64-lane wavefront maps one read.
//
//   stage A  read ASCII -> 2-bit packed words in this wave's LDS slice (coalesced byte loads, one
//            ds_write_b8 per 4 bases; byte address ^7 makes the words first-base-most-significant so a
//            window read out of LDS is the reference's k-mer integer, utils.cpp:117-129).
//   stage B  anchors (aligner.cpp:345-378 getNOverlap): lane i owns read position base+i, builds the
//            forward and reverse-complement (k-1)-mers from LDS, takes the smaller, and walks the MPHF
//            cascade (one dwordx4 per level) + one u64 key compare.  __ballot orders the hits by position,
//            exactly the sequential scan's order; anchors are tried as soon as they are found (the
//            reference collects `tryNumber` first, but collecting has no side effect).
//   stage C  greedy extension (alignerGreedy.cpp:167-364): wave-uniform walk; at each step the <=4
//            neighbour unitigs are scored in parallel, 16 lanes per candidate, each lane XOR-ing 32-base
//            chunks of the packed unitig (HBM/L2) against the packed read (LDS) and popcounting.
//            argmin with lowest-slot tie-break == the reference's "first zero wins, else strict min".
//   stage D  the path (LDS) is appended to a global arena with one atomicAdd per read.
//
// Integer/byte work only: no MFMA anywhere (there is no dense contraction on this path).
#include "align_kernels.h"

#include "../../include/bgreat_gpu.h"

namespace bgr {

namespace {

typedef uint64_t u64;

#define EVEN_BITS 0x5555555555555555ULL

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t rl32(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
__device__ __forceinline__ u64 rl64(u64 v, int lane) {
    return ((u64)rl32((uint32_t)(v >> 32), lane) << 32) | rl32((uint32_t)v, lane);
}
// 32 bases starting at base p of a first-base-most-significant packed array (needs A[p/32 + 1] readable)
template <typename P>
__device__ __forceinline__ u64 win32(P A, u64 p) {
    u64 w = p >> 5;
    uint32_t s = (uint32_t)(p & 31) * 2;
    u64 hi = A[w], lo = A[w + 1];
    return s ? (hi << s) | (lo >> (64 - s)) : hi;
}
// 32 BITS starting at bit q of a 1-bit-per-base plane, most significant first
__device__ __forceinline__ uint32_t plane32(const u64* P, u64 q) {
    u64 w = q >> 6;
    uint32_t s = (uint32_t)(q & 63);
    u64 hi = P[w], lo = P[w + 1];
    u64 x = s ? (hi << s) | (lo >> (64 - s)) : hi;
    return (uint32_t)(x >> 32);
}
// keep the even-position bits of x (bit 62-2j -> bit 31-j)
__device__ __forceinline__ uint32_t compress_even(u64 x) {
    x &= EVEN_BITS;
    x = (x | (x >> 1)) & 0x3333333333333333ULL;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0FULL;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFULL;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFULL;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFULL;
    return (uint32_t)x;
}

// MPHF cascade walk for one key per lane (graph_layout.h: 2-bit position states).  Returns the minimal index
// or BGR_NONE; the caller compares keys[idx].  Straight-line body, wave-uniform trip count: the loop runs
// until no lane is still on a "several keys here" position -- about 2-3 levels at gamma 2, because a lane
// that lands on an empty position (most read positions are not overlaps) is rejected at once.
// LV = level descriptors {units, base} staged in LDS.
template <typename UP>
__device__ __forceinline__ uint32_t mphf_lookup(const BgrDeviceGraph& g, const uint2* LV, UP units, u64 key, bool active) {
    u64 m = bgr_mix64(key);
    uint32_t hl = (uint32_t)m;
    const uint32_t hb = (uint32_t)(m >> 32) | 1u;
    uint32_t res = BGR_NONE;
    const uint32_t nl = g.n_levels;
    for (uint32_t l = 0; l < nl; ++l) {
        if (!__any(active)) break;
        const uint2 lv = LV[l];
        const uint32_t u = lv.y + __umulhi(hl, lv.x);
        const uint32_t p = bgr_level_pos(hl);
        const uint4 q = reinterpret_cast<const uint4*>(units)[u];
        const uint32_t wi = p >> 4, sh = (p & 15) * 2;
        const uint32_t w = wi == 0 ? q.x : (wi == 1 ? q.y : q.z);
        const uint32_t st = (w >> sh) & 3u;
        uint32_t r = q.w + __popc(bgr_unique_mask(w) & ((1u << sh) - 1u));
        r += wi >= 1 ? __popc(bgr_unique_mask(q.x)) : 0;
        r += wi >= 2 ? __popc(bgr_unique_mask(q.y)) : 0;
        if (active && st == 1u) res = r;
        active = active && st == 3u;
        hl += hb;
    }
    if (g.n_fallback && __any(active)) {
        if (active) {  // bisection in the (tiny) sorted fallback list
            uint32_t lo = 0, hi = g.n_fallback;
            while (lo < hi) {
                uint32_t mid = (lo + hi) >> 1;
                if (g.fallback[mid] < key) lo = mid + 1; else hi = mid;
            }
            if (lo < g.n_fallback && g.fallback[lo] == key) res = g.n_placed + lo;
        }
    }
    return res;
}

struct Step {  // wave-uniform result of one extension step
    bool found, fits;
    int32_t sid;
    uint32_t miss, ext, next_rec;
    bool next_canon;
};

// One extension step of the greedy walks.  DIR 0: left (checkBeginGreedy / mapOnLeftEndGreedy, alignerGreedy.cpp
// :268-319 / :167-218), 1: first right step (checkEndGreedy :322-364), 2: later right steps (mapOnRightEndGreedy
// :221-265, whose read slice INCLUDES the k-1 overlap).  Candidates = the <=4 slots of the neighbour record
// (getEnd / getBegin, aligner.cpp:147-267), scored 16 lanes each.
template <int DIR>
__device__ __forceinline__ Step greedy_step(const BgrDeviceGraph& g, const u64* CMP, const u64* NM, bool useN, uint32_t L,
                                            uint32_t K1, uint32_t rec, bool canon, uint32_t pos, uint32_t budget, int lane) {
    Step out;
    out.found = false; out.fits = false; out.sid = 0; out.miss = 0; out.ext = 0; out.next_rec = BGR_NONE; out.next_canon = false;
    if (rec == BGR_NONE) return out;  // key not in the table: getBegin/getEnd return an empty list
    const int c = lane >> 4, sub = lane & 15;
    // getEnd(bin): bin<=rc ? rightIndices : leftIndices ; getBegin(bin): bin<=rc ? leftIndices : rightIndices
    const bool useR = (DIR == 0) ? canon : !canon;
    const uint32_t fbit = canon ? BGR_SLOT_F0 : BGR_SLOT_F1;
    uint32_t slot = g.recs[(u64)rec * 8 + (useR ? 4 : 0) + c];
    uint32_t id = slot & BGR_SLOT_ID_MASK;
    // the reference's nested ifs stop at the first empty slot
    u64 zmask = __ballot(id == 0);
    int first_zero = zmask ? (__ffsll((long long)zmask) - 1) >> 4 : 4;
    bool valid = c < first_zero;
    bool fwd = (slot & fbit) != 0;
    u64 S = 0;
    uint32_t len = 0, mflags = 0, rec_beg = 0, rec_end = 0;
    if (valid) {
        const uint4* mp = reinterpret_cast<const uint4*>(g.meta + id);
        uint4 m0 = mp[0];
        uint2 m1 = reinterpret_cast<const uint2*>(mp + 1)[0];
        S = ((u64)m0.y << 32) | m0.x;
        len = m0.z; mflags = m0.w; rec_beg = m1.x; rec_end = m1.y;
        if (!fwd) S += len;
    }
    uint32_t ext = len - K1;
    bool fits;
    uint32_t n, ustart, rstart;
    if (DIR == 0) {
        fits = ext >= pos;
        n = fits ? pos : ext;
        ustart = fits ? ext - pos : 0;
        rstart = fits ? 0 : pos - ext;
    } else if (DIR == 1) {
        uint32_t rl = L - pos - K1;
        fits = ext >= rl;
        n = fits ? rl : ext;
        ustart = K1;
        rstart = pos + K1;
    } else {
        uint32_t rl = L - pos;
        fits = ext >= rl;
        n = fits ? rl : (len < rl ? len : rl);  // read.substr(pos, |u|) is clipped at |read|
        ustart = 0;
        rstart = pos;
    }
    uint32_t cnt = 0;
    if (valid) {
        for (uint32_t t = sub; t * 32 < n; t += 16) {
            u64 ub = S + ustart + (u64)t * 32;
            u64 x = win32(g.seq, ub) ^ win32(CMP, (u64)rstart + t * 32);
            u64 mm = (x | (x >> 1)) & EVEN_BITS;
            u64 nm = 0;
            if (useN) { nm = win32(NM, (u64)rstart + t * 32) & EVEN_BITS; mm |= nm; }
            if (g.has_exc) {  // forward-strand unitig bases outside ACGT: never equal, except N == N
                uint32_t e = plane32(g.exc, ub);
                if (e) {
                    uint32_t en = plane32(g.excn, ub);
                    uint32_t m1 = compress_even(mm) | e;
                    m1 &= ~(en & compress_even(nm));
                    uint32_t v1 = n - t * 32;
                    if (v1 < 32) m1 &= ~(0xFFFFFFFFu >> v1);
                    cnt += __popc(m1);
                    continue;
                }
            }
            uint32_t v = n - t * 32;
            if (v < 32) mm &= ~(~0ULL >> (2 * v));
            cnt += __popcll(mm);
        }
    }
    cnt += __shfl_xor(cnt, 8);
    cnt += __shfl_xor(cnt, 4);
    cnt += __shfl_xor(cnt, 2);
    cnt += __shfl_xor(cnt, 1);
    // best = smallest miss, lowest slot on ties, only if miss <= budget
    uint32_t best = budget + 1;
    int bc = -1;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        uint32_t tcc = rl32(cnt, cc * 16);
        if (cc < first_zero && tcc < best) { best = tcc; bc = cc; }
    }
    if (bc < 0) return out;
    const int bl = bc * 16;
    uint32_t bid = rl32(id, bl);
    bool bfwd = rl32(fwd ? 1u : 0u, bl) != 0;
    uint32_t bflags = rl32(mflags, bl);
    out.found = true;
    out.fits = rl32(fits ? 1u : 0u, bl) != 0;
    out.sid = bfwd ? (int32_t)bid : -(int32_t)bid;
    out.miss = best;
    out.ext = rl32(ext, bl);
    if (DIR == 0) {
        out.next_rec = bfwd ? rl32(rec_beg, bl) : rl32(rec_end, bl);
        out.next_canon = (bflags & (bfwd ? BGR_META_CANON_BEG : BGR_META_CANON_RCEND)) != 0;
    } else {
        out.next_rec = bfwd ? rl32(rec_end, bl) : rl32(rec_beg, bl);
        out.next_canon = (bflags & (bfwd ? BGR_META_CANON_END : BGR_META_CANON_RCBEG)) != 0;
    }
    return out;
}

// Greedy extension from one anchor (alignReadGreedy's loop body, alignerGreedy.cpp:41-52).  On success the
// path sits in PATH[*p_lo .. *p_lo + *p_n).
__device__ __forceinline__ bool greedy_from_anchor(const BgrDeviceGraph& g, const u64* CMP, const u64* NM, bool useN, uint32_t L,
                                                   uint32_t K1, uint32_t a_rec, bool a_canon, uint32_t a_pos, uint32_t m,
                                                   int32_t* PATH, uint32_t* p_lo, uint32_t* p_n, int lane) {
    const uint32_t mid = a_pos + 2;  // left pushes grow downwards from mid-1 (at most a_pos+1 of them), right upwards from mid
    uint32_t nl = 0, nr = 0, budget = m;
    uint32_t pos = a_pos, rec = a_rec;
    bool canon = a_canon;
    for (;;) {  // left walk
        if (pos == 0) { if (lane == 0) PATH[mid - 1 - nl] = 0; ++nl; break; }
        Step s = greedy_step<0>(g, CMP, NM, useN, L, K1, rec, canon, pos, budget, lane);
        if (!s.found) return false;
        if (lane == 0) PATH[mid - 1 - nl] = s.sid;
        ++nl;
        budget -= s.miss;
        if (s.fits) { if (lane == 0) PATH[mid - 1 - nl] = (int32_t)(s.ext - pos); ++nl; break; }
        pos -= s.ext; rec = s.next_rec; canon = s.next_canon;
    }
    pos = a_pos; rec = a_rec; canon = a_canon;
    bool first = true;
    for (;;) {  // right walk
        if (first) { if (L - pos - K1 == 0) break; } else { if (L - pos < K1 + 1) break; }
        Step s = first ? greedy_step<1>(g, CMP, NM, useN, L, K1, rec, canon, pos, budget, lane)
                       : greedy_step<2>(g, CMP, NM, useN, L, K1, rec, canon, pos, budget, lane);
        if (!s.found) return false;
        if (lane == 0) PATH[mid + nr] = s.sid;
        ++nr;
        budget -= s.miss;
        if (s.fits) break;
        pos += s.ext; rec = s.next_rec; canon = s.next_canon;
        first = false;
    }
    *p_lo = mid - nl;
    *p_n = nl + nr;
    return true;
}

template <bool STAGE>
__global__ void __launch_bounds__(1024) bgr_align_greedy_kernel(BgrDeviceGraph g, BatchIO io, KernelParams prm) {
    extern __shared__ u64 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const uint32_t W = io.words_per_read;
    const uint32_t K1 = g.k - 1;
    // LDS: [level descriptors 512 B][optional MPHF copy][per-wave: FW3 | FWQ | RCW | NM | PATH]
    uint2* LV = reinterpret_cast<uint2*>(lds);
    if (threadIdx.x < BGR_MAX_LEVELS) LV[threadIdx.x] = make_uint2(g.levels[threadIdx.x].units, g.levels[threadIdx.x].base);
    const uint32_t mphf_words = STAGE ? (g.units_bytes_lo + 7) / 8 : 0;
    const uint32_t* units = g.units;
    if (STAGE) {
        const uint4* src = reinterpret_cast<const uint4*>(g.units);
        uint4* dst = reinterpret_cast<uint4*>(lds + 64);
        for (uint32_t i = threadIdx.x; i < g.units_bytes_lo / 16; i += blockDim.x) dst[i] = src[i];
        units = reinterpret_cast<const uint32_t*>(lds + 64);
    }
    __syncthreads();
    const uint32_t per_wave_words = 4 * W + io.path_cap / 2;
    u64* FW3 = lds + 64 + mphf_words + (u64)wave * per_wave_words;
    u64* FWQ = FW3 + W;
    u64* RCW = FWQ + W;
    u64* NM = RCW + W;
    int32_t* PATH = reinterpret_cast<int32_t*>(NM + W);
    unsigned char* FW3b = reinterpret_cast<unsigned char*>(FW3);
    unsigned char* NMb = reinterpret_cast<unsigned char*>(NM);

    uint32_t c_reads = 0, c_noov = 0, c_al = 0, c_na = 0;
    uint32_t chunk_pos = 0, chunk_end = 0;  // this wave's slice of the path arena

    for (uint32_t r = blockIdx.x * waves + wave; r < io.n_reads; r += gridDim.x * waves) {
        const u64 off = io.read_offs[r];
        const uint32_t L = (uint32_t)(io.read_offs[r + 1] - off);
        // ---- stage A: ASCII -> packed (4 bases per lane per round) ------------------------------
        bool sawN = false;
        for (uint32_t bi = lane; bi < 8 * W; bi += 64) {
            uint32_t b0 = bi * 4, code = 0, nmask = 0;
            if (b0 < L) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t cc = 0, nn = 0;
                    if (b0 + j < L) {
                        unsigned char ch = io.reads[off + b0 + j];
                        cc = ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : 3u;  // str2num, utils.cpp:117-129
                        nn = (cc == 3u && ch != 'T') ? 3u : 0u;                      // 'N' (parser admits only ACGTN)
                    }
                    code = (code << 2) | cc;
                    nmask = (nmask << 2) | nn;
                }
            }
            FW3b[bi ^ 7] = (unsigned char)code;
            NMb[bi ^ 7] = (unsigned char)nmask;
            sawN |= nmask != 0;
        }
        const bool hasN = __any(sawN);
        wave_sync();
        // ---- reverse-complement stream + rolling-update quirk stream ----------------------------
        for (uint32_t w = lane; w < W; w += 64) {
            // RCW word w = bases 32w..32w+31 of reverseComplements(read) (utils.cpp:66-73: non-ACG -> 'A' == 3 - 3)
            long long p = (long long)L - 32 * ((long long)w + 1);
            u64 rcw = 0;
            if (p >= 0) {
                rcw = ~bgr_rev2(win32(FW3, (u64)p));
            } else if (p > -32) {
                uint32_t v = (uint32_t)(32 + p);  // valid bases
                u64 x = FW3[0] >> (64 - 2 * v);
                rcw = (~bgr_rev2(x)) & (~0ULL << (64 - 2 * v));
            }
            RCW[w] = rcw;
            // FWQ: what getNOverlap's rolling `num` holds: str2num codes inside the first window (N->3),
            // nuc2int codes (N->0) for every base entered by update() (aligner.cpp:305-309, utils.cpp:132-140)
            u64 ge;
            if (32 * w >= K1) ge = ~0ULL;
            else if (32 * (w + 1) <= K1) ge = 0;
            else ge = ~0ULL >> (2 * (K1 - 32 * w));
            FWQ[w] = FW3[w] & ~(NM[w] & ge);
        }
        wave_sync();

        // ---- passes: forward read, then its reverse complement (alignerGreedy.cpp:54) -----------
        uint32_t status = BGR_ST_NOANCHOR, p_lo = 0, p_n = 0;
        const uint32_t npos = L >= K1 ? L - K1 + 1 : 0;
        for (int pass = 0; pass < 2; ++pass) {
            const u64* A = pass ? RCW : FWQ;   // forward-strand k-mers of this pass
            const u64* B = pass ? FW3 : RCW;   // reverse-strand k-mers (rolling nuc2intrc: N -> 0)
            const u64* CMP = pass ? RCW : FW3; // characters compared by missmatchNumber
            const bool useN = (pass == 0) && hasN;
            uint32_t tried = 0;
            bool done = false;
            for (uint32_t base = 0; base < npos && !done && tried < prm.effort; base += 64) {
                const uint32_t i = base + lane;
                const bool valid = i < npos;
                u64 num = 0, rcn = 0;
                if (valid) {
                    num = win32(A, i) >> (64 - 2 * K1);
                    rcn = win32(B, L - K1 - i) >> (64 - 2 * K1);
                }
                const u64 rep = num < rcn ? num : rcn;
                uint32_t idx = mphf_lookup(g, LV, units, rep, valid);
                bool hit = false;
                if (idx != BGR_NONE) hit = g.keys[idx] == rep;   // aligner.cpp:353,361 key check
                u64 mask = __ballot(hit);
                while (mask && tried < prm.effort) {
                    const int src = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    ++tried;
                    const u64 a_num = rl64(num, src), a_rcn = rl64(rcn, src);
                    uint32_t a_rec = rl32(idx, src);
                    // getBegin/getEnd recompute rc = rcb(num) (aligner.cpp:149,211); it differs from the
                    // rolling rcnum only when an N was rolled into the window.
                    const u64 rc2 = bgr_rcb(a_num, K1);
                    if (rc2 != a_rcn) {
                        const u64 key2 = a_num < rc2 ? a_num : rc2;
                        uint32_t i2 = mphf_lookup(g, LV, units, key2, true);
                        bool ok2 = false;
                        if (i2 != BGR_NONE) ok2 = g.keys[i2] == key2;
                        a_rec = ok2 ? i2 : BGR_NONE;
                    }
                    if (greedy_from_anchor(g, CMP, NM, useN, L, K1, a_rec, a_num <= rc2, base + src, prm.max_mismatch, PATH, &p_lo, &p_n, lane)) {
                        done = true;
                        break;
                    }
                }
            }
            if (done) { status = BGR_ST_ALIGNED | (pass ? BGR_ST_RC : 0); break; }
            if (tried == 0) { status = BGR_ST_NOANCHOR | (pass ? BGR_ST_RC : 0); break; }  // ++noOverlapRead, no retry
            status = BGR_ST_FAILED | BGR_ST_RC;  // all anchors failed: retry on the reverse complement once
        }
        // ---- stage D: publish ---------------------------------------------------------------------
        wave_sync();
        uint32_t abase = 0;
        if ((status & BGR_ST_MASK) == BGR_ST_ALIGNED) {
            // Arena space comes in per-wave chunks: ONE global atomic per ~50 reads instead of one per read
            // (a single-address atomic saturates near 90 M/s chip-wide, MI355X_MICROARCH.md "dequeue").
            if (p_n > chunk_end - chunk_pos) {
                uint32_t want = p_n > io.arena_chunk ? p_n : io.arena_chunk;
                uint32_t got = 0;
                if (lane == 0) got = atomicAdd(io.cursor, want);
                chunk_pos = rl32(got, 0);
                chunk_end = chunk_pos + want;
            }
            abase = chunk_pos;
            chunk_pos += p_n;
            if (abase + p_n <= io.arena_cap) {
                for (uint32_t j = lane; j < p_n; j += 64) io.arena[abase + j] = PATH[p_lo + j];
            } else if (lane == 0) {
                io.cursor[1] = 1;  // overflow: reported by the host as an error
            }
        } else {
            p_n = 0;
        }
        if (lane == 0) {
            io.status[r] = (uint8_t)status;
            io.path_off[r] = abase;
            io.path_len[r] = p_n;
        }
        ++c_reads;
        c_noov += (status & BGR_ST_MASK) == BGR_ST_NOANCHOR;
        c_al += (status & BGR_ST_MASK) == BGR_ST_ALIGNED;
        c_na += (status & BGR_ST_MASK) == BGR_ST_FAILED;
        wave_sync();
    }
    if (lane == 0 && c_reads) {
        atomicAdd(&io.counters[0], (unsigned long long)c_reads);
        if (c_noov) atomicAdd(&io.counters[1], (unsigned long long)c_noov);
        if (c_al) atomicAdd(&io.counters[2], (unsigned long long)c_al);
        if (c_na) atomicAdd(&io.counters[3], (unsigned long long)c_na);
    }
}

}  // namespace

hipError_t launch_align(const BgrDeviceGraph& g, const BatchIO& io, const KernelParams& p, const LaunchCfg& cfg, hipStream_t stream) {
    if (io.n_reads == 0) return hipSuccess;
    dim3 grid(cfg.blocks), block(cfg.waves_per_block * 64);
    if (p.mode != 0) return hipErrorNotSupported;
    if (cfg.stage_mphf) {
        if (cfg.lds_bytes > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bgr_align_greedy_kernel<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)cfg.lds_bytes);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(bgr_align_greedy_kernel<true>, grid, block, cfg.lds_bytes, stream, g, io, p);
    } else {
        if (cfg.lds_bytes > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bgr_align_greedy_kernel<false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)cfg.lds_bytes);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(bgr_align_greedy_kernel<false>, grid, block, cfg.lds_bytes, stream, g, io, p);
    }
    return hipGetLastError();
}

}  // namespace bgr
