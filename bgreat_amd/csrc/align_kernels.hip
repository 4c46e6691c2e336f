// align_kernels.hip -- gfx950 kernels of the BGREAT read-mapping path.  One 64-lane wavefront maps one read.
//
//   stage A  read ASCII -> 2-bit packed words in this wave's LDS slice (coalesced byte loads, one
//            ds_write_b8 per 4 bases; byte address ^7 makes the words first-base-most-significant so a
//            window read out of LDS is the reference's k-mer integer, utils.cpp:117-129).
//   stage B  anchors (aligner.cpp:345-378 getNOverlap): lane i owns read position base+i, builds the
//            forward and reverse-complement (k-1)-mers from LDS, takes the smaller, and walks the MPHF
//            cascade (one dwordx4 per level; from LDS when the cascade is staged there, else from L2/HBM with
//            the next level's unit requested ahead) + one u64 key compare.  __ballot orders the hits by
//            position, exactly the sequential scan's order; anchors are tried as soon as they are found (the
//            reference collects `tryNumber` first, but collecting has no side effect).
//   stage C  extension.  Greedy (alignerGreedy.cpp:167-364): wave-uniform walk; at each step the <=4
//            neighbour unitigs are scored in parallel, 16 lanes per candidate, each lane XOR-ing 32-base
//            chunks of the packed unitig (HBM/L2) against the packed read (LDS) and popcounting; argmin with
//            lowest-slot tie-break == the reference's "first zero wins, else strict min".
//            Exhaustive (alignerExhaustive.cpp:61-259): the same scoring inside a depth-first search with
//            an explicit frame stack in LDS (see exh_search); a second pass keeps that state in HBM for the
//            rare deep search and for reads too long for LDS.
//            Anchors mode (-G, alignerGreedy.cpp:60-164): k-mer anchors through BooPHF's exact structure, the
//            anchoring unitig placed on the read, then the greedy walks from its two ends.
//   stage D  the path (LDS) is appended to a global arena, space reserved per wave in chunks; launch_csr turns
//            (results, arena) into input-ordered CSR arrays on the device.
//
// Integer/byte work only: no MFMA anywhere (there is no dense contraction on this path).
#include "align_kernels.h"

#include <algorithm>

#include "../../include/bgreat_gpu.h"

namespace bgr {

namespace {

typedef uint64_t u64;
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;

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
// Branch-free funnel shift: both words are always loaded (side by side: one 16-byte access), and the low word's
// contribution vanishes by itself when the window is word aligned.
template <typename P>
__device__ __forceinline__ u64 win32(P A, u64 p) {
    const u64 w = p >> 5;
    const uint32_t s = (uint32_t)(p & 31) * 2;
    const u64 hi = A[w], lo = A[w + 1];
    return (hi << s) | ((lo >> 1) >> (63 - s));
}
// the same for the per-wave LDS streams, whose base positions fit 32 bits
__device__ __forceinline__ u64 lds_win32(const u64* A, uint32_t p) {
    const uint32_t s = (p & 31) * 2;
    const u64 hi = A[p >> 5], lo = A[(p >> 5) + 1];
    return (hi << s) | ((lo >> 1) >> (63 - s));
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
// Sum over each aligned group of 16 lanes (a DPP "row"), result in every lane of the group: four v_add_u32_dpp,
// no LDS round trip (ds_bpermute shuffles cost ~7 instructions + an LDS wait each).
__device__ __forceinline__ uint32_t row16_sum(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]  (lane ^ 1)
    x += (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]  (lane ^ 2)
    x += (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x141, 0xF, 0xF, true);  // row_half_mirror: the other quad of the half row
    x += (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x140, 0xF, 0xF, true);  // row_mirror: the other half of the row
    return x;
}

// reverse the order of the 32 two-bit digits of x: full bit reversal (v_bfrev_b32 x2), then swap the bits of each pair
__device__ __forceinline__ u64 rev2_fast(u64 x) {
    u64 y = __builtin_bitreverse64(x);
    return ((y >> 1) & EVEN_BITS) | ((y & EVEN_BITS) << 1);
}
__device__ __forceinline__ u64 rcb_fast(u64 x, uint32_t n) { return (~rev2_fast(x)) >> (64 - 2 * n); }

// MPHF cascade walk for one key per lane (graph_layout.h: 2-bit position states).  Returns the minimal index
// or BGR_NONE; the caller compares keys[idx].  Straight-line body, wave-uniform trip count: the loop runs
// until no lane is still on a "several keys here" position -- about 2-3 levels at gamma 2, because a lane
// that lands on an empty position (most read positions are not overlaps) is rejected at once.
// LV = level descriptors {units, base} staged in LDS.
// SPEC (cascade in L2/HBM, not staged in LDS): the state word of level l+1 is requested together with level l's -- its
// address needs only the hash (double hashing), not level l's answer -- so two dependent-looking loads are in flight
// at once; lanes that stop on level l simply drop it.
template <bool SPEC, typename UP>
__device__ __forceinline__ uint32_t mphf_lookup(const BgrDeviceGraph& g, const uint2* LV, UP units, u64 key, bool active) {
    u64 m = bgr_mix64(key);
    uint32_t hl = (uint32_t)m;
    const uint32_t hb = (uint32_t)(m >> 32) | 1u;
    const uint32_t nl = g.n_levels;
    uint32_t res = BGR_NONE;
    if (SPEC) {
        // cascade in L2/HBM: one 16-byte unit per level answers membership candidate AND minimal index, the next level's unit
        // requested ahead (a second dependent access for the rank would cost a full L2 round trip: measured 7.98 -> 9.63 ms)
        uint4 qn = make_uint4(0, 0, 0, 0);
        if (active && nl) { const uint2 lv = LV[0]; qn = reinterpret_cast<const uint4*>(units)[lv.y + __umulhi(hl, lv.x)]; }
        for (uint32_t l = 0; l < nl; ++l) {
            if (!__any(active)) break;
            const uint4 q = qn;
            if (active && l + 1 < nl) { const uint2 lv1 = LV[l + 1]; qn = reinterpret_cast<const uint4*>(units)[lv1.y + __umulhi(hl + hb, lv1.x)]; }
            const uint32_t p = bgr_level_pos(hl);
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
    } else {
        // cascade in LDS: per level only the state word that holds the position is read (a dword: a quarter of the unit); the
        // position where the walk ends on state 1 is remembered (unit << 6 | position) and its rank worked out once, behind
        // the loop (4.49 -> 4.09 ms per 5 M reads: the loop body halves)
        uint32_t hit = BGR_NONE;
        for (uint32_t l = 0; l < nl; ++l) {
            if (!__any(active)) break;
            const uint2 lv = LV[l];
            const uint32_t u = lv.y + __umulhi(hl, lv.x);
            const uint32_t p = bgr_level_pos(hl);
            const uint32_t w = units[(size_t)u * 4 + (p >> 4)];
            const uint32_t st = (w >> ((p & 15) * 2)) & 3u;
            if (active && st == 1u) hit = u << 6 | p;
            active = active && st == 3u;
            hl += hb;
        }
        if (hit != BGR_NONE) {  // minimal index = rank of the position among the placed ones: the unit's running rank + the placed states before it
            const uint4 q = reinterpret_cast<const uint4*>(units)[hit >> 6];
            const uint32_t p = hit & 63u, wi = p >> 4, sh = (p & 15) * 2;
            const uint32_t w = wi == 0 ? q.x : (wi == 1 ? q.y : q.z);
            uint32_t r = q.w + __popc(bgr_unique_mask(w) & ((1u << sh) - 1u));
            r += wi >= 1 ? __popc(bgr_unique_mask(q.x)) : 0;
            r += wi >= 2 ? __popc(bgr_unique_mask(q.y)) : 0;
            res = r;
        }
    }
    if ((g.flags & BGR_GF_HAS_FALLBACK) && __any(active)) {
        if (active) {  // bisection in the (tiny) sorted fallback list; its location comes from the blob header
            const uint32_t nfb = (uint32_t)g.hdr->n_fallback;
            const u64* fb = reinterpret_cast<const u64*>(reinterpret_cast<const char*>(g.hdr) + g.hdr->off_fallback);
            uint32_t lo = 0, hi = nfb;
            while (lo < hi) {
                uint32_t mid = (lo + hi) >> 1;
                if (fb[mid] < key) lo = mid + 1; else hi = mid;
            }
            if (lo < nfb && fb[lo] == key) res = (uint32_t)g.hdr->n_placed + lo;
        }
    }
    return res;
}
// membership: MPHF index of key if key is an overlap of the graph, else BGR_NONE (aligner.cpp:158,219,353,361)
template <bool SPEC, typename UP>
__device__ __forceinline__ uint32_t find_key(const BgrDeviceGraph& g, const uint2* LV, UP units, u64 key, bool active) {
    uint32_t idx = mphf_lookup<SPEC>(g, LV, units, key, active);
    if (idx != BGR_NONE && g.keys[idx] != key) idx = BGR_NONE;
    return idx;
}

// ---- stage A: the read's 2-bit words, from the planes the pre-pass (bgr_pack_reads_kernel) or the host packer wrote ----
// FW3: str2num codes (N->3).  NM: 3 on every N.  RCW: reverseComplements(read) (utils.cpp:66-73, non-ACG -> 'A').
// FWQ: what the rolling `num` of getNOverlap/getListOverlap holds: str2num codes inside the first window,
//      nuc2int codes (N->0) for bases entered by update() (aligner.cpp:305-309, utils.cpp:132-140).
// Read r of a batch owns words [woff, woff + ceil(L/32)) of both planes, woff = (read_offs[r] >> 5) + r: computable from the
// ASCII offsets alone (no scan), never overlapping, at most one spare word per read.  The N plane of a read is valid
// only if its bit in `hasn` is set.
__device__ __forceinline__ uint32_t packed_word_offset(u64 off, uint32_t r) { return (uint32_t)(off >> 5) + r; }

__device__ __forceinline__ bool load_packed(const BatchIO& io, uint32_t r, u64 off, uint32_t L, uint32_t W, u64* FW3, u64* NM, int lane) {
    const uint32_t woff = packed_word_offset(off, r), Wr = (L + 31) >> 5;
    const bool hasN = (io.hasn[r >> 5] >> (r & 31)) & 1u;
    for (uint32_t j = lane; j < W; j += 64) {
        u64 f = 0, m = 0;
        if (j < Wr) { f = io.fw3[woff + j]; if (hasN) m = io.nmw[woff + j]; }
        FW3[j] = f;
        NM[j] = m;
    }
    wave_sync();
    return hasN;
}

// 4 ASCII bases starting at byte 4*bi of the read as one dword (first base in the low byte), 0 past the end
__device__ __forceinline__ uint32_t load4(const uint8_t* rd, uint32_t L, uint32_t bi) {
    const uint32_t b0 = bi * 4;
    uint32_t x = 0;
    if (b0 < L) {
        const uint32_t nb = L - b0;
        if (nb >= 4) {
            x = *reinterpret_cast<const u32_unaligned*>(rd + b0);  // possibly unaligned dword
        } else {  // the last 1..3 bases: never touch bytes past the read (they may be past the buffer)
            x = rd[b0];
            if (nb > 1) x |= (uint32_t)rd[b0 + 1] << 8;
            if (nb > 2) x |= (uint32_t)rd[b0 + 2] << 16;
        }
    }
    return x;
}

// 4 ASCII bases (one dword, first base in the low byte; a zero byte = past the end) -> one byte of 2-bit codes, first base
// in the top two bits, and the same for the N mask (3 on 'N').  A0 C1 G2 T3 = ((c>>1)^(c>>2))&3; exact for the
// alphabet ACGTN the parser admits (aligner.cpp:56-61).
__device__ __forceinline__ void pack4(uint32_t x, uint32_t* code, uint32_t* nmask) {
    uint32_t c = ((x >> 1) ^ (x >> 2)) & 0x03030303u;
    const uint32_t t = x ^ 0x4E4E4E4Eu;                                 // zero byte <=> 'N'
    const uint32_t isn = ((t - 0x01010101u) & ~t & 0x80808080u) >> 7;  // 0x01 per 'N' byte
    const uint32_t n3 = isn * 3u;
    c |= n3;
    *code = ((c << 6) | (c >> 4) | (c >> 14) | (c >> 24)) & 0xFFu;
    *nmask = ((n3 << 6) | (n3 >> 4) | (n3 >> 14) | (n3 >> 24)) & 0xFFu;
}

// 32 bases [32j, 32j+32) of a read -> its FW3 word (first base most significant) and N-mask word
__device__ __forceinline__ void pack32(const uint8_t* rd, uint32_t L, uint32_t j, u64 abs_end_ok, u64* fw, u64* nm) {
    uint32_t xs[8];
    if (abs_end_ok) {  // all 32 bytes lie inside the batch buffer: two (unaligned) 16-byte loads, bytes past the read masked off
        typedef uint32_t __attribute__((ext_vector_type(4), aligned(1))) u32x4_unaligned;
        const u32x4_unaligned v0 = *reinterpret_cast<const u32x4_unaligned*>(rd + 32 * j);
        const u32x4_unaligned v1 = *reinterpret_cast<const u32x4_unaligned*>(rd + 32 * j + 16);
        xs[0] = v0.x; xs[1] = v0.y; xs[2] = v0.z; xs[3] = v0.w; xs[4] = v1.x; xs[5] = v1.y; xs[6] = v1.z; xs[7] = v1.w;
        const uint32_t valid = L - 32 * j;  // >= 1
        if (valid < 32) {
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const uint32_t lo = 4u * d;
                if (valid <= lo) xs[d] = 0;
                else if (valid < lo + 4) xs[d] &= 0xFFFFFFFFu >> (8 * (lo + 4 - valid));
            }
        }
    } else {
#pragma unroll
        for (int d = 0; d < 8; ++d) xs[d] = load4(rd, L, 8 * j + d);
    }
    u64 w = 0, n = 0;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        uint32_t c, m;
        pack4(xs[d], &c, &m);
        w = (w << 8) | c;
        n = (n << 8) | m;
    }
    *fw = w;
    *nm = n;
}

// RCW (reverse-complement stream) and FWQ (rolling-update quirk stream) from FW3/NM.  Only needed when the read
// contains N (the rolling k-mers then differ from plain windows) or for the reverse-complement retry; a read
// without N maps from FW3 alone (FWQ == FW3, and the reverse k-mer of a window is rcb of its forward k-mer).
__device__ __forceinline__ void derive_streams(uint32_t L, uint32_t W, uint32_t K1, const u64* FW3, u64* FWQ, u64* RCW, const u64* NM, int lane) {
    for (uint32_t w = lane; w < W; w += 64) {
        long long p = (long long)L - 32 * ((long long)w + 1);
        u64 rcw = 0;
        if (p >= 0) {
            rcw = ~rev2_fast(lds_win32(FW3, (uint32_t)p));
        } else if (p > -32) {
            uint32_t v = (uint32_t)(32 + p);  // valid bases
            u64 x = FW3[0] >> (64 - 2 * v);
            rcw = (~rev2_fast(x)) & (~0ULL << (64 - 2 * v));
        }
        RCW[w] = rcw;
        u64 ge;
        if (32 * w >= K1) ge = ~0ULL;
        else if (32 * (w + 1) <= K1) ge = 0;
        else ge = ~0ULL >> (2 * (K1 - 32 * w));
        FWQ[w] = FW3[w] & ~(NM[w] & ge);
    }
    wave_sync();
}

// ---- candidate scoring shared by the greedy and the exhaustive extension -------------------------------
// The <=4 slots of the neighbour record (getEnd / getBegin, aligner.cpp:147-267) are scored 16 lanes each.
// DIR 0: left step (alignerGreedy.cpp:167-218,268-319; alignerExhaustive.cpp:109-203)
// DIR 1: right step whose read slice starts AFTER the k-1 overlap (checkEndGreedy :322-364; every exhaustive
//        right step, alignerExhaustive.cpp:61-106,206-259)
// DIR 2: later greedy right steps, whose slice INCLUDES the overlap (mapOnRightEndGreedy :221-265)
struct Scored {        // per lane; lanes 16c..16c+15 describe candidate c
    uint32_t cnt;      // Hamming distance over the compared window (full count, not clipped)
    int32_t sid;       // +id forward, -id reversed (what the reference pushes on the path)
    uint32_t ext;      // len - (k-1)
    uint32_t nrec;     // neighbour record at the far end of this unitig in walking direction
    uint32_t info;     // bit 0: the walk ends inside this unitig ("fits"), bit 1: the far-end k-mer is canonical
    int first_zero;    // number of candidates (the reference's nested ifs stop at the first empty slot)
};

// 32 bases of the packed unitig store starting `ub` bases after the start of seq word `fw` (32-bit arithmetic)
__device__ __forceinline__ u64 seq_win32(const u64* seq, uint32_t fw, uint32_t ub) {
    const uint32_t boff = (fw + (ub >> 5)) << 3;  // byte offset < 4 GiB (checked when the graph is built)
    const u64* q = reinterpret_cast<const u64*>(reinterpret_cast<const char*>(seq) + boff);
    const uint32_t s = (ub & 31) * 2;
    const u64 hi = q[0], lo = q[1];
    return (hi << s) | ((lo >> 1) >> (63 - s));
}

// mismatches of one 32-base chunk (v = valid bases in it, 1..32)
__device__ __forceinline__ uint32_t ham_chunk(const BgrDeviceGraph& g, const u64* CMP, const u64* NM, bool useN, uint32_t fw, uint32_t ub,
                                              uint32_t rb, uint32_t v) {
    const u64 x = seq_win32(g.seq, fw, ub) ^ lds_win32(CMP, rb);
    u64 mm = (x | (x >> 1)) & EVEN_BITS;
    u64 nm = 0;
    if (useN) { nm = lds_win32(NM, rb) & EVEN_BITS; mm |= nm; }
    if (g.flags & BGR_GF_HAS_EXC) {  // forward-strand unitig bases outside ACGT: never equal, except N == N
        const u64 abs_base = (u64)fw * 32 + ub;
        const u64* exc = reinterpret_cast<const u64*>(reinterpret_cast<const char*>(g.hdr) + g.hdr->off_exc);
        const uint32_t e = plane32(exc, abs_base);
        if (e) {
            const u64* excn = reinterpret_cast<const u64*>(reinterpret_cast<const char*>(g.hdr) + g.hdr->off_excn);
            const uint32_t en = plane32(excn, abs_base);
            uint32_t m1 = compress_even(mm) | e;
            m1 &= ~(en & compress_even(nm));
            if (v < 32) m1 &= ~(0xFFFFFFFFu >> v);
            return __popc(m1);
        }
    }
    if (v < 32) mm &= ~(~0ULL >> (2 * v));
    return __popcll(mm);
}

template <int DIR>
__device__ __forceinline__ Scored score_candidates(const BgrDeviceGraph& g, const u64* CMP, const u64* NM, bool useN, uint32_t L,
                                                   uint32_t K1, uint32_t rec, bool canon, uint32_t pos, int lane) {
    Scored sc;
    const int c = lane >> 4, sub = lane & 15;
    // getEnd(bin): bin<=rc ? rightIndices : leftIndices ; getBegin(bin): bin<=rc ? leftIndices : rightIndices
    const bool useR = (DIR == 0) ? canon : !canon;
    const uint32_t fbit = canon ? BGR_SLOT_F0 : BGR_SLOT_F1;
    // one 32-byte slot per candidate (graph_layout.h BgrSlot): id + orientation bits, length, sequence address |
    // the unitig's flags and end records (what the NEXT step needs)
    const uint4* sp = reinterpret_cast<const uint4*>(g.recs) + (size_t)(rec * 8u + (useR ? 4u : 0u) + (uint32_t)c) * 2;
    const uint4 sl = sp[0];
    const uint4 m0 = sp[1];  // x = BGR_META_* flags, y = rec_beg, z = rec_end
    const uint32_t id = sl.x & BGR_SLOT_ID_MASK;
    const u64 zmask = __ballot(id == 0);
    sc.first_zero = zmask ? (__ffsll((long long)zmask) - 1) >> 4 : 4;
    const bool valid = c < sc.first_zero;
    const bool fwd = (sl.x & fbit) != 0;
    const uint32_t len = valid ? sl.y : 0;
    // oriented strand start: forward at (Fw, Fo), reverse complement `len` bases further
    const uint32_t fw = sl.z, fo = sl.w + (fwd ? 0u : len);
    sc.sid = fwd ? (int32_t)id : -(int32_t)id;
    sc.ext = len - K1;
    bool fits;
    uint32_t n, ustart, rstart;
    if (DIR == 0) {
        fits = sc.ext >= pos;
        n = fits ? pos : sc.ext;
        ustart = fits ? sc.ext - pos : 0;
        rstart = fits ? 0 : pos - sc.ext;
        sc.nrec = fwd ? m0.y : m0.z;
        sc.info = (fits ? 1u : 0u) | ((m0.x & (fwd ? BGR_META_CANON_BEG : BGR_META_CANON_RCEND)) ? 2u : 0u);
    } else {
        if (DIR == 1) {
            const uint32_t rl = L - pos - K1;
            fits = sc.ext >= rl;
            n = fits ? rl : sc.ext;
            ustart = K1;
            rstart = pos + K1;
        } else {
            const uint32_t rl = L - pos;
            fits = sc.ext >= rl;
            n = fits ? rl : (len < rl ? len : rl);  // read.substr(pos, |u|) is clipped at |read|
            ustart = 0;
            rstart = pos;
        }
        sc.nrec = fwd ? m0.z : m0.y;
        sc.info = (fits ? 1u : 0u) | ((m0.x & (fwd ? BGR_META_CANON_END : BGR_META_CANON_RCBEG)) ? 2u : 0u);
    }
    if (!valid) n = 0;
    // (requesting the next step's slot line here, ahead of the choice, was measured: 1-4 % slower on all workloads)
    // lane `sub` of the candidate's 16 compares bases [32*sub, 32*sub+32); windows longer than 512 bases loop on
    uint32_t cnt = 0;
    const uint32_t b0 = (uint32_t)sub * 32;
    if (b0 < n) cnt = ham_chunk(g, CMP, NM, useN, fw, fo + ustart + b0, rstart + b0, n - b0);
    if (__any(n > 512)) {
        for (uint32_t b = b0 + 512; b < n; b += 512) cnt += ham_chunk(g, CMP, NM, useN, fw, fo + ustart + b, rstart + b, n - b);
    }
    sc.cnt = row16_sum(cnt);
    return sc;
}

// ============================================== greedy ================================================
struct Step {  // wave-uniform result of one extension step
    bool found, fits;
    int32_t sid;
    uint32_t miss, ext, next_rec;
    bool next_canon;
};

template <int DIR>
__device__ __forceinline__ Step greedy_step(const BgrDeviceGraph& g, const u64* CMP, const u64* NM, bool useN, uint32_t L,
                                            uint32_t K1, uint32_t rec, bool canon, uint32_t pos, uint32_t budget, int lane) {
    Step out;
    out.found = false; out.fits = false; out.sid = 0; out.miss = 0; out.ext = 0; out.next_rec = BGR_NONE; out.next_canon = false;
    if (rec == BGR_NONE) return out;  // key not in the table: getBegin/getEnd return an empty list
    const Scored sc = score_candidates<DIR>(g, CMP, NM, useN, L, K1, rec, canon, pos, lane);
    // best = smallest miss, lowest slot on ties, only if miss <= budget (== "first zero wins, else strict min"):
    // the minimum of (miss << 2 | slot) over the candidates
    uint32_t key = 0xFFFFFFFFu;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        const uint32_t kc = (rl32(sc.cnt, cc * 16) << 2) | (uint32_t)cc;
        if (cc < sc.first_zero && kc < key) key = kc;
    }
    if ((key >> 2) > budget) return out;  // includes "no candidate"
    const int bl = (int)(key & 3u) * 16;
    const uint32_t info = rl32(sc.info, bl);
    out.found = true;
    out.fits = (info & 1u) != 0;
    out.next_canon = (info & 2u) != 0;
    out.sid = (int32_t)rl32((uint32_t)sc.sid, bl);
    out.miss = key >> 2;
    out.ext = rl32(sc.ext, bl);
    out.next_rec = rl32(sc.nrec, bl);
    return out;
}

// Left walk (checkBeginGreedy / mapOnLeftEndGreedy, alignerGreedy.cpp:268-319,167-218) from the (k-1)-mer `rec` at
// read position `pos`: ints are stored downwards from PATH[mid-1] (near -> far, offset last), *nl counts them.
// Returns false when a step has no candidate within the budget; the budget is reduced by what the walk spent.
__device__ __forceinline__ bool walk_left(const BgrDeviceGraph& g, const u64* CMP, const u64* NM, bool useN, uint32_t L, uint32_t K1,
                                          uint32_t rec, bool canon, uint32_t pos, uint32_t* budget, int32_t* PATH, uint32_t mid,
                                          uint32_t* nl_out, int lane) {
    uint32_t nl = 0;
    for (;;) {
        if (pos == 0) { if (lane == 0) PATH[mid - 1 - nl] = 0; ++nl; break; }
        Step s = greedy_step<0>(g, CMP, NM, useN, L, K1, rec, canon, pos, *budget, lane);
        if (!s.found) return false;
        if (lane == 0) PATH[mid - 1 - nl] = s.sid;
        ++nl;
        *budget -= s.miss;
        if (s.fits) { if (lane == 0) PATH[mid - 1 - nl] = (int32_t)(s.ext - pos); ++nl; break; }
        pos -= s.ext; rec = s.next_rec; canon = s.next_canon;
    }
    *nl_out = nl;
    return true;
}
// Right walk (checkEndGreedy, then mapOnRightEndGreedy: alignerGreedy.cpp:322-364,221-265): ints stored upwards from
// PATH[at], *nr counts them.
__device__ __forceinline__ bool walk_right(const BgrDeviceGraph& g, const u64* CMP, const u64* NM, bool useN, uint32_t L, uint32_t K1,
                                           uint32_t rec, bool canon, uint32_t pos, uint32_t* budget, int32_t* PATH, uint32_t at,
                                           uint32_t* nr_out, int lane) {
    uint32_t nr = 0;
    bool first = true;
    for (;;) {
        if (first) { if (L - pos - K1 == 0) break; } else { if (L - pos < K1 + 1) break; }
        Step s = first ? greedy_step<1>(g, CMP, NM, useN, L, K1, rec, canon, pos, *budget, lane)
                       : greedy_step<2>(g, CMP, NM, useN, L, K1, rec, canon, pos, *budget, lane);
        if (!s.found) return false;
        if (lane == 0) PATH[at + nr] = s.sid;
        ++nr;
        *budget -= s.miss;
        if (s.fits) break;
        pos += s.ext; rec = s.next_rec; canon = s.next_canon;
        first = false;
    }
    *nr_out = nr;
    return true;
}

// Greedy extension from one anchor (alignReadGreedy's loop body, alignerGreedy.cpp:41-52).  On success the
// path sits in PATH[*p_lo .. *p_lo + *p_n).
__device__ __forceinline__ bool greedy_from_anchor(const BgrDeviceGraph& g, const u64* CMP, const u64* NM, bool useN, uint32_t L,
                                                   uint32_t K1, uint32_t a_rec, bool a_canon, uint32_t a_pos, uint32_t m,
                                                   int32_t* PATH, uint32_t* p_lo, uint32_t* p_n, int lane) {
    // (the two loops are walk_left / walk_right written out: sharing them through out-parameters cost this kernel two
    // spilled registers)
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

// ============================================ exhaustive ==============================================
// alignerExhaustive.cpp:61-259.  The reference recursion explores a candidate only if its own mismatches are
// below the best total found so far in that call, adds the best total of the rest of the walk, and keeps the
// first candidate (slot order) on ties.  That is exactly: among all complete walks of cost <= budget, the
// one of minimal total cost, ties broken by slot order at the shallowest differing step -- i.e. what a
// depth-first search in slot order with ONE running best and strict `<` acceptance returns.  Pruning with the
// running best only skips walks that could not be accepted anyway.
//
// Frame (20 u32, in LDS): [0] rec  [1] pos  [2] cost so far  [3] cursor | ncand<<8 | scored<<16 | canon<<17
//                         [4+4c..] candidate c: sid, next_rec, aux (non-fitting: ext; fitting: the path int
//                                               emitted when the walk ends there), miss | fits<<16 | next_canon<<17
#ifndef BGR_GREEDY_OCC
#define BGR_GREEDY_OCC 6
#endif
#ifndef BGR_ANC_OCC
#define BGR_ANC_OCC 4 /* 128 VGPRs, no spills: 220 vs 200 Mreads/s at 6 */
#endif
#ifndef BGR_DP_OCC
#define BGR_DP_OCC 5 /* 96 VGPRs: at 6 (80 VGPRs) the level search spills 38 VGPRs into scratch inside its loops (3.4 KB written per
                        read, round 1); at 5 three registers are parked once per kernel.  14.4 vs 14.7 ms per 2 M reads; at 4: 16.5 */
#endif
#ifndef BGR_EXH_OCC
#define BGR_EXH_OCC 6 /* waves per SIMD the exhaustive kernel is compiled for */
#endif
#define FR_WORDS 20
#define EXH_OVERFLOW 0xFFFFFFFFu

template <int DIR>
__device__ __forceinline__ uint32_t exh_search(const BgrDeviceGraph& g, const u64* CMP, const u64* NM, bool useN, uint32_t L, uint32_t K1,
                                               uint32_t a_rec, bool a_canon, uint32_t a_pos, uint32_t budget, bool partial,
                                               uint32_t* FR, uint32_t max_frames, int32_t* CUR, int32_t* BEST, uint32_t* best_n, int lane) {
    // returns the best total (budget+1 if none; EXH_OVERFLOW if the search needs more than max_frames frames);
    // the best walk's ints are BEST[0..*best_n) in output order
    uint32_t best = budget + 1;
    *best_n = 0;
    int depth = 0;
    if (lane == 0) { FR[0] = a_rec; FR[1] = a_pos; FR[2] = 0; FR[3] = a_canon ? (1u << 17) : 0u; }
    wave_sync();
    while (depth >= 0) {
        uint32_t* F = FR + (uint32_t)depth * FR_WORDS;
        const uint32_t rec = F[0], pos = F[1], cost = F[2];
        uint32_t ctl = F[3];
        if (!((ctl >> 16) & 1u)) {
            // ---- first visit: base cases, then score the candidates once --------------------------
            const bool end_here = (DIR == 0) ? (pos == 0) : (L - pos - K1 == 0);
            if (end_here) {
                // left: checkBeginExhaustive pushes 0 only at the top (:159 vs :112); right: every depth pushes 0 (:64,:210)
                if (cost < best) {
                    best = cost;
                    if (DIR == 0) {
                        for (int j = lane; j < depth; j += 64) BEST[j] = CUR[depth - 1 - j];  // far -> near
                        *best_n = (uint32_t)depth;
                        if (depth == 0) { if (lane == 0) BEST[0] = 0; *best_n = 1; }
                    } else {
                        for (int j = lane; j < depth; j += 64) BEST[j] = CUR[j];              // near -> far
                        if (lane == 0) BEST[depth] = 0;
                        *best_n = (uint32_t)depth + 1;
                    }
                }
                wave_sync();
                --depth;
                continue;
            }
            uint32_t ncand = 0;
            if (rec != BGR_NONE) {
                const bool canon = (ctl >> 17) & 1u;
                const Scored sc = score_candidates<DIR>(g, CMP, NM, useN, L, K1, rec, canon, pos, lane);
                ncand = (uint32_t)sc.first_zero;
                if ((lane & 15) == 0 && (lane >> 4) < sc.first_zero) {
                    const int c = lane >> 4;
                    const uint32_t miss = sc.cnt > 0xFFFFu ? 0xFFFFu : sc.cnt;
                    const bool fits = (sc.info & 1u) != 0;
                    // fitting: left emits the offset in the last unitig (ext-pos, :126,:175), right |readLeft|+k-1 (:99,:231)
                    const uint32_t aux = fits ? ((DIR == 0) ? sc.ext - pos : L - pos) : sc.ext;
                    F[4 + 4 * c] = (uint32_t)sc.sid;
                    F[5 + 4 * c] = sc.nrec;
                    F[6 + 4 * c] = aux;
                    F[7 + 4 * c] = miss | (fits ? 1u << 16 : 0u) | ((sc.info & 2u) ? 1u << 17 : 0u);
                }
            }
            if (DIR == 1 && depth == 0 && partial && ncand == 0) {  // alignerExhaustive.cpp:217-221 (-i)
                *best_n = 0;
                wave_sync();
                return 0;
            }
            ctl = (ctl & (1u << 17)) | (1u << 16) | (ncand << 8);
            if (lane == 0) F[3] = ctl;
            wave_sync();
        }
        const uint32_t cur = ctl & 0xFFu, ncand = (ctl >> 8) & 0xFFu;
        if (cur >= ncand) { --depth; continue; }
        const uint32_t sid = F[4 + 4 * cur], nrec = F[5 + 4 * cur], aux = F[6 + 4 * cur], pk = F[7 + 4 * cur];
        wave_sync();
        if (lane == 0) F[3] = ctl + 1;
        const uint32_t total = cost + (pk & 0xFFFFu);
        if (total >= best) { wave_sync(); continue; }  // the reference explores a candidate only if miss < best so far
        if ((pk >> 16) & 1u) {
            // the walk ends inside this unitig: a complete solution, strictly better than the running best
            best = total;
            if (DIR == 0) {  // [offset, this (farthest) unitig, ..., nearest unitig]
                for (int j = lane; j < depth; j += 64) BEST[2 + j] = CUR[depth - 1 - j];
                if (lane == 0) { BEST[0] = (int32_t)aux; BEST[1] = (int32_t)sid; }
            } else {         // [nearest ... this (farthest) unitig, end offset]
                for (int j = lane; j < depth; j += 64) BEST[j] = CUR[j];
                if (lane == 0) { BEST[depth] = (int32_t)sid; BEST[depth + 1] = (int32_t)aux; }
            }
            *best_n = (uint32_t)depth + 2;
            wave_sync();
            continue;
        }
        // descend
        if ((uint32_t)depth + 1 >= max_frames) { wave_sync(); return EXH_OVERFLOW; }
        if (lane == 0) {
            CUR[depth] = (int32_t)sid;
            uint32_t* N = F + FR_WORDS;
            N[0] = nrec;
            N[1] = (DIR == 0) ? pos - aux : pos + aux;
            N[2] = total;
            N[3] = ((pk >> 17) & 1u) ? (1u << 17) : 0u;
        }
        wave_sync();
        ++depth;
    }
    return best;
}

// ---- the same search, level by level ------------------------------------------------------------------------
// exh_search visits one (record, read position) node per loop iteration and comes back to it once per candidate.
// A node's best continuation does not depend on how the walk got there, so the search can also run as a dynamic
// programme over the levels of the walk (level = number of unitigs taken):
//   forward   all nodes of a level at once -- up to 4 nodes x 4 slots = 16 candidates, 4 lanes each -- are scored;
//             candidates within the budget that do not end inside their unitig give the next level's nodes, and
//             candidates reaching the same (record, position, strand) share one node (what the depth-first search
//             re-explores once per way of getting there);
//   backward  cost(node) = min over its candidates, first slot on ties, of mismatches + cost(child): the value and the
//             choice the reference's recursion arrives at (its strict `<` keeps the first minimum; a candidate it
//             does not explore because miss >= best-so-far could not have been strictly better);
//   then the walk is read off from the root.
// The result is that of exh_search.  Levels are ~13 for a 250 bp read where exh_search takes ~500 iterations.
// A level with more than 4 distinct nodes, or more than `max_levels` levels, returns EXH_OVERFLOW (the read then
// goes to the depth-first kernel's second pass).
// Tables (u32 words, LDS): T[0..32) two node arrays {rec, pos, strand, cheapest way here} x 4; per level 52 words:
// 16 candidates x {sid, aux, miss | fits<<16 | alive<<17 | child<<18}, 4 node words {cost | argmin<<16 | end<<18 | used<<19};
// then max_levels words of scratch for reading the walk off.
#define DP_LEVEL_WORDS 52
#define DP_FITS (1u << 16)
#define DP_ALIVE (1u << 17)
#define DP_END (1u << 18)
#define DP_USED (1u << 19)

__device__ __forceinline__ uint32_t quad_xor1(uint32_t x) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t quad_xor2(uint32_t x) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true); }

template <int DIR>
__device__ __forceinline__ uint32_t exh_dp(const BgrDeviceGraph& g, const u64* CMP, const u64* NM, bool useN, uint32_t L, uint32_t K1,
                                           uint32_t a_rec, bool a_canon, uint32_t a_pos, uint32_t budget, bool partial,
                                           uint32_t* T, uint32_t max_levels, int32_t* BEST, uint32_t* best_n, int lane) {
    const int i = lane >> 4, c = (lane >> 2) & 3, sub = lane & 3, k = lane >> 2;
    const bool leader = sub == 0;
    *best_n = 0;
    if (lane == 0) { T[0] = a_rec; T[1] = a_pos; T[2] = a_canon ? 1u : 0u; T[3] = 0; }
    wave_sync();
    uint32_t n_cur = 1, lvl = 0;
    for (; n_cur; ++lvl) {
        if (lvl >= max_levels) { wave_sync(); return EXH_OVERFLOW; }
        const uint4 nd = reinterpret_cast<const uint4*>(T)[(lvl & 1) * 4 + i];
        const bool nvalid = (uint32_t)i < n_cur;
        const uint32_t rec = nd.x, pos = nd.y, prefix = nd.w;
        const bool canon = (nd.z & 1u) != 0;
        const bool end_here = nvalid && ((DIR == 0) ? (pos == 0) : (L - pos - K1 == 0));
        const bool has_rec = nvalid && !end_here && rec != BGR_NONE;
        const bool useR = (DIR == 0) ? canon : !canon;
        const uint32_t fbit = canon ? BGR_SLOT_F0 : BGR_SLOT_F1;
        uint4 sl = make_uint4(0, 0, 0, 0), m0 = make_uint4(0, 0, 0, 0);
        if (has_rec) {
            const uint4* sp = reinterpret_cast<const uint4*>(g.recs) + (size_t)(rec * 8u + (useR ? 4u : 0u) + (uint32_t)c) * 2;
            sl = sp[0];
            m0 = sp[1];
        }
        const uint32_t id = sl.x & BGR_SLOT_ID_MASK;
        const u64 zmask = __ballot(leader && (!has_rec || id == 0));
        const uint32_t nb = (uint32_t)(zmask >> (16 * i)) & 0x1111u;
        const uint32_t first_zero = nb ? (uint32_t)(__ffs((int)nb) - 1) >> 2 : 4u;  // the reference stops at the first empty slot
        if (DIR == 1 && partial && lvl == 0) {  // alignerExhaustive.cpp:217-221 (-i): nothing starts here, nothing to pay
            const bool none = rl32(first_zero, 0) == 0 && !(rl32(end_here ? 1u : 0u, 0));
            if (none) { wave_sync(); return 0; }
        }
        const bool valid = has_rec && (uint32_t)c < first_zero;
        const bool fwd = (sl.x & fbit) != 0;
        const uint32_t len = valid ? sl.y : 0;
        const uint32_t fw = sl.z, fo = sl.w + (fwd ? 0u : len);
        const int32_t sid = fwd ? (int32_t)id : -(int32_t)id;
        const uint32_t ext = len - K1;
        bool fits;
        uint32_t n, ustart, rstart, nrec;
        bool ncanon;
        if (DIR == 0) {
            fits = ext >= pos;
            n = fits ? pos : ext;
            ustart = fits ? ext - pos : 0;
            rstart = fits ? 0 : pos - ext;
            nrec = fwd ? m0.y : m0.z;
            ncanon = (m0.x & (fwd ? BGR_META_CANON_BEG : BGR_META_CANON_RCEND)) != 0;
        } else {
            const uint32_t rl = L - pos - K1;
            fits = ext >= rl;
            n = fits ? rl : ext;
            ustart = K1;
            rstart = pos + K1;
            nrec = fwd ? m0.z : m0.y;
            ncanon = (m0.x & (fwd ? BGR_META_CANON_END : BGR_META_CANON_RCBEG)) != 0;
        }
        if (!valid) n = 0;
        uint32_t cnt = 0;
        for (uint32_t b = (uint32_t)sub * 32; b < n; b += 128) cnt += ham_chunk(g, CMP, NM, useN, fw, fo + ustart + b, rstart + b, n - b);
        cnt += quad_xor1(cnt);
        cnt += quad_xor2(cnt);
        const uint32_t miss = cnt > 0xFFFFu ? 0xFFFFu : cnt;
        const uint32_t ptotal = prefix + miss;
        const bool alive = valid && ptotal <= budget;  // a walk through here costs at least this much
        const uint32_t aux = fits ? ((DIR == 0) ? ext - pos : L - pos) : 0u;
        const bool need = alive && !fits;
        const uint32_t npos = (DIR == 0) ? pos - ext : pos + ext;
        const u64 key = (u64)nrec << 32 | (u64)(npos << 1) | (ncanon ? 1u : 0u);
        // candidates that reach the same node share it: the first of them (slot order over the level) creates it
        const u64 needmask = __ballot(need && leader);
        uint32_t first_k = (uint32_t)k;
        for (u64 mm = needmask; mm; mm &= mm - 1) {
            const int src = __ffsll((long long)mm) - 1;
            const u64 kk = rl64(key, src);
            if (need && kk == key && (uint32_t)(src >> 2) < first_k) first_k = (uint32_t)(src >> 2);
        }
        const bool is_first = need && leader && first_k == (uint32_t)k;
        const u64 fmask = __ballot(is_first);
        const uint32_t n_next = (uint32_t)__popcll(fmask);
        if (n_next > 4) { wave_sync(); return EXH_OVERFLOW; }
        const uint32_t child = (uint32_t)__popcll(fmask & ((1ULL << (4 * first_k)) - 1));
        uint32_t* NX = T + ((lvl + 1) & 1) * 16;
        if (is_first) { NX[child * 4] = nrec; NX[child * 4 + 1] = npos; NX[child * 4 + 2] = ncanon ? 1u : 0u; NX[child * 4 + 3] = 0xFFFFFFFFu; }
        uint32_t* LV_ = T + 32 + lvl * DP_LEVEL_WORDS;
        if (leader && nvalid) {
            LV_[3 * k] = (uint32_t)sid;
            LV_[3 * k + 1] = aux;
            LV_[3 * k + 2] = miss | (fits ? DP_FITS : 0u) | (alive ? DP_ALIVE : 0u) | (child << 18);
        }
        if (leader && c == 0) LV_[48 + i] = (nvalid ? DP_USED : 0u) | (end_here ? DP_END : 0u);
        wave_sync();
        if (need && leader) atomicMin(&NX[child * 4 + 3], ptotal);
        wave_sync();
        n_cur = n_next;
    }
    const uint32_t levels = lvl;
    // ---- backward: cheapest continuation of every node, first slot on ties --------------------------------
    for (int l = (int)levels - 1; l >= 0; --l) {
        uint32_t* LV_ = T + 32 + (uint32_t)l * DP_LEVEL_WORDS;
        uint32_t sel = 0xFFFFFFFFu;
        if (lane < 16) {
            const uint32_t used = LV_[48 + (lane >> 2)] & DP_USED;
            uint32_t total = 0xFFFFu;
            if (used) {
                const uint32_t pk = LV_[3 * lane + 2];
                if (pk & DP_ALIVE) {
                    uint32_t cc = 0;
                    if (!(pk & DP_FITS)) cc = ((uint32_t)l + 1 < levels) ? (T[32 + ((uint32_t)l + 1) * DP_LEVEL_WORDS + 48 + ((pk >> 18) & 3u)] & 0xFFFFu) : 0xFFFFu;
                    total = (pk & 0xFFFFu) + cc;
                    if (total > 0xFFFFu) total = 0xFFFFu;
                }
            }
            sel = total << 2 | (uint32_t)(lane & 3);
        }
        uint32_t o = quad_xor1(sel);
        sel = o < sel ? o : sel;
        o = quad_xor2(sel);
        sel = o < sel ? o : sel;
        wave_sync();
        if (lane < 16 && (lane & 3) == 0) {
            const uint32_t fl = LV_[48 + (lane >> 2)];
            const uint32_t cost = (fl & DP_END) ? 0u : (sel >> 2);
            LV_[48 + (lane >> 2)] = (fl & (DP_END | DP_USED)) | cost | ((sel & 3u) << 16);
        }
        wave_sync();
    }
    const uint32_t root = T[32 + 48];
    const uint32_t s = (root & DP_END) ? 0u : (root & 0xFFFFu);
    if (s > budget) return budget + 1;
    // ---- read the walk off (lane 0; at most `levels` steps) -------------------------------------------------
    if (lane == 0) {
        int32_t* W_ = reinterpret_cast<int32_t*>(T + 32 + max_levels * DP_LEVEL_WORDS);
        uint32_t d = 0, node = 0, n_out = 0;
        for (;;) {
            const uint32_t* LV_ = T + 32 + d * DP_LEVEL_WORDS;
            const uint32_t w = LV_[48 + node];
            if (w & DP_END) {
                // left: checkBeginExhaustive pushes 0 only at the top (:159 vs :112); right: every depth pushes 0 (:64,:210)
                if (DIR == 0) {
                    if (d == 0) { BEST[0] = 0; n_out = 1; }
                    else { for (uint32_t j = 0; j < d; ++j) BEST[j] = W_[d - 1 - j]; n_out = d; }
                } else {
                    for (uint32_t j = 0; j < d; ++j) BEST[j] = W_[j];
                    BEST[d] = 0;
                    n_out = d + 1;
                }
                break;
            }
            const uint32_t cc = (w >> 16) & 3u;
            const uint32_t* C = LV_ + 3 * (node * 4 + cc);
            const int32_t sid = (int32_t)C[0];
            const uint32_t aux = C[1], pk = C[2];
            if (pk & DP_FITS) {
                if (DIR == 0) {  // [offset, this (farthest) unitig, ..., nearest unitig]
                    BEST[0] = (int32_t)aux; BEST[1] = sid;
                    for (uint32_t j = 0; j < d; ++j) BEST[2 + j] = W_[d - 1 - j];
                } else {         // [nearest ... this (farthest) unitig, end offset]
                    for (uint32_t j = 0; j < d; ++j) BEST[j] = W_[j];
                    BEST[d] = sid; BEST[d + 1] = (int32_t)aux;
                }
                n_out = d + 2;
                break;
            }
            W_[d] = sid;
            node = (pk >> 18) & 3u;
            if (++d >= levels) break;  // (cannot happen: a finite cost ends in a fitting candidate or an end node)
        }
        T[0] = n_out;
    }
    wave_sync();
    *best_n = rl32(T[0], 0);
    wave_sync();
    return s;
}

// ================================================ kernels ===============================================
template <bool STAGE>
__device__ __forceinline__ const uint32_t* block_prologue(const BgrDeviceGraph& g, u64* lds, uint2** LVout, uint32_t* mphf_words) {
    // LDS: [level descriptors 512 B][optional MPHF copy][per-wave regions]
    uint2* LV = reinterpret_cast<uint2*>(lds);
    if (threadIdx.x < BGR_MAX_LEVELS) LV[threadIdx.x] = make_uint2(g.hdr->levels[threadIdx.x].units, g.hdr->levels[threadIdx.x].base);
    *mphf_words = STAGE ? (g.units_bytes + 7) / 8 : 0;
    const uint32_t* units = g.units;
    if (STAGE) {
        const uint4* src = reinterpret_cast<const uint4*>(g.units);
        uint4* dst = reinterpret_cast<uint4*>(lds + 64);
        for (uint32_t i = threadIdx.x; i < g.units_bytes / 16; i += blockDim.x) dst[i] = src[i];
        units = reinterpret_cast<const uint32_t*>(lds + 64);
    }
    __syncthreads();
    *LVout = LV;
    return units;
}

// Arena space comes in per-wave chunks: ONE global atomic per ~50 reads instead of one per read (a
// single-address atomic saturates near 90 M/s chip-wide, MI355X_MICROARCH.md "dequeue").
__device__ __forceinline__ uint32_t publish_path(const BatchIO& io, const int32_t* PATH, uint32_t p_lo, uint32_t p_n,
                                                 uint32_t* chunk_pos, uint32_t* chunk_end, int lane) {
    if (p_n > *chunk_end - *chunk_pos) {
        const uint32_t want = p_n > io.arena_chunk ? p_n : io.arena_chunk;
        uint32_t got = 0;
        if (lane == 0) got = atomicAdd(io.cursor, want);
        *chunk_pos = rl32(got, 0);
        *chunk_end = *chunk_pos + want;
    }
    const uint32_t abase = *chunk_pos;
    *chunk_pos += p_n;
    if (abase + p_n <= io.arena_cap) {
        for (uint32_t j = lane; j < p_n; j += 64) io.arena[abase + j] = PATH[p_lo + j];
    } else if (lane == 0) {
        io.cursor[1] = 1;  // overflow: reported by the host as an error
    }
    return abase;
}

template <bool STAGE>
__global__ void __launch_bounds__(1024, BGR_GREEDY_OCC) bgr_align_greedy_kernel(BgrDeviceGraph g, BatchIO io, KernelParams prm) {
    extern __shared__ u64 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const uint32_t W = io.words_per_read;
    const uint32_t K1 = g.k - 1;
    // as the second pass behind bgr_align_greedy4_kernel it maps only the reads that kernel listed (count in cursor[subset_ctr])
    const uint32_t total = io.subset ? io.cursor[io.subset_ctr] : io.n_reads;
    if ((uint32_t)(blockIdx.x * waves) >= total) return;  // nothing for this workgroup (before it copies the cascade into LDS)
    uint2* LV;
    uint32_t mphf_words;
    const uint32_t* units = block_prologue<STAGE>(g, lds, &LV, &mphf_words);
    const uint32_t per_wave_words = 4 * W + io.path_cap / 2;
    u64* FW3 = lds + 64 + mphf_words + (u64)wave * per_wave_words;
    u64* FWQ = FW3 + W;
    u64* RCW = FWQ + W;
    u64* NM = RCW + W;
    int32_t* PATH = reinterpret_cast<int32_t*>(NM + W);

    uint32_t c_lane = 0;                    // per-lane status counter (a VGPR: the kernel is short of SGPRs, not of VGPRs)
    uint32_t chunk_pos = 0, chunk_end = 0;  // this wave's slice of the path arena
    // getNOverlap(read, 0) still looks at position 0 before testing the count (aligner.cpp:349-368)
    const uint32_t effort = prm.effort ? prm.effort : 1;

    for (uint32_t it = blockIdx.x * waves + wave; it < total; it += gridDim.x * waves) {
        const uint32_t r = io.subset ? io.subset[it] : it;
        const u64 off = io.read_offs[r];
        const uint32_t L = (uint32_t)(io.read_offs[r + 1] - off);
        // (prefetching the next read one iteration ahead was measured: no gain at 24 waves/CU, it only added spills)
        const bool hasN = load_packed(io, r, off, L, W, FW3, NM, lane);
        bool derived = false;

        // ---- passes: forward read, then its reverse complement (alignerGreedy.cpp:54) -----------
        uint32_t status = BGR_ST_NOANCHOR, p_lo = 0, p_n = 0;
        uint32_t npos = L >= K1 ? L - K1 + 1 : 0;
        if (!prm.effort && npos > 1) npos = 1;
#ifdef BGR_PHASE_TIMING  /* diagnostic builds: env BGR_DEBUG_STOP = 1 stops after packing, 2 after the position scan */
        if (prm.debug_stop == 1) npos = 0;
#endif
        for (int pass = 0; pass < 2; ++pass) {
            if ((pass == 1 || hasN) && !derived) { derive_streams(L, W, K1, FW3, FWQ, RCW, NM, lane); derived = true; }
            // A read without N: FWQ == FW3 and the rolling reverse k-mer == rcb(forward k-mer), so pass 0 needs FW3 only.
            const bool plain = (pass == 0) && !hasN;
            const u64* A = pass ? RCW : (plain ? FW3 : FWQ);   // forward-strand k-mers of this pass
            const u64* B = pass ? FW3 : RCW;                   // reverse-strand k-mers (rolling nuc2intrc: N -> 0)
            const u64* CMP = pass ? RCW : FW3;                 // characters compared by missmatchNumber
            const bool useN = (pass == 0) && hasN;
            uint32_t tried = 0;
            bool done = false;
            for (uint32_t base = 0; base < npos && !done && tried < effort; base += 64) {
                const uint32_t i = base + lane;
                const bool valid = i < npos;
                u64 num = 0, rcn = 0;
                if (valid) {
                    num = lds_win32(A, i) >> (64 - 2 * K1);
                    rcn = plain ? rcb_fast(num, K1) : lds_win32(B, L - K1 - i) >> (64 - 2 * K1);
                }
                const u64 rep = num < rcn ? num : rcn;
                const uint32_t idx = find_key<!STAGE>(g, LV, units, rep, valid);
                u64 mask = __ballot(idx != BGR_NONE);
#ifdef BGR_PHASE_TIMING
                if (prm.debug_stop == 2) { if (mask) { ++tried; done = true; p_n = 0; } mask = 0; }
#endif
                while (mask && tried < effort) {
                    const int src = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    ++tried;
                    // the anchor's k-mers are re-read from LDS (uniform) rather than kept in two 64-bit VGPRs across the walk
                    const uint32_t a_pos = base + (uint32_t)src;
                    const u64 a_num = rl64(lds_win32(A, a_pos) >> (64 - 2 * K1), 0);
                    const u64 a_rcn = plain ? rcb_fast(a_num, K1) : rl64(lds_win32(B, L - K1 - a_pos) >> (64 - 2 * K1), 0);
                    uint32_t a_rec = rl32(idx, src);
                    // getBegin/getEnd recompute rc = rcb(num) (aligner.cpp:149,211); it differs from the
                    // rolling rcnum only when an N was rolled into the window.
                    const u64 rc2 = rcb_fast(a_num, K1);
                    if (rc2 != a_rcn) a_rec = find_key<false>(g, LV, units, a_num < rc2 ? a_num : rc2, true);
                    if (greedy_from_anchor(g, CMP, NM, useN, L, K1, a_rec, a_num <= rc2, a_pos, prm.max_mismatch, PATH, &p_lo, &p_n, lane)) {
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
        if ((status & BGR_ST_MASK) == BGR_ST_ALIGNED) abase = publish_path(io, PATH, p_lo, p_n, &chunk_pos, &chunk_end, lane);
        else p_n = 0;
        if (lane == 0) io.results[r] = make_uint2(abase, p_n | (status << 24));
        c_lane += (uint32_t)lane == (status & BGR_ST_MASK);  // lane s counts the reads that ended with status s
        wave_sync();
    }
    {   // aligner.h:68 counters: [0] readNumber [1] noOverlapRead [2] alignedRead [3] notAligned
        unsigned long long* counters = reinterpret_cast<unsigned long long*>(io.cursor + 16);
        const uint32_t total = rl32(c_lane, BGR_ST_NOANCHOR) + rl32(c_lane, BGR_ST_FAILED) + rl32(c_lane, BGR_ST_ALIGNED);
        if (lane == 0 && total) atomicAdd(&counters[0], (unsigned long long)total);
        if (lane == BGR_ST_NOANCHOR && c_lane) atomicAdd(&counters[1], (unsigned long long)c_lane);
        if (lane == BGR_ST_ALIGNED && c_lane) atomicAdd(&counters[2], (unsigned long long)c_lane);
        if (lane == BGR_ST_FAILED && c_lane) atomicAdd(&counters[3], (unsigned long long)c_lane);
    }
}

// ================================= greedy, four reads per wavefront ====================================
// bgr_align_greedy_kernel above walks one read per wave: a walk step is two dependent loads (slot, bases) scored by at
// most 4 candidates x a few 32-base chunks, i.e. a handful of the 64 lanes, and per read there are ~4 such steps in a row.
// Here a wave takes FOUR reads: their position scans still run one after the other on all 64 lanes (a scan is lane-
// efficient: one (k-1)-mer per lane), then the four extensions run side by side, 16 lanes each (4 candidates x 4 chunk
// lanes = 128 bases per step), so four slot/base load chains are in flight per wave and every wave instruction of a walk
// step serves four reads.  Path ints stay in registers (lane j of a group holds int j of each direction).
// The kernel settles the common case only -- no N in the read, first anchor extends within the budget (or there is
// no anchor at all), at most G4_PATH ints per direction.  Every other read (N, failed first anchor: the reference then
// tries further anchors and the reverse complement, alignerGreedy.cpp:41-56; long paths) is put on a list and mapped by
// bgr_align_greedy_kernel right behind, so results are the reference's for every read.
#ifndef BGR_G4_OCC
#define BGR_G4_OCC 8
#endif
#define G4_PATH 16

__device__ __forceinline__ uint32_t row_ror4(uint32_t x) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x124, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t row_ror8(uint32_t x) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x128, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t lane_get(uint32_t v, uint32_t src_lane) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v); }

// One extension step for up to four walks, one per 16-lane group.  `phase` (uniform within a group): 0 = the group sits
// out, 1 = left step (checkBeginGreedy / mapOnLeftEndGreedy), 2 = first right step (checkEndGreedy: the read slice starts
// behind the k-1 overlap), 3 = later right step (mapOnRightEndGreedy: the slice includes the overlap).  alignerGreedy.cpp:167-364.
// Result, uniform within a group: next record | next canonical << 28 | fits << 29 | found << 30; miss; ext; sid.
#define G4_REC_MASK 0x0FFFFFFFu
#define G4_CANON (1u << 28)
#define G4_FITS (1u << 29)
#define G4_FOUND (1u << 30)
__device__ __forceinline__ uint32_t g4_step(const BgrDeviceGraph& g, const u64* FW, uint32_t L, uint32_t K1, uint32_t phase, uint32_t rec, uint32_t canon,
                                            uint32_t pos, uint32_t budget, int lane, uint32_t* miss, uint32_t* ext_o, int32_t* sid_o) {
    const uint32_t c = ((uint32_t)lane >> 2) & 3u, q = (uint32_t)lane & 3u;
    const uint32_t left = phase == 1 ? 1u : 0u;
    // getEnd(bin): bin<=rc ? rightIndices : leftIndices ; getBegin(bin): bin<=rc ? leftIndices : rightIndices
    const uint32_t useR = canon == left ? 1u : 0u;
    uint4 sl = make_uint4(0, 0, 0, 0), m0 = make_uint4(0, 0, 0, 0);
    if (phase != 0 && rec != G4_REC_MASK) {
        const uint4* sp = reinterpret_cast<const uint4*>(g.recs) + (size_t)(rec * 8u + useR * 4u + c) * 2;
        sl = sp[0];
        m0 = sp[1];
    }
    const uint32_t id = sl.x & BGR_SLOT_ID_MASK;
    const u64 zmask = __ballot(id == 0);  // (all lanes of a candidate agree; a group that sits out reads as "no candidate")
    const uint32_t nb = (uint32_t)(zmask >> ((uint32_t)lane & 48u)) & 0x1111u;
    const uint32_t first_zero = nb ? (uint32_t)(__ffs((int)nb) - 1) >> 2 : 4u;  // the reference stops at the first empty slot
    const uint32_t fwd = (sl.x & (canon ? BGR_SLOT_F0 : BGR_SLOT_F1)) ? 1u : 0u;
    const uint32_t len = sl.y;
    const uint32_t fw = sl.z, fo = sl.w + (fwd ? 0u : len);
    const uint32_t ext = len - K1;
    // left: `rl` bases of the read lie left of the overlap; right: behind it (first step) / from its start (later steps)
    const uint32_t kk = phase == 2 ? K1 : 0u;
    const uint32_t rl = left ? pos : L - pos - kk;
    const uint32_t fits = ext >= rl ? 1u : 0u;
    const uint32_t span = left ? ext : ext + K1 - kk;  // what is compared when the walk goes on: the unitig beyond the overlap, or all of it
    uint32_t n = fits ? rl : (span < rl ? span : rl);  // (later right steps: read.substr(pos, |u|) is clipped at |read|)
    const uint32_t ustart = left ? ext - n : kk;
    const uint32_t rstart = left ? rl - n : pos + kk;
    const uint32_t nrec = fwd == left ? m0.y : m0.z;
    const uint32_t cbit = left ? (fwd ? BGR_META_CANON_BEG : BGR_META_CANON_RCEND) : (fwd ? BGR_META_CANON_END : BGR_META_CANON_RCBEG);
    if (c >= first_zero) n = 0;
    uint32_t cnt = 0;
    for (uint32_t b = q * 32; __any(b < n); b += 128)
        if (b < n) cnt += ham_chunk(g, FW, nullptr, false, fw, fo + ustart + b, rstart + b, n - b);
    cnt += quad_xor1(cnt);
    cnt += quad_xor2(cnt);
    // best = smallest miss, lowest slot on ties, only if miss <= budget (== "first zero wins, else strict min")
    uint32_t key = c < first_zero ? ((cnt > 0x0FFFFFFFu ? 0x0FFFFFFFu : cnt) << 2) | c : 0xFFFFFFFFu;
    uint32_t o = row_ror4(key);
    key = o < key ? o : key;
    o = row_ror8(key);
    key = o < key ? o : key;
    const uint32_t src = ((uint32_t)lane & 48u) | ((key & 3u) << 2);
    const uint32_t pk = nrec | ((m0.x & cbit) ? G4_CANON : 0u) | (fits ? G4_FITS : 0u);
    const uint32_t w1 = lane_get(pk, src);
    *ext_o = lane_get(ext, src);
    *sid_o = (int32_t)lane_get(fwd ? id : 0u - id, src);
    *miss = key >> 2;
    return (key >> 2) <= budget ? w1 | G4_FOUND : 0u;  // an empty record gives key 0xFFFFFFFF: not found
}

// A read the kernel cannot finish in this launch is listed with where to go on: which strand (the reference maps the
// reverse complement once every forward anchor has failed, alignerGreedy.cpp:54), how many anchors of that strand have
// been tried (getNOverlap hands out the first `effort` of them) and the position the scan resumes from.
#define G4_ST_RC (1u << 31)
#define G4_ST_TRIED_SHIFT 20
#define G4_ST_POS_MASK 0xFFFFFu

// LIST: the launch maps the reads an earlier launch listed (io.subset) instead of all reads of the batch.
template <bool STAGE, bool LIST>
__global__ void __launch_bounds__(1024, BGR_G4_OCC) bgr_align_greedy4_kernel(BgrDeviceGraph g, BatchIO io, KernelParams prm) {
    extern __shared__ u64 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const uint32_t W = io.words_per_read;  // <= 16 (checked by the host): one lane per word of a read
    const uint32_t K1 = g.k - 1;
    // later passes map the reads an earlier pass listed (count in cursor[subset_ctr]), from the state it left in g4_state
    const uint32_t total = LIST ? io.cursor[io.subset_ctr] : io.n_reads;
    if ((uint32_t)(blockIdx.x * waves) * 4u >= total) return;  // nothing for this workgroup (before it copies the cascade into LDS)
    uint2* LV;
    uint32_t mphf_words;
    const uint32_t* units = block_prologue<STAGE>(g, lds, &LV, &mphf_words);
    u64* RD = lds + 64 + mphf_words + (u64)wave * (8 * W);  // the four reads of this wave: forward words | reverse-complement words
    const uint32_t grp = (uint32_t)lane >> 4, sub = (uint32_t)lane & 15u;
    const uint32_t m = prm.max_mismatch;
    const uint32_t eff = prm.effort ? prm.effort : 1;  // getNOverlap(read, 0) still takes a hit at position 0 (aligner.cpp:349-368)

    uint32_t c_noov = 0, c_al = 0, c_na = 0;  // wave-uniform counts of the reads settled here
    uint32_t chunk_pos = 0, chunk_end = 0;    // this wave's slice of the path arena
    uint32_t lst_pos = 0, lst_end = 0;        // this wave's slice of the list for the next pass (reserved io.list_chunk entries at a
                                              // time: one single-address atomic per listed read caps a launch near 300 M/s)

    // (per-lane flags are kept as 0/1 words in VGPRs on purpose: as `bool`s they become 64-bit lane masks in SGPRs, and this
    // kernel is short of SGPRs, not of VGPRs)
    for (uint32_t ibase = (blockIdx.x * waves + wave) * 4; ibase < total; ibase += gridDim.x * waves * 4) {
        const uint32_t it = ibase + grp;
        uint32_t have = it < total ? 1u : 0u;
        uint32_t r = 0, st = 0;
        u64 off = 0;
        uint32_t L = 0, fast = 0;
        if (have && LIST) {
            r = io.subset[it];
            if (r == BGR_NONE) have = 0;  // a hole: the unused tail of some wave's chunk of the list
            else st = io.g4_state[r];
        } else {
            r = it;
        }
        if (have) {
            off = io.read_offs[r];
            L = (uint32_t)(io.read_offs[r + 1] - off);
            fast = ((io.hasn[r >> 5] >> (r & 31)) & 1u) ^ 1u;  // a read with an N goes to the general kernel
        }
        uint32_t rc = st >> 31;
        uint32_t tried = (st >> G4_ST_TRIED_SHIFT) & 0x7FFu;
        uint32_t s_from = st & G4_ST_POS_MASK;
        u64* F = RD + grp * (2 * W);
        {   // stage the 2-bit words: lane `sub` of a group brings word `sub` of its read
            u64 f = 0;
            if (fast && sub < ((L + 31) >> 5)) f = io.fw3[packed_word_offset(off, r) + sub];
            if (sub < W) F[sub] = f;
        }
        wave_sync();
        uint32_t act = fast;                 // the group takes part in the current round
        uint32_t outcome = 4, nst = 0, rc_out = rc;  // what became of the read (below); 4 = general kernel
        uint32_t nl = 0, nr = 0;
        int32_t pl = 0, pr = 0;  // lane `sub` keeps path int number `sub` of the left walk (near -> far, offset last) / right walk
        for (uint32_t round = 0;; ++round) {
            if (__any(act && rc)) {  // reverseComplements(read) (utils.cpp:66-73) of the groups that are on their second strand
                if (act && rc && sub < W) {
                    const long long p = (long long)L - 32 * ((long long)sub + 1);
                    u64 w = 0;
                    if (p >= 0) w = ~rev2_fast(lds_win32(F, (uint32_t)p));
                    else if (p > -32) { const uint32_t v = (uint32_t)(32 + p); w = (~rev2_fast(F[0] >> (64 - 2 * v))) & (~0ULL << (64 - 2 * v)); }
                    F[W + sub] = w;
                }
                wave_sync();
            }
            const u64* FW = F + (rc ? W : 0);  // the strand this pass maps

            // ---- anchors (getNOverlap, aligner.cpp:345-378): the next overlap (k-1)-mer of each read from where its scan stands and,
            // when it lies in the same 64 positions, the one after it; record | canonical << 28
            uint32_t a_pos = 0, a_rec = BGR_NONE, b_pos = 0, b_rec = BGR_NONE;
            for (uint32_t q = 0; q < 4; ++q) {
                if (!rl32(act, (int)(16 * q))) continue;
                const uint32_t Lq = rl32(L, (int)(16 * q));
                const u64* A = RD + q * (2 * W) + (rl32(rc, (int)(16 * q)) ? W : 0);
                const uint32_t left_q = eff - rl32(tried, (int)(16 * q));  // anchors this strand may still try (>= 1)
                uint32_t npos = Lq >= K1 ? Lq - K1 + 1 : 0;
                if (!prm.effort && npos > 1) npos = 1;
                for (uint32_t base = rl32(s_from, (int)(16 * q)); base < npos; base += 64) {
                    const uint32_t i = base + (uint32_t)lane;
                    const bool valid = i < npos;
                    u64 num = 0;
                    if (valid) num = lds_win32(A, i) >> (64 - 2 * K1);
                    const u64 rcn = rcb_fast(num, K1);  // no N in the read: the rolling reverse k-mer is rcb of the forward one
                    uint32_t idx = find_key<!STAGE>(g, LV, units, num < rcn ? num : rcn, valid);
                    const u64 mask = __ballot(idx != BGR_NONE);
                    if (mask) {
                        if (idx != BGR_NONE && num <= rcn) idx |= G4_CANON;
                        const int s1 = __ffsll((long long)mask) - 1;
                        const u64 mask2 = mask & (mask - 1);
                        const uint32_t h1 = rl32(idx, s1);
                        uint32_t h2 = BGR_NONE, p2 = 0;
                        if (mask2 && left_q >= 2) {  // a second anchor is tried when the first fails
                            const int s2 = __ffsll((long long)mask2) - 1;
                            h2 = rl32(idx, s2);
                            p2 = base + (uint32_t)s2;
                        }
                        if (grp == q) { a_pos = base + (uint32_t)s1; a_rec = h1; b_pos = p2; b_rec = h2; }
                        break;
                    }
                }
            }

            // ---- extension (alignReadGreedy's loop body, alignerGreedy.cpp:41-52), four reads abreast; a group whose anchor fails
            // starts over from the next one, if the scan has seen it, while the others go on ----
            uint32_t phase = (act && a_rec != BGR_NONE) ? 1u : 0u;
            uint32_t pos = a_pos, rec = a_rec & G4_REC_MASK, canon = (a_rec >> 28) & 1u, budget = m;
            uint32_t bad = 0, failed = 0;
            if (act) { nl = 0; nr = 0; }
            for (;;) {
                if (phase == 1 && pos == 0) {  // the left walk reached the read's first base: push 0, then the right side of the anchor
                    if (sub == nl) pl = 0;
                    ++nl;
                    phase = 2; pos = a_pos; rec = a_rec & G4_REC_MASK; canon = (a_rec >> 28) & 1u;
                }
                if (phase == 2 && L - pos - K1 == 0) phase = 0;  // nothing right of the anchor: aligned
                if (phase == 3 && L - pos < K1 + 1) phase = 0;   // |readLeft| < k: aligned
                if ((phase == 1 && nl > G4_PATH - 2) || (phase >= 2 && nr > G4_PATH - 1)) { bad = 1; phase = 0; }  // path too long for the registers
                if (!__any(phase != 0)) break;
                uint32_t miss, ext;
                int32_t sid;
                const uint32_t w1 = g4_step(g, FW, L, K1, phase, rec, canon, pos, budget, lane, &miss, &ext, &sid);
                if (phase != 0) {
                    if (!(w1 & G4_FOUND)) {
                        ++tried;
                        if (b_rec != BGR_NONE) {  // next anchor of getNOverlap's list, from scratch
                            a_pos = b_pos; a_rec = b_rec; b_rec = BGR_NONE;
                            nl = 0; nr = 0; budget = m;
                            phase = 1; pos = a_pos; rec = a_rec & G4_REC_MASK; canon = (a_rec >> 28) & 1u;
                        } else { failed = 1; phase = 0; }
                    } else if (phase == 1) {
                        if (sub == nl) pl = sid;
                        ++nl;
                        budget -= miss;
                        if (w1 & G4_FITS) {
                            if (sub == nl) pl = (int32_t)(ext - pos);
                            ++nl;
                            phase = 2; pos = a_pos; rec = a_rec & G4_REC_MASK; canon = (a_rec >> 28) & 1u;
                        } else { pos -= ext; rec = w1 & G4_REC_MASK; canon = (w1 >> 28) & 1u; }
                    } else {
                        if (sub == nr) pr = sid;
                        ++nr;
                        budget -= miss;
                        if (w1 & G4_FITS) phase = 0;
                        else { pos += ext; rec = w1 & G4_REC_MASK; canon = (w1 >> 28) & 1u; phase = 3; }
                    }
                }
            }

            // ---- what became of each read (alignerGreedy.cpp:35-57) ---------------------------------------------------------------
            // 0 = aligned, 1 = no anchor on this strand and none tried before (++noOverlapRead), 2 = not aligned (both strands done),
            // 3 = goes on in a later pass with `nst`, 4 = general kernel (N in the read, path too long for the registers)
            uint32_t o_now = 4, n_now = 0;
            {
                const uint32_t npos_g = (L >= K1 ? L - K1 + 1 : 0);
                const uint32_t npos_e = (!prm.effort && npos_g > 1) ? 1u : npos_g;
                if (bad) o_now = 4;
                else if (a_rec != BGR_NONE && !failed) o_now = 0;
                else if (a_rec == BGR_NONE && tried == 0) o_now = 1;
                else {
                    // the strand's anchors are used up when `effort` of them have been tried or the scan has passed the last position
                    const uint32_t resume = a_pos + 1;  // (a_pos = the anchor tried last; unused when the scan found none)
                    const uint32_t used_up = (a_rec == BGR_NONE || tried >= eff || resume >= npos_e) ? 1u : 0u;
                    if (!used_up) { o_now = 3; n_now = (rc << 31) | (tried << G4_ST_TRIED_SHIFT) | resume; }
                    else if (!rc) { o_now = 3; n_now = G4_ST_RC; }  // the reverse complement, from its first position
                    else o_now = 2;
                    if (tried > 0x7FFu) o_now = 4;
                }
            }
            if (act) { outcome = o_now; nst = n_now; rc_out = rc; }
            // A launch over a LIST goes straight on to the reverse complement of the reads whose forward anchors are used up now
            // (most of such a launch's reads: a second round is as densely packed as the first); the launch over all reads leaves
            // them to the next one (7 % of its reads: three of four groups would sit idle).
            const uint32_t again = (LIST && round == 0 && act && o_now == 3 && n_now == G4_ST_RC) ? 1u : 0u;
            if (!LIST) break;
            if (!__any(again != 0)) break;
            act = again; rc = 1; tried = 0; s_from = 0;
        }
        if (outcome == 3 && io.g4_last) outcome = 4;  // no further pass of this kernel: the general kernel maps the read from scratch

        // ---- publish: reverse(left) ++ right into the arena ----------------------------------------------------------------------
        const uint32_t aligned = outcome == 0 ? 1u : 0u;
        const uint32_t p_n = aligned ? nl + nr : 0;
        const uint32_t n0 = rl32(p_n, 0), n1 = rl32(p_n, 16), n2 = rl32(p_n, 32), n3 = rl32(p_n, 48);
        const uint32_t tot = n0 + n1 + n2 + n3;
        if (tot > chunk_end - chunk_pos) {  // one global atomic per ~50 reads (see publish_path)
            const uint32_t want = tot > io.arena_chunk ? tot : io.arena_chunk;
            uint32_t got = 0;
            if (lane == 0) got = atomicAdd(io.cursor, want);
            chunk_pos = rl32(got, 0);
            chunk_end = chunk_pos + want;
        }
        const uint32_t gbase = chunk_pos + (grp > 0 ? n0 : 0u) + (grp > 1 ? n1 : 0u) + (grp > 2 ? n2 : 0u);
        const bool room = chunk_pos + tot <= io.arena_cap;
        chunk_pos += tot;
#pragma unroll
        for (uint32_t jj = 0; jj < 2; ++jj) {
            const uint32_t j = sub + 16 * jj;
            const uint32_t vl = lane_get((uint32_t)pl, ((uint32_t)lane & 48u) | ((nl - 1 - j) & 15u));
            const uint32_t vr = lane_get((uint32_t)pr, ((uint32_t)lane & 48u) | ((j - nl) & 15u));
            if (j < p_n && room) io.arena[gbase + j] = (int32_t)(j < nl ? vl : vr);
        }
        if (!room && lane == 0 && tot) io.cursor[1] = 1;  // overflow: reported by the host as an error
        if (sub == 0 && have) {
            if (outcome <= 2) {
                const uint32_t code = (outcome == 0 ? BGR_ST_ALIGNED : outcome == 1 ? BGR_ST_NOANCHOR : BGR_ST_FAILED) | (rc_out ? BGR_ST_RC : 0u);
                io.results[r] = make_uint2(aligned ? gbase : 0u, p_n | (code << 24));
            } else if (outcome == 3) {
                io.g4_state[r] = nst;
            } else {
                io.gen_list[atomicAdd(io.cursor + io.gen_ctr, 1u)] = r;
            }
        }
        {   // the reads that go on in the next pass: appended to this wave's slice of the list
            const u64 lm = __ballot(sub == 0 && have && outcome == 3);
            if (lm) {
                const uint32_t cnt = (uint32_t)__popcll(lm);
                if (cnt > lst_end - lst_pos) {  // what is left of the old slice becomes holes
                    for (uint32_t j = lst_pos + (uint32_t)lane; j < lst_end; j += 64) io.ovf_list[j] = BGR_NONE;
                    uint32_t got = 0;
                    if (lane == 0) got = atomicAdd(io.cursor + io.ovf_ctr, io.list_chunk);
                    lst_pos = rl32(got, 0);
                    lst_end = lst_pos + io.list_chunk;
                }
                if (sub == 0 && have && outcome == 3) io.ovf_list[lst_pos + (uint32_t)__popcll(lm & ((1ULL << lane) - 1))] = r;
                lst_pos += cnt;
            }
        }
        c_al += (uint32_t)__popcll(__ballot(sub == 0 && have && outcome == 0));
        c_noov += (uint32_t)__popcll(__ballot(sub == 0 && have && outcome == 1));
        c_na += (uint32_t)__popcll(__ballot(sub == 0 && have && outcome == 2));
        wave_sync();
    }
    for (uint32_t j = lst_pos + (uint32_t)lane; j < lst_end; j += 64) io.ovf_list[j] = BGR_NONE;
    if (lane == 0 && (c_al | c_noov | c_na)) {  // aligner.h:68 counters: [0] readNumber [1] noOverlapRead [2] alignedRead [3] notAligned
        unsigned long long* counters = reinterpret_cast<unsigned long long*>(io.cursor + 16);
        atomicAdd(&counters[0], (unsigned long long)(c_al + c_noov + c_na));
        if (c_noov) atomicAdd(&counters[1], (unsigned long long)c_noov);
        if (c_al) atomicAdd(&counters[2], (unsigned long long)c_al);
        if (c_na) atomicAdd(&counters[3], (unsigned long long)c_na);
    }
}

// ======================================== anchors mode (-G) ==========================================
// alignReadGreedyAnchors (alignerGreedy.cpp:60-164) over getNAnchors (aligner.cpp:381-405).  The k-mer anchors come
// from BooPHF's exact structure (graph_layout.h "anchors index"): the reference takes whatever lookup() answers,
// key or not, so the lookup below is boomphf::mphf::lookup (BooPHF.h:783-818) instruction for instruction in its
// arithmetic: hash64 with the two seeds, xorshift128+ for the later levels, `% domain`, the 512-bit rank blocks.
struct AncView {
    const u64* bits;
    const u64* ranks;
    const u64* fin;
    const u64* pos;
    u64 n_final, last_rank;
    u64 lv_domain, lv_word_base, lv_rank_base, lv_magic;  // lane l < n_active holds level l
    uint32_t n_active;                                    // levels that hold set bits (the rest cannot answer)
};

__device__ __forceinline__ AncView anc_view(const BgrDeviceGraph& g, int lane) {
    AncView a;
    const BgrBlobHeader* h = g.hdr;
    const char* base = reinterpret_cast<const char*>(h);
    a.bits = reinterpret_cast<const u64*>(base + h->off_anc_bits);
    a.ranks = reinterpret_cast<const u64*>(base + h->off_anc_ranks);
    a.fin = reinterpret_cast<const u64*>(base + h->off_anc_final);
    a.pos = reinterpret_cast<const u64*>(base + h->off_anc_pos);
    a.n_final = h->anc_n_final;
    a.last_rank = h->anc_last_rank;
    a.n_active = (uint32_t)h->anc_active_levels;
    const int l = lane < (int)a.n_active ? lane : 0;
    a.lv_domain = h->anc_levels[l].domain;
    a.lv_word_base = h->anc_levels[l].word_base;
    a.lv_rank_base = h->anc_levels[l].rank_base;
    a.lv_magic = h->anc_levels[l].magic;
    return a;
}

// wave-uniform key -> index or ~0.  Lane l probes level l; the first level whose bit is set answers with its rank.
__device__ __forceinline__ u64 anc_lookup(const AncView& a, u64 key, int lane) {
    u64 s0 = bgr_boo_hash64(key, BGR_BOO_SEED0), s1 = bgr_boo_hash64(key, BGR_BOO_SEED1);
    u64 hv = lane == 0 ? s0 : s1;
    for (int i = 2; i < (int)a.n_active; ++i) {  // BooPHF.h:336-356: the level hashes are a sequence, walked in step
        const u64 v = bgr_boo_next(&s0, &s1);
        if (lane == i) hv = v;
    }
    bool hit = false;
    u64 pos = 0;
    if (lane < (int)a.n_active) {
        pos = bgr_mod_magic(hv, a.lv_domain, a.lv_magic);
        hit = (a.bits[a.lv_word_base + (pos >> 6)] >> (pos & 63)) & 1;
    }
    const u64 mask = __ballot(hit);
    if (mask) {  // BooPHF.h:609-622 rank: sample of the 512-bit block + popcount of the words before the bit
        const int f = __ffsll((long long)mask) - 1;
        const u64 fpos = rl64(pos, f), wb = rl64(a.lv_word_base, f), rb = rl64(a.lv_rank_base, f);
        const u64 widx = fpos >> 6, blk = fpos >> 9;
        uint32_t cnt = 0;
        if (lane < 8) {
            const u64 wi = blk * 8 + (u64)lane;
            if (wi < widx) cnt = (uint32_t)__popcll(a.bits[wb + wi]);
            else if (wi == widx) cnt = (uint32_t)__popcll(a.bits[wb + wi] & ((1ULL << (fpos & 63)) - 1));
        }
        cnt = rl32(row16_sum(cnt), 0);
        return a.ranks[rb + blk] + cnt;
    }
    u64 lo = 0, hi = a.n_final;  // what 24 levels could not place (repeated k-mers): exact, sorted {key, index}
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if (a.fin[2 * mid] < key) lo = mid + 1; else hi = mid;
    }
    if (lo < a.n_final && a.fin[2 * lo] == key) return a.last_rank + a.fin[2 * lo + 1];
    return ~0ULL;
}

// Hamming distance of read[rb, rb+n) against the packed store from base `ub` after word `fw` (one strand of one
// unitig), whole wave: lane l takes bases [32l, 32l+32) of every 2048.
__device__ __forceinline__ uint32_t ham_span(const BgrDeviceGraph& g, const u64* CMP, const u64* NM, bool useN, uint32_t fw, uint32_t ub,
                                             uint32_t rb, uint32_t n, int lane) {
    uint32_t cnt = 0;
    for (uint32_t b = (uint32_t)lane * 32; b < n; b += 2048) cnt += ham_chunk(g, CMP, NM, useN, fw, ub + b, rb + b, n - b);
    cnt = row16_sum(cnt);
    return rl32(cnt, 0) + rl32(cnt, 16) + rl32(cnt, 32) + rl32(cnt, 48);
}

__global__ void __launch_bounds__(1024, BGR_ANC_OCC) bgr_align_anchors_kernel(BgrDeviceGraph g, BatchIO io, KernelParams prm) {
    extern __shared__ u64 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const uint32_t W = io.words_per_read;
    const uint32_t K = g.k, K1 = g.k - 1;
    const uint32_t per_wave_words = 4 * W + io.path_cap / 2;
    u64* FW3 = lds + 64 + (u64)wave * per_wave_words;  // same per-wave layout as the greedy kernel, no MPHF staging
    u64* FWQ = FW3 + W;
    u64* RCW = FWQ + W;
    u64* NM = RCW + W;
    int32_t* PATH = reinterpret_cast<int32_t*>(NM + W);
    const AncView av = anc_view(g, lane);
    const u64 km1_mask = (1ULL << (2 * K1)) - 1;  // offsetUpdate - 1 (aligner.h:101-102): update() keeps k-1 digits

    uint32_t c_reads = 0, c_noov = 0, c_al = 0, c_na = 0;
    uint32_t chunk_pos = 0, chunk_end = 0;
    const uint32_t effort = prm.effort ? prm.effort : 1;  // getNAnchors(read, 0) still takes a hit at position 0

    for (uint32_t r = blockIdx.x * waves + wave; r < io.n_reads; r += gridDim.x * waves) {
        const u64 off = io.read_offs[r];
        const uint32_t L = (uint32_t)(io.read_offs[r + 1] - off);
        const bool hasN = load_packed(io, r, off, L, W, FW3, NM, lane);
        bool derived = false;
        uint32_t status = BGR_ST_NOANCHOR, p_lo = 0, p_n = 0;
        const uint32_t dk = L < K ? L : K;                       // read.substr(0, k) of a shorter read is the whole read
        const uint32_t last_i = !prm.effort ? 0 : (L > K ? L - K : 0);  // the loop leaves after position i when i + k >= |read|
        for (int pass = 0; pass < 2 && L > 0; ++pass) {
            if (pass == 1 && !derived) { derive_streams(L, W, K1, FW3, FWQ, RCW, NM, lane); derived = true; }
            const u64* S = pass ? RCW : FW3;        // the characters of this pass's read (reverseComplements: N -> 'A')
            const bool useN = (pass == 0) && hasN;
            // getNAnchors (aligner.cpp:381-405): the first k-mer by str2num, then the (k-1)-mer rolling updates applied to it
            u64 num = lds_win32(S, 0) >> (64 - 2 * dk);
            u64 rcnum = rcb_fast(num, K);
            uint32_t tried = 0;
            bool done = false;
            for (uint32_t i = 0;; ++i) {
                const u64 rep = num < rcnum ? num : rcnum;
                const u64 idx = anc_lookup(av, rep, lane);
                if (idx != ~0ULL) {
                    ++tried;
                    // ---- alignReadGreedyAnchors loop body for this anchor (alignerGreedy.cpp:68-161) ------------
                    const u64 pv = av.pos[idx];
                    const uint32_t un = (uint32_t)(pv >> 32);
                    uint32_t pU = (uint32_t)pv;
                    const uint32_t pR = i;
                    const BgrUnitigMeta mt = g.meta[un];
                    const uint32_t len = mt.len;
                    if (len >= K) {  // :72-75 (an index nobody wrote holds unitig 0, the empty string)
                        const uint32_t fw = (uint32_t)(mt.F >> 5);
                        uint32_t fo = (uint32_t)(mt.F & 31);
                        const u64 ukm = seq_win32(g.seq, fw, fo + pU) >> (64 - 2 * K);
                        const uint32_t rdk = L - pR < K ? L - pR : K;
                        const u64 rkm = lds_win32(S, pR) >> (64 - 2 * rdk);
                        const bool returned = ukm != rkm;  // :76-83: any difference means "take the reverse complement"
                        if (returned) { fo += len; pU = len - K - pU; }
                        const int32_t uid = returned ? -(int32_t)un : (int32_t)un;
                        // the oriented unitig's end (k-1)-mers as neighbour records (what str2num + getEnd/getBegin find)
                        const uint32_t rec_b = returned ? mt.rec_end : mt.rec_beg, rec_e = returned ? mt.rec_beg : mt.rec_end;
                        const bool can_b = (mt.flags & (returned ? BGR_META_CANON_RCEND : BGR_META_CANON_BEG)) != 0;
                        const bool can_e = (mt.flags & (returned ? BGR_META_CANON_RCBEG : BGR_META_CANON_END)) != 0;
                        const uint32_t m = prm.max_mismatch;
                        if (pR >= pU) {
                            const uint32_t start = pR - pU;  // read position of the unitig's first base
                            if (L - pR >= len - pU) {
                                // CASE 1: unitig inside the read (:87-110)
                                const uint32_t errors = ham_span(g, S, NM, useN, fw, fo, start, len, lane);
                                if (errors <= m) {
                                    uint32_t budget = m - errors, nl = 0, nr = 0;
                                    const uint32_t mid = start + 2;
                                    if (walk_left(g, S, NM, useN, L, K1, rec_b, can_b, start, &budget, PATH, mid, &nl, lane)) {
                                        if (lane == 0) PATH[mid] = uid;
                                        if (walk_right(g, S, NM, useN, L, K1, rec_e, can_e, start + len - K1, &budget, PATH, mid + 1, &nr, lane)) {
                                            p_lo = mid - nl; p_n = nl + 1 + nr; done = true;
                                        }
                                    }
                                }
                            } else {
                                // CASE 2: the unitig runs past the read's end (:111-130)
                                const uint32_t errors = ham_span(g, S, NM, useN, fw, fo, start, L - start, lane);
                                if (errors <= m) {
                                    uint32_t budget = m - errors, nl = 0;
                                    const uint32_t mid = start + 2;
                                    if (walk_left(g, S, NM, useN, L, K1, rec_b, can_b, start, &budget, PATH, mid, &nl, lane)) {
                                        if (lane == 0) PATH[mid] = uid;
                                        p_lo = mid - nl; p_n = nl + 1; done = true;
                                    }
                                }
                            }
                        } else {
                            const uint32_t uoff = pU - pR;  // unitig position of the read's first base
                            if (L - pR >= len - pU) {
                                // CASE 3: the read starts inside the unitig and runs past its end (:133-148)
                                const uint32_t errors = ham_span(g, S, NM, useN, fw, fo + uoff, 0, len - uoff, lane);
                                if (errors <= m) {
                                    uint32_t budget = m - errors, nr = 0;
                                    if (lane == 0) { PATH[0] = (int32_t)uoff; PATH[1] = uid; }
                                    if (walk_right(g, S, NM, useN, L, K1, rec_e, can_e, len - uoff - K1, &budget, PATH, 2, &nr, lane)) {
                                        p_lo = 0; p_n = 2 + nr; done = true;
                                    }
                                }
                            } else {
                                // CASE 4: read inside the unitig (:149-160)
                                const uint32_t errors = ham_span(g, S, NM, useN, fw, fo + uoff, 0, L, lane);
                                if (errors <= m) {
                                    if (lane == 0) { PATH[0] = (int32_t)uoff; PATH[1] = uid; }
                                    p_lo = 0; p_n = 2; done = true;
                                }
                            }
                        }
                    }
                    if (done || tried >= effort) break;
                }
                if (i >= last_i) break;
                // update() / updateRC() (aligner.cpp:305-315) with read[i + k]
                const uint32_t d = (uint32_t)(lds_win32(S, i + K) >> 62);
                const bool isn = useN && (lds_win32(NM, i + K) >> 62) != 0;
                const u64 fwd_code = isn ? 0 : d, rc_code = isn ? 0 : 3 - d;  // nuc2int / nuc2intrc (utils.cpp:132-151)
                num = ((num << 2) + fwd_code) & km1_mask;
                rcnum = (rcnum >> 2) + (rc_code << (2 * K - 4));
            }
            if (done) { status = BGR_ST_ALIGNED | (pass ? BGR_ST_RC : 0); break; }
            if (tried == 0) { status = BGR_ST_NOANCHOR | (pass ? BGR_ST_RC : 0); break; }  // ++noOverlapRead, no retry
            status = BGR_ST_FAILED | BGR_ST_RC;  // every anchor failed: once more on the reverse complement (:162)
        }
        wave_sync();
        uint32_t abase = 0;
        if ((status & BGR_ST_MASK) == BGR_ST_ALIGNED) abase = publish_path(io, PATH, p_lo, p_n, &chunk_pos, &chunk_end, lane);
        else p_n = 0;
        if (lane == 0) io.results[r] = make_uint2(abase, p_n | (status << 24));
        ++c_reads;
        c_noov += (status & BGR_ST_MASK) == BGR_ST_NOANCHOR;
        c_al += (status & BGR_ST_MASK) == BGR_ST_ALIGNED;
        c_na += (status & BGR_ST_MASK) == BGR_ST_FAILED;
        wave_sync();
    }
    if (lane == 0 && c_reads) {
        unsigned long long* counters = reinterpret_cast<unsigned long long*>(io.cursor + 16);
        atomicAdd(&counters[0], (unsigned long long)c_reads);
        if (c_noov) atomicAdd(&counters[1], (unsigned long long)c_noov);
        if (c_al) atomicAdd(&counters[2], (unsigned long long)c_al);
        if (c_na) atomicAdd(&counters[3], (unsigned long long)c_na);
    }
}

// alignReadExhaustive (alignerExhaustive.cpp:35-58): every read position is an anchor candidate
// (getListOverlap, aligner.cpp:318-342, keeps them all); per anchor the best left walk with budget m, then
// the best right walk with what is left; no reverse-complement retry.  Only position 0 and positions whose
// (k-1)-mer is an overlap of the graph can succeed (anywhere else getEnd() is empty), so the position scan
// is the same lane-parallel membership test as in the greedy kernel.
// DEEP (pass 2): the search state (OUT | CUR | BEST | frames) of every wave lives in HBM (io.deep_scratch) instead
// of LDS, sized for the worst case, so neither the depth of the search nor the read length is bounded by LDS.
template <bool STAGE, bool DEEP>
__global__ void __launch_bounds__(1024, BGR_EXH_OCC) bgr_align_exhaustive_kernel(BgrDeviceGraph g, BatchIO io, KernelParams prm) {
    extern __shared__ u64 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const uint32_t W = io.words_per_read;
    const uint32_t K1 = g.k - 1;
    uint2* LV;
    uint32_t mphf_words;
    const uint32_t* units = block_prologue<STAGE>(g, lds, &LV, &mphf_words);
    // per wave: FW3 | FWQ | RCW | NM | OUT | CUR | BEST | frames   (the last four in HBM when DEEP)
    const uint32_t per_wave_words = DEEP ? 4 * W : 4 * W + 3 * (io.path_cap / 2) + (io.frames_per_wave * FR_WORDS) / 2;
    u64* FW3 = lds + 64 + mphf_words + (u64)wave * per_wave_words;
    u64* FWQ = FW3 + W;
    u64* RCW = FWQ + W;
    u64* NM = RCW + W;
    int32_t* OUT = DEEP ? reinterpret_cast<int32_t*>(io.deep_scratch + (u64)(blockIdx.x * waves + wave) * io.deep_stride)
                        : reinterpret_cast<int32_t*>(NM + W);
    int32_t* CUR = OUT + io.path_cap;
    int32_t* BEST = CUR + io.path_cap;
    uint32_t* FR = reinterpret_cast<uint32_t*>(BEST + io.path_cap);

    uint32_t c_reads = 0, c_al = 0, c_na = 0;
    unsigned long long c_ov = 0;
    uint32_t chunk_pos = 0, chunk_end = 0;
    const uint32_t m = prm.max_mismatch;

    // pass 2 maps only the reads that pass 1 listed as needing a deeper stack (count left in cursor[subset_ctr] by the pass before)
    const uint32_t total = io.subset ? io.cursor[io.subset_ctr] : io.n_reads;
    for (uint32_t it = blockIdx.x * waves + wave; it < total; it += gridDim.x * waves) {
        const uint32_t r = io.subset ? io.subset[it] : it;
        const u64 off = io.read_offs[r];
        const uint32_t L = (uint32_t)(io.read_offs[r + 1] - off);
        const bool hasN = load_packed(io, r, off, L, W, FW3, NM, lane);
        if (hasN) derive_streams(L, W, K1, FW3, FWQ, RCW, NM, lane);
        const u64* ROLL = hasN ? FWQ : FW3;  // the rolling `num` stream
        uint32_t p_n = 0;
        const uint32_t npos = L >= K1 ? L - K1 + 1 : 0;
        bool done = false, overflow = false;
        for (uint32_t base = 0; base < npos && !done && !overflow; base += 64) {
            const uint32_t i = base + lane;
            const bool valid = i < npos;
            u64 num = 0;
            if (valid) num = lds_win32(ROLL, i) >> (64 - 2 * K1);  // the rolling `num` (aligner.cpp:321,334)
            const u64 rc = rcb_fast(num, K1);                  // getBegin/getEnd use rcb(num) (aligner.cpp:149,211)
            const uint32_t idx = find_key<!STAGE>(g, LV, units, num < rc ? num : rc, valid);
            u64 mask = __ballot(idx != BGR_NONE);
            if (base == 0) mask |= 1;  // position 0: the left side is trivially [0] whatever the k-mer
            while (mask) {
                const int src = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const uint32_t a_rec = rl32(idx, src), a_pos = base + (uint32_t)src;
                const u64 a_num = rl64(lds_win32(ROLL, a_pos) >> (64 - 2 * K1), 0);  // re-read (uniform) instead of keeping `num` live across the search
                const bool a_canon = a_num <= rcb_fast(a_num, K1);
                uint32_t nl = 0, nr = 0, eb = 0;
                if (a_pos == 0) {  // checkBeginExhaustive at position 0 is [0] at no cost (alignerExhaustive.cpp:159): no search
                    if (lane == 0) OUT[0] = 0;
                    nl = 1;
                } else {
                    eb = exh_search<0>(g, FW3, NM, hasN, L, K1, a_rec, a_canon, a_pos, m, false, FR, io.frames_per_wave, CUR, BEST, &nl, lane);
                    if (eb == EXH_OVERFLOW) { overflow = true; break; }
                    if (eb > m) continue;
                    for (uint32_t j = lane; j < nl; j += 64) OUT[j] = BEST[j];
                }
                wave_sync();
                // position 0 is tried whatever its (k-1)-mer: when that is no overlap of the graph getBegin() is empty, and only an
                // empty right side or -i can make the anchor succeed (alignerExhaustive.cpp:206-221): no search either
                if (a_rec == BGR_NONE && !prm.partial && L - a_pos - K1 != 0) continue;
                const uint32_t ee = exh_search<1>(g, FW3, NM, hasN, L, K1, a_rec, a_canon, a_pos, m - eb, prm.partial != 0, FR, io.frames_per_wave, CUR, BEST, &nr, lane);
                if (ee == EXH_OVERFLOW) { overflow = true; break; }
                if (ee > m - eb) continue;
                for (uint32_t j = lane; j < nr; j += 64) OUT[nl + j] = BEST[j];
                p_n = nl + nr;
                done = true;
                break;
            }
        }
        wave_sync();
        if (overflow) {  // leave this read to pass 2 (full-depth stack); nothing is written or counted for it here
            if (lane == 0) io.ovf_list[atomicAdd(io.cursor + io.ovf_ctr, 1u)] = r;
            continue;
        }
        c_ov += npos;
        uint32_t abase = 0;
        if (done) abase = publish_path(io, OUT, 0, p_n, &chunk_pos, &chunk_end, lane);
        if (lane == 0) io.results[r] = make_uint2(abase, p_n | ((uint32_t)(done ? BGR_ST_ALIGNED : BGR_ST_FAILED) << 24));
        ++c_reads;
        c_al += done ? 1 : 0;
        c_na += done ? 0 : 1;
        wave_sync();
    }
    if (lane == 0 && c_reads) {
        unsigned long long* counters = reinterpret_cast<unsigned long long*>(io.cursor + 16);
        atomicAdd(&counters[0], (unsigned long long)c_reads);
        if (c_al) atomicAdd(&counters[2], (unsigned long long)c_al);
        if (c_na) atomicAdd(&counters[3], (unsigned long long)c_na);
        atomicAdd(&counters[4], c_ov);
    }
}

// ============================== exhaustive, four reads per wavefront ===================================
// The level search (exh_dp) gives a wave ONE read, and after de-duplication a level of the walk rarely holds more than one
// node: 4 candidate slots x 4 chunk lanes = 16 of the 64 lanes work.  Here a wave takes four reads, 16 lanes each, and runs
// their searches side by side, for the shape nearly every read has: every level has exactly ONE node (all candidates that go
// on lead to the same (record, position, strand): bubbles that close again), at most X4_LEVELS levels per side, no N, and
// the first anchor that can succeed does.  Per side: a forward sweep scores the node's <= 4 candidates per level (kept:
// id, offset, mismatches, fits, alive), a backward sweep settles cost(level) = first minimum over the slots of mismatches
// [+ cost(level + 1) when the walk goes on] -- the value and the choice of the reference's recursion
// (alignerExhaustive.cpp:61-259; see exh_dp) -- and the walk is read off.  Anything else (a level with two nodes, a failing
// first anchor, N, -i) is listed for bgr_align_exhaustive_dp_kernel / the depth-first passes, which map it from scratch.
#ifndef BGR_X4_OCC
#define BGR_X4_OCC 6
#endif
#define X4_LEVELS 16
#define X4_LV_WORDS 16  // per level: [0..3] sid, [4..7] aux (the path int a walk ending there emits), [8..11] miss | fits<<16 | alive<<17, [12] node flags, [13] chosen slot
#define X4_END 1u
#define X4_INF 0xFFFFu

// One side of the search for up to four reads (act = the group takes part).  DIR 0: left of the anchor (exL), DIR 1: right
// (exR).  On return, for the groups that took part: *cost = best total (X4_INF: none within the budget; the caller compares
// with its budget), *n_out ints written to OUTG[o_off ...] in output order, *fb = the search left the shape this kernel
// handles (the read goes on the list).
template <int DIR>
__device__ __forceinline__ void x4_search(const BgrDeviceGraph& g, const u64* FW, uint32_t L, uint32_t K1, uint32_t act, uint32_t a_rec, uint32_t a_canon,
                                          uint32_t a_pos, uint32_t budget, uint32_t* LVT, int32_t* OUTG, uint32_t o_off, int lane, uint32_t* cost_o,
                                          uint32_t* n_out, uint32_t* fb_o) {
    const uint32_t c = ((uint32_t)lane >> 2) & 3u, q = (uint32_t)lane & 3u, sub = (uint32_t)lane & 15u;
    uint32_t fwd = act, fb = 0, lvl = 0, nlev = 0, prefix = 0;
    uint32_t pos = a_pos, rec = a_rec, canon = a_canon;
    // ---- forward: one node per level ----
    for (;;) {
        if (fwd && lvl >= X4_LEVELS) { fb = 1; fwd = 0; }
        const uint32_t end_here = (fwd && ((DIR == 0) ? (pos == 0) : (L - pos - K1 == 0))) ? 1u : 0u;
        if (end_here) {  // left: the read's first base is reached; right: nothing is left of the read
            if (sub == 0) LVT[lvl * X4_LV_WORDS + 12] = X4_END;
            nlev = lvl + 1;
            fwd = 0;
        }
        if (!__any(fwd != 0)) break;
        const uint32_t useR = ((DIR == 0) ? canon : (canon ^ 1u));
        uint4 sl = make_uint4(0, 0, 0, 0), m0 = make_uint4(0, 0, 0, 0);
        if (fwd && rec != G4_REC_MASK) {
            const uint4* sp = reinterpret_cast<const uint4*>(g.recs) + (size_t)(rec * 8u + useR * 4u + c) * 2;
            sl = sp[0];
            m0 = sp[1];
        }
        const uint32_t id = sl.x & BGR_SLOT_ID_MASK;
        const u64 zmask = __ballot(id == 0);
        const uint32_t zb = (uint32_t)(zmask >> ((uint32_t)lane & 48u)) & 0x1111u;
        const uint32_t first_zero = zb ? (uint32_t)(__ffs((int)zb) - 1) >> 2 : 4u;  // the reference stops at the first empty slot
        const uint32_t valid = c < first_zero ? 1u : 0u;
        const uint32_t fwdu = (sl.x & (canon ? BGR_SLOT_F0 : BGR_SLOT_F1)) ? 1u : 0u;
        const uint32_t len = sl.y;
        const uint32_t fw = sl.z, fo = sl.w + (fwdu ? 0u : len);
        const uint32_t ext = len - K1;
        uint32_t fits, n, ustart, rstart, nrec, cbit, aux, npos;
        if (DIR == 0) {
            fits = ext >= pos ? 1u : 0u;
            n = fits ? pos : ext;
            ustart = fits ? ext - pos : 0;
            rstart = fits ? 0 : pos - ext;
            nrec = fwdu ? m0.y : m0.z;
            cbit = fwdu ? BGR_META_CANON_BEG : BGR_META_CANON_RCEND;
            aux = ext - pos;   // offset in the last unitig (:126,:175)
            npos = pos - ext;
        } else {
            const uint32_t rl = L - pos - K1;
            fits = ext >= rl ? 1u : 0u;
            n = fits ? rl : ext;
            ustart = K1;
            rstart = pos + K1;
            nrec = fwdu ? m0.z : m0.y;
            cbit = fwdu ? BGR_META_CANON_END : BGR_META_CANON_RCBEG;
            aux = L - pos;     // |readLeft| + k-1 (:99,:231)
            npos = pos + ext;
        }
        if (!valid) n = 0;
        uint32_t cnt = 0;
        for (uint32_t b = q * 32; __any(b < n); b += 128)
            if (b < n) cnt += ham_chunk(g, FW, nullptr, false, fw, fo + ustart + b, rstart + b, n - b);
        cnt += quad_xor1(cnt);
        cnt += quad_xor2(cnt);
        const uint32_t miss = cnt > 0xFFFFu ? 0xFFFFu : cnt;
        const uint32_t ptotal = prefix + miss;
        const uint32_t alive = (valid && ptotal <= budget) ? 1u : 0u;  // a walk through here costs at least this much
        const uint32_t need = (alive && !fits) ? 1u : 0u;
        if (fwd && q == 0) {
            uint32_t* R = LVT + lvl * X4_LV_WORDS;
            R[c] = fwdu ? id : 0u - id;
            R[4 + c] = aux;
            R[8 + c] = miss | (fits << 16) | (alive << 17);
            if (c == 0) R[12] = 0;
        }
        // the candidates that go on must all reach the same node
        const u64 nmask = __ballot(need && q == 0 && fwd);
        const uint32_t nb = (uint32_t)(nmask >> ((uint32_t)lane & 48u)) & 0x1111u;
        const uint32_t src = ((uint32_t)lane & 48u) | (nb ? (uint32_t)(__ffs((int)nb) - 1) : 0u);
        const uint32_t kpk = nrec | ((m0.x & cbit) ? G4_CANON : 0u);
        const uint32_t k_rec = lane_get(kpk, src), k_pos = lane_get(npos, src);
        const u64 dmask = __ballot(need && fwd && (kpk != k_rec || npos != k_pos));
        uint32_t pmin = need ? ptotal : 0xFFFFFFFFu;
        uint32_t o = row_ror4(pmin);
        pmin = o < pmin ? o : pmin;
        o = row_ror8(pmin);
        pmin = o < pmin ? o : pmin;
        if (fwd) {
            nlev = lvl + 1;
            if ((uint32_t)(dmask >> ((uint32_t)lane & 48u)) & 0xFFFFu) { fb = 1; fwd = 0; }  // a level with two nodes
            else if (!nb) fwd = 0;                                                      // every candidate ends here or is too dear
            else { prefix = pmin; pos = k_pos; rec = k_rec & G4_REC_MASK; canon = (k_rec >> 28) & 1u; ++lvl; }
        }
    }
    wave_sync();
    // ---- backward: cost of every level, first slot on ties ----
    const uint32_t ok = (act && !fb) ? 1u : 0u;
    const uint32_t n0 = rl32(ok ? nlev : 0u, 0), n1 = rl32(ok ? nlev : 0u, 16), n2 = rl32(ok ? nlev : 0u, 32), n3 = rl32(ok ? nlev : 0u, 48);
    const uint32_t maxl = max(max(n0, n1), max(n2, n3));
    uint32_t cnext = X4_INF;
    for (int l = (int)maxl - 1; l >= 0; --l) {
        const uint32_t in = (ok && (uint32_t)l < nlev) ? 1u : 0u;
        uint32_t key = 0xFFFFFFFFu, flags = 0;
        if (in) {
            const uint32_t* R = LVT + (uint32_t)l * X4_LV_WORDS;
            flags = R[12];
            if (sub < 4) {
                const uint32_t pk = R[8 + sub];
                uint32_t total = X4_INF;
                if (pk & (1u << 17)) {
                    total = (pk & 0xFFFFu) + ((pk & (1u << 16)) ? 0u : cnext);
                    if (total > X4_INF) total = X4_INF;
                }
                key = total << 2 | sub;
            }
        }
        uint32_t o = quad_xor1(key);
        key = o < key ? o : key;
        o = quad_xor2(key);
        key = o < key ? o : key;
        key = lane_get(key, (uint32_t)lane & 48u);
        if (in) {
            cnext = (flags & X4_END) ? 0u : (key >> 2);
            if (sub == 0) LVT[(uint32_t)l * X4_LV_WORDS + 13] = key & 3u;
        }
    }
    wave_sync();
    // ---- read the walk off: level j's chosen slot, down to the first level that ends the walk ----
    uint32_t endj = 0, sid = 0, auxv = 0, isend = 0;
    if (ok && sub < nlev) {
        const uint32_t* R = LVT + sub * X4_LV_WORDS;
        const uint32_t a = R[13];
        isend = R[12] & X4_END;
        const uint32_t pk = R[8 + a];
        endj = (isend || (pk & (1u << 16))) ? 1u : 0u;
        sid = R[a];
        auxv = R[4 + a];
    }
    const u64 emask = __ballot(endj != 0);
    const uint32_t eb16 = (uint32_t)(emask >> ((uint32_t)lane & 48u)) & 0xFFFFu;
    const uint32_t d = eb16 ? (uint32_t)(__ffs((int)eb16) - 1) : 0u;           // depth of the level that ends the walk
    const uint32_t d_end = lane_get(isend, ((uint32_t)lane & 48u) | d);       // ... by reaching the read's end (no unitig taken there)
    uint32_t n = 0;
    const uint32_t good = (ok && cnext <= budget && eb16) ? 1u : 0u;
    if (good) {
        int32_t* O = OUTG + o_off;
        if (DIR == 0) {
            // left: checkBeginExhaustive pushes 0 only at the top (:159 vs :112); else [offset, farthest unitig, ..., nearest]
            if (d_end) { n = d == 0 ? 1u : d; if (d == 0) { if (sub == 0) O[0] = 0; } else if (sub < d) O[d - 1 - sub] = (int32_t)sid; }
            else { n = d + 2; if (sub == d) { O[0] = (int32_t)auxv; O[1] = (int32_t)sid; } else if (sub < d) O[1 + (d - sub)] = (int32_t)sid; }
        } else {
            // right: every depth pushes 0 at the read's end (:64,:210); else [nearest ... farthest unitig, end offset]
            if (d_end) { n = d + 1; if (sub < d) O[sub] = (int32_t)sid; if (sub == d) O[d] = 0; }
            else { n = d + 2; if (sub <= d) O[sub] = (int32_t)sid; if (sub == d) O[d + 1] = (int32_t)auxv; }
        }
    }
    wave_sync();
    *cost_o = good ? cnext : X4_INF;
    *n_out = n;
    *fb_o = fb;
}

template <bool STAGE>
__global__ void __launch_bounds__(1024, BGR_X4_OCC) bgr_align_exhaustive4_kernel(BgrDeviceGraph g, BatchIO io, KernelParams prm) {
    extern __shared__ u64 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const uint32_t W = io.words_per_read;  // <= 16 (checked by the host)
    const uint32_t K1 = g.k - 1;
    uint2* LV;
    uint32_t mphf_words;
    const uint32_t* units = block_prologue<STAGE>(g, lds, &LV, &mphf_words);
    // per wave: 4 x { read words W | level table X4_LEVELS x X4_LV_WORDS u32 | out ints 2 x (X4_LEVELS + 2) }
    const uint32_t out_ints = 2 * (X4_LEVELS + 2);
    const uint32_t grp_words = W + (X4_LEVELS * X4_LV_WORDS + out_ints + 1) / 2;
    const uint32_t grp = (uint32_t)lane >> 4, sub = (uint32_t)lane & 15u;
    u64* WV = lds + 64 + mphf_words + (u64)wave * (4 * grp_words);
    u64* F = WV + grp * grp_words;
    uint32_t* LVT = reinterpret_cast<uint32_t*>(F + W);
    int32_t* OUTG = reinterpret_cast<int32_t*>(LVT + X4_LEVELS * X4_LV_WORDS);
    const uint32_t m = prm.max_mismatch;

    uint32_t c_al = 0, c_na = 0;
    unsigned long long c_ov = 0;
    uint32_t chunk_pos = 0, chunk_end = 0;

    for (uint32_t rbase = (blockIdx.x * waves + wave) * 4; rbase < io.n_reads; rbase += gridDim.x * waves * 4) {
        const uint32_t r = rbase + grp;
        const uint32_t have = r < io.n_reads ? 1u : 0u;
        u64 off = 0;
        uint32_t L = 0, fast = 0;
        if (have) {
            off = io.read_offs[r];
            L = (uint32_t)(io.read_offs[r + 1] - off);
            fast = ((io.hasn[r >> 5] >> (r & 31)) & 1u) ^ 1u;
            if (L <= K1) fast = 0;  // (a read of k-1 bases or fewer: the general kernel)
        }
        {
            u64 f = 0;
            if (fast && sub < ((L + 31) >> 5)) f = io.fw3[packed_word_offset(off, r) + sub];
            if (sub < W) F[sub] = f;
        }
        wave_sync();
        // ---- the first position that can anchor the read (getListOverlap keeps every position, aligner.cpp:318-342; only
        // position 0 and overlap (k-1)-mers of the graph can succeed): position 0 when its k-mer is an overlap, else the first hit
        uint32_t a_pos = 0, a_rec = BGR_NONE;
        for (uint32_t qq = 0; qq < 4; ++qq) {
            if (!rl32(fast, (int)(16 * qq))) continue;
            const uint32_t Lq = rl32(L, (int)(16 * qq));
            const u64* A = WV + qq * grp_words;
            const uint32_t npos = Lq - K1 + 1;
            for (uint32_t base = 0; base < npos; base += 64) {
                const uint32_t i = base + (uint32_t)lane;
                const bool valid = i < npos;
                u64 num = 0;
                if (valid) num = lds_win32(A, i) >> (64 - 2 * K1);
                const u64 rcn = rcb_fast(num, K1);
                uint32_t idx = find_key<!STAGE>(g, LV, units, num < rcn ? num : rcn, valid);
                const u64 mask = __ballot(idx != BGR_NONE);
                if (mask) {
                    if (idx != BGR_NONE && num <= rcn) idx |= G4_CANON;
                    const int s1 = __ffsll((long long)mask) - 1;
                    const uint32_t h1 = rl32(idx, s1);
                    if (grp == qq) { a_pos = base + (uint32_t)s1; a_rec = h1; }
                    break;
                }
            }
        }
        const uint32_t npos_g = L >= K1 ? L - K1 + 1 : 0;
        const uint32_t anchored = (fast && a_rec != BGR_NONE) ? 1u : 0u;
        // left side: [0] at position 0 (no search), else the search with the whole budget
        uint32_t eb = 0, nl = 0, fbl = 0;
        {
            const uint32_t actl = (anchored && a_pos != 0) ? 1u : 0u;
            uint32_t cl = 0, nll = 0;
            x4_search<0>(g, F, L, K1, actl, a_rec & G4_REC_MASK, (a_rec >> 28) & 1u, a_pos, m, LVT, OUTG, 0, lane, &cl, &nll, &fbl);
            if (actl) { eb = cl; nl = nll; }
            else if (anchored) { if (sub == 0) OUTG[0] = 0; nl = 1; }
        }
        wave_sync();
        uint32_t ee = 0, nr = 0, fbr = 0;
        {
            const uint32_t actr = (anchored && !fbl && eb <= m) ? 1u : 0u;
            uint32_t cr = 0, nrr = 0;
            x4_search<1>(g, F, L, K1, actr, a_rec & G4_REC_MASK, (a_rec >> 28) & 1u, a_pos, actr ? m - eb : 0u, LVT, OUTG, nl, lane, &cr, &nrr, &fbr);
            if (actr) { ee = cr; nr = nrr; } else ee = X4_INF;
        }
        // 0 = aligned; 2 = not aligned for sure (no overlap (k-1)-mer anywhere in the read: every position fails); 4 = the list
        uint32_t outcome = 4;
        if (fast && !anchored) outcome = 2;
        else if (anchored && !fbl && !fbr && eb <= m && ee != X4_INF && eb + ee <= m) outcome = 0;
        const uint32_t aligned = outcome == 0 ? 1u : 0u;
        const uint32_t p_n = aligned ? nl + nr : 0;
        const uint32_t n0 = rl32(p_n, 0), n1 = rl32(p_n, 16), n2 = rl32(p_n, 32), n3 = rl32(p_n, 48);
        const uint32_t tot = n0 + n1 + n2 + n3;
        if (tot > chunk_end - chunk_pos) {
            const uint32_t want = tot > io.arena_chunk ? tot : io.arena_chunk;
            uint32_t got = 0;
            if (lane == 0) got = atomicAdd(io.cursor, want);
            chunk_pos = rl32(got, 0);
            chunk_end = chunk_pos + want;
        }
        const uint32_t gbase = chunk_pos + (grp > 0 ? n0 : 0u) + (grp > 1 ? n1 : 0u) + (grp > 2 ? n2 : 0u);
        const bool room = chunk_pos + tot <= io.arena_cap;
        chunk_pos += tot;
        for (uint32_t j = sub; j < p_n; j += 16)
            if (room) io.arena[gbase + j] = OUTG[j];
        if (!room && lane == 0 && tot) io.cursor[1] = 1;
        if (sub == 0 && have) {
            if (outcome == 0) io.results[r] = make_uint2(gbase, p_n | ((uint32_t)BGR_ST_ALIGNED << 24));
            else if (outcome == 2) io.results[r] = make_uint2(0u, (uint32_t)BGR_ST_FAILED << 24);
            else io.ovf_list[atomicAdd(io.cursor + io.ovf_ctr, 1u)] = r;
        }
        const u64 fin = __ballot(sub == 0 && have && outcome != 4);
        c_al += (uint32_t)__popcll(__ballot(sub == 0 && have && outcome == 0));
        c_na += (uint32_t)__popcll(__ballot(sub == 0 && have && outcome == 2));
        for (int gq = 0; gq < 4; ++gq)
            if ((fin >> (16 * gq)) & 1) c_ov += rl32(npos_g, 16 * gq);  // overlaps += listOverlap.size() (alignerExhaustive.cpp:38)
        wave_sync();
    }
    if (lane == 0 && (c_al | c_na)) {
        unsigned long long* counters = reinterpret_cast<unsigned long long*>(io.cursor + 16);
        atomicAdd(&counters[0], (unsigned long long)(c_al + c_na));
        if (c_al) atomicAdd(&counters[2], (unsigned long long)c_al);
        if (c_na) atomicAdd(&counters[3], (unsigned long long)c_na);
        atomicAdd(&counters[4], c_ov);
    }
}

// Pass 1 of exhaustive mode with the level-by-level search (exh_dp); what it cannot hold goes to the overflow list and
// through bgr_align_exhaustive_kernel<false, true>.
template <bool STAGE>
__global__ void __launch_bounds__(1024, BGR_DP_OCC) bgr_align_exhaustive_dp_kernel(BgrDeviceGraph g, BatchIO io, KernelParams prm) {
    extern __shared__ u64 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const uint32_t W = io.words_per_read;
    const uint32_t K1 = g.k - 1;
    uint2* LV;
    uint32_t mphf_words;
    const uint32_t* units = block_prologue<STAGE>(g, lds, &LV, &mphf_words);
    // per wave: FW3 | FWQ | RCW | NM | OUT | BEST | tables (32 + levels * 52 + levels words, see exh_dp)
    const uint32_t table_words = 32 + io.frames_per_wave * (DP_LEVEL_WORDS + 1);
    const uint32_t per_wave_words = 4 * W + 2 * (io.path_cap / 2) + ((table_words + 3) / 4) * 2;  // whole 16-byte units
    u64* FW3 = lds + 64 + mphf_words + (u64)wave * per_wave_words;
    u64* FWQ = FW3 + W;
    u64* RCW = FWQ + W;
    u64* NM = RCW + W;
    int32_t* OUT = reinterpret_cast<int32_t*>(NM + W);
    int32_t* BEST = OUT + io.path_cap;
    uint32_t* T = reinterpret_cast<uint32_t*>(BEST + io.path_cap);

    uint32_t c_reads = 0, c_al = 0, c_na = 0;
    unsigned long long c_ov = 0;
    uint32_t chunk_pos = 0, chunk_end = 0;
    const uint32_t m = prm.max_mismatch;

    // pass 2 maps only the reads that pass 1 listed as needing a deeper stack (count left in cursor[subset_ctr] by the pass before)
    const uint32_t total = io.subset ? io.cursor[io.subset_ctr] : io.n_reads;
    for (uint32_t it = blockIdx.x * waves + wave; it < total; it += gridDim.x * waves) {
        const uint32_t r = io.subset ? io.subset[it] : it;
        const u64 off = io.read_offs[r];
        const uint32_t L = (uint32_t)(io.read_offs[r + 1] - off);
        const bool hasN = load_packed(io, r, off, L, W, FW3, NM, lane);
        if (hasN) derive_streams(L, W, K1, FW3, FWQ, RCW, NM, lane);
        const u64* ROLL = hasN ? FWQ : FW3;  // the rolling `num` stream
        uint32_t p_n = 0;
        const uint32_t npos = L >= K1 ? L - K1 + 1 : 0;
        bool done = false, overflow = false;
        for (uint32_t base = 0; base < npos && !done && !overflow; base += 64) {
            const uint32_t i = base + lane;
            const bool valid = i < npos;
            u64 num = 0;
            if (valid) num = lds_win32(ROLL, i) >> (64 - 2 * K1);  // the rolling `num` (aligner.cpp:321,334)
            const u64 rc = rcb_fast(num, K1);                  // getBegin/getEnd use rcb(num) (aligner.cpp:149,211)
            const uint32_t idx = find_key<!STAGE>(g, LV, units, num < rc ? num : rc, valid);
            u64 mask = __ballot(idx != BGR_NONE);
            if (base == 0) mask |= 1;  // position 0: the left side is trivially [0] whatever the k-mer
            while (mask) {
                const int src = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const uint32_t a_rec = rl32(idx, src), a_pos = base + (uint32_t)src;
                const u64 a_num = rl64(lds_win32(ROLL, a_pos) >> (64 - 2 * K1), 0);  // re-read (uniform) instead of keeping `num` live across the search
                const bool a_canon = a_num <= rcb_fast(a_num, K1);
                uint32_t nl = 0, nr = 0, eb = 0;
                if (a_pos == 0) {  // checkBeginExhaustive at position 0 is [0] at no cost (alignerExhaustive.cpp:159): no search
                    if (lane == 0) OUT[0] = 0;
                    nl = 1;
                } else {
                    eb = exh_dp<0>(g, FW3, NM, hasN, L, K1, a_rec, a_canon, a_pos, m, false, T, io.frames_per_wave, BEST, &nl, lane);
                    if (eb == EXH_OVERFLOW) { overflow = true; break; }
                    if (eb > m) continue;
                    for (uint32_t j = lane; j < nl; j += 64) OUT[j] = BEST[j];
                }
                wave_sync();
                // position 0 is tried whatever its (k-1)-mer: when that is no overlap of the graph getBegin() is empty, and only an
                // empty right side or -i can make the anchor succeed (alignerExhaustive.cpp:206-221): no search either
                if (a_rec == BGR_NONE && !prm.partial && L - a_pos - K1 != 0) continue;
                const uint32_t ee = exh_dp<1>(g, FW3, NM, hasN, L, K1, a_rec, a_canon, a_pos, m - eb, prm.partial != 0, T, io.frames_per_wave, BEST, &nr, lane);
                if (ee == EXH_OVERFLOW) { overflow = true; break; }
                if (ee > m - eb) continue;
                for (uint32_t j = lane; j < nr; j += 64) OUT[nl + j] = BEST[j];
                p_n = nl + nr;
                done = true;
                break;
            }
        }
        wave_sync();
        if (overflow) {  // leave this read to pass 2 (full-depth stack); nothing is written or counted for it here
            if (lane == 0) io.ovf_list[atomicAdd(io.cursor + io.ovf_ctr, 1u)] = r;
            continue;
        }
        c_ov += npos;
        uint32_t abase = 0;
        if (done) abase = publish_path(io, OUT, 0, p_n, &chunk_pos, &chunk_end, lane);
        if (lane == 0) io.results[r] = make_uint2(abase, p_n | ((uint32_t)(done ? BGR_ST_ALIGNED : BGR_ST_FAILED) << 24));
        ++c_reads;
        c_al += done ? 1 : 0;
        c_na += done ? 0 : 1;
        wave_sync();
    }
    if (lane == 0 && c_reads) {
        unsigned long long* counters = reinterpret_cast<unsigned long long*>(io.cursor + 16);
        atomicAdd(&counters[0], (unsigned long long)c_reads);
        if (c_al) atomicAdd(&counters[2], (unsigned long long)c_al);
        if (c_na) atomicAdd(&counters[3], (unsigned long long)c_na);
        atomicAdd(&counters[4], c_ov);
    }
}

template <typename K>
hipError_t launch_one(K kernel, const BgrDeviceGraph& g, const BatchIO& io, const KernelParams& p, const LaunchCfg& cfg, hipStream_t stream) {
    if (cfg.lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cfg.lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3(cfg.blocks), dim3(cfg.waves_per_block * 64), cfg.lds_bytes, stream, g, io, p);
    return hipGetLastError();
}

}  // namespace

// ======================================= pre-pass: ASCII reads -> 2-bit planes ================================
// Streaming kernel in front of every mapping launch that is handed ASCII reads (what getReads yields, aligner.cpp:46-117):
// str2num codes (utils.cpp:117-129: A0 C1 G2, anything else 3) 32 bases per u64, first base most significant, plus
// the N mask for the few reads that hold an N (their bit is set in `hasn`, which the caller zeroes).  8 lanes per read,
// 32 bases per lane and step.  ~150 B in + 48 B out per 150 bp read: HBM-streaming bound.
__global__ void __launch_bounds__(256) bgr_pack_reads_kernel(const uint8_t* reads, const u64* read_offs, uint32_t n, u64 total_bytes, u64* fw3,
                                                             u64* nmw, uint32_t* hasn) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t r = t >> 3, j0 = t & 7;
    const int lane = threadIdx.x & 63;
    bool sawN = false;
    u64 off = 0;
    uint32_t L = 0, Wr = 0, woff = 0;
    if (r < n) {
        off = read_offs[r];
        L = (uint32_t)(read_offs[r + 1] - off);
        Wr = (L + 31) >> 5;
        woff = packed_word_offset(off, r);
        for (uint32_t j = j0; j < Wr; j += 8) {
            u64 w, nm;
            pack32(reads + off, L, j, off + 32ull * j + 32 <= total_bytes, &w, &nm);
            fw3[woff + j] = w;
            sawN |= nm != 0;
        }
    }
    // a read with an N: all its N-plane words are written (the mapping kernels read the plane only for such reads)
    const u64 any = __ballot(sawN);
    if ((any >> (lane & ~7)) & 0xFFu) {
        for (uint32_t j = j0; j < Wr; j += 8) {
            u64 w, nm;
            pack32(reads + off, L, j, off + 32ull * j + 32 <= total_bytes, &w, &nm);
            nmw[woff + j] = nm;
        }
        if (j0 == 0) atomicOr(&hasn[r >> 5], 1u << (r & 31));
    }
}

hipError_t launch_pack_reads(const uint8_t* reads, const uint64_t* read_offs, uint32_t n, uint64_t total_bytes, uint64_t* fw3, uint64_t* nmw,
                             uint32_t* hasn, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)(((uint64_t)n * 8 + 255) / 256);
    hipLaunchKernelGGL(bgr_pack_reads_kernel, dim3(blocks), dim3(256), 0, stream, reads, read_offs, n, total_bytes, fw3, nmw, hasn);
    return hipGetLastError();
}

// plane[index[i]] = value[i]: the N-mask words of a host-packed batch (bgr_align_batch_packed)
__global__ void __launch_bounds__(256) bgr_scatter_words_kernel(const uint32_t* index, const u64* value, uint64_t n, u64* plane, uint64_t plane_words) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && index[i] < plane_words) plane[index[i]] = value[i];
}
hipError_t launch_scatter_words(const uint32_t* index, const uint64_t* value, uint64_t n, uint64_t* plane, uint64_t plane_words, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(bgr_scatter_words_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, index, value, n, plane, plane_words);
    return hipGetLastError();
}

// ======================================= results -> CSR, on the device =======================================
// The mapping kernels leave every path where its wave found room in the arena.  These three small kernels turn
// (results, arena) into what the C-ABI hands out -- input-ordered path_offsets[n+1], dense paths, status bytes -- so
// the host neither loops over the reads nor copies the arena: 4096 reads per workgroup (4 per thread), block sums,
// one-workgroup scan of the sums, then the gather.
constexpr uint32_t kCsrThreads = 1024, kCsrItems = 4, kCsrTile = kCsrThreads * kCsrItems;

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds_waves, uint32_t* block_total) {
    // inclusive scan inside the wave by DPP-free shuffles, then across the 16 waves through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)inc, d, 64);
        if (lane >= d) inc += up;
    }
    if (lane == 63) lds_waves[wave] = inc;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) {
        const uint32_t t = lds_waves[w];
        if (w < (uint32_t)wave) before += t;
        total += t;
    }
    __syncthreads();
    *block_total = total;
    return before + inc - v;
}

__global__ void __launch_bounds__(kCsrThreads) bgr_csr_block_sums(const uint2* results, uint32_t n, uint32_t* block_sums) {
    __shared__ uint32_t lw[16];
    const uint32_t base = blockIdx.x * kCsrTile + threadIdx.x * kCsrItems;
    uint32_t s = 0;
#pragma unroll
    for (uint32_t j = 0; j < kCsrItems; ++j) if (base + j < n) s += results[base + j].y & 0xFFFFFFu;
    uint32_t total;
    (void)block_exclusive_scan(s, lw, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// one workgroup: block_sums[b] -> ints before tile b; total[0] = all ints
__global__ void __launch_bounds__(kCsrThreads) bgr_csr_scan_sums(uint32_t* block_sums, uint32_t nb, unsigned long long* total_out) {
    __shared__ uint32_t lw[16];
    __shared__ unsigned long long carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nb; b0 += kCsrThreads) {
        const uint32_t i = b0 + threadIdx.x;
        const uint32_t v = i < nb ? block_sums[i] : 0;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan(v, lw, &total);
        const unsigned long long carry = carry_s;
        if (i < nb) block_sums[i] = (uint32_t)(carry + ex);  // < 2^32: the arena holds fewer than 2^32 ints
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry_s;
}

__global__ void __launch_bounds__(kCsrThreads) bgr_csr_gather(const uint2* results, const int32_t* arena, uint32_t n, const uint32_t* block_offs,
                                                             unsigned long long* path_offsets, int32_t* paths, uint8_t* status, uint32_t paths_cap) {
    __shared__ uint32_t lw[16];
    const uint32_t base = blockIdx.x * kCsrTile + threadIdx.x * kCsrItems;
    uint2 r[kCsrItems];
    uint32_t s = 0;
#pragma unroll
    for (uint32_t j = 0; j < kCsrItems; ++j) {
        r[j] = base + j < n ? results[base + j] : make_uint2(0, 0);
        s += r[j].y & 0xFFFFFFu;
    }
    uint32_t total;
    uint32_t w = block_offs[blockIdx.x] + block_exclusive_scan(s, lw, &total);
    if (blockIdx.x == 0 && threadIdx.x == 0) path_offsets[0] = 0;
#pragma unroll
    for (uint32_t j = 0; j < kCsrItems; ++j) {
        if (base + j >= n) break;
        const uint32_t len = r[j].y & 0xFFFFFFu;
        if (w + len <= paths_cap)
            for (uint32_t q = 0; q < len; ++q) paths[w + q] = arena[r[j].x + q];
        w += len;
        path_offsets[base + j + 1] = w;
        status[base + j] = (uint8_t)(r[j].y >> 24);
    }
}

hipError_t launch_csr(const uint2* results, const int32_t* arena, uint32_t n, uint32_t* block_sums, unsigned long long* total,
                      unsigned long long* path_offsets, int32_t* paths, uint8_t* status, uint32_t paths_cap, int phase, hipStream_t stream) {
    const uint32_t nb = (n + kCsrTile - 1) / kCsrTile;
    if (phase == 0) {  // lengths -> tile offsets + total
        hipLaunchKernelGGL(bgr_csr_block_sums, dim3(nb), dim3(kCsrThreads), 0, stream, results, n, block_sums);
        hipLaunchKernelGGL(bgr_csr_scan_sums, dim3(1), dim3(kCsrThreads), 0, stream, block_sums, nb, total);
    } else {           // gather (the caller has sized `paths` from the total)
        hipLaunchKernelGGL(bgr_csr_gather, dim3(nb), dim3(kCsrThreads), 0, stream, results, arena, n, block_sums, path_offsets, paths, status, paths_cap);
    }
    return hipGetLastError();
}

uint32_t resident_waves_per_cu(uint32_t mode) {
    hipFuncAttributes fa;
    const void* fn = mode == 0 ? reinterpret_cast<const void*>(&bgr_align_greedy_kernel<true>)
                   : mode == 2 ? reinterpret_cast<const void*>(&bgr_align_anchors_kernel)
                   : mode == 3 ? reinterpret_cast<const void*>(&bgr_align_exhaustive_dp_kernel<false>)
                   : mode == 4 ? reinterpret_cast<const void*>(&bgr_align_greedy4_kernel<true, true>)
                   : mode == 5 ? reinterpret_cast<const void*>(&bgr_align_exhaustive4_kernel<false>)
                               : reinterpret_cast<const void*>(&bgr_align_exhaustive_kernel<true, false>);
    if (hipFuncGetAttributes(&fa, fn) != hipSuccess || fa.numRegs <= 0) return 16;
    // MI355X_MICROARCH.md "Register files": 512 VGPRs per SIMD lane, allocation granule 8, at most 8 waves per SIMD;
    // the kernels use ~106 SGPRs, which caps a SIMD at 6 waves (800 / (7*16 + 16)); compiling for 7 (72 VGPRs,
    // spills) measured 381 vs 532 Mreads/s greedy, 28 vs 37 exhaustive.
    const uint32_t alloc = ((uint32_t)fa.numRegs + 7) / 8 * 8;
    return 4 * std::min<uint32_t>(mode == 4 ? 8 : 6, 512 / alloc);
}

hipError_t launch_align(const BgrDeviceGraph& g, const BatchIO& io, const KernelParams& p, const LaunchCfg& cfg, hipStream_t stream) {
    if (io.n_reads == 0) return hipSuccess;
    if (p.mode == 0) {
        if (io.greedy4 && io.subset) return cfg.stage_mphf ? launch_one(bgr_align_greedy4_kernel<true, true>, g, io, p, cfg, stream)
                                                           : launch_one(bgr_align_greedy4_kernel<false, true>, g, io, p, cfg, stream);
        if (io.greedy4) return cfg.stage_mphf ? launch_one(bgr_align_greedy4_kernel<true, false>, g, io, p, cfg, stream)
                                              : launch_one(bgr_align_greedy4_kernel<false, false>, g, io, p, cfg, stream);
        return cfg.stage_mphf ? launch_one(bgr_align_greedy_kernel<true>, g, io, p, cfg, stream)
                              : launch_one(bgr_align_greedy_kernel<false>, g, io, p, cfg, stream);
    }
    if (p.mode == 2) return launch_one(bgr_align_anchors_kernel, g, io, p, cfg, stream);
    if (io.exh4) return cfg.stage_mphf ? launch_one(bgr_align_exhaustive4_kernel<true>, g, io, p, cfg, stream)
                                       : launch_one(bgr_align_exhaustive4_kernel<false>, g, io, p, cfg, stream);
    if (io.deep_scratch) return launch_one(bgr_align_exhaustive_kernel<false, true>, g, io, p, cfg, stream);
    if (io.level_search) return cfg.stage_mphf ? launch_one(bgr_align_exhaustive_dp_kernel<true>, g, io, p, cfg, stream)
                                               : launch_one(bgr_align_exhaustive_dp_kernel<false>, g, io, p, cfg, stream);
    return cfg.stage_mphf ? launch_one(bgr_align_exhaustive_kernel<true, false>, g, io, p, cfg, stream)
                          : launch_one(bgr_align_exhaustive_kernel<false, false>, g, io, p, cfg, stream);
}

}  // namespace bgr
