// anchors_kernel.hip -- anchors mode (-G: getNAnchors + alignReadGreedyAnchors, aligner.cpp:381-405, alignerGreedy.cpp:60-164).
#include "device_common.h"

namespace bgr {
namespace {

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

    // as the second pass behind bgr_align_anchors4_kernel it maps only the reads that kernel listed (count in cursor[subset_ctr])
    const uint32_t total = io.subset ? io.cursor[io.subset_ctr] : io.n_reads;
    for (uint32_t it = blockIdx.x * waves + wave; it < total; it += gridDim.x * waves) {
        const uint32_t r = io.subset ? io.subset[it] : it;
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

// ============================ anchors mode, several reads per wavefront ===================================
// The kernel above gives a wave one read; per read it does a few BooPHF lookups (lane = level: a handful of the 64 lanes),
// one placement compare and the greedy walks (a handful of lanes again).  Here a wave takes 64 / GL reads, GL lanes each (8
// when the index has at most 8 active levels -- eight reads per wave -- else 16):
// lane `sub` of a group probes level `sub` of the lookup, compares bases [32 sub, 32 sub + 32) of the placement, and the
// walks run through g4_step like the greedy kernel's.  All four groups step through the reference's loop in lockstep --
// lookup at the current position, placement of the anchoring unitig if the lookup answered, walks, next position / next
// strand -- which is efficient because every read does about the same work (the first `effort` answers are tried, then
// the reverse complement).  Reads with an N, paths longer than GL ints per side and graphs with more than 16 active
// BooPHF levels go to bgr_align_anchors_kernel (listed / not launched).
#ifndef BGR_ANC4_OCC
#define BGR_ANC4_OCC 5 /* 96 VGPRs: 632 Mreads/s; 4 (110 VGPRs): 574; 6 (80 VGPRs): 621 */
#endif

__device__ __forceinline__ u64 lane_get64(u64 v, uint32_t src) {
    return ((u64)lane_get((uint32_t)(v >> 32), src) << 32) | lane_get((uint32_t)v, src);
}

// sum over the GL (16 or 8) lanes of a group, in every lane
template <int GL>
__device__ __forceinline__ uint32_t group_sum(uint32_t x) {
    if (GL == 16) return row16_sum(x);
    x += quad_xor1(x);
    x += quad_xor2(x);
    x += half_row_mirror(x);
    return x;
}

// boomphf::mphf::lookup (BooPHF.h:783-818) for one key per GL-lane group (need = the group looks up): index or ~0.
// Lane j of a group probes level j: the index must have at most GL active levels.
template <int GL>
__device__ __forceinline__ u64 anc_lookup4(const AncView& a, u64 key, uint32_t need, int lane) {
    const uint32_t sub = (uint32_t)lane % GL, gb = (uint32_t)lane & (64u - GL);
    u64 s0 = bgr_boo_hash64(key, BGR_BOO_SEED0), s1 = bgr_boo_hash64(key, BGR_BOO_SEED1);
    u64 hv = sub == 0 ? s0 : s1;
    for (uint32_t i = 2; i < a.n_active; ++i) {  // BooPHF.h:336-356: the level hashes are a sequence, walked in step
        const u64 v = bgr_boo_next(&s0, &s1);
        if (sub == i) hv = v;
    }
    bool hit = false;
    u64 pos = 0;
    if (need && sub < a.n_active) {
        pos = bgr_mod_magic(hv, a.lv_domain, a.lv_magic);
        hit = (a.bits[a.lv_word_base + (pos >> 6)] >> (pos & 63)) & 1;
    }
    const u64 mask = __ballot(hit);
    const uint32_t m16 = (uint32_t)(mask >> gb) & (GL == 16 ? 0xFFFFu : 0xFFu);
    const uint32_t src = gb | (m16 ? (uint32_t)(__ffs((int)m16) - 1) : 0u);  // the first level whose bit is set answers
    const u64 fpos = lane_get64(pos, src), wb = lane_get64(a.lv_word_base, src), rb = lane_get64(a.lv_rank_base, src);
    const u64 widx = fpos >> 6, blk = fpos >> 9;
    uint32_t cnt = 0;
    if (m16 && sub < 8) {  // BooPHF.h:609-622 rank: sample of the 512-bit block + popcount of the words before the bit
        const u64 wi = blk * 8 + (u64)sub;
        if (wi < widx) cnt = (uint32_t)__popcll(a.bits[wb + wi]);
        else if (wi == widx) cnt = (uint32_t)__popcll(a.bits[wb + wi] & ((1ULL << (fpos & 63)) - 1));
    }
    cnt = group_sum<GL>(cnt);
    u64 res = ~0ULL;
    if (m16) res = a.ranks[rb + blk] + cnt;
    const uint32_t slow = (need && !m16) ? 1u : 0u;  // what 24 levels could not place (repeated k-mers): exact, sorted {key, index}
    if (a.n_final && __any(slow != 0)) {
        u64 lo = 0, hi = slow ? a.n_final : 0;
        while (__any(lo < hi)) {
            if (lo < hi) {
                const u64 mid = (lo + hi) >> 1;
                if (a.fin[2 * mid] < key) lo = mid + 1; else hi = mid;
            }
        }
        if (slow && lo < a.n_final && a.fin[2 * lo] == key) res = a.last_rank + a.fin[2 * lo + 1];
    }
    return res;
}

template <int GL>
__global__ void __launch_bounds__(1024, BGR_ANC4_OCC) bgr_align_anchors4_kernel(BgrDeviceGraph g, BatchIO io, KernelParams prm) {
    constexpr uint32_t RPW = 64 / GL;  // reads per wave
    extern __shared__ u64 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t W = io.words_per_read;  // <= 16 (checked by the host)
    const uint32_t K = g.k, K1 = g.k - 1;
    const uint32_t grp = (uint32_t)lane / GL, sub = (uint32_t)lane % GL;
    u64* RD = lds + 64 + (u64)wave * (RPW * 2 * W);  // the reads of this wave: forward words | reverse-complement words
    u64* F = RD + grp * (2 * W);
    const AncView av = anc_view(g, (int)sub);
    const u64 km1_mask = (1ULL << (2 * K1)) - 1;  // offsetUpdate - 1 (aligner.h:101-102): update() keeps k-1 digits
    const uint32_t m = prm.max_mismatch;
    const uint32_t effort = prm.effort ? prm.effort : 1;  // getNAnchors(read, 0) still takes a hit at position 0

    uint32_t c_noov = 0, c_al = 0, c_na = 0;
    // Path ints go straight into the arena, as in bgr_align_greedy_multi_kernel: read r owns the row arena[r * kA4PathInts ...]; left int
    // number i (near -> far, offset last) at row[LH - 1 - i], then the anchoring unitig ([offset,] id) from row[LH] on, the right ints
    // behind it: reverse(left) ++ [offset,] unitig ++ right is the slice row[LH - nl, LH + nmid + nr)
    constexpr uint32_t LH = kA4PathInts / 2 - 1;
    unsigned long long* wg_counts = wg_counts_init(lds);
    task_stock_init(lds, (io.n_reads + RPW - 1) / RPW);
    __syncthreads();

    // (a wave claims its next RPW reads at run time -- claim_task, device_common.h -- instead of a share dealt out by wave number)
    const uint32_t n_tasks = (io.n_reads + RPW - 1) / RPW;
    for (uint32_t task; (task = claim_task(lds, io.cursor + io.task_ctr, n_tasks, lane)) != BGR_NONE;) {
        const uint32_t rbase = task * RPW;
        const uint32_t r = rbase + grp;
        const uint32_t have = r < io.n_reads ? 1u : 0u;
        u64 off = 0;
        uint32_t L = 0, fast = 0;
        if (have) {
            off = io.read_offs[r];
            L = (uint32_t)(io.read_offs[r + 1] - off);
            fast = ((io.hasn[r >> 5] >> (r & 31)) & 1u) ^ 1u;  // a read with an N goes to the general kernel
            if (((L + 31) >> 5) >= W) fast = 0;                // so does a read too long for one lane per word
        }
        for (uint32_t j = sub; j < W; j += GL) {
            u64 f = 0;
            if (fast && j < ((L + 31) >> 5)) f = io.fw3[packed_word_offset(off, r) + j];
            F[j] = f;
        }
        wave_sync();
        const uint32_t dk = L < K ? L : K;                                  // read.substr(0, k) of a shorter read is the whole read
        const uint32_t last_i = !prm.effort ? 0u : (L > K ? L - K : 0u);    // the loop leaves after position i when i + k >= |read|
        // what became of the read: 0 aligned, 1 no anchor (++noOverlapRead), 2 not aligned, 4 general kernel
        uint32_t outcome = fast ? 1u : 4u, rc = 0;
        uint32_t active = (fast && L > 0) ? 1u : 0u;   // (an empty read: no anchor, no strand switch, alignerGreedy.cpp:60-67)
        // getNAnchors (aligner.cpp:381-405): the first k-mer by str2num, then the (k-1)-mer rolling updates applied to it
        u64 num = active ? lds_win32(F, 0) >> (64 - 2 * dk) : 0;
        u64 rcnum = rcb_fast(num, K);
        uint32_t i = 0, tried = 0;
        uint32_t nl = 0, nr = 0, nmid = 0;
        int32_t* PT = io.arena + (size_t)(have ? r : 0u) * kA4PathInts;  // the read's row; written by the group's first lane
        while (__any(active != 0)) {
            const u64* S = F + (rc ? W : 0);  // the characters of this pass's read (reverseComplements: N -> 'A')
            // ---- one lookup per group at its current position ----
            const u64 idx = anc_lookup4<GL>(av, num < rcnum ? num : rcnum, active, lane);
            const uint32_t found = (active && idx != ~0ULL) ? 1u : 0u;
            uint32_t success = 0, bad = 0;
            if (__any(found != 0)) {
                tried += found;
                // ---- alignReadGreedyAnchors loop body for this anchor (alignerGreedy.cpp:68-161): place the unitig on the read ----
                u64 pv = 0;
                BgrUnitigMeta mt;
                mt.len = 0; mt.flags = 0; mt.rec_beg = 0; mt.rec_end = 0; mt.F = 0; mt.hw[0] = mt.hw[1] = 0;
                if (found) { pv = av.pos[idx]; mt = g.meta[(uint32_t)(pv >> 32)]; }
                const uint32_t un = (uint32_t)(pv >> 32);
                uint32_t pU = (uint32_t)pv;
                const uint32_t pR = i, len = mt.len;
                const uint32_t okp = (found && len >= K) ? 1u : 0u;  // :72-75 (an index nobody wrote holds unitig 0, the empty string)
                const uint32_t fw = (uint32_t)(mt.F >> 5);
                uint32_t fo = (uint32_t)(mt.F & 31);
                u64 ukm = 0, rkm = 0;
                if (okp) {
                    ukm = seq_win32(g.seq, fw, fo + pU) >> (64 - 2 * K);
                    const uint32_t rdk = L - pR < K ? L - pR : K;
                    rkm = lds_win32(S, pR) >> (64 - 2 * rdk);
                }
                const uint32_t returned = ukm != rkm ? 1u : 0u;  // :76-83: any difference means "take the reverse complement"
                if (returned) { fo += len; pU = len - K - pU; }
                const int32_t uid = returned ? -(int32_t)un : (int32_t)un;
                // the oriented unitig's end (k-1)-mers as neighbour records (what str2num + getEnd/getBegin find)
                const uint32_t can_b = (mt.flags & (returned ? BGR_META_CANON_RCEND : BGR_META_CANON_BEG)) ? 1u : 0u;
                const uint32_t can_e = (mt.flags & (returned ? BGR_META_CANON_RCBEG : BGR_META_CANON_END)) ? 1u : 0u;
                // (as the HALVES the first steps read: getEnd for the left walk, getBegin for the right one -- handles, graph_layout.h)
                uint32_t rec_b = G4_REC_MASK, rec_e = G4_REC_MASK;
                if (found && okp) {
                    rec_b = half_handle(g, returned ? mt.rec_end : mt.rec_beg, can_b != 0, true);
                    rec_e = half_handle(g, returned ? mt.rec_beg : mt.rec_end, can_e != 0, false);
                }
                const uint32_t c12 = pR >= pU ? 1u : 0u;                       // the unitig starts inside the read (cases 1, 2)
                const uint32_t longr = (L - pR >= len - pU) ? 1u : 0u;         // the read reaches the unitig's end (cases 1, 3)
                const uint32_t start = pR - pU, uoff = pU - pR;               // (whichever the case uses)
                uint32_t ub, rb, n;
                if (c12) { ub = fo; rb = start; n = longr ? len : L - start; }  // CASE 1 (:87-110) / CASE 2 (:111-130)
                else { ub = fo + uoff; rb = 0; n = longr ? len - uoff : L; }   // CASE 3 (:133-148) / CASE 4 (:149-160)
                if (!okp) n = 0;
                uint32_t errors = 0;
                for (uint32_t b = sub * 32; __any(b < n); b += 32 * GL)
                    if (b < n) errors += ham_chunk(g, S, nullptr, false, fw, ub + b, rb + b, n - b);
                errors = group_sum<GL>(errors);
                const uint32_t good = (okp && errors <= m) ? 1u : 0u;
                // ---- the walks from the unitig's ends: left (cases 1, 2), right (cases 1, 3) ----
                const uint32_t want_left = (good && c12) ? 1u : 0u, want_right = (good && longr) ? 1u : 0u;
                const uint32_t r_pos = c12 ? start + len - K1 : len - uoff - K1;
                uint32_t phase = want_left ? 1u : (want_right ? 2u : 0u);
                uint32_t pos = want_left ? start : r_pos, rec = want_left ? rec_b : rec_e, canon = want_left ? can_b : can_e, budget = m - errors;
                uint32_t wfail = 0;
                if (found) {
                    nl = 0; nr = 0;
                    if (c12) { if (sub == 0) PT[LH] = uid; nmid = 1; } else { if (sub == 0) { PT[LH] = (int32_t)uoff; PT[LH + 1] = uid; } nmid = 2; }
                }
                for (;;) {
                    if (phase == 1 && pos == 0) {  // the left walk reached the read's first base: push 0
                        if (sub == 0) PT[LH - 1 - nl] = 0;
                        ++nl;
                        phase = want_right ? 2u : 0u; pos = r_pos; rec = rec_e; canon = can_e;
                    }
                    if (phase == 2 && L - pos - K1 == 0) phase = 0;  // nothing right of the unitig
                    if (phase == 3 && L - pos < K1 + 1) phase = 0;   // |readLeft| < k
                    if ((phase == 1 && nl > LH - 2) || (phase >= 2 && nr > LH - 1)) { bad = 1; phase = 0; }  // path too long for the row (a left step may push two ints)
                    if (!__any(phase != 0)) break;
                    uint32_t miss, ext;
                    int32_t sid;
                    const uint32_t w1 = g4_step<GL>(g, S, L, K1, phase, rec, canon, pos, budget, lane, &miss, &ext, &sid);
                    if (phase != 0) {
                        if (!(w1 & G4_FOUND)) { wfail = 1; phase = 0; }
                        else if (phase == 1) {
                            if (sub == 0) PT[LH - 1 - nl] = sid;
                            ++nl;
                            budget -= miss;
                            if (w1 & G4_FITS) {
                                if (sub == 0) PT[LH - 1 - nl] = (int32_t)(ext - pos);
                                ++nl;
                                phase = want_right ? 2u : 0u; pos = r_pos; rec = rec_e; canon = can_e;
                            } else { pos -= ext; rec = w1 & G4_REC_MASK; canon = (w1 >> 28) & 1u; }
                        } else {
                            if (sub == 0) PT[LH + nmid + nr] = sid;
                            ++nr;
                            budget -= miss;
                            if (w1 & G4_FITS) phase = 0;
                            else { pos += ext; rec = w1 & G4_REC_MASK; canon = (w1 >> 28) & 1u; phase = 3; }
                        }
                    }
                }
                success = (good && !wfail && !bad) ? 1u : 0u;
            }
            // ---- where each group goes from here (alignerGreedy.cpp:60-164, aligner.cpp:381-405) ----
            uint32_t pass_end = 0;
            if (active) {
                if (bad) { outcome = 4; active = 0; }
                else if (success) { outcome = 0; active = 0; }
                else if ((found && tried >= effort) || i >= last_i) pass_end = 1;
                else {  // update() / updateRC() (aligner.cpp:305-315) with read[i + k]
                    const u64 d = lds_win32(S, i + K) >> 62;
                    num = ((num << 2) + d) & km1_mask;
                    rcnum = (rcnum >> 2) + ((3 - d) << (2 * K - 4));
                    ++i;
                }
            }
            uint32_t sw = 0;
            if (pass_end) {
                if (tried == 0) { outcome = 1; active = 0; }          // ++noOverlapRead, no retry
                else if (rc) { outcome = 2; active = 0; }             // every anchor failed on both strands
                else sw = 1;                                          // once more on the reverse complement (:162)
            }
            if (__any(sw != 0)) {
                if (sw) {
                    for (uint32_t j = sub; j < W; j += GL) {
                        const long long p = (long long)L - 32 * ((long long)j + 1);
                        u64 w = 0;
                        if (p >= 0) w = ~rev2_fast(lds_win32(F, (uint32_t)p));
                        else if (p > -32) { const uint32_t v = (uint32_t)(32 + p); w = (~rev2_fast(F[0] >> (64 - 2 * v))) & (~0ULL << (64 - 2 * v)); }
                        F[W + j] = w;
                    }
                }
                wave_sync();
                if (sw) {
                    rc = 1; i = 0; tried = 0;
                    num = lds_win32(F + W, 0) >> (64 - 2 * dk);
                    rcnum = rcb_fast(num, K);
                }
            }
        }
        // ---- publish: the path is a slice of the read's row ----
        const uint32_t aligned = (have && outcome == 0) ? 1u : 0u;
        const uint32_t p_n = aligned ? nl + nmid + nr : 0;
        const uint32_t gbase = (have ? r : 0u) * kA4PathInts + LH - nl;
        if (sub == 0 && have) {
            if (outcome <= 2) {
                const uint32_t code = (outcome == 0 ? BGR_ST_ALIGNED : outcome == 1 ? BGR_ST_NOANCHOR : BGR_ST_FAILED) | (rc ? BGR_ST_RC : 0u);
                io.results[r] = make_uint2(aligned ? gbase : 0u, p_n | (code << 24));
            } else {
                io.ovf_list[atomicAdd(io.cursor + io.ovf_ctr, 1u)] = r;
            }
        }
        c_al += (uint32_t)__popcll(__ballot(sub == 0 && have && outcome == 0));
        c_noov += (uint32_t)__popcll(__ballot(sub == 0 && have && outcome == 1));
        c_na += (uint32_t)__popcll(__ballot(sub == 0 && have && outcome == 2));
        wave_sync();
    }
    wg_counts_flush(io, wg_counts, lane, c_al + c_noov + c_na, c_noov, c_al, c_na, 0);
}

}  // namespace

hipError_t launch_anchors(const BgrDeviceGraph& g, const BatchIO& io, const KernelParams& p, const LaunchCfg& cfg, hipStream_t stream) {
    if (io.anc4 == 8) return launch_one(bgr_align_anchors4_kernel<8>, g, io, p, cfg, stream);  // (the value = lanes per read)
    if (io.anc4) return launch_one(bgr_align_anchors4_kernel<16>, g, io, p, cfg, stream);
    return launch_one(bgr_align_anchors_kernel, g, io, p, cfg, stream);
}
const void* anchors_kernel_fn(bool four_reads) {
    return four_reads ? reinterpret_cast<const void*>(&bgr_align_anchors4_kernel<8>) : reinterpret_cast<const void*>(&bgr_align_anchors_kernel);
}

}  // namespace bgr
