// exhaustive_kernels.hip -- exhaustive mode (-b: alignReadExhaustive, alignerExhaustive.cpp:35-259) on gfx950.
//   bgr_align_exhaustive4_kernel    eight reads per wavefront (8 lanes per read), one node per level of the walk (x4_search)
//   bgr_align_exhaustive_dp_kernel  the level search (exh_dp): all nodes of a level at once, backward cost pass
//   bgr_align_exhaustive_kernel     depth-first search in slot order (exh_search); DEEP (the last pass): the same recursion memoised on the
//                                   node (exh_memo), search state in HBM -- polynomial whatever the unitig set looks like
#include "device_common.h"
#ifndef BGR_EXH_MMX
#define BGR_EXH_MMX 1
#endif

namespace bgr {
namespace {

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
#define FR_WORDS 20
#define EXH_OVERFLOW 0xFFFFFFFFu

template <int DIR>
__device__ __forceinline__ uint32_t exh_search(const BgrDeviceGraph& g, const u64* CMP, const u64* NM, bool useN, uint32_t L, uint32_t K1,
                                               uint32_t a_rec, bool a_canon, uint32_t a_pos, uint32_t budget, bool partial,
                                               uint32_t* FR, uint32_t max_frames, int32_t* CUR, int32_t* BEST, uint32_t* best_n, int lane, uint32_t max_iters = 0) {
    // returns the best total (budget+1 if none; EXH_OVERFLOW if the search needs more than max_frames frames -- or, max_iters != 0, more
    // than that many loop iterations: on unitig sets that duplicate each other's k-mers the recursion enumerates exponentially many walks
    // (DESIGN 8 item 6); the read then goes to the last pass, whose level search over tables in HBM is polynomial);
    // the best walk's ints are BEST[0..*best_n) in output order
    uint32_t best = budget + 1;
    *best_n = 0;
    int depth = 0;
    uint32_t iters = 0;
    {   // (a frame names the HALF its candidates come from -- a handle, graph_layout.h; the anchor is a key entry)
        const uint32_t h0 = half_handle(g, a_rec, a_canon, DIR == 0);
        if (lane == 0) { FR[0] = h0; FR[1] = a_pos; FR[2] = 0; FR[3] = a_canon ? (1u << 17) : 0u; }
    }
    wave_sync();
    while (depth >= 0) {
        if (max_iters && ++iters > max_iters) { wave_sync(); return EXH_OVERFLOW; }
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
            if (rec != BGR_HNONE) {
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

template <int DIR>
__device__ __forceinline__ uint32_t exh_dp(const BgrDeviceGraph& g, const u64* CMP, const u64* NM, bool useN, uint32_t L, uint32_t K1,
                                           uint32_t a_rec, bool a_canon, uint32_t a_pos, uint32_t budget, bool partial,
                                           uint32_t* T, uint32_t max_levels, int32_t* BEST, uint32_t* best_n, int lane) {
    const int i = lane >> 4, c = (lane >> 2) & 3, sub = lane & 3, k = lane >> 2;
    const bool leader = sub == 0;
    *best_n = 0;
    {   // (a node names the HALF its candidates come from -- a handle, graph_layout.h; the anchor is a key entry)
        const uint32_t h0 = half_handle(g, a_rec, a_canon, DIR == 0);
        if (lane == 0) { T[0] = h0; T[1] = a_pos; T[2] = a_canon ? 1u : 0u; T[3] = 0; }
    }
    wave_sync();
    uint32_t n_cur = 1, lvl = 0;
    for (; n_cur; ++lvl) {
        if (lvl >= max_levels) { wave_sync(); return EXH_OVERFLOW; }
        const uint4 nd = reinterpret_cast<const uint4*>(T)[(lvl & 1) * 4 + i];
        const bool nvalid = (uint32_t)i < n_cur;
        const uint32_t rec = nd.x, pos = nd.y, prefix = nd.w;
        const bool canon = (nd.z & 1u) != 0;
        const bool end_here = nvalid && ((DIR == 0) ? (pos == 0) : (L - pos - K1 == 0));
        const bool has_rec = nvalid && !end_here && rec != BGR_HNONE;
        const uint32_t fbit = canon ? BGR_SLOT_F0 : BGR_SLOT_F1;
        uint4 sl = make_uint4(0, 0, 0, 0), m0 = make_uint4(0, 0, 0, 0);
        if (has_rec) {
            const uint4* sp = reinterpret_cast<const uint4*>(g.recs) + (size_t)(rec + (uint32_t)c) * 2;
            sl = sp[0];
            m0 = sp[1];
        }
        const uint32_t id = sl.x & BGR_SLOT_ID_MASK;
        const u64 lmask = __ballot(leader && has_rec && (sl.w & BGR_SLOT_LAST) != 0);
        const uint32_t nb = (uint32_t)(lmask >> (16 * i)) & 0x1111u;
        // candidates = the half's slots up to and including the first flagged one (the reference stops at the first empty slot)
        const uint32_t first_zero = nb ? ((uint32_t)(__ffs((int)nb) - 1) >> 2) + 1u : 0u;
        if (DIR == 1 && partial && lvl == 0) {  // alignerExhaustive.cpp:217-221 (-i): nothing starts here, nothing to pay
            const bool none = rl32(first_zero, 0) == 0 && !(rl32(end_here ? 1u : 0u, 0));
            if (none) { wave_sync(); return 0; }
        }
        const bool valid = has_rec && (uint32_t)c < first_zero;
        const bool fwd = (sl.x & fbit) != 0;
        const uint32_t len = valid ? sl.y : 0;
        const uint32_t fw = sl.z, fo = (sl.w & BGR_SLOT_FO_MASK) + (fwd ? 0u : len);
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
        } else {
            const uint32_t rl = L - pos - K1;
            fits = ext >= rl;
            n = fits ? rl : ext;
            ustart = K1;
            rstart = pos + K1;
        }
        {
            const uint32_t nx = canon ? m0.y : m0.z;  // next half | canonical << 28 (graph_layout.h nx0 / nx1)
            nrec = nx & BGR_HNONE;
            ncanon = (nx & BGR_H_CANON) != 0;
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

// ---- the same search, memoised on the node (the last pass) ----------------------------------------------------
// The reference's recursion (alignerExhaustive.cpp:61-259) re-explores a node -- (overlap k-mer, read position) = (half, strand, position) here --
// once per way of getting there.  On a compacted de Bruijn graph that is harmless; on a unitig set that duplicates its own k-mers (a homopolymer
// on both strands, copied unitigs) the ways multiply and the recursion is exponential (150 s for one 79-base read in the compiled reference).
// What a call returns depends on the node and its budget only:
//   f(node) = min over the candidates c in slot order, first minimum kept (strict `<`), of miss_c [+ f(child_c) when the walk goes on]
//   call(node, b) = f(node) if f(node) <= b, else "none" (b + 1 in the reference) -- a candidate is explored only if miss_c < best so far
//                   (<= b at first), its child is called with b - miss_c (:85,:134,:185,:243), and which candidate wins does not depend on b.
// So every call is remembered per read: {node -> exact f} or {node -> "none within bmax"}; a later call with b <= bmax is answered from the table,
// one with a larger budget computes again (at most budget + 1 times per node).  Nodes <= positions x halves x 2: polynomial.  A position strictly
// advances with every unitig taken (a unitig has at least k bases), so the recursion never meets a node that is still being computed.
// The walk is read off afterwards from the winners kept in the table.  Table: open addressing in the wave's HBM scratch, entries of one read
// carry its generation number; a table that fills up returns EXH_OVERFLOW and the host runs the read again with a larger one (capi.hip).
// Frame (32 u32, HBM): [0] half  [1] pos  [2] budget of this call  [3] cursor | ncand<<8 | scored<<16 | canon<<17  [4] best total so far
//                      [5] its slot  [6] table slot  [8+5c..] candidate c: sid, next half, ext, miss, fits | next canon<<1
// Entry (8 u32): [0] half | canon<<28 | DIR<<29 | 1<<31  [1] pos  [2] generation  [3] 1 = exact  [4] f  [5] bmax  [6] winning slot
#define MF_WORDS BGR_MEMO_FRAME_WORDS
#define MM_WORDS BGR_MEMO_ENTRY_WORDS
#define EXH_NONE 0xFFFFFFFEu
struct MemoTab {
    uint32_t* E;
    uint32_t mask, gen, used, limit;
};
__device__ __forceinline__ uint32_t memo_find(const MemoTab& t, uint32_t k0, uint32_t pos, bool* found) {
    uint32_t h = (k0 * 0x9E3779B1u) ^ (pos * 0x85EBCA6Bu);
    h = (h ^ (h >> 15)) & t.mask;
    for (;;) {  // (the table is never full: `limit` < its size)
        const uint32_t* e = t.E + (size_t)h * MM_WORDS;
        if (e[2] != t.gen) { *found = false; return h; }
        if (e[0] == k0 && e[1] == pos) { *found = true; return h; }
        h = (h + 1) & t.mask;
    }
}

template <int DIR>
__device__ __forceinline__ uint32_t exh_memo(const BgrDeviceGraph& g, const u64* CMP, const u64* NM, bool useN, uint32_t L, uint32_t K1,
                                             uint32_t a_rec, bool a_canon, uint32_t a_pos, uint32_t budget, bool partial,
                                             uint32_t* FR, uint32_t max_frames, MemoTab& mt, int32_t* CUR, int32_t* BEST, uint32_t* best_n, int lane) {
    // returns the best total (budget + 1 if none; EXH_OVERFLOW: the table is full); the best walk's ints are BEST[0..*best_n) in output order
    *best_n = 0;
    const uint32_t h0 = half_handle(g, a_rec, a_canon, DIR == 0);
    const uint32_t kdir = ((uint32_t)DIR << 29) | (1u << 31);
    int depth = 0;
    uint32_t ret = EXH_NONE;
    bool have_ret = false;
    if (lane == 0) { FR[0] = h0; FR[1] = a_pos; FR[2] = budget; FR[3] = a_canon ? (1u << 17) : 0u; }
    wave_sync();
    while (depth >= 0) {
        uint32_t* F = FR + (size_t)depth * MF_WORDS;
        const uint32_t hnd = F[0], pos = F[1], b = F[2];
        uint32_t ctl = F[3];
        if (!((ctl >> 16) & 1u)) {
            // ---- first visit: base cases, the table, then score the candidates once -----------------------------
            const bool end_here = (DIR == 0) ? (pos == 0) : (L - pos - K1 == 0);
            if (end_here) { ret = 0; have_ret = true; --depth; continue; }  // (:64,:112,:159,:210: nothing left to pay)
            if (hnd == BGR_HNONE) {  // getBegin / getEnd return an empty list
                if (DIR == 1 && depth == 0 && partial) return 0;  // alignerExhaustive.cpp:217-221 (-i)
                ret = EXH_NONE; have_ret = true; --depth;
                continue;
            }
            const bool canon = (ctl >> 17) & 1u;
            const uint32_t k0 = hnd | (canon ? 1u << 28 : 0u) | kdir;
            bool found;
            const uint32_t slot = memo_find(mt, k0, pos, &found);
            uint32_t* E = mt.E + (size_t)slot * MM_WORDS;
            if (found) {
                const uint32_t exact = E[3], val = E[4], bmax = E[5];
                if (exact) { ret = val <= b ? val : EXH_NONE; have_ret = true; --depth; continue; }
                if (b <= bmax) { ret = EXH_NONE; have_ret = true; --depth; continue; }
            } else {
                if (mt.used >= mt.limit) { wave_sync(); return EXH_OVERFLOW; }
                ++mt.used;
                if (lane == 0) { E[0] = k0; E[1] = pos; E[2] = mt.gen; E[3] = 0; E[4] = 0; E[5] = 0; E[6] = 0; }
            }
            const Scored sc = score_candidates<DIR>(g, CMP, NM, useN, L, K1, hnd, canon, pos, lane);
            const uint32_t ncand = (uint32_t)sc.first_zero;
            if ((lane & 15) == 0 && (lane >> 4) < sc.first_zero) {
                uint32_t* C = F + 8 + 5 * (lane >> 4);
                C[0] = (uint32_t)sc.sid; C[1] = sc.nrec; C[2] = sc.ext; C[3] = sc.cnt; C[4] = sc.info;
            }
            if (DIR == 1 && depth == 0 && partial && ncand == 0) { wave_sync(); return 0; }
            ctl = (ctl & (1u << 17)) | (1u << 16) | (ncand << 8);
            if (lane == 0) { F[3] = ctl; F[4] = EXH_NONE; F[5] = 0; F[6] = slot; }
            wave_sync();
        }
        uint32_t best = F[4], bslot = F[5];
        uint32_t cur = ctl & 0xFFu;
        const uint32_t ncand = (ctl >> 8) & 0xFFu;
        if (have_ret) {  // the child of candidate cur - 1 has come back
            have_ret = false;
            if (ret != EXH_NONE) {
                const uint32_t total = F[8 + 5 * (cur - 1) + 3] + ret;
                if (total < best) { best = total; bslot = cur - 1; }
            }
        }
        bool descended = false;
        while (cur < ncand) {
            const uint32_t* C = F + 8 + 5 * cur;
            const uint32_t c = cur++;
            const uint32_t miss = C[3], info = C[4];
            if (miss > b || miss >= best) continue;  // the reference explores a candidate only if miss < minMiss (errors + 1 at first)
            if (info & 1u) { best = miss; bslot = c; continue; }  // the walk ends inside this unitig
            if ((uint32_t)depth + 1 >= max_frames) { wave_sync(); return EXH_OVERFLOW; }  // (cannot happen: the frames are sized for |read| - (k-1) + 3 levels)
            const uint32_t ext = C[2], nrec = C[1];
            wave_sync();
            if (lane == 0) {
                F[3] = (ctl & ~0xFFu) | cur; F[4] = best; F[5] = bslot;
                uint32_t* N = F + MF_WORDS;
                N[0] = nrec;
                N[1] = (DIR == 0) ? pos - ext : pos + ext;
                N[2] = b - miss;
                N[3] = (info & 2u) ? (1u << 17) : 0u;
            }
            wave_sync();
            ++depth;
            descended = true;
            break;
        }
        if (descended) continue;
        // ---- every candidate seen: remember the answer ------------------------------------------------------------
        const uint32_t slot = F[6];
        wave_sync();
        if (lane == 0) {
            uint32_t* E = mt.E + (size_t)slot * MM_WORDS;
            if (best <= b) { E[3] = 1; E[4] = best; E[6] = bslot; }
            else E[5] = b;
        }
        wave_sync();
        ret = best <= b ? best : EXH_NONE;
        have_ret = true;
        --depth;
    }
    if (ret == EXH_NONE || ret > budget) return budget + 1;
    // ---- read the walk off: every node on it has its exact entry (its call's budget covered its f) -----------------
    {
        uint32_t hnd = h0, pos = a_pos, d = 0;
        bool canon = a_canon;
        for (;;) {
            const bool end_here = (DIR == 0) ? (pos == 0) : (L - pos - K1 == 0);
            if (end_here) {
                wave_sync();
                // left: checkBeginExhaustive pushes 0 only at the top (:159 vs :112); right: every depth pushes 0 (:64,:210)
                if (DIR == 0) {
                    for (uint32_t j = lane; j < d; j += 64) BEST[j] = CUR[d - 1 - j];  // far -> near
                    *best_n = d;
                    if (d == 0) { if (lane == 0) BEST[0] = 0; *best_n = 1; }
                } else {
                    for (uint32_t j = lane; j < d; j += 64) BEST[j] = CUR[j];          // near -> far
                    if (lane == 0) BEST[d] = 0;
                    *best_n = d + 1;
                }
                break;
            }
            bool found;
            const uint32_t slot = memo_find(mt, hnd | (canon ? 1u << 28 : 0u) | kdir, pos, &found);
            const uint32_t bs = mt.E[(size_t)slot * MM_WORDS + 6];
            const Scored sc = score_candidates<DIR>(g, CMP, NM, useN, L, K1, hnd, canon, pos, lane);
            const int bl = (int)bs * 16;
            const int32_t sid = (int32_t)rl32((uint32_t)sc.sid, bl);
            const uint32_t ext = rl32(sc.ext, bl), nrec = rl32(sc.nrec, bl), info = rl32(sc.info, bl);
            if (info & 1u) {
                wave_sync();
                // fitting: left emits the offset in the last unitig (ext-pos, :126,:175), right |readLeft|+k-1 (:99,:231)
                if (DIR == 0) {  // [offset, this (farthest) unitig, ..., nearest unitig]
                    for (uint32_t j = lane; j < d; j += 64) BEST[2 + j] = CUR[d - 1 - j];
                    if (lane == 0) { BEST[0] = (int32_t)(ext - pos); BEST[1] = sid; }
                } else {         // [nearest ... this (farthest) unitig, end offset]
                    for (uint32_t j = lane; j < d; j += 64) BEST[j] = CUR[j];
                    if (lane == 0) { BEST[d] = sid; BEST[d + 1] = (int32_t)(L - pos); }
                }
                *best_n = d + 2;
                break;
            }
            if (lane == 0) CUR[d] = sid;
            ++d;
            pos = (DIR == 0) ? pos - ext : pos + ext;
            hnd = nrec;
            canon = (info & 2u) != 0;
        }
        wave_sync();
    }
    return ret;
}

// alignReadExhaustive (alignerExhaustive.cpp:35-58): every read position is an anchor candidate
// (getListOverlap, aligner.cpp:318-342, keeps them all); per anchor the best left walk with budget m, then
// the best right walk with what is left; no reverse-complement retry.  Only position 0 and positions whose
// (k-1)-mer is an overlap of the graph can succeed (anywhere else getEnd() is empty), so the position scan
// is the same lane-parallel membership test as in the greedy kernel.
// DEEP (the last pass): the search state (OUT | CUR | BEST | frames | table of remembered calls) of every wave lives in HBM (io.deep_scratch)
// instead of LDS, frames sized for the worst case, so neither the depth of the search nor the read length is bounded by LDS, and the search is
// exh_memo: polynomial on any unitig set.  A read whose table fills up is listed (io.ovf_list, count at cursor[io.ovf_ctr]) for a run with a larger one.
template <bool STAGE, bool DEEP>
__global__ void __launch_bounds__(1024, DEEP ? BGR_EXH_DEEP_OCC : BGR_EXH_OCC) bgr_align_exhaustive_kernel(BgrDeviceGraph g, BatchIO io, KernelParams prm) {
    extern __shared__ u64 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const uint32_t W = io.words_per_read;
    const uint32_t K1 = g.k - 1;
    uint32_t ktab_words;
    const uint32_t* ktab = block_prologue<STAGE>(g, lds, &ktab_words);
    // per wave: FW3 | FWQ | RCW | NM | OUT | CUR | BEST | frames   (the last four in HBM when DEEP)
    const uint32_t per_wave_words = DEEP ? 4 * W : 4 * W + 3 * (io.path_cap / 2) + (io.frames_per_wave * FR_WORDS) / 2;
    u64* FW3 = lds + 64 + ktab_words + (u64)wave * per_wave_words;
    u64* FWQ = FW3 + W;
    u64* RCW = FWQ + W;
    u64* NM = RCW + W;
    int32_t* OUT = DEEP ? reinterpret_cast<int32_t*>(io.deep_scratch + (u64)(blockIdx.x * waves + wave) * io.deep_stride)
                        : reinterpret_cast<int32_t*>(NM + W);
    int32_t* CUR = OUT + io.path_cap;
    int32_t* BEST = CUR + io.path_cap;
    uint32_t* FR = reinterpret_cast<uint32_t*>(BEST + io.path_cap);
    MemoTab mt;
    mt.E = nullptr; mt.mask = 0; mt.gen = 0; mt.used = 0; mt.limit = 0;
    if (DEEP) {  // (deep_scratch_words, align_kernels.h)
        mt.E = FR + (size_t)io.frames_per_wave * MF_WORDS;
        mt.mask = io.deep_memo_cap - 1;
        mt.limit = io.deep_memo_cap - (io.deep_memo_cap >> 2);
        for (uint32_t j = lane; j < io.deep_memo_cap; j += 64) mt.E[(size_t)j * MM_WORDS + 2] = 0;  // no entry carries generation 0
        wave_sync();
    }

    uint32_t c_reads = 0, c_al = 0, c_na = 0;
    unsigned long long c_ov = 0;
    uint32_t chunk_pos = 0, chunk_end = 0;
    const uint32_t m = prm.max_mismatch;

    // pass 2 maps only the reads that pass 1 listed as needing a deeper stack (count left in cursor[subset_ctr] by the pass before)
    const uint32_t total = io.subset ? io.cursor[io.subset_ctr] : io.n_reads;
    for (uint32_t it = blockIdx.x * waves + wave; it < total; it += gridDim.x * waves) {
        ++mt.gen;      // (what the searches of one read remember holds for all of its anchors: f depends on the node and the read only)
        mt.used = 0;
        const uint32_t r = io.subset ? io.subset[it] : it;
        const u64 off = io.read_offs[r];
        const uint32_t L = (uint32_t)(io.read_offs[r + 1] - off);
        const bool hasN = load_packed(io, r, off, L, W, FW3, NM, lane);
        if (hasN) derive_streams(L, W, K1, FW3, FWQ, RCW, NM, lane);
        const u64* ROLL = hasN ? FWQ : FW3;  // the rolling `num` stream
        uint32_t p_n = 0;
        const uint32_t npos = L >= K1 ? L - K1 + 1 : 0;
        bool done = false, overflow = false;
        // (minimizer filter in front of a key table that is not staged: a scan step covers 65 - w positions, device_common.h)
        const uint32_t mmx_w = (BGR_EXH_MMX && !STAGE && g.bloom && g.filter_kind == BGR_FILTER_MINIMIZER) ? K1 + 1 - BGR_MMX_BASES : 0u;
        const uint32_t scan_step = mmx_w ? 65 - mmx_w : 64;
        for (uint32_t base = 0; base < npos && !done && !overflow; base += scan_step) {
            const uint32_t i = base + lane;
            const bool valid = i < npos && (uint32_t)lane < scan_step;
            u64 num = 0, win = 0;
            if (valid || (mmx_w && i + BGR_MMX_BASES <= L)) win = lds_win32(ROLL, i);
            if (valid) num = win >> (64 - 2 * K1);             // the rolling `num` (aligner.cpp:321,334)
            const u64 rc = rcb_fast(num, K1);                  // getBegin/getEnd use rcb(num) (aligner.cpp:149,211)
            // (the key is the canonical form of the window of ROLL itself, so its 16-mers are the windows' leading 16 bases)
            const uint32_t mblock = (!STAGE && mmx_w) ? scan_mblock(g, win, i + BGR_MMX_BASES <= L, mmx_w) : 0u;
            const uint32_t idx = find_key<!STAGE>(g, ktab, num < rc ? num : rc, valid, mblock);
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
                    eb = DEEP ? exh_memo<0>(g, FW3, NM, hasN, L, K1, a_rec, a_canon, a_pos, m, false, FR, io.frames_per_wave, mt, CUR, BEST, &nl, lane)
                              : exh_search<0>(g, FW3, NM, hasN, L, K1, a_rec, a_canon, a_pos, m, false, FR, io.frames_per_wave, CUR, BEST, &nl, lane, io.search_iters);
                    if (eb == EXH_OVERFLOW) { overflow = true; break; }
                    if (eb > m) continue;
                    for (uint32_t j = lane; j < nl; j += 64) OUT[j] = BEST[j];
                }
                wave_sync();
                // position 0 is tried whatever its (k-1)-mer: when that is no overlap of the graph getBegin() is empty, and only an
                // empty right side or -i can make the anchor succeed (alignerExhaustive.cpp:206-221): no search either
                if (a_rec == BGR_NONE && !prm.partial && L - a_pos - K1 != 0) continue;
                const uint32_t ee = DEEP ? exh_memo<1>(g, FW3, NM, hasN, L, K1, a_rec, a_canon, a_pos, m - eb, prm.partial != 0, FR, io.frames_per_wave, mt, CUR, BEST, &nr, lane)
                                         : exh_search<1>(g, FW3, NM, hasN, L, K1, a_rec, a_canon, a_pos, m - eb, prm.partial != 0, FR, io.frames_per_wave, CUR, BEST, &nr, lane, io.search_iters);
                if (ee == EXH_OVERFLOW) { overflow = true; break; }
                if (ee > m - eb) continue;
                for (uint32_t j = lane; j < nr; j += 64) OUT[nl + j] = BEST[j];
                p_n = nl + nr;
                done = true;
                break;
            }
        }
        wave_sync();
        if (overflow) {  // leave this read to the next pass (the last one: to its next run with a larger table); nothing is written or counted for it here
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

// ============================== exhaustive, several reads per wavefront ===================================
// The level search (exh_dp) gives a wave ONE read, and after de-duplication a level of the walk rarely holds more than one
// node: 4 candidate slots x 4 chunk lanes = 16 of the 64 lanes work.  Here a wave takes 64 / GL reads, GL lanes each (GL = 8:
// eight reads, 4 slots x 2 chunk lanes; written for GL = 16), and runs
// their searches side by side, for the shape nearly every read has: every level has exactly ONE node (all candidates that go
// on lead to the same (record, position, strand): bubbles that close again), at most 8 or 16 levels per side (chosen per launch), no N, and
// the first anchor that can succeed does.  Per side: a forward sweep scores the node's <= 4 candidates per level (kept:
// id, offset, mismatches, fits, alive), a backward sweep settles cost(level) = first minimum over the slots of mismatches
// [+ cost(level + 1) when the walk goes on] -- the value and the choice of the reference's recursion
// (alignerExhaustive.cpp:61-259; see exh_dp) -- and the walk is read off.  Anything else (a level with two nodes, a failing
// first anchor, N, -i) is listed for bgr_align_exhaustive_dp_kernel / the depth-first passes, which map it from scratch.
#ifndef BGR_X4_OCC
#define BGR_X4_OCC 6
#endif
// per level of a walk, nine words: [0..3] the slots' path ints (+-id), [4..7] per slot miss (8 bits; the pass takes budgets <= 254, a dearer
// candidate is dead anyway) | fits << 8 | alive << 9 | aux << 10 (aux: the path int a walk ending in this slot emits, < 2^22: the host keeps
// graphs with longer unitigs off this pass), [8] node flags | chosen slot << 8.  Round 3 kept 16 words (64 B) per level; at nine a wave's
// eight reads take 6.4 instead of 9.9 KB with 16 levels per side, and a CU holds 24 instead of 16 waves (the register limit)
#define X4_LV_WORDS BGR_X4_LEVEL_WORDS
#define X4_END 1u
#define X4_INF 0xFFFFu

// One side of the search for the wave's reads (act = the group takes part).  DIR 0: left of the anchor (exL), DIR 1: right
// (exR).  On return, for the groups that took part: *cost = best total (X4_INF: none within the budget; the caller compares
// with its budget), *n_out ints written to OUTG[o_off ...] in output order, *fb = the search left the shape this kernel
// handles (the read goes on the list).
template <int DIR, bool NEAR, int GL, int XLV>
__device__ __forceinline__ void x4_search(const BgrDeviceGraph& g, const u64* FW, uint32_t L, uint32_t K1, uint32_t act, uint32_t a_rec, uint32_t a_canon,
                                          uint32_t a_pos, uint32_t budget, uint32_t* LVT, int32_t* OUTG, uint32_t o_off, int lane, uint32_t* cost_o,
                                          uint32_t* n_out, uint32_t* fb_o) {
    constexpr uint32_t QL = GL / 4, XL = XLV, RPW = 64 / GL;  // lanes per candidate slot; levels per side; reads per wave
    constexpr uint32_t NH = XL / GL;  // the walk is read off GL levels at a time
    constexpr uint32_t Q0 = GL == 16 ? 0x1111u : 0x55u, GM = GL == 16 ? 0xFFFFu : 0xFFu;  // the slots' first lanes / all lanes of a group
    const uint32_t c = ((uint32_t)lane / QL) & 3u, q = (uint32_t)lane % QL, sub = (uint32_t)lane % GL;
    const uint32_t gl0 = (uint32_t)lane & (64u - GL);  // first lane of the group
    uint32_t fwd = act, fb = 0, lvl = 0, nlev = 0, prefix = 0;
    // rec: the HANDLE of the half the level's candidates come from (graph_layout.h); the anchor a_rec is a key entry
    uint32_t pos = a_pos, canon = a_canon, rec = G4_REC_MASK;
    if (act && a_rec != G4_REC_MASK) rec = half_handle(g, a_rec, a_canon != 0, DIR == 0);
    // ---- forward: one node per level ----
    for (;;) {
        if (fwd && lvl >= XL) { fb = 1; fwd = 0; }
        const uint32_t end_here = (fwd && ((DIR == 0) ? (pos == 0) : (L - pos - K1 == 0))) ? 1u : 0u;
        if (end_here) {  // left: the read's first base is reached; right: nothing is left of the read
            if (sub == 0) LVT[lvl * X4_LV_WORDS + 8] = X4_END;
            nlev = lvl + 1;
            fwd = 0;
        }
        if (!__any(fwd != 0)) break;
        uint4 sl = make_uint4(0, 0, 0, 0), m0 = make_uint4(0, 0, 0, 0);
        const bool reads = fwd && rec != G4_REC_MASK;
        if (reads) {
            const uint4* sp = reinterpret_cast<const uint4*>(g.recs) + (size_t)(rec + c) * 2;
            sl = sp[0];
            m0 = sp[1];
        }
        const uint32_t id = reads ? sl.x & BGR_SLOT_ID_MASK : 0u;
        const u64 lmask = __ballot(reads && (sl.w & BGR_SLOT_LAST) != 0);
        const uint32_t zb = (uint32_t)(lmask >> gl0) & Q0;
        // candidates = the half's slots up to and including the first flagged one (the reference stops at the first empty slot)
        const uint32_t first_zero = (reads && zb) ? (uint32_t)(__ffs((int)zb) - 1) / QL + 1u : 0u;
        const uint32_t valid = c < first_zero ? 1u : 0u;
        const uint32_t fwdu = (sl.x & (canon ? BGR_SLOT_F0 : BGR_SLOT_F1)) ? 1u : 0u;
        const uint32_t len = sl.y;
        const uint32_t fw = sl.z, fo = (sl.w & BGR_SLOT_FO_MASK) + (fwdu ? 0u : len);
        const uint32_t ext = len - K1;
        uint32_t fits, n, ustart, rstart, aux, npos;
        if (DIR == 0) {
            fits = ext >= pos ? 1u : 0u;
            n = fits ? pos : ext;
            ustart = fits ? ext - pos : 0;
            rstart = fits ? 0 : pos - ext;
            aux = ext - pos;   // offset in the last unitig (:126,:175)
            npos = pos - ext;
        } else {
            const uint32_t rl = L - pos - K1;
            fits = ext >= rl ? 1u : 0u;
            n = fits ? rl : ext;
            ustart = K1;
            rstart = pos + K1;
            aux = L - pos;     // |readLeft| + k-1 (:99,:231)
            npos = pos + ext;
        }
        if (!valid) n = 0;
        // at most 32 bases next to the overlap sit in the slot itself (graph_layout.h `near`): no load from seq (the exhaustive walks
        // never compare the overlap again, alignerExhaustive.cpp:99,231)
        const uint32_t both = BGR_SLOT_F0 | BGR_SLOT_F1;
        // (NEAR: graphs that live in L2/HBM; issue-bound launches over a small graph are better off without the second compare path)
        const bool near_ok = NEAR && n <= 32 && (sl.x & both) != both && !(g.flags & BGR_GF_HAS_EXC);
        uint32_t cnt = 0;
        if (NEAR && near_ok && n && q == 0) cnt = ham_near(FW, bgr_slot_near(sl.w, m0.x, m0.w), DIR == 0, n, rstart);
        for (uint32_t b = q * 32; wave_any(b < n && !near_ok); b += 32 * QL)
            if (b < n && !near_ok) cnt += ham_chunk(g, FW, nullptr, false, fw, fo + ustart + b, rstart + b, n - b);
        cnt += quad_xor1(cnt);
        if (QL == 4) cnt += quad_xor2(cnt);
        const uint32_t miss = cnt > 0xFFFFu ? 0xFFFFu : cnt;
        const uint32_t ptotal = prefix + miss;
        const uint32_t alive = (valid && ptotal <= budget) ? 1u : 0u;  // a walk through here costs at least this much
        const uint32_t need = (alive && !fits) ? 1u : 0u;
        if (fwd && q == 0) {
            uint32_t* R = LVT + lvl * X4_LV_WORDS;
            R[c] = fwdu ? id : 0u - id;
            R[4 + c] = (miss > 255u ? 255u : miss) | (fits << 8) | (alive << 9) | (aux << 10);
            if (c == 0) R[8] = 0;
        }
        // the candidates that go on must all reach the same node
        const u64 nmask = __ballot(need && q == 0 && fwd);
        const uint32_t nb = (uint32_t)(nmask >> gl0) & Q0;
        const uint32_t src = gl0 | (nb ? (uint32_t)(__ffs((int)nb) - 1) : 0u);
        const uint32_t kpk = (canon ? m0.y : m0.z) & (G4_REC_MASK | G4_CANON);  // next half | canonical << 28 (graph_layout.h nx0 / nx1)
        const uint32_t k_rec = lane_get(kpk, src), k_pos = lane_get(npos, src);
        const u64 dmask = __ballot(need && fwd && (kpk != k_rec || npos != k_pos));
        uint32_t pmin = need ? ptotal : 0xFFFFFFFFu;
        uint32_t o = GL == 16 ? row_ror4(pmin) : quad_xor2(pmin);
        pmin = o < pmin ? o : pmin;
        o = GL == 16 ? row_ror8(pmin) : half_row_mirror(pmin);
        pmin = o < pmin ? o : pmin;
        if (fwd) {
            nlev = lvl + 1;
            if ((uint32_t)(dmask >> gl0) & GM) { fb = 1; fwd = 0; }  // a level with two nodes
            else if (!nb) fwd = 0;                                                      // every candidate ends here or is too dear
            else { prefix = pmin; pos = k_pos; rec = k_rec & G4_REC_MASK; canon = (k_rec >> 28) & 1u; ++lvl; }
        }
    }
    wave_sync();
    // ---- backward: cost of every level, first slot on ties ----
    const uint32_t ok = (act && !fb) ? 1u : 0u;
    uint32_t maxl = 0;
#pragma unroll
    for (uint32_t i = 0; i < RPW; ++i) maxl = max(maxl, rl32(ok ? nlev : 0u, (int)(GL * i)));
    uint32_t cnext = X4_INF;
    for (int l = (int)maxl - 1; l >= 0; --l) {
        const uint32_t in = (ok && (uint32_t)l < nlev) ? 1u : 0u;
        uint32_t key = 0xFFFFFFFFu, flags = 0;
        if (in) {
            const uint32_t* R = LVT + (uint32_t)l * X4_LV_WORDS;
            flags = R[8] & 0xFFu;
            if (sub < 4) {
                const uint32_t pk = R[4 + sub];
                uint32_t total = X4_INF;
                if (pk & (1u << 9)) {
                    total = (pk & 0xFFu) + ((pk & (1u << 8)) ? 0u : cnext);
                    if (total > X4_INF) total = X4_INF;
                }
                key = total << 2 | sub;
            }
        }
        uint32_t o = quad_xor1(key);
        key = o < key ? o : key;
        o = quad_xor2(key);
        key = o < key ? o : key;
        key = lane_get(key, gl0);
        if (in) {
            cnext = (flags & X4_END) ? 0u : (key >> 2);
            if (sub == 0) LVT[(uint32_t)l * X4_LV_WORDS + 8] = flags | ((key & 3u) << 8);
        }
    }
    wave_sync();
    // ---- read the walk off: level j's chosen slot, down to the first level that ends the walk (lane `sub` takes the levels
    // sub, sub + GL, ...) ----
    uint32_t sidv[NH], auxvv[NH], isendv[NH];
    uint32_t ebits = 0;  // bit j = level j ends the walk
#pragma unroll
    for (uint32_t h = 0; h < NH; ++h) {
        const uint32_t j = sub + h * GL;
        uint32_t endj = 0;
        sidv[h] = 0; auxvv[h] = 0; isendv[h] = 0;
        if (ok && j < nlev) {
            const uint32_t* R = LVT + j * X4_LV_WORDS;
            const uint32_t fc = R[8], a = fc >> 8;
            isendv[h] = fc & X4_END;
            const uint32_t pk = R[4 + a];
            endj = (isendv[h] || (pk & (1u << 8))) ? 1u : 0u;
            sidv[h] = R[a];
            auxvv[h] = pk >> 10;
        }
        const u64 emask = __ballot(endj != 0);
        ebits |= ((uint32_t)(emask >> gl0) & GM) << (h * GL);
    }
    const uint32_t d = ebits ? (uint32_t)(__ffs((int)ebits) - 1) : 0u;  // depth of the level that ends the walk
    uint32_t d_end = 0;                                                 // ... by reaching the read's end (no unitig taken there)
#pragma unroll
    for (uint32_t h = 0; h < NH; ++h) {
        const uint32_t v = lane_get(isendv[h], gl0 | (d % GL));
        if (d / GL == h) d_end = v;
    }
    uint32_t n = 0;
    const uint32_t good = (ok && cnext <= budget && ebits) ? 1u : 0u;
    if (good) {
        int32_t* O = OUTG + o_off;
#pragma unroll
        for (uint32_t h = 0; h < NH; ++h) {
            const uint32_t j = sub + h * GL;
            const int32_t sid = (int32_t)sidv[h];
            if (DIR == 0) {
                // left: checkBeginExhaustive pushes 0 only at the top (:159 vs :112); else [offset, farthest unitig, ..., nearest]
                if (d_end) { n = d == 0 ? 1u : d; if (d == 0) { if (j == 0) O[0] = 0; } else if (j < d) O[d - 1 - j] = sid; }
                else { n = d + 2; if (j == d) { O[0] = (int32_t)auxvv[h]; O[1] = sid; } else if (j < d) O[1 + (d - j)] = sid; }
            } else {
                // right: every depth pushes 0 at the read's end (:64,:210); else [nearest ... farthest unitig, end offset]
                if (d_end) { n = d + 1; if (j < d) O[j] = sid; if (j == d) O[d] = 0; }
                else { n = d + 2; if (j <= d) O[j] = sid; if (j == d) O[d + 1] = (int32_t)auxvv[h]; }
            }
        }
    }
    wave_sync();
    *cost_o = good ? cnext : X4_INF;
    *n_out = n;
    *fb_o = fb;
}

// XLV = levels (unitigs) of a walk per side the level table holds: 8 for short walks (less LDS per read: more waves), else 16
template <bool STAGE, int GL, int XLV>
__global__ void __launch_bounds__(1024, BGR_X4_OCC) bgr_align_exhaustive4_kernel(BgrDeviceGraph g, BatchIO io, KernelParams prm) {
    constexpr uint32_t RPW = 64 / GL, XL = XLV;  // reads per wave; levels per side
    extern __shared__ u64 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const uint32_t W = io.words_per_read;  // <= 16 (checked by the host)
    const uint32_t K1 = g.k - 1;
    // (minimizer filter in front of a key table that is not staged: a scan step covers 65 - w positions, device_common.h)
    const uint32_t mmx_w = (BGR_EXH_MMX && !STAGE && g.bloom && g.filter_kind == BGR_FILTER_MINIMIZER) ? K1 + 1 - BGR_MMX_BASES : 0u;
    const bool wide_scan = io.wide_scan && mmx_w && scan_mblock_is_wide(mmx_w);   // (k = 31 / 32: the filter block in all 64 lanes, scan_mblock_wide)
    const uint32_t scan_step = (mmx_w && !wide_scan) ? 65 - mmx_w : 64;
    unsigned long long* wg_counts = wg_counts_init(lds);
    task_stock_init(lds, (io.n_reads + RPW - 1) / RPW);
#ifdef BGR_PHASE_TIMING
    const unsigned long long wt0 = wall_clock64();
#endif
    uint32_t ktab_words;
    const uint32_t* ktab = block_prologue<STAGE>(g, lds, &ktab_words);
#ifdef BGR_PHASE_TIMING
    const unsigned long long wt1 = wall_clock64();
#endif
    // per wave: RPW x { read words W | level table XL x X4_LV_WORDS u32 | out ints 2 x (XL + 2) }  (x4_group_words, align_kernels.h)
    const uint32_t out_ints = 2 * (XL + 2);
    const uint32_t grp_words = W + (XL * X4_LV_WORDS + out_ints + 1) / 2;
    const uint32_t grp = (uint32_t)lane / GL, sub = (uint32_t)lane % GL;
    u64* WV = lds + 64 + ktab_words + (u64)wave * (RPW * grp_words);
    u64* F = WV + grp * grp_words;
    uint32_t* LVT = reinterpret_cast<uint32_t*>(F + W);
    int32_t* OUTG = reinterpret_cast<int32_t*>(LVT + XL * X4_LV_WORDS);
    const uint32_t m = prm.max_mismatch;

    uint32_t c_al = 0, c_na = 0;
    unsigned long long c_ov = 0;
    // (the wave's first arena chunk is its own by number: the host starts the cursor behind them, see bgr_align_greedy_multi_kernel)
    uint32_t chunk_pos = (uint32_t)(blockIdx.x * waves + wave) * io.arena_chunk, chunk_end = chunk_pos + io.arena_chunk;

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
            fast = io.ascii ? 1u : ((io.hasn[r >> 5] >> (r & 31)) & 1u) ^ 1u;
            if (L <= K1) fast = 0;  // (a read of k-1 bases or fewer: the general kernel)
            if (((L + 31) >> 5) >= W) fast = 0;  // (or too long for one lane per word)
        }
        if (io.ascii) {  // no pre-pass, no planes (round 5): the words straight from the read's characters; a read with an N goes on the list
            if (stage_group_ascii<GL>(io, have ? ascii_start(io, r, off) : 0, L, W, fast, F, sub)) fast = 0;
        } else
        for (uint32_t j = sub; j < W; j += GL) {
            u64 f = 0;
            if (fast && j < ((L + 31) >> 5)) f = io.fw3[packed_word_offset(off, r) + j];
            F[j] = f;
        }
        wave_sync();
        // ---- the first position that can anchor the read (getListOverlap keeps every position, aligner.cpp:318-342; only
        // position 0 and overlap (k-1)-mers of the graph can succeed): position 0 when its k-mer is an overlap, else the first hit
        uint32_t a_pos = 0, a_rec = BGR_NONE;
        for (uint32_t qq = 0; qq < RPW; ++qq) {
            if (!rl32(fast, (int)(GL * qq))) continue;
#ifdef BGR_PHASE_TIMING
            if (prm.debug_stop == 1) continue;  // 1 = stops behind the staging of the reads
#endif
            const uint32_t Lq = rl32(L, (int)(GL * qq));
            const u64* A = WV + qq * grp_words;
            const uint32_t npos = Lq - K1 + 1;
            for (uint32_t base = 0; base < npos; base += scan_step) {
                const uint32_t i = base + (uint32_t)lane;
                const bool valid = i < npos && (uint32_t)lane < scan_step;
                u64 num = 0, win = 0;
                if (valid || (mmx_w && i + BGR_MMX_BASES <= Lq)) win = lds_win32(A, i);
                if (valid) num = win >> (64 - 2 * K1);
                const u64 rcn = rcb_fast(num, K1);
                const uint32_t mblock = (STAGE || !mmx_w) ? 0u
                                      : wide_scan ? scan_mblock_wide(g, win, i + BGR_MMX_BASES <= Lq, i + 2 * BGR_MMX_BASES <= Lq, mmx_w)
                                                  : scan_mblock(g, win, i + BGR_MMX_BASES <= Lq, mmx_w);
                uint32_t idx = find_key<!STAGE>(g, ktab, num < rcn ? num : rcn, valid, mblock);
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
#ifdef BGR_PHASE_TIMING
        if (prm.debug_stop == 2) a_rec = BGR_NONE;  // 2 = stops behind the anchor scan (every read "without anchor")
#endif
        const uint32_t anchored = (fast && a_rec != BGR_NONE) ? 1u : 0u;
        // left side: [0] at position 0 (no search), else the search with the whole budget
        uint32_t eb = 0, nl = 0, fbl = 0;
        {
            const uint32_t actl = (anchored && a_pos != 0) ? 1u : 0u;
            uint32_t cl = 0, nll = 0;
            x4_search<0, !STAGE, GL, XLV>(g, F, L, K1, actl, a_rec & G4_REC_MASK, (a_rec >> 28) & 1u, a_pos, m, LVT, OUTG, 0, lane, &cl, &nll, &fbl);
            if (actl) { eb = cl; nl = nll; }
            else if (anchored) { if (sub == 0) OUTG[0] = 0; nl = 1; }
        }
        wave_sync();
        uint32_t ee = 0, nr = 0, fbr = 0;
        {
            const uint32_t actr = (anchored && !fbl && eb <= m) ? 1u : 0u;
            uint32_t cr = 0, nrr = 0;
            x4_search<1, !STAGE, GL, XLV>(g, F, L, K1, actr, a_rec & G4_REC_MASK, (a_rec >> 28) & 1u, a_pos, actr ? m - eb : 0u, LVT, OUTG, nl, lane, &cr, &nrr, &fbr);
            if (actr) { ee = cr; nr = nrr; } else ee = X4_INF;
        }
        // 0 = aligned; 2 = not aligned for sure (no overlap (k-1)-mer anywhere in the read: every position fails); 4 = the list
        uint32_t outcome = 4;
        if (fast && !anchored) outcome = 2;
        else if (anchored && !fbl && !fbr && eb <= m && ee != X4_INF && eb + ee <= m) outcome = 0;
        const uint32_t aligned = outcome == 0 ? 1u : 0u;
        const uint32_t p_n = aligned ? nl + nr : 0;
        uint32_t tot = 0, before = 0;
#pragma unroll
        for (uint32_t i = 0; i < RPW; ++i) {
            const uint32_t ni = rl32(p_n, (int)(GL * i));
            if (grp > i) before += ni;
            tot += ni;
        }
        if (tot > chunk_end - chunk_pos) {
            const uint32_t want = tot > io.arena_chunk ? tot : io.arena_chunk;
            uint32_t got = 0;
            if (lane == 0) got = io.arena_own + atomicAdd(io.cursor, want);
            chunk_pos = rl32(got, 0);
            chunk_end = chunk_pos + want;
        }
        const uint32_t gbase = chunk_pos + before;
        const bool room = chunk_pos + tot <= io.arena_cap;
        chunk_pos += tot;
        for (uint32_t j = sub; j < p_n; j += GL)
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
        for (int gq = 0; gq < (int)RPW; ++gq)
            if ((fin >> (GL * gq)) & 1) c_ov += rl32(npos_g, (int)(GL * gq));  // overlaps += listOverlap.size() (alignerExhaustive.cpp:38)
        wave_sync();
    }
#ifdef BGR_PHASE_TIMING
    if (io.wave_times && lane == 0) {
        unsigned long long* w = io.wave_times + 4ull * (blockIdx.x * (blockDim.x >> 6) + wave);
        w[0] = wt0; w[1] = wt1; w[2] = w[3] = wall_clock64();
    }
#endif
    wg_counts_flush(io, wg_counts, lane, c_al + c_na, 0, c_al, c_na, c_ov);
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
    uint32_t ktab_words;
    const uint32_t* ktab = block_prologue<STAGE>(g, lds, &ktab_words);
    // per wave: FW3 | FWQ | RCW | NM | OUT | BEST | tables (32 + levels * 52 + levels words, see exh_dp)
    const uint32_t table_words = 32 + io.frames_per_wave * (DP_LEVEL_WORDS + 1);
    const uint32_t per_wave_words = 4 * W + 2 * (io.path_cap / 2) + ((table_words + 3) / 4) * 2;  // whole 16-byte units
    u64* FW3 = lds + 64 + ktab_words + (u64)wave * per_wave_words;
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
        // (minimizer filter in front of a key table that is not staged: a scan step covers 65 - w positions, device_common.h)
        const uint32_t mmx_w = (BGR_EXH_MMX && !STAGE && g.bloom && g.filter_kind == BGR_FILTER_MINIMIZER) ? K1 + 1 - BGR_MMX_BASES : 0u;
        const uint32_t scan_step = mmx_w ? 65 - mmx_w : 64;
        for (uint32_t base = 0; base < npos && !done && !overflow; base += scan_step) {
            const uint32_t i = base + lane;
            const bool valid = i < npos && (uint32_t)lane < scan_step;
            u64 num = 0, win = 0;
            if (valid || (mmx_w && i + BGR_MMX_BASES <= L)) win = lds_win32(ROLL, i);
            if (valid) num = win >> (64 - 2 * K1);             // the rolling `num` (aligner.cpp:321,334)
            const u64 rc = rcb_fast(num, K1);                  // getBegin/getEnd use rcb(num) (aligner.cpp:149,211)
            // (the key is the canonical form of the window of ROLL itself, so its 16-mers are the windows' leading 16 bases)
            const uint32_t mblock = (!STAGE && mmx_w) ? scan_mblock(g, win, i + BGR_MMX_BASES <= L, mmx_w) : 0u;
            const uint32_t idx = find_key<!STAGE>(g, ktab, num < rc ? num : rc, valid, mblock);
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

}  // namespace

hipError_t launch_exhaustive(const BgrDeviceGraph& g, const BatchIO& io, const KernelParams& p, const LaunchCfg& cfg, hipStream_t stream) {
    if (io.exh4) {
        constexpr int GL = (int)kX4GroupLanes;
        if (io.exh4 == 8) return cfg.stage_mphf ? launch_one(bgr_align_exhaustive4_kernel<true, GL, 8>, g, io, p, cfg, stream)
                                                : launch_one(bgr_align_exhaustive4_kernel<false, GL, 8>, g, io, p, cfg, stream);
        return cfg.stage_mphf ? launch_one(bgr_align_exhaustive4_kernel<true, GL, 16>, g, io, p, cfg, stream)
                              : launch_one(bgr_align_exhaustive4_kernel<false, GL, 16>, g, io, p, cfg, stream);
    }
    if (io.deep_scratch) return launch_one(bgr_align_exhaustive_kernel<false, true>, g, io, p, cfg, stream);
    if (io.level_search) return cfg.stage_mphf ? launch_one(bgr_align_exhaustive_dp_kernel<true>, g, io, p, cfg, stream)
                                               : launch_one(bgr_align_exhaustive_dp_kernel<false>, g, io, p, cfg, stream);
    return cfg.stage_mphf ? launch_one(bgr_align_exhaustive_kernel<true, false>, g, io, p, cfg, stream)
                          : launch_one(bgr_align_exhaustive_kernel<false, false>, g, io, p, cfg, stream);
}
const void* exhaustive_kernel_fn(uint32_t which) {  // 0 depth-first, 1 level search, 2 several reads per wave
    return which == 1 ? reinterpret_cast<const void*>(&bgr_align_exhaustive_dp_kernel<false>)
         : which == 2 ? reinterpret_cast<const void*>(&bgr_align_exhaustive4_kernel<false, (int)kX4GroupLanes, 16>)
                      : reinterpret_cast<const void*>(&bgr_align_exhaustive_kernel<true, false>);
}

}  // namespace bgr
