// greedy_kernels.hip -- greedy mode (alignReadGreedy, alignerGreedy.cpp:35-57,167-364) on gfx950.
//   bgr_align_greedy_multi_kernel  sixteen reads per wavefront (lanes per read: a template parameter): the position scans one after the
//                             other on all 64 lanes, the extensions side by side, 8 lanes each; settles the common shapes, lists the rest
//   bgr_align_greedy_kernel   the general kernel: one read per wavefront, every anchor, both strands, N planes, any path length
#include "device_common.h"

namespace bgr {
namespace {

template <bool STAGE>
__global__ void __launch_bounds__(1024, BGR_GREEDY_OCC) bgr_align_greedy_kernel(BgrDeviceGraph g, BatchIO io, KernelParams prm) {
    extern __shared__ u64 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const uint32_t W = io.words_per_read;
    const uint32_t K1 = g.k - 1;
    // as the second pass behind bgr_align_greedy_multi_kernel it maps only the reads that kernel listed (count in cursor[subset_ctr])
    const uint32_t total = io.subset ? io.cursor[io.subset_ctr] : io.n_reads;
    if ((uint32_t)(blockIdx.x * waves) >= total) return;  // nothing for this workgroup (before it copies the cascade into LDS)
    uint32_t ktab_words;
    const uint32_t* ktab = block_prologue<STAGE>(g, lds, &ktab_words);
    const uint32_t per_wave_words = 4 * W + io.path_cap / 2;
    u64* FW3 = lds + 64 + ktab_words + (u64)wave * per_wave_words;
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
        // (minimizer filter in front of a key table that is not staged: a scan step covers 65 - w positions, device_common.h)
        const uint32_t mmx_w = (!STAGE && g.bloom && g.filter_kind == BGR_FILTER_MINIMIZER) ? K1 + 1 - BGR_MMX_BASES : 0u;
        const uint32_t scan_step = mmx_w ? 65 - mmx_w : 64;
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
            for (uint32_t base = 0; base < npos && !done && tried < effort; base += scan_step) {
                const uint32_t i = base + lane;
                const bool valid = i < npos && (uint32_t)lane < scan_step;
                u64 num = 0, rcn = 0, win = 0;
                if (valid || (mmx_w && i + BGR_MMX_BASES <= L)) win = lds_win32(A, i);
                if (valid) {
                    num = win >> (64 - 2 * K1);
                    rcn = plain ? rcb_fast(num, K1) : lds_win32(B, L - K1 - i) >> (64 - 2 * K1);
                }
                const u64 rep = num < rcn ? num : rcn;
                uint32_t mblock = 0;
                if (!STAGE && mmx_w) {  // (a read with an N: the key looked up need not be the canonical form of the window -- its own 16-mers then)
                    mblock = plain ? scan_mblock(g, win, i + BGR_MMX_BASES <= L, mmx_w) : bgr_mmx_block(bgr_mmx_of_key(rep, K1), g.bloom_mask);
                }
                const uint32_t idx = find_key<!STAGE>(g, ktab, rep, valid, mblock);
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
                    if (rc2 != a_rcn) {
                        const u64 key2 = a_num < rc2 ? a_num : rc2;
                        a_rec = find_key<!STAGE>(g, ktab, key2, true, (!STAGE && mmx_w) ? bgr_mmx_block(bgr_mmx_of_key(key2, K1), g.bloom_mask) : 0u);
                    }
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

// ================================= greedy, several reads per wavefront ====================================
// bgr_align_greedy_kernel above walks one read per wave: a walk step is two dependent loads (slot, bases) scored by at
// most 4 candidates x a few 32-base chunks, i.e. a handful of the 64 lanes, and per read there are ~3 such steps in a row.
// Here a wave takes 64 / GL reads, GL lanes each (GL = 4: SIXTEEN reads; round 2 started with GL = 16, four reads, and ended with 8):
// their position scans still run one after the other on all 64 lanes (a scan is lane-efficient: one (k-1)-mer per lane), then
// the extensions run side by side, GL lanes each (4 candidate slots x GL/4 chunk lanes of 32 bases; at GL = 4 one lane per slot, two
// when a half has at most two candidates), so sixteen slot/base load chains are in flight per wave and every wave instruction of a
// walk step serves sixteen reads (a quad of reads needs max-over-4 = 3.4 steps, an octet 3.7: the instructions per read nearly halve
// with every doubling).  Path ints go straight into the read's own row of the arena (kG4PathInts ints: left walk downwards from the
// middle, right walk upwards), so there are no path registers, no per-wave arena chunks and no copy when a walk ends.
//
// The reference's retry ladder (alignerGreedy.cpp:41-56: the next anchors of getNOverlap's list, then once the reverse
// complement) runs INSIDE the launch (round 3; round 2 re-launched the kernel twice over lists): an ITEM is one strand of
// one read from one scan position on.  A wave first takes its share of the batch (items = whole reads, forward strand, position
// 0); an item whose anchors fail leaves a follow-up item -- the same strand from behind the anchor tried last, or the reverse
// complement from its first position -- in the wave's own queue (a ring in HBM, written and read by this wave only), and once
// the wave's share of the batch is done it works its queue off in dense groups until nothing is left.  A reverse-complement item
// stages reverseComplements(read) (utils.cpp:66-73) as its words, so both strands run the same code.  What the kernel does not
// take at all -- N in the read, more than kG4PathInts / 2 path ints per direction, reads of a mixed batch that are too long for one lane per
// word -- goes on a list for bgr_align_greedy_kernel, which maps those reads from scratch right behind.
#ifndef BGR_G4_OCC
#define BGR_G4_OCC 8
#endif

// where an item stands: which strand (the reference maps the reverse complement once every forward anchor has failed,
// alignerGreedy.cpp:54), how many anchors of that strand have been tried (getNOverlap hands out the first `effort` of them)
// and the position the scan resumes from
#define G4_ST_RC (1u << 31)
#define G4_ST_TRIED_SHIFT 20
#define G4_ST_POS_MASK 0xFFFFFu

// GL = lanes per read (kG4GroupLanes, align_kernels.h): 16 = four reads per wave, 8 = eight, 4 = sixteen.
template <bool STAGE, int GL, bool ASCII>
__global__ void __launch_bounds__(1024, BGR_G4_OCC) bgr_align_greedy_multi_kernel(BgrDeviceGraph g, BatchIO io, KernelParams prm) {
    constexpr uint32_t RPW = 64 / GL;  // reads per wave
    extern __shared__ u64 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const uint32_t W = io.words_per_read;  // <= 16 (set by the host): one lane per word of a read
    const uint32_t K1 = g.k - 1;
    const uint32_t total = io.n_reads;
    if ((uint32_t)(blockIdx.x * waves) * RPW >= total) return;  // nothing for this workgroup (before it copies the key table into LDS)
    // the workgroup's counts of settled reads: four words at the start of the LDS (summed into HBM once per workgroup at the end: one
    // global atomic per wave and counter, all on the same four addresses, cost a launch of 131 k reads a third of its time)
    uint32_t* wg_counts = reinterpret_cast<uint32_t*>(lds);
    if (threadIdx.x < 4) wg_counts[threadIdx.x] = 0;
    task_stock_init(lds, (total + RPW - 1) / RPW);
#ifdef BGR_PHASE_TIMING  /* tools/wave_times.sh: when does a wave start, have its table, finish its share of the batch, finish its queue */
    const unsigned long long wt0 = wall_clock64();
    unsigned long long wt2 = 0;
#endif
    uint32_t ktab_words;
    const uint32_t* ktab = block_prologue<STAGE>(g, lds, &ktab_words);
#ifdef BGR_PHASE_TIMING
    const unsigned long long wt1 = wall_clock64();
#endif
    u64* RD = lds + 64 + ktab_words + (u64)wave * (RPW * W);
    const uint32_t grp = (uint32_t)lane / GL, sub = (uint32_t)lane % GL;
    const uint32_t m = prm.max_mismatch;
    const uint32_t eff = prm.effort ? prm.effort : 1;  // getNOverlap(read, 0) still takes a hit at position 0 (aligner.cpp:349-368)
    // (minimizer filter in front of a key table that is not staged: a scan step covers 65 - w positions, device_common.h)
    const uint32_t mmx_w = (!STAGE && g.bloom && g.filter_kind == BGR_FILTER_MINIMIZER) ? K1 + 1 - BGR_MMX_BASES : 0u;
    const bool wide_scan = io.wide_scan && mmx_w && scan_mblock_is_wide(mmx_w);   // (k = 31 / 32: the filter block in all 64 lanes, scan_mblock_wide)
    const uint32_t scan_step = (mmx_w && !wide_scan) ? 65 - mmx_w : 64;

    // Path ints go straight into the arena: read r owns the row arena[r * kG4PathInts ...] (the host keeps n_reads rows in front of the
    // chunks the cursor serves to the other kernels).  Left int number i (near -> far, offset last) at row[PH - 1 - i], right int number
    // i at row[PH + i], so reverse(left) ++ right is the slice row[PH - nl, PH + nr) -- no allocation, no copy when a walk ends.
    const uint32_t wid = (uint32_t)(blockIdx.x * waves + wave);
    // this wave's queue of follow-up items {read, state}: a ring of io.q_cap (>= 2 RPW) entries in HBM, written and read by this wave only.
    // A wave takes a dense group of RPW items off its queue as soon as it holds that many, else its next task of fresh reads (claim_task:
    // whichever wave is free takes the next one), and what is left of the queue when the launch has no task left.  An item leaves at most
    // one follow-up item behind, so the ring never holds more than 2 RPW - 1 entries.
    const uint32_t q_base = wid * io.q_cap;
    uint32_t q_rd = 0, q_wr = 0, q_cnt = 0;
    const uint32_t n_tasks = (total + RPW - 1) / RPW;
    uint32_t pool_open = 1;

    // (per-lane flags are kept as 0/1 words in VGPRs on purpose: as `bool`s they become 64-bit lane masks in SGPRs, and this
    // kernel is short of SGPRs, not of VGPRs)
    for (;;) {
        uint32_t r = BGR_NONE, st = 0;  // r == BGR_NONE: the group has no item
        uint32_t task = BGR_NONE;
        if (q_cnt < RPW && pool_open) {
            task = claim_task(lds, io.cursor + io.task_ctr, n_tasks, lane);
            if (task == BGR_NONE) {
                pool_open = 0;
#ifdef BGR_PHASE_TIMING
                if (!wt2) wt2 = wall_clock64();
#endif
            }
        }
        if (task != BGR_NONE) {  // fresh reads
            if (task * RPW + grp < total) r = task * RPW + grp;
        } else {                 // the queue: a full group, or -- no task left -- what remains
            if (q_cnt == 0) break;
            const uint32_t take = q_cnt < RPW ? q_cnt : RPW;
            if (grp < take) {
                uint32_t at = q_rd + grp;
                if (at >= io.q_cap) at -= io.q_cap;
                // (written by this wave some iterations ago; read past the CU's vector cache all the same)
                r = __hip_atomic_load(&io.queue[q_base + at].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                st = __hip_atomic_load(&io.queue[q_base + at].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            q_rd += take;
            if (q_rd >= io.q_cap) q_rd -= io.q_cap;
            q_cnt -= take;
        }
        uint32_t L = 0, act = 0;  // act: the group takes part (its item is one this kernel maps)
        u64* F = RD + grp * W;
        {   // stage the item's 2-bit words: the lanes of a group bring the words of its read -- or of reverseComplements(read)
            // (utils.cpp:66-73), word j of which is the reversed complement of bases [L - 32 (j + 1), L - 32 j)
            const u64* src = io.fw3;
            u64 a0 = 0;  // (ASCII staging: where the read's characters start)
            if (r != BGR_NONE) {
                const u64 off = io.read_offs[r];
                L = (uint32_t)(io.read_offs[r + 1] - off);
                act = ASCII ? 1u : ((io.hasn[r >> 5] >> (r & 31)) & 1u) ^ 1u;  // a read with an N goes to the general kernel
                if (((L + 31) >> 5) >= W) act = 0;               // so does a read too long for one lane per word (a batch of mixed lengths)
                if (ASCII) a0 = ascii_start(io, r, off);
                else src += packed_word_offset(off, r);
            }
            const uint32_t Wr = (L + 31) >> 5;
            uint32_t chars = 0;
            for (uint32_t j = sub; j < W; j += GL) {
                u64 f = 0;
                if (act && j < Wr) {
                    if (ASCII) {
                        // no pre-pass, no planes: the words straight from the characters (a launch handed ASCII reads: 150 B in instead of 40 B of planes
                        // that a pre-pass wrote and this kernel read back; ~11 vector instructions per read)
                        // (reads end to end: what follows a read in its last 32-byte window are the next read's bases -- no character is masked;
                        // reads scattered in a text: a newline and a header follow, masked off precisely)
                        u64 fw;
                        const int32_t p = (st >> 31) ? (int32_t)L - 32 * ((int32_t)j + 1) : 0;
                        const u64 at = (st >> 31) ? a0 + (uint32_t)(p > 0 ? p : 0) : a0 + 32ull * j;
                        const uint32_t vv = (st >> 31) ? (p >= 0 ? 32u : (uint32_t)(32 + p)) : L - 32 * j;
                        if (io.ascii_src) fw = ascii_word<false>(io.ascii, at, vv, io.ascii_bytes, &chars);
                        else fw = ascii_word<true>(io.ascii, at, vv, io.ascii_bytes, &chars);
                        if (!(st >> 31)) f = fw;
                        else if (p >= 0) f = ~rev2_fast(fw);
                        else f = (~rev2_fast(fw >> (64 - 2 * vv))) & (~0ULL << (64 - 2 * vv));
                    } else if (!(st >> 31)) f = src[j];
                    else {
                        const int32_t p = (int32_t)L - 32 * ((int32_t)j + 1);
                        if (p >= 0) f = ~rev2_fast(win32(src, (u64)(uint32_t)p));
                        else { const uint32_t v = (uint32_t)(32 + p); f = (~rev2_fast(src[0] >> (64 - 2 * v))) & (~0ULL << (64 - 2 * v)); }
                    }
                }
                F[j] = f;
            }
            if (ASCII) {  // an N anywhere in the read (bit 3 of a character): the general kernel
                uint32_t n8 = chars & 0x08080808u;
                n8 |= quad_xor1(n8);
                if (GL >= 4) n8 |= quad_xor2(n8);
                if (GL >= 8) n8 |= half_row_mirror(n8);
                if (GL == 16) n8 |= (uint32_t)__builtin_amdgcn_mov_dpp((int)n8, 0x140, 0xF, 0xF, true);  // row_mirror
                if (n8) act = 0;
            }
        }
        wave_sync();

        // ---- anchors (getNOverlap, aligner.cpp:345-378): the next overlap (k-1)-mer of each item from where its scan stands and,
        // when it lies in the same 64 positions, the one after it; record | canonical << 28
        uint32_t a_pos = 0, a_rec = BGR_NONE, b_pos = 0;  // (b_pos: the hit after a_pos when the scan's step saw one, else 0)
        for (uint32_t q = 0; q < RPW; ++q) {
            if (!rl32(act, (int)(GL * q))) continue;
#ifdef BGR_PHASE_TIMING
            if (prm.debug_stop == 1) continue;  // 1 = stops behind the staging of the reads
#endif
            const uint32_t Lq = rl32(L, (int)(GL * q)), stq = rl32(st, (int)(GL * q));
            const u64* A = RD + q * W;
            const uint32_t left_q = eff - ((stq >> G4_ST_TRIED_SHIFT) & 0x7FFu);  // anchors this strand may still try (>= 1)
            uint32_t npos = Lq >= K1 ? Lq - K1 + 1 : 0;
            if (!prm.effort && npos > 1) npos = 1;
            for (uint32_t base = stq & G4_ST_POS_MASK; base < npos; base += scan_step) {
                const uint32_t i = base + (uint32_t)lane;
                const bool valid = i < npos && (uint32_t)lane < scan_step;
                u64 num = 0, win = 0;
                if (valid || (mmx_w && i + BGR_MMX_BASES <= Lq)) win = lds_win32(A, i);
                if (valid) num = win >> (64 - 2 * K1);
                const u64 rcn = rcb_fast(num, K1);  // no N in the read: the rolling reverse k-mer is rcb of the forward one
                const uint32_t mblock = (STAGE || !mmx_w) ? 0u
                                      : wide_scan ? scan_mblock_wide(g, win, i + BGR_MMX_BASES <= Lq, i + 2 * BGR_MMX_BASES <= Lq, mmx_w)
                                                  : scan_mblock(g, win, i + BGR_MMX_BASES <= Lq, mmx_w);
                uint32_t idx = find_key<!STAGE>(g, ktab, num < rcn ? num : rcn, valid, mblock);
                const u64 mask = __ballot(idx != BGR_NONE);
                if (mask) {
                    if (idx != BGR_NONE && num <= rcn) idx |= G4_CANON;
                    const int s1 = __ffsll((long long)mask) - 1;
                    const u64 mask2 = mask & (mask - 1);
                    const uint32_t h1 = rl32(idx, s1);
                    uint32_t p2 = 0;
                    if (mask2 && left_q >= 2) p2 = base + (uint32_t)(__ffsll((long long)mask2) - 1);  // a second anchor is tried when the first fails: where a follow-up item resumes
                    if (grp == q) { a_pos = base + (uint32_t)s1; a_rec = h1; b_pos = p2; }
                    break;
                }
            }
        }

        // ---- extension (alignReadGreedy's loop body, alignerGreedy.cpp:41-52), the wave's items abreast; a group whose anchor fails
        // starts over from the next one, if the scan has seen it, while the others go on ----
        // phase: 1 left walk, 2 first right step, 3 later right steps; 0 the walk is over (aligned, or there was no anchor),
        // 4 the path outgrew the registers, 5 every anchor seen failed.  `tried` counts on in the item's state word.
        uint32_t nl = 0, nr = 0;
        constexpr uint32_t PH = kG4PathInts / 2;
        int32_t* PT = io.arena + (size_t)(r == BGR_NONE ? 0u : r) * kG4PathInts;  // the read's row; written by the group's first lane (only while the group walks)
        uint32_t phase = (act && a_rec != BGR_NONE) ? 1u : 0u;
#ifdef BGR_PHASE_TIMING  /* diagnostic builds (tools/phase_cost.sh): knob DEBUG_STOP = 2 stops behind the anchor scan */
        if (prm.debug_stop == 2) phase = 0;
#endif
        // rec: the half the next step reads (handle | canonical << 28).  An anchor is a key entry: its left walk starts from the half that
        // getEnd reads for it, its right walk from the one getBegin reads -- both handles sit in the key entry, fetched by ONE 8-byte load
        // when the anchor is taken up (a_right waits for the left walk to end: a load at that point would stall all sixteen walks)
        uint32_t a_right = G4_REC_MASK;
        auto take_anchor = [&](uint32_t anchor) -> uint32_t {  // -> the left start; sets a_right
            const uint32_t cn = (anchor >> 28) & 1u;
            const uint2 h = *reinterpret_cast<const uint2*>(&g.keys[anchor & G4_REC_MASK].hL);
            a_right = (cn ? h.x : h.y) | (cn ? G4_CANON : 0u);
            return (cn ? h.y : h.x) | (cn ? G4_CANON : 0u);
        };
        uint32_t pos = a_pos, rec = phase ? take_anchor(a_rec) : G4_REC_MASK, budget = m;
        for (;;) {
            if (phase == 1 && pos == 0) {  // the left walk reached the read's first base: push 0, then the right side of the anchor
                if (sub == 0) PT[PH - 1 - nl] = 0;
                ++nl;
                phase = 2; pos = a_pos; rec = a_right;
            }
            if (phase == 2 && L - pos - K1 == 0) phase = 0;  // nothing right of the anchor: aligned
            if (phase == 3 && L - pos < K1 + 1) phase = 0;   // |readLeft| < k: aligned
            if ((phase == 1 && nl > PH - 2) || ((phase == 2 || phase == 3) && nr > PH - 1)) phase = 4;  // path too long for the row (a left step may push two ints)
            const uint32_t on = (phase - 1u < 3u) ? phase : 0u;
            if (!__any(on != 0)) break;
            uint32_t miss, ext;
            int32_t sid;
            const uint32_t w1 = g4_step<GL>(g, F, L, K1, on, rec & G4_REC_MASK, (rec >> 28) & 1u, pos, budget, lane, &miss, &ext, &sid);
            if (on != 0) {
                if (!(w1 & G4_FOUND)) {
                    st += 1u << G4_ST_TRIED_SHIFT;
                    // (the next anchor of getNOverlap's list, when the scan saw it, is taken up by a follow-up item that resumes AT it -- rounds 3-4
                    // restarted the walk in this loop, which kept all sixteen groups' loop going for one walk: profiles/r05_scan_schemes.txt)
                    if (b_pos) { a_pos = b_pos - 1; b_pos = 0; }
                    phase = 5;
                } else if (phase == 1) {
                    if (sub == 0) PT[PH - 1 - nl] = sid;
                    ++nl;
                    budget -= miss;
                    if (w1 & G4_FITS) {
                        if (sub == 0) PT[PH - 1 - nl] = (int32_t)(ext - pos);
                        ++nl;
                        phase = 2; pos = a_pos; rec = a_right;
                    } else { pos -= ext; rec = w1; }
                } else {
                    if (sub == 0) PT[PH + nr] = sid;
                    ++nr;
                    budget -= miss;
                    if (w1 & G4_FITS) phase = 0;
                    else { pos += ext; rec = w1; phase = 3; }
                }
            }
        }

        // ---- what became of each item (alignerGreedy.cpp:35-57) ---------------------------------------------------------------
        // 0 = aligned, 1 = no anchor on this strand and none tried before (++noOverlapRead), 2 = not aligned (both strands done),
        // 3 = a follow-up item `nst` goes into the wave's queue, 4 = general kernel (N in the read, path too long for the registers)
        uint32_t outcome = 4, nst = 0;
        const uint32_t rc = st >> 31;
        if (act) {
            const uint32_t tried = (st >> G4_ST_TRIED_SHIFT) & 0x7FFu;
            const uint32_t npos_g = (L >= K1 ? L - K1 + 1 : 0);
            const uint32_t npos_e = (!prm.effort && npos_g > 1) ? 1u : npos_g;
            if (phase == 4) outcome = 4;
            else if (a_rec != BGR_NONE && phase != 5) outcome = 0;
            else if (a_rec == BGR_NONE && tried == 0) outcome = 1;
            else {
                // the strand's anchors are used up when `effort` of them have been tried or the scan has passed the last position
                const uint32_t resume = a_pos + 1;  // (a_pos = the anchor tried last; unused when the scan found none)
                const uint32_t used_up = (a_rec == BGR_NONE || tried >= eff || resume >= npos_e) ? 1u : 0u;
                if (!used_up) { outcome = 3; nst = (st & ~G4_ST_POS_MASK) | resume; }
                else if (!rc) { outcome = 3; nst = G4_ST_RC; }  // the reverse complement, from its first position
                else outcome = 2;
                if (tried >= 0x7FEu || resume > G4_ST_POS_MASK) outcome = 4;  // (an item tries at most two anchors: the count stays inside its field)
            }
        }
        const uint32_t have = r != BGR_NONE ? 1u : 0u;

        // ---- publish: the path is the slice [PH - nl, PH + nr) of the read's row ----------------------------------------------------
        const uint32_t aligned = outcome == 0 ? 1u : 0u;
        const uint32_t p_n = aligned ? nl + nr : 0;
        const uint32_t gbase = (r == BGR_NONE ? 0u : r) * kG4PathInts + PH - nl;
        if (sub == 0 && have) {
            if (outcome <= 2) {
                const uint32_t code = (outcome == 0 ? BGR_ST_ALIGNED : outcome == 1 ? BGR_ST_NOANCHOR : BGR_ST_FAILED) | (rc ? BGR_ST_RC : 0u);
                io.results[r] = make_uint2(aligned ? gbase : 0u, p_n | (code << 24));
            } else if (outcome == 4) {
                io.gen_list[atomicAdd(io.cursor + io.gen_ctr, 1u)] = r;
            }
        }
        {   // follow-up items into the wave's queue
            const bool listed = sub == 0 && have && outcome == 3;
            const u64 lm = __ballot(listed);
            if (lm) {
                const uint32_t cnt = (uint32_t)__popcll(lm);
                if (listed) {
                    uint32_t at = q_wr + __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u));  // listed lanes below this one
                    if (at >= io.q_cap) at -= io.q_cap;
                    io.queue[q_base + at] = make_uint2(r, nst);
                }
                q_wr += cnt;
                if (q_wr >= io.q_cap) q_wr -= io.q_cap;
                q_cnt += cnt;
            }
        }
        // what became of the read: counted per workgroup in LDS (wg_counts[outcome]: aligned, no anchor, not aligned, follow-up item)
        if (sub == 0 && have && outcome <= 3) atomicAdd(&wg_counts[outcome], 1u);
        wave_sync();
    }
#ifdef BGR_PHASE_TIMING
    if (io.wave_times && lane == 0) {
        unsigned long long* w = io.wave_times + 4ull * (blockIdx.x * waves + wave);
        w[0] = wt0; w[1] = wt1; w[2] = wt2; w[3] = wall_clock64();
    }
#endif
    __syncthreads();
    if (threadIdx.x == 0) {  // aligner.h:68 counters: [0] readNumber [1] noOverlapRead [2] alignedRead [3] notAligned
        unsigned long long* counters = reinterpret_cast<unsigned long long*>(io.cursor + 16);
        const uint32_t al = wg_counts[0], noov = wg_counts[1], na = wg_counts[2], qa = wg_counts[3];  // (indexed by `outcome`)
        if (al | noov | na) atomicAdd(&counters[0], (unsigned long long)(al + noov + na));
        if (noov) atomicAdd(&counters[1], (unsigned long long)noov);
        if (al) atomicAdd(&counters[2], (unsigned long long)al);
        if (na) atomicAdd(&counters[3], (unsigned long long)na);
        if (qa) atomicAdd(io.cursor + 2, qa);  // follow-up items of the launch (bgr_aligner_pass_counts)
    }
}

}  // namespace

hipError_t launch_greedy(const BgrDeviceGraph& g, const BatchIO& io, const KernelParams& p, const LaunchCfg& cfg, hipStream_t stream) {
    constexpr int GL = (int)kG4GroupLanes;
    if (io.greedy_multi) {
        if (io.ascii) return cfg.stage_mphf ? launch_one(bgr_align_greedy_multi_kernel<true, GL, true>, g, io, p, cfg, stream)
                                            : launch_one(bgr_align_greedy_multi_kernel<false, GL, true>, g, io, p, cfg, stream);
        return cfg.stage_mphf ? launch_one(bgr_align_greedy_multi_kernel<true, GL, false>, g, io, p, cfg, stream)
                              : launch_one(bgr_align_greedy_multi_kernel<false, GL, false>, g, io, p, cfg, stream);
    }
    return cfg.stage_mphf ? launch_one(bgr_align_greedy_kernel<true>, g, io, p, cfg, stream)
                          : launch_one(bgr_align_greedy_kernel<false>, g, io, p, cfg, stream);
}
const void* greedy_kernel_fn(bool many_reads) {
    return many_reads ? reinterpret_cast<const void*>(&bgr_align_greedy_multi_kernel<true, (int)kG4GroupLanes, true>) : reinterpret_cast<const void*>(&bgr_align_greedy_kernel<true>);
}

}  // namespace bgr
