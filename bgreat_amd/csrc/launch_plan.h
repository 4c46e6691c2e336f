// launch_plan.h -- the geometry of one mapping launch, as a pure function of numbers: the graph's header, the device (CUs, LDS per CU,
// register-limited resident waves per kernel), the tuning the caller set, and the batch (mode, budget, reads, bases, longest read).
// No HIP in here: capi.hip calls plan_launch() and enqueues what it says; tests drive it on the CPU over synthetic headers
// (bgr_plan_launch, tests/test_launch_plan.py: the plan never fails on a mappable batch and never stages a table it cannot hold).
// The constants in here are measured ones; the comment next to each says on what.
#ifndef BGREAT_AMD_LAUNCH_PLAN_H
#define BGREAT_AMD_LAUNCH_PLAN_H

#include <stdint.h>

#include <algorithm>

#include "../../include/bgreat_gpu.h"
#include "align_kernels.h"

namespace bgr {

const uint32_t kLdsFixed = 512;  // reserved at the start of a workgroup's dynamic LDS (counters, task stock)

struct PlanGraph {    // from BgrBlobHeader
    uint32_t k = 0, slot_fill_x100 = 100, table_bytes = 0;
    uint64_t total_bases = 0, n_unitigs = 0, n_buckets = 0, max_unitig_len = 0, anc_n = 0, anc_active_levels = 0;
    bool has_exc = false;
};
struct PlanDevice {
    uint32_t num_cus = 256;
    uint64_t lds_per_cu = 160 * 1024;
    // waves one CU keeps resident, by kernel (resident_waves_per_cu's modes: 0 greedy general, 1 exhaustive depth-first, 2 anchors, 3 exhaustive
    // level search, 4 greedy sixteen reads per wave, 5 exhaustive eight reads per wave, 6 anchors four reads per wave)
    uint32_t resident[7] = {24, 24, 16, 20, 32, 24, 16};
};
struct PlanTuning {   // bgr_aligner_configure / bgr_aligner_set_knob
    uint32_t cfg_waves = 0, cfg_blocks_per_cu = 0, cfg_lds_mphf = 0;
    uint32_t frame_cap = 0, search = 0;  // BGR_KNOB_EXH_FRAME_CAP, BGR_KNOB_EXH_SEARCH
    bool no_greedy_fast = false, no_exh_fast = false, no_anc_fast = false;
    uint32_t memo_cap = 0;               // BGR_KNOB_EXH_MEMO_CAP: entries of the last pass's table per wave in its first run (0: from the read length)
};
struct PlanBatch {
    uint32_t mode = 0, max_mismatch = 0, partial = 0, max_read_len = 0;
    uint64_t n_reads = 0, total_bases = 0;
};

struct LaunchPlan {
    const char* error = nullptr;  // the batch cannot be mapped (read too long, batch too large): what to tell the caller
    LaunchCfg cfg, cfg_deep, cfg_mid, cfg_fast, cfg_x4, cfg_a4;
    bool level_search = false, two_pass = false, deep_only = false, mid_pass = false, fast_pass = false, x4_pass = false, a4_pass = false, wide_scan = false;
    uint32_t words = 0, wfast = 0, path_cap = 0, frames = 0, frames_deep = 0, frames_mid = 0, x4_levels = 16, a4_lanes = 16;
    uint32_t arena_chunk = 0, q_cap = 0, search_iters = 0, memo_cap = 0;
    uint64_t deep_stride = 0;     // u32 words of one wave's region of the last pass's scratch
    uint64_t arena_cap = 0, fast_rows = 0, plane_words = 0;
};

// smallest power of two >= x (x >= 1)
inline uint32_t pow2_at_least(uint64_t x) {
    uint32_t p = 1;
    while (p < x && p < (1u << 31)) p <<= 1;
    return p;
}

// b workgroups per CU of w waves each (every staged workgroup holds its own copy of the key table): does it fit a CU's LDS?
struct PlanGeometry {
    const PlanGraph& g;
    const PlanDevice& d;
    const PlanTuning& t;
    uint32_t mode, cap_default;
    bool fits(uint32_t b, uint32_t w, bool st, uint32_t pw) const {
        const uint64_t lds_fit = d.lds_per_cu - 64;  // keep a little slack for alignment
        return (uint64_t)b * (kLdsFixed + (st ? ((g.table_bytes + 15) / 16) * 16 : 0) + (uint64_t)w * pw) <= lds_fit;
    }
    // Resident waves per CU are bounded by registers; LDS decides how they are grouped.  More resident waves hide more of the walk's
    // dependent-load latency (measured 16 -> 24 waves/CU: +18 %), and a grid of exactly CUs x b workgroups avoids a partial last round.
    // stage_pct: the staged grouping is taken when it keeps at least this share of the resident waves of the best grouping without staging.
    bool operator()(uint32_t pw, uint64_t n_items, bool allow_tuning, bool allow_stage, LaunchCfg& cfg, uint32_t cap_override = 0, uint32_t stage_pct = 100) const {
        const uint32_t cap = cap_override ? cap_override : cap_default;
        uint32_t waves = 0, bpc = 0;
        bool stage = false;
        if (allow_tuning && (t.cfg_waves || t.cfg_blocks_per_cu)) {  // explicit tuning through bgr_aligner_configure
            stage = allow_stage && mode != BGR_MODE_ANCHORS && (t.cfg_lds_mphf == 2 || (t.cfg_lds_mphf == 0 && fits(1, 1, true, pw)));
            if (stage && !fits(1, 1, true, pw)) stage = false;  // (staging asked for a table no CU can hold next to one wave: probed in L2, at the waves asked for -- the
                                                                // loops below would otherwise shrink the workgroup to one wave first: 0.06 of the rate, tools/geometry_sweep.py)
            waves = t.cfg_waves ? t.cfg_waves : (stage ? 12 : 4);
            bpc = t.cfg_blocks_per_cu ? t.cfg_blocks_per_cu : std::max<uint32_t>(1, cap / waves);
            while (bpc > 1 && !fits(bpc, waves, stage, pw)) --bpc;
            while (waves > 1 && !fits(bpc, waves, stage, pw)) --waves;
            if (!fits(bpc, waves, stage, pw) && stage) stage = false;  // (also when staging was asked for: a table beyond the LDS is probed in L2)
        } else {
            uint32_t best_res = 0;
            if (allow_stage && t.cfg_lds_mphf != 1 && mode != BGR_MODE_ANCHORS && g.n_buckets * 4 < 0xFFFFFFFFull) {
                const uint32_t bs[] = {1, 2, 3, 4, 6};
                for (uint32_t b : bs) {
                    uint32_t w = std::min<uint32_t>(16, cap / b);
                    while (w > 0 && !fits(b, w, true, pw)) --w;
                    if (w > 4) w -= w % 4;  // whole waves per SIMD: 5-, 7-wave workgroups measured up to 40 % slower (E. coli-scale graph, round 2)
                    if (w && b * w > best_res) { best_res = b * w; waves = w; bpc = b; stage = true; }
                }
            }
            // without staging: as many small workgroups as the registers admit; when the per-wave LDS region is large
            // (long reads, exhaustive frame stacks) fewer, larger workgroups keep more waves resident
            uint32_t wn = 0, bn = 0, res_n = 0;
            // (4-wave workgroups first: a workgroup whose wave count is not a multiple of the 4 SIMDs measured far slower)
            const uint32_t bs2[] = {6, 5, 4, 3, 2, 1};
            uint32_t w4 = 0, b4 = 0;  // the best grouping made of 4-wave workgroups
            for (uint32_t b : bs2) {
                uint32_t w = std::min<uint32_t>(b >= 5 ? 4 : 16, std::max<uint32_t>(1, cap / b));
                while (w > 0 && !fits(b, w, false, pw)) --w;
                if (w > 4) w -= w % 4;
                if (w && b * w > res_n) { res_n = b * w; wn = w; bn = b; }
                const uint32_t wq = std::min<uint32_t>(w, 4);
                if (wq && b * wq > b4 * w4) { w4 = wq; b4 = b; }
            }
            // one wave per SIMD and workgroup schedules best (8-wave workgroups measured 87 vs 123 Mreads/s at 24 vs 20
            // resident waves, round 1): take that grouping unless it gives up more than a fifth of the resident waves
            if (w4 == 4 && b4 * w4 * 5 >= res_n * 4) { res_n = b4 * w4; wn = w4; bn = b4; }
            // (lds_mphf = 2 asks for staging: where there is nothing to stage -- anchors mode probes its own index -- or the table does not fit a CU's
            // LDS next to one wave, the launch runs without; launch_info says which it was)
            if (!allow_stage || t.cfg_lds_mphf == 1 || best_res == 0 || (t.cfg_lds_mphf == 0 && res_n * stage_pct > best_res * 100)) { stage = false; waves = wn; bpc = bn; best_res = res_n; }
            if (best_res == 0) waves = 0;
        }
        if (waves == 0 || !fits(bpc ? bpc : 1, waves, stage, pw)) return false;
        cfg.lds_bytes = kLdsFixed + (stage ? ((g.table_bytes + 15) / 16) * 16 : 0) + waves * pw;
        cfg.blocks = (uint32_t)std::min<uint64_t>((n_items + waves - 1) / waves, (uint64_t)d.num_cus * bpc);
        cfg.waves_per_block = waves;
        cfg.stage_mphf = stage ? 1 : 0;
        return true;
    }
};

// The last pass's scratch of one run: `waves` waves with a table of `memo_cap` entries each.
inline uint64_t deep_scratch_words(uint32_t path_cap, uint32_t frames, uint32_t memo_cap) {
    return 3ull * path_cap + (uint64_t)frames * BGR_MEMO_FRAME_WORDS + (uint64_t)memo_cap * BGR_MEMO_ENTRY_WORDS;
}

inline LaunchPlan plan_launch(const PlanGraph& g, const PlanDevice& d, const PlanTuning& t, const PlanBatch& b) {
    LaunchPlan P;
    const uint64_t n_reads = b.n_reads;
    const uint32_t max_read_len = b.max_read_len;
    // Exhaustive mode runs in passes: the first ones give every wave a SHALLOW search stack (kExhFrameCap frames) in LDS so that many waves
    // fit a CU; the rare read whose search goes deeper is listed and mapped by the last pass, which keeps the worst-case search state in HBM.
    // Reads too long for the LDS layouts all go through the last pass's kernel directly.  Greedy mode is one pass.
    const bool fc_set = t.frame_cap != 0;  // BGR_KNOB_EXH_FRAME_CAP: tests shrink it to push most reads through the last pass
    const uint32_t kExhFrameCap = fc_set ? std::max<uint32_t>(2, t.frame_cap) : 24;
    const uint32_t lmode = b.mode == BGR_MODE_EXHAUSTIVE ? 1u : 0u;  // anchors mode uses the greedy per-wave layout
    lds_bytes_per_wave(lmode, g.k, max_read_len, &P.words, &P.path_cap, &P.frames_deep, 0);
    const uint32_t per_wave_deep = deep_lds_bytes_per_wave(max_read_len);
    const bool exhaustive = b.mode == BGR_MODE_EXHAUSTIVE;
    // pass 1 of exhaustive mode runs the level-by-level search (exh_dp) or the depth-first one (BGR_KNOB_EXH_SEARCH forces either).
    // Which one is faster depends on how much the walks branch within the mismatch budget: the depth-first search wins
    // on a graph with an occasional 2-way bubble (about 1.2x), the level search where a read crosses many multi-way
    // sites (3.6x at 4 alleles every ~36 bp, m=5).  Estimate: (extra candidates per record) x (m+1) x (unitigs per read).
    const double mean_ext = std::max(1.0, (double)g.total_bases / (2.0 * (double)std::max<uint64_t>(1, g.n_unitigs)) - (double)(g.k - 1));
    // Scans behind the minimizer filter (large key tables): 64 positions per step for k = 31 / 32 (scan_mblock_wide: ~15 instructions more per step) where a
    // read's next overlap usually lies beyond the step's 50 positions -- unitigs longer than a step (chr1-scale graph, 150 bp reads: 3 -> 2 steps, 1 760 ->
    // 1 873 Mreads/s); on a graph that branches every ~36 bp the first step nearly always holds a hit and the narrow step is the cheaper one (1 004 vs 980).
    P.wide_scan = mean_ext >= 64.0;
    if (exhaustive) {
        const double branching = (g.slot_fill_x100 / 100.0 - 1.0) * (double)(b.max_mismatch + 1) * ((double)max_read_len / mean_ext);
        P.level_search = t.search ? t.search == 2 : branching >= 15.0;
        // short walks (E. coli-scale graph, 150 bp: 2-3 unitigs per side): half the table, twice the waves per CU (1 200 vs 1 440 Mreads/s)
        if (2.0 * (double)max_read_len / mean_ext <= 8.0) P.x4_levels = 8;
    }
    // level search: a level is one unitig of the walk; 16 levels cover 250 bp reads on a graph that branches every ~36 bp
    const uint32_t level_cap = fc_set ? kExhFrameCap : std::max<uint32_t>(16, (max_read_len / 64) * 4);
    const uint32_t per_wave = lds_bytes_per_wave(P.level_search ? 2u : lmode, g.k, max_read_len, &P.words, &P.path_cap, &P.frames, P.level_search ? level_cap : kExhFrameCap);
    // (every exhaustive launch has its last pass behind it: the depth-first passes with their stack in LDS bound their work per search -- search_iters --
    // and hand on what exceeds it, besides what outgrows their frames; the level search can also overflow on a wide level)
    P.two_pass = exhaustive;
    PlanGeometry geometry{g, d, t, b.mode, std::max<uint32_t>(4, d.resident[P.level_search ? 3u : (b.mode <= 2 ? b.mode : 0u)])};
    // level search: what it cannot hold (a level wider than 4 nodes, too many levels) goes to the depth-first kernel with
    // its LDS stack first, and only what overflows that one to the last pass
    const uint32_t per_wave_mid = lds_bytes_per_wave(1u, g.k, max_read_len, nullptr, nullptr, &P.frames_mid, kExhFrameCap);
    if (!geometry(per_wave, n_reads, true, true, P.cfg)) {
        if (!exhaustive) { P.error = "bgr_align_device: read too long for the per-wave LDS staging (limit ~30 kb)"; return P; }
        P.deep_only = P.two_pass = true;
    }
    if (P.two_pass) {
        if (!geometry(per_wave_deep, n_reads, false, false, P.cfg_deep)) {  // the last pass never stages the key table
            P.error = "bgr_align_device: read too long for the per-wave LDS staging (limit ~160 kb)";
            return P;
        }
        // The last pass sees few reads (none on a compacted de Bruijn graph unless the batch holds very long reads: deep_only): two waves per CU,
        // and a table of remembered calls that starts small -- a read that fills it is run again with a larger one (capi.hip, settle_launch) --
        // so that the scratch every exhaustive aligner carries stays in the tens of megabytes.
        P.memo_cap = t.memo_cap ? pow2_at_least(std::max<uint32_t>(8, t.memo_cap)) : pow2_at_least(std::max<uint64_t>(1024, 4ull * max_read_len));
        P.deep_stride = deep_scratch_words(P.path_cap, P.frames_deep, P.memo_cap);
        if (P.deep_stride > 0xFFFFFFFFull) { P.error = "bgr_align_device: read too long"; return P; }
        const uint64_t max_waves = std::max<uint64_t>(1, std::min<uint64_t>(2ull * d.num_cus, (1ull << 30) / (P.deep_stride * 4)));
        if ((uint64_t)P.cfg_deep.blocks * P.cfg_deep.waves_per_block > max_waves) {
            P.cfg_deep.waves_per_block = (uint32_t)std::min<uint64_t>(std::min<uint32_t>(P.cfg_deep.waves_per_block, 2), max_waves);
            P.cfg_deep.blocks = (uint32_t)std::max<uint64_t>(1, max_waves / P.cfg_deep.waves_per_block);
            P.cfg_deep.lds_bytes = kLdsFixed + P.cfg_deep.waves_per_block * per_wave_deep;
        }
        if (P.deep_only) P.cfg = P.cfg_deep;
        P.mid_pass = P.level_search && !P.deep_only && P.frames_mid < P.frames_deep && geometry(per_wave_mid, n_reads, false, true, P.cfg_mid);
    }
    // Greedy mode, first pass: sixteen reads per wave (bgr_align_greedy_multi_kernel, the reference's retry ladder inside the launch)
    // when a read fits one lane per word and the graph has no exception planes; what it does not take (N reads, very long paths)
    // is listed and mapped by the general kernel (cfg) right behind.
    P.wfast = std::min<uint32_t>(P.words, 16);  // the many-reads-per-wave kernels take reads of < 16 words; longer ones of a mixed batch are listed
    P.fast_pass = b.mode == BGR_MODE_GREEDY && !t.no_greedy_fast && !g.has_exc &&
                  geometry(kG4ReadsPerWave * 8 * P.wfast, (n_reads + kG4ReadsPerWave - 1) / kG4ReadsPerWave, true, true, P.cfg_fast, std::max<uint32_t>(4, d.resident[4]),
                           50);  // sixteen reads per wave, E. coli-scale table (72 KB): 2 x 12 waves with the table in LDS 1 877 Mreads/s, 1 x 16: 1 543, 32 waves probing it in L2: 1 381
    // Exhaustive mode, first pass: eight reads per wave (bgr_align_exhaustive4_kernel) for the shape nearly every read has (one
    // node per level of the walk); what it does not settle is listed and goes through the passes above from scratch.
    P.x4_pass = exhaustive && !P.deep_only && !t.no_exh_fast && !b.partial && !g.has_exc && b.max_mismatch <= kX4MaxMismatch && g.max_unitig_len <= kX4MaxUnitigLen &&
                geometry(kX4ReadsPerWave * 8 * (P.wfast + x4_group_words(P.x4_levels)), (n_reads + kX4ReadsPerWave - 1) / kX4ReadsPerWave, true, true, P.cfg_x4,
                         std::max<uint32_t>(4, d.resident[5]), 50);  // (E. coli-scale table, 150 bp: one staged workgroup of 16 waves 1 496 Mreads/s, 28 waves probing the table in L2 1 395)
    // Anchors mode, first pass: four reads per wave (bgr_align_anchors4_kernel); reads with an N and very long paths are listed
    // for the one-read-per-wave kernel.  Lanes per read: a lookup spreads BooPHF's active levels over the lanes of the read's group (8 when they fit, else 16)
    P.a4_lanes = g.anc_active_levels <= 8 ? 8u : 16u;
    const uint32_t a4_rpw = 64 / P.a4_lanes;
    P.a4_pass = b.mode == BGR_MODE_ANCHORS && !t.no_anc_fast && !g.has_exc && g.anc_active_levels <= 16 &&
                geometry(a4_rpw * 16 * P.wfast, (n_reads + a4_rpw - 1) / a4_rpw, true, false, P.cfg_a4, std::max<uint32_t>(4, d.resident[6]));
    // Path arena: every path int consumes at least one read base (+8 per read for offsets / short reads), plus
    // the unused tail of the per-wave chunks the kernels reserve with one atomic each.
    // A chunk is at least twice the longest possible path, so an abandoned chunk is more than half used.
    P.arena_chunk = std::max<uint32_t>(256, 2 * P.path_cap);
    // the several-reads-per-wave greedy kernel writes a read's path ints where they are found, into the read's own row of kG4PathInts ints
    // at the start of the arena (no per-wave chunks, no copy at the end of a walk); the cursor-served chunks of the other kernels follow
    P.fast_rows = P.fast_pass ? n_reads * kG4PathInts : 0;
    const uint64_t deep_waves = P.two_pass ? (uint64_t)P.cfg_deep.blocks * P.cfg_deep.waves_per_block : 0;
    P.arena_cap = 2 * (b.total_bases + 8 * n_reads) + (uint64_t)P.cfg.blocks * P.cfg.waves_per_block * P.arena_chunk +
                  (P.two_pass ? (P.deep_only ? kDeepRuns - 1 : kDeepRuns) * deep_waves * P.arena_chunk : 0) +  // (every run of the last pass starts its waves on fresh chunks)
                  (P.mid_pass ? (uint64_t)P.cfg_mid.blocks * P.cfg_mid.waves_per_block * P.arena_chunk : 0) +
                  P.fast_rows +
                  (P.x4_pass ? (uint64_t)P.cfg_x4.blocks * P.cfg_x4.waves_per_block * P.arena_chunk : 0) +
                  (P.a4_pass ? n_reads * kA4PathInts : 0);  // (per-read rows, as for the greedy kernel)
    if (P.arena_cap >= 0xFFFFFFFFull) { P.error = "bgr_align_device: batch too large (2*(bases + 16*reads) must stay below 2^32); split it"; return P; }
    P.plane_words = (b.total_bases >> 5) + n_reads + 4;
    if (P.plane_words >= 0xFFFFFFFFull) { P.error = "bgr_align_device: batch too large; split it"; return P; }
    // a depth-first search in LDS takes ~2 iterations per read base on a branching graph (500 for 250 bp, 4 alleles every 36 bp, m = 5); beyond 64 x that
    // its read goes to the last pass (polynomial: exh_memo) -- never reached on a graph of unique k-mers
    P.search_iters = (exhaustive && P.two_pass && !P.deep_only) ? 128u * (max_read_len + 64u) : 0u;
    // the sixteen-reads-per-wave greedy kernel keeps a ring of follow-up items per wave (it drains the ring whenever it holds a full group)
    P.q_cap = P.fast_pass ? 2 * kG4ReadsPerWave : 0;
    return P;
}

}  // namespace bgr

#endif
