// align_kernels.h -- launch interface between the C-ABI (capi.hip) and the gfx950 kernels (greedy_kernels.hip,
// exhaustive_kernels.hip, anchors_kernel.hip, batch_kernels.hip; shared device code in device_common.h).
#ifndef BGREAT_AMD_ALIGN_KERNELS_H
#define BGREAT_AMD_ALIGN_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "graph_layout.h"

namespace bgr {

// lanes per read of the many-reads-per-wave greedy kernel (bgr_align_greedy_multi_kernel): 16 = four reads per wave, 8 = eight, 4 = sixteen
// (sixteen: the step instructions of a wave serve twice the reads and twice as many walks wait on memory at once: E. coli-scale 1 772 -> 1 877
// Mreads/s, chr1-scale 1 202 -> 1 320, configs[1] 1 791 -> 2 035)
#ifndef BGR_G4_GROUP_LANES
#define BGR_G4_GROUP_LANES 4
#endif
constexpr uint32_t kG4GroupLanes = BGR_G4_GROUP_LANES, kG4ReadsPerWave = 64 / BGR_G4_GROUP_LANES;
constexpr uint32_t kA4PathInts = 20;  // ... of the several-reads-per-wave anchors kernel: 8 left + [offset,] unitig + 8 right (9 + 2 + 9 slots)
constexpr uint32_t kG4PathInts = 16;  // path ints of a read's row in the arena: 8 of the left walk (offset included), 8 of the right; longer paths go to the general kernel
// the same for the exhaustive first pass (bgr_align_exhaustive4_kernel); its level table holds one level per lane of a read's group
#ifndef BGR_X4_GROUP_LANES
#define BGR_X4_GROUP_LANES 8
#endif
constexpr uint32_t kX4GroupLanes = BGR_X4_GROUP_LANES, kX4ReadsPerWave = 64 / BGR_X4_GROUP_LANES;
// u64 words per read besides the read's own: level table (8 or 16 levels of a walk per side, BGR_X4_LEVEL_WORDS u32 each: exhaustive_kernels.hip) + out ints
#define BGR_X4_LEVEL_WORDS 9
constexpr uint32_t kX4MaxMismatch = 254, kX4MaxUnitigLen = (1u << 22) - 1;  // what the level table's packed fields hold
constexpr uint32_t x4_group_words(uint32_t levels) { return (levels * BGR_X4_LEVEL_WORDS + 2 * (levels + 2) + 1) / 2; }

struct BatchIO {
    const uint64_t* fw3;         // 2-bit plane of the batch (bgr_pack_reads_kernel / host packer): read r at word (read_offs[r] >> 5) + r
    const uint64_t* nmw;         // N-mask plane, same addressing; valid only for reads whose bit is set in hasn
    const uint32_t* hasn;        // bitmap: read holds an N
    const uint64_t* read_offs;   // n+1 base offsets of the reads (lengths; packed-plane addressing)
    const uint8_t* ascii;        // greedy mode: the reads' characters in HBM -- the mapping kernels stage from them (no pre-pass, no planes); else nullptr
    const uint32_t* ascii_src;   //   where read r's characters start when the reads lie scattered in a text (text route), else nullptr: at read_offs[r]
    uint64_t ascii_bytes;        //   bytes of that buffer (a 32-byte load never reaches beyond it)
    uint2* results;              // n: x = path offset in the arena, y = path length | status << 24
    int32_t* arena;
    uint32_t* cursor;            // [0] ints used, [1] overflow flag; counters (5 x u64) start at cursor + 16
    uint32_t n_reads;
    uint32_t words_per_read;     // u64 words of each packed per-wave LDS array (max_read_len/32 + 2)
    uint32_t path_cap;           // ints of the per-wave LDS path buffer
    uint32_t arena_cap;          // ints
    uint32_t arena_chunk;        // ints a wave reserves per global atomic
    uint32_t frames_per_wave;    // exhaustive mode: DFS frames (20 u32 each) in the per-wave LDS region
    uint32_t* ovf_list;          // exhaustive pass 1: reads whose search outgrew frames_per_wave are listed here (count at cursor[2])
    const uint32_t* subset;      // exhaustive pass 2: map reads subset[0 .. cursor[2]) instead of 0 .. n_reads
    uint32_t* deep_scratch;      // exhaustive, last pass: per-wave search state in HBM (OUT | CUR | BEST | frames | table), else nullptr
    uint32_t deep_stride;        // u32 words of one wave's region in deep_scratch
    uint32_t level_search;       // exhaustive pass 1: level-by-level search (exh_dp), frames_per_wave = its level cap
    uint32_t search_iters;       // exhaustive, depth-first passes with their stack in LDS: loop iterations one search may take before its read is handed to the
                                 //   last pass (0 = no bound): the recursion is exponential where unitigs duplicate each other's k-mers
    uint32_t arena_own;          // ints at the start of the arena that waves own by their number: what the cursor hands out lies behind them (the cursor itself starts at 0:
                                 // one fill less per launch than starting it at this value)
    uint32_t wide_scan;          // scans behind the minimizer filter: 64 positions per step where k allows (launch_plan.h)
    uint32_t deep_memo_cap;      // exhaustive, last pass: entries (a power of two) of a wave's table of remembered calls in deep_scratch (exh_memo); a read that
                                 //   fills it is put on ovf_list and run again by the host with a larger table
    uint32_t greedy_multi;       // greedy mode: launch the sixteen-reads-per-wave kernel; reads it does not take go on gen_list (for the general kernel)
    uint2* queue;                // its per-wave rings of follow-up items {read, state}: q_cap entries per wave of the grid
    uint32_t q_cap;
    uint32_t* gen_list;          // reads for the general kernel (count at cursor[gen_ctr])
    uint32_t gen_ctr;
    uint32_t anc4;               // anchors mode: launch the four-reads-per-wave kernel (what it does not settle goes on ovf_list)
    uint32_t exh4;               // exhaustive mode: launch the several-reads-per-wave kernel (what it does not settle goes on ovf_list);
                                 //   the value = levels per side of its level table (8 or 16)
    uint32_t subset_ctr, ovf_ctr; // which words of `cursor` count the reads of `subset` / collect the reads put on `ovf_list`
    uint32_t task_ctr;            // which word of `cursor` hands out the tasks of a several-reads-per-wave launch (claim_task, device_common.h)
    unsigned long long* wave_times;  // diagnostic builds (-DBGR_PHASE_TIMING) only, else null: four 100 MHz time stamps per wave of the several-reads-per-wave greedy kernel
};

struct KernelParams {
    uint32_t max_mismatch, effort, partial, mode;
    uint32_t debug_stop;  // diagnostic builds only (env BGR_DEBUG_STOP): 1 = stop after packing, 2 = after the position scan
};

struct LaunchCfg {
    uint32_t waves_per_block;  // 1..16
    uint32_t blocks;           // grid
    uint32_t lds_bytes;        // dynamic LDS
    uint32_t stage_mphf;       // 1: copy the MPHF cascade into LDS at block start
};

// the last exhaustive pass (exh_memo, exhaustive_kernels.hip): u32 words of a frame of its explicit stack / of an entry of its table, both in HBM;
// runs of that pass one launch's arena has room for (the first one and its repeats with larger tables: 16 x per repeat, capi.hip settle_launch)
#define BGR_MEMO_FRAME_WORDS 32
#define BGR_MEMO_ENTRY_WORDS 8
constexpr uint32_t kDeepRuns = 8;

// Per-wave LDS bytes for a batch whose longest read has max_len bases (mode 0 greedy, 1 exhaustive).
// frame_cap > 0 limits the exhaustive DFS stack (the last pass, with the full stack in HBM, maps the few reads that need more; its LDS need is
// deep_lds_bytes_per_wave(), its HBM need deep_scratch_words(), launch_plan.h).
inline uint32_t deep_lds_bytes_per_wave(uint32_t max_len) { return 4 * 8 * (max_len / 32 + 2); }
inline uint32_t lds_bytes_per_wave(uint32_t mode, uint32_t k, uint32_t max_len, uint32_t* words, uint32_t* path_cap, uint32_t* frames, uint32_t frame_cap = 0) {
    uint32_t w = max_len / 32 + 2;
    uint32_t pc = max_len + 8;
    pc = (pc + 3) & ~3u;  // multiple of 4 ints: what lies behind the path buffers stays 16-byte aligned
    uint32_t fr = 0;
    uint32_t bytes = 4 * 8 * w + 4 * pc;  // FW3 | FWQ | RCW | NM | PATH
    if (mode != 0) {
        // every DFS descent consumes at least one read base outside the anchor's k-1 window
        fr = (max_len >= k - 1 ? max_len - (k - 1) : 0) + 3;
        if (frame_cap && fr > frame_cap) fr = frame_cap;
        bytes = 4 * 8 * w + 3 * 4 * pc + fr * 20 * 4;  // ... OUT | CUR | BEST | frames
    }
    if (mode == 2) {  // exhaustive pass 1 with the level-by-level search: FW3 | FWQ | RCW | NM | OUT | BEST | tables
        fr = (max_len >= k - 1 ? max_len - (k - 1) : 0) + 3;
        if (frame_cap && fr > frame_cap) fr = frame_cap;
        const uint32_t table_words = 32 + fr * (52 + 1);
        bytes = 4 * 8 * w + 2 * 4 * pc + 16 * ((table_words + 3) / 4);
    }
    if (words) *words = w;
    if (path_cap) *path_cap = pc;
    if (frames) *frames = fr;
    return bytes;
}

// Waves of the mapping kernel that one CU can keep resident (register-limited; mode 0 greedy, 1 exhaustive depth-first,
// 2 anchors, 3 exhaustive level search, 4 greedy sixteen-reads-per-wave, 5 exhaustive eight-reads-per-wave, 6 anchors four-reads-per-wave).
uint32_t resident_waves_per_cu(uint32_t mode);

// (results, arena) of the last mapping launch -> input-ordered CSR on the device.  phase 0: block_sums[ceil(n/4096)] and
// *total (all path ints); phase 1: path_offsets[n+1], paths[total] (nothing is stored past paths_cap), status[n].
hipError_t launch_csr(const uint2* results, const int32_t* arena, uint32_t n, uint32_t* block_sums, unsigned long long* total,
                      unsigned long long* path_offsets, int32_t* paths, uint8_t* status, uint32_t paths_cap, int phase, hipStream_t stream);

// ASCII reads (ACGTN) -> the 2-bit planes the mapping kernels read.  `hasn` (ceil(n/32) words) must be zero on entry;
// planes need (total_bytes >> 5) + n + 2 words each.
hipError_t launch_pack_reads(const uint8_t* reads, const uint64_t* read_offs, uint32_t n, uint64_t total_bytes, uint64_t* fw3, uint64_t* nmw,
                             uint32_t* hasn, hipStream_t stream);

hipError_t launch_scatter_words(const uint32_t* index, const uint64_t* value, uint64_t n, uint64_t* plane, uint64_t plane_words, hipStream_t stream);

hipError_t launch_align(const BgrDeviceGraph& g, const BatchIO& io, const KernelParams& p, const LaunchCfg& cfg, hipStream_t stream);

}  // namespace bgr

#endif
