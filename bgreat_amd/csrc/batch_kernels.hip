// batch_kernels.hip -- the kernels around the mapping passes of a launch: the pre-pass that packs ASCII reads into 2-bit
// planes, the scatter of a host-packed batch's N-mask words, results -> CSR on the device, and the launch dispatch.
#include <algorithm>

#include "device_common.h"
#include "text_kernels.h"

namespace bgr {

// ======================================= pre-pass: ASCII reads -> 2-bit planes ================================
// Streaming kernel in front of every mapping launch that is handed ASCII reads (what getReads yields, aligner.cpp:46-117):
// str2num codes (utils.cpp:117-129: A0 C1 G2, anything else 3) 32 bases per u64, first base most significant, plus
// the N mask for the few reads that hold an N (their bit is set in `hasn`, which the caller zeroes).  `lpr` lanes per
// read (the batch's mean words per read, rounded up), 32 bases per lane and step.  ~150 B in + 40 B out per 150 bp read.

// (src_off: null = read r's characters start at reads + read_offs[r], the reads of a batch end to end; else at reads + src_off[r]:
// reads scattered in a FASTA text, text_kernels.hip -- read_offs still numbers the bases of the batch and addresses the planes)
__global__ void __launch_bounds__(256) bgr_pack_reads_kernel(const uint8_t* reads, const uint32_t* src_off, const u64* read_offs, uint32_t n, u64 total_bytes, u64* fw3,
                                                             u64* nmw, uint32_t* hasn, uint32_t lpr, uint32_t inv_lpr, uint32_t reads_per_block) {
    const uint32_t slot = (threadIdx.x * inv_lpr) >> 16;  // threadIdx.x / lpr (inv_lpr checked exact for 0..255 by the launcher)
    const uint32_t j0 = threadIdx.x - slot * lpr;
    const uint32_t r = blockIdx.x * reads_per_block + slot;
    if (slot >= reads_per_block || r >= n) return;
    const u64 off = read_offs[r];
    const uint32_t L = (uint32_t)(read_offs[r + 1] - off);
    const uint32_t Wr = (L + 31) >> 5;
    const uint32_t woff = packed_word_offset(off, r);
    const u64 src = src_off ? (u64)src_off[r] : off;
    uint32_t sawN = 0;
    for (uint32_t j = j0; j < Wr; j += lpr) {
        uint32_t xs[8], c[8];
        load32(reads + src, L, j, src + 32ull * j + 32 <= total_bytes, xs);
#pragma unroll
        for (int d = 0; d < 8; ++d) { c[d] = codes4(xs[d]); sawN |= xs[d]; }  // (bit 3 of a character: set for N only)
        fw3[woff + j] = pack_codes(c);
    }
    // a read with an N: all its N-plane words are written (the mapping kernels read the plane only for such reads).  Rare:
    // whichever lane saw an N writes the read's whole N plane (two lanes of one read write the same words).
    if (sawN & 0x08080808u) {
        for (uint32_t j = 0; j < Wr; ++j) {
            uint32_t xs[8], c[8];
            load32(reads + src, L, j, src + 32ull * j + 32 <= total_bytes, xs);
#pragma unroll
            for (int d = 0; d < 8; ++d) c[d] = ((xs[d] >> 3) & 0x01010101u) * 3u;
            nmw[woff + j] = pack_codes(c);
        }
        atomicOr(&hasn[r >> 5], 1u << (r & 31));
    }
}

static hipError_t launch_pack_impl(const uint8_t* reads, const uint32_t* src_off, const uint64_t* read_offs, uint32_t n, uint64_t total_bytes, uint64_t total_bases,
                                   uint64_t* fw3, uint64_t* nmw, uint32_t* hasn, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    // lanes per read: the mean number of 32-base words per read, rounded up (150 bp: 5), at most 16; longer reads loop
    uint32_t lpr = (uint32_t)std::min<uint64_t>(16, std::max<uint64_t>(1, (total_bases / n + 31) / 32));
    uint32_t inv = (65536 + lpr - 1) / lpr;
    for (uint32_t t = 0; t < 256; ++t)
        if (((t * inv) >> 16) != t / lpr) { lpr = 8; inv = 65536 / 8; break; }  // (never taken for lpr <= 16; kept as a guard)
    const uint32_t rpb = 256 / lpr;
    const uint32_t blocks = (n + rpb - 1) / rpb;
    hipLaunchKernelGGL(bgr_pack_reads_kernel, dim3(blocks), dim3(256), 0, stream, reads, src_off, read_offs, n, total_bytes, fw3, nmw, hasn, lpr, inv, rpb);
    return hipGetLastError();
}
hipError_t launch_pack_reads(const uint8_t* reads, const uint64_t* read_offs, uint32_t n, uint64_t total_bytes, uint64_t* fw3, uint64_t* nmw,
                             uint32_t* hasn, hipStream_t stream) {
    return launch_pack_impl(reads, nullptr, read_offs, n, total_bytes, total_bytes, fw3, nmw, hasn, stream);
}
hipError_t launch_pack_reads_at(const uint8_t* text, const uint32_t* src_off, const uint64_t* read_offs, uint32_t n, uint64_t text_bytes, uint64_t total_bases,
                                uint64_t* fw3, uint64_t* nmw, uint32_t* hasn, hipStream_t stream) {
    return launch_pack_impl(text, src_off, read_offs, n, text_bytes, total_bases, fw3, nmw, hasn, stream);
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

hipError_t launch_greedy(const BgrDeviceGraph& g, const BatchIO& io, const KernelParams& p, const LaunchCfg& cfg, hipStream_t stream);
hipError_t launch_exhaustive(const BgrDeviceGraph& g, const BatchIO& io, const KernelParams& p, const LaunchCfg& cfg, hipStream_t stream);
hipError_t launch_anchors(const BgrDeviceGraph& g, const BatchIO& io, const KernelParams& p, const LaunchCfg& cfg, hipStream_t stream);
const void* greedy_kernel_fn(bool four_reads);
const void* exhaustive_kernel_fn(uint32_t which);
const void* anchors_kernel_fn(bool four_reads);

uint32_t resident_waves_per_cu(uint32_t mode) {
    hipFuncAttributes fa;
    const void* fn = mode == 0 ? greedy_kernel_fn(false)
                   : mode == 2 ? anchors_kernel_fn(false)
                   : mode == 6 ? anchors_kernel_fn(true)
                   : mode == 3 ? exhaustive_kernel_fn(1)
                   : mode == 4 ? greedy_kernel_fn(true)
                   : mode == 5 ? exhaustive_kernel_fn(2)
                               : exhaustive_kernel_fn(0);
    if (hipFuncGetAttributes(&fa, fn) != hipSuccess || fa.numRegs <= 0) return 16;
    // MI355X_MICROARCH.md "Register files": 512 VGPRs per SIMD lane, allocation granule 8, at most 8 waves per SIMD;
    // the one-read-per-wave kernels use ~106 SGPRs and are compiled for at most 6 waves per SIMD (compiling for 7: 72 VGPRs,
    // spills: 381 vs 532 Mreads/s greedy, 28 vs 37 exhaustive, round 1); the four-reads-per-wave greedy kernel runs 8.
    const uint32_t alloc = ((uint32_t)fa.numRegs + 7) / 8 * 8;
    return 4 * std::min<uint32_t>(mode == 4 ? 8 : 6, 512 / alloc);
}

hipError_t launch_align(const BgrDeviceGraph& g, const BatchIO& io, const KernelParams& p, const LaunchCfg& cfg, hipStream_t stream) {
    if (io.n_reads == 0) return hipSuccess;
    if (p.mode == 0) return launch_greedy(g, io, p, cfg, stream);
    if (p.mode == 2) return launch_anchors(g, io, p, cfg, stream);
    return launch_exhaustive(g, io, p, cfg, stream);
}

}  // namespace bgr
