// text_kernels.hip -- the two ends of Aligner::alignPartGreedy's per-batch work on the device (gfx950), so that a batch crosses
// PCIe as the file's own bytes and comes back as the bytes to write:
//   in   getReads (aligner.cpp:46-117) for the shape nearly every FASTA piece has -- header line, ONE sequence line, next header --
//        records found, checked (alphabet ACGTN, size > 2, size > k) and handed to the mapping kernels as packed planes;
//        a piece of any other shape (multi-line sequences, empty lines, a last record without its newline ...) raises a flag and
//        the caller parses it on the host with the exact state machine (fastx.cpp), so the accepted records are the reference's;
//   out  the records as the reference writes them: mapped -> header '\n' printPath (aligner.cpp:600-609: to_string(int) + '.')
//        '\n' (alignerGreedy.cpp:406-411), the others -> header '\n' read '\n' (alignerGreedy.cpp:421-427), each stream in input
//        order at offsets from a device-wide scan.
// Streaming byte work: HBM-bound by design, no MFMA.
#include "device_common.h"
#include "text_kernels.h"

namespace bgr {
namespace {

typedef uint32_t __attribute__((ext_vector_type(4), aligned(1))) u32x4_unaligned;

constexpr uint32_t kTxtThreads = 1024, kTxtTile = kTxtThreads * 16;  // bytes per workgroup of the marking kernel

__device__ __forceinline__ uint32_t block_exclusive_scan32(uint32_t v, uint32_t* lds_waves, uint32_t* block_total) {
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

// ---- device-wide exclusive scan of a u32 array (4096 items per workgroup; sums, one-workgroup scan of the sums, apply) ----------
constexpr uint32_t kScanItems = 4, kScanTile = kTxtThreads * kScanItems;
__global__ void __launch_bounds__(kTxtThreads) bgr_scan_block_sums(const uint32_t* in, uint32_t n, uint32_t* sums) {
    __shared__ uint32_t lw[16];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t s = 0;
#pragma unroll
    for (uint32_t j = 0; j < kScanItems; ++j) if (base + j < n) s += in[base + j];
    uint32_t total;
    (void)block_exclusive_scan32(s, lw, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
__global__ void __launch_bounds__(kTxtThreads) bgr_scan_sums(uint32_t* sums, uint32_t nb, uint32_t* total_out) {
    __shared__ uint32_t lw[16];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nb; b0 += kTxtThreads) {
        const uint32_t i = b0 + threadIdx.x;
        const uint32_t v = i < nb ? sums[i] : 0;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan32(v, lw, &total);
        const uint32_t carry = carry_s;
        if (i < nb) sums[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry_s;
}
__global__ void __launch_bounds__(kTxtThreads) bgr_scan_apply(const uint32_t* in, uint32_t n, const uint32_t* sums, uint32_t* out) {
    __shared__ uint32_t lw[16];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t v[kScanItems], s = 0;
#pragma unroll
    for (uint32_t j = 0; j < kScanItems; ++j) { v[j] = base + j < n ? in[base + j] : 0; s += v[j]; }
    uint32_t total;
    uint32_t w = sums[blockIdx.x] + block_exclusive_scan32(s, lw, &total);
#pragma unroll
    for (uint32_t j = 0; j < kScanItems; ++j) {
        if (base + j < n) out[base + j] = w;
        w += v[j];
    }
}

// ---- the same for TWO arrays of equal length in one go, over the first min(n, *n_dev) entries (n_dev may be null) -------------------------
// (round 5: the text form scanned four arrays per piece with three launches each, over all rec_cap entries of arrays of which a sixth is used --
// the record count is only known on the device; now two pairs, and workgroups beyond the count leave at once)
__global__ void __launch_bounds__(kTxtThreads) bgr_scan2_block_sums(const uint32_t* inA, const uint32_t* inB, uint32_t n, const uint32_t* n_dev, uint32_t* sums, uint32_t nb) {
    __shared__ uint32_t lw[16];
    if (n_dev && *n_dev < n) n = *n_dev;
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    if (blockIdx.x * kScanTile >= n) { if (threadIdx.x == 0) { sums[blockIdx.x] = 0; sums[nb + blockIdx.x] = 0; } return; }
    uint32_t a = 0, b = 0;
#pragma unroll
    for (uint32_t j = 0; j < kScanItems; ++j) if (base + j < n) { a += inA[base + j]; b += inB[base + j]; }
    uint32_t ta, tb;
    (void)block_exclusive_scan32(a, lw, &ta);
    (void)block_exclusive_scan32(b, lw, &tb);
    if (threadIdx.x == 0) { sums[blockIdx.x] = ta; sums[nb + blockIdx.x] = tb; }
}
__global__ void __launch_bounds__(kTxtThreads) bgr_scan2_sums(uint32_t* sums, uint32_t nb, uint32_t* totalA, uint32_t* totalB) {
    __shared__ uint32_t lw[16];
    __shared__ uint32_t carry_s[2];
    if (threadIdx.x < 2) carry_s[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t half = 0; half < 2; ++half) {
        uint32_t* S = sums + half * nb;
        for (uint32_t b0 = 0; b0 < nb; b0 += kTxtThreads) {
            const uint32_t i = b0 + threadIdx.x;
            const uint32_t v = i < nb ? S[i] : 0;
            uint32_t total;
            const uint32_t ex = block_exclusive_scan32(v, lw, &total);
            const uint32_t carry = carry_s[half];
            if (i < nb) S[i] = carry + ex;
            __syncthreads();
            if (threadIdx.x == 0) carry_s[half] = carry + total;
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) { *totalA = carry_s[0]; *totalB = carry_s[1]; }
}
__global__ void __launch_bounds__(kTxtThreads) bgr_scan2_apply(const uint32_t* inA, const uint32_t* inB, uint32_t n, const uint32_t* n_dev, const uint32_t* sums, uint32_t nb,
                                                               uint32_t* outA, uint32_t* outB) {
    __shared__ uint32_t lw[16];
    if (n_dev && *n_dev < n) n = *n_dev;
    if (blockIdx.x * kScanTile >= n) return;
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t va[kScanItems], vb[kScanItems], a = 0, b = 0;
#pragma unroll
    for (uint32_t j = 0; j < kScanItems; ++j) { va[j] = base + j < n ? inA[base + j] : 0; vb[j] = base + j < n ? inB[base + j] : 0; a += va[j]; b += vb[j]; }
    uint32_t total;
    uint32_t wa = sums[blockIdx.x] + block_exclusive_scan32(a, lw, &total);
    uint32_t wb = sums[nb + blockIdx.x] + block_exclusive_scan32(b, lw, &total);
#pragma unroll
    for (uint32_t j = 0; j < kScanItems; ++j) {
        if (base + j < n) { outA[base + j] = wa; outB[base + j] = wb; }
        wa += va[j]; wb += vb[j];
    }
}

// ---- record starts: a '>' at the start of a line ---------------------------------------------------------------------------
// bit 7 of every byte of x that equals the byte replicated in `pat`
__device__ __forceinline__ uint32_t eq_bytes(uint32_t x, uint32_t pat) { return bgr_zero_bytes(x ^ pat); }

// phase 0: record starts per 16 KB tile -> sums[tile]; phase 1: rec_start[] (byte offsets, ascending) from the scanned sums
// (rec_cap: room in rec_start; a piece with more record starts than that is left to the host, the caller sees it from the count)
// FASTQ (-q): record j is lines 4j .. 4j+3 whatever they contain (aligner.cpp:51-68), so the marks are the NEWLINES; the byte behind
// every fourth one starts a record (the piece holds whole records and starts at one: the caller cuts it so)
// (FASTQ = lines per record: 4, or 2 when the caller has left the '+' and quality lines on the host; 0 = FASTA)
template <int PHASE, int FASTQ>
__global__ void __launch_bounds__(kTxtThreads) bgr_text_mark_kernel(const uint8_t* text, uint32_t n, uint32_t* sums, uint32_t* rec_start, uint32_t rec_cap) {
    __shared__ uint32_t lw[16];
    constexpr uint32_t LN = FASTQ ? (uint32_t)FASTQ : 1u;  // lines per FASTQ record (a power of two)
    const uint32_t pos = blockIdx.x * kTxtTile + threadIdx.x * 16;
    uint32_t rs = 0;  // bit i: byte pos + i starts a record (FASTQ: is a newline)
    if (FASTQ) {
        if (pos < n) {
            const uint4 v = *reinterpret_cast<const uint4*>(text + pos);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint32_t nl = eq_bytes(w[d], 0x0A0A0A0Au) >> 7;
                rs |= (((nl & 1u) | ((nl >> 7) & 2u) | ((nl >> 14) & 4u) | ((nl >> 21) & 8u)) << (4 * d));
            }
            if (pos + 16 > n) rs &= (1u << (n - pos)) - 1u;
        }
    } else if (pos < n) {
        const uint4 v = *reinterpret_cast<const uint4*>(text + pos);  // (the buffer is zero padded: a zero byte is neither '>' nor '\n')
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t prev_nl = pos == 0 ? 1u : (text[pos - 1] == '\n' ? 1u : 0u);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t gt = eq_bytes(w[d], 0x3E3E3E3Eu) >> 7, nl = eq_bytes(w[d], 0x0A0A0A0Au) >> 7;  // bit 0 of each byte
            const uint32_t after_nl = (nl << 8) | prev_nl;   // byte i follows a newline
            const uint32_t hit = gt & after_nl;
            rs |= (((hit & 1u) | ((hit >> 7) & 2u) | ((hit >> 14) & 4u) | ((hit >> 21) & 8u)) << (4 * d));
            prev_nl = nl >> 24;
        }
        if (pos + 16 > n) rs &= (1u << (n - pos)) - 1u;
    }
    const uint32_t cnt = (uint32_t)__popc(rs);
    uint32_t total;
    const uint32_t ex = block_exclusive_scan32(cnt, lw, &total);
    if (PHASE == 0) {
        if (threadIdx.x == 0) sums[blockIdx.x] = total;
    } else {
        uint32_t at = sums[blockIdx.x] + ex;
        if (FASTQ && blockIdx.x == 0 && threadIdx.x == 0 && n) rec_start[0] = 0;
        while (rs) {
            const uint32_t i = (uint32_t)__ffs((int)rs) - 1;
            rs &= rs - 1;
            if (!FASTQ) { if (at < rec_cap) rec_start[at] = pos + i; }
            else if (((at + 1) & (LN - 1u)) == 0 && ((at + 1) / LN) < rec_cap && pos + i + 1 < n) rec_start[(at + 1) / LN] = pos + i + 1;  // behind the 4th, 8th ... (2nd, 4th ...) newline
            ++at;
        }
    }
}
// FASTQ: newlines counted -> records (a piece of whole records ends with a newline: 4 per record)
__global__ void bgr_text_fastq_count_kernel(uint32_t* n_rec, uint32_t lines) { *n_rec = *n_rec / lines; }

// ---- records: header / sequence extents, shape check, accept test ------------------------------------------------------------
// min over the 16 lanes of a row, result in every lane
__device__ __forceinline__ uint32_t row16_min(uint32_t x) {
    uint32_t o = quad_xor1(x); x = o < x ? o : x;
    o = quad_xor2(x); x = o < x ? o : x;
    o = half_row_mirror(x); x = o < x ? o : x;
    o = (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x140, 0xF, 0xF, true); x = o < x ? o : x;  // row_mirror
    return x;
}
// 16 bytes from `p` (any alignment; bytes at or beyond `end` read as zero)
__device__ __forceinline__ void load16(const uint8_t* text, uint32_t p, uint32_t end, uint32_t w[4]) {
    w[0] = w[1] = w[2] = w[3] = 0;
    if (p >= end) return;
    const u32x4_unaligned v = *reinterpret_cast<const u32x4_unaligned*>(text + p);  // (the buffer is padded by 32 bytes)
    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    const uint32_t valid = end - p;
    if (valid < 16) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t lo = 4u * d;
            if (valid <= lo) w[d] = 0;
            else if (valid < lo + 4) w[d] &= 0xFFFFFFFFu >> (8 * (lo + 4 - valid));
        }
    }
}
// bit i (0..15) = byte i of the 16 loaded equals c
__device__ __forceinline__ uint32_t mask16(const uint32_t w[4], uint32_t pat) {
    uint32_t m = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const uint32_t h = eq_bytes(w[d], pat) >> 7;
        m |= ((h & 1u) | ((h >> 7) & 2u) | ((h >> 14) & 4u) | ((h >> 21) & 8u)) << (4 * d);
    }
    return m;
}

// One 16-lane group per record j: bytes [rec_start[j], rec_start[j+1] or n).  The shape this route takes: exactly two
// newlines, the second one the record's last byte (header line + one sequence line).  Anything else sets *irregular.
// rec[j] = {header offset, header length, sequence offset, sequence length | accepted << 31}
// FASTQ: a record is four lines; header = the first, read = the second; accepted when size > 2 and ACGTN only (aligner.cpp:54-66: no
// size > k test), and there is no "other shape" -- record j is lines 4j .. 4j+3 whatever they hold.
template <bool FASTQ>
__global__ void __launch_bounds__(256) bgr_text_records_kernel(const uint8_t* text, uint32_t n, const uint32_t* rec_start, const uint32_t* n_rec_p, uint32_t k,
                                                               uint4* rec, uint32_t* acc_flag, uint32_t* acc_len, uint32_t* info, uint32_t rec_cap) {
    const uint32_t R = *n_rec_p;
    const uint32_t sub = threadIdx.x & 15;
    // (more record starts than rec_start holds: the caller hands the piece to the host once it has read the count; the scans behind this kernel
    // stop at min(count, rec_cap) and the compaction leaves at once -- nothing reads what this kernel did not write.  Round 4 launched one group
    // per rec_cap entry, five in six of them only to write zeroes for scans that ran over all of them: 111 us per 44 MB piece, 0.40 TB/s)
    if (R > rec_cap) return;
    for (uint32_t j = (blockIdx.x * blockDim.x + threadIdx.x) >> 4; j < R; j += (gridDim.x * blockDim.x) >> 4) {
    const uint32_t p = rec_start[j], q = j + 1 < R ? rec_start[j + 1] : n;
    // pass 1: the first two newlines and the number of newlines
    uint32_t first = 0xFFFFFFFFu, second = 0xFFFFFFFFu, count = 0;
    for (uint32_t base = p; base < q; base += 256) {
        uint32_t w[4];
        const uint32_t at = base + 16 * sub;
        load16(text, at, q, w);
        uint32_t m = mask16(w, 0x0A0A0A0Au);
        count += row16_sum((uint32_t)__popc(m));
        if (first == 0xFFFFFFFFu) {
            const uint32_t mine = m ? at + (uint32_t)__ffs((int)m) - 1 : 0xFFFFFFFFu;
            first = row16_min(mine);
            if (m && mine == first) m &= m - 1;  // the lane that holds it looks past it; every other newline lies behind it anyway
        }
        if (first != 0xFFFFFFFFu && second == 0xFFFFFFFFu) second = row16_min(m ? at + (uint32_t)__ffs((int)m) - 1 : 0xFFFFFFFFu);
        if (count > 2 || (FASTQ && count >= 2)) break;
    }
    const bool regular = FASTQ ? (first != 0xFFFFFFFFu && second != 0xFFFFFFFFu)
                               : (count == 2 && second == q - 1 && (j != 0 || p == 0));  // (bytes in front of the piece's first record start: not this route's shape)
    uint32_t hl = 0, so = 0, L = 0, ok = 0;
    if (regular) {
        hl = first - p;
        so = first + 1;
        L = second - so;
        // pass 2: every character of the read in ACGTN (aligner.cpp:79-84)
        uint32_t bad = 0;
        for (uint32_t base = so; base < second; base += 256) {
            uint32_t w[4];
            const uint32_t at = base + 16 * sub;
            load16(text, at, second, w);
            const uint32_t valid = at < second ? (second - at < 16 ? second - at : 16u) : 0u;  // bytes of the read in this lane's 16
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint32_t x = w[d];
                const uint32_t good = eq_bytes(x, 0x41414141u) | eq_bytes(x, 0x43434343u) | eq_bytes(x, 0x47474747u) | eq_bytes(x, 0x54545454u) | eq_bytes(x, 0x4E4E4E4Eu);
                const uint32_t nv = valid > 4u * d ? (valid - 4u * d < 4u ? valid - 4u * d : 4u) : 0u;
                const uint32_t in_read = nv == 4 ? 0x80808080u : (0x80808080u & ((1u << (8 * nv)) - 1u));
                bad |= (good ^ 0x80808080u) & in_read;
            }
        }
        bad = row16_sum(bad ? 1u : 0u);
        ok = (L > 2 && bad == 0 && (FASTQ || L > k)) ? 1u : 0u;  // aligner.cpp:78-88: size > 2, alphabet, size > k (FASTA only)
    }
    if (sub == 0) {
        if (!regular && !__hip_atomic_load(&info[TXT_INFO_IRREGULAR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&info[TXT_INFO_IRREGULAR], 1u);
        rec[j] = make_uint4(p, hl, so, L | (ok << 31));
        acc_flag[j] = ok;
        acc_len[j] = ok ? L : 0u;
        // (one atomic per record on one word would cap the kernel near 90 M records/s: only a read longer than what the word holds adds)
        if (ok && L > __hip_atomic_load(&info[TXT_INFO_MAX_LEN], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&info[TXT_INFO_MAX_LEN], L);
    }
    }
}

// what became of every record, for a caller that reproduces the reference's -b progress blocks: kept << 31 | mapped << 30 | read length
__global__ void __launch_bounds__(256) bgr_text_record_info_kernel(const uint4* rec, const uint32_t* acc_idx, const uint2* results, uint32_t n_rec, uint32_t* out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_rec) return;
    const uint32_t w = rec[j].w;
    uint32_t v = 0;
    if (w >> 31) {
        const uint32_t st = results[acc_idx[j]].y >> 24;
        v = 0x80000000u | ((st & BGR_ST_MASK) == BGR_ST_ALIGNED ? 0x40000000u : 0u) | (w & 0x3FFFFFFFu);
    }
    out[j] = v;
}

// accepted records, compacted in input order: which record, where its sequence starts in the text, base offsets of the batch
__global__ void __launch_bounds__(256) bgr_text_compact_kernel(const uint4* rec, const uint32_t* n_rec_p, const uint32_t* acc_idx, const uint32_t* base_off,
                                                               uint32_t* acc_rec, uint32_t* acc_src, u64* read_offs, const uint32_t* n_acc_p, const uint32_t* bases_p, uint32_t rec_cap) {
    const uint32_t R = *n_rec_p;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (R > rec_cap) return;  // (rec[] was not written: the piece goes to the host parser)
    if (j == 0) read_offs[*n_acc_p] = *bases_p;
    if (j >= R) return;
    const uint4 r = rec[j];
    if (!(r.w >> 31)) return;
    const uint32_t a = acc_idx[j];
    acc_rec[a] = j;
    acc_src[a] = r.z;
    read_offs[a] = base_off[j];
}

// ---- out: sizes of the formatted records -----------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t dec_len(int32_t v) {  // characters of to_string(v) + '.'
    uint32_t u = (uint32_t)v, len = 1;
    if (v < 0) { u = 0u - u; len = 2; }
    len += u < 10 ? 1 : u < 100 ? 2 : u < 1000 ? 3 : u < 10000 ? 4 : u < 100000 ? 5 : u < 1000000 ? 6 : u < 10000000 ? 7 : u < 100000000 ? 8 : u < 1000000000 ? 9 : 10;
    return len;
}
__global__ void __launch_bounds__(256) bgr_text_sizes_kernel(const uint2* results, const int32_t* arena, const uint4* rec, const uint32_t* acc_rec, uint32_t n_acc,
                                                             uint32_t* psz, uint32_t* nsz) {
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_acc) return;
    const uint2 res = results[a];
    const uint4 r = rec[acc_rec[a]];
    const uint32_t np = res.y & 0xFFFFFFu;
    uint32_t ps = 0, ns = 0;
    if (np) {
        ps = r.y + 2;
        for (uint32_t i = 0; i < np; ++i) ps += dec_len(arena[res.x + i]);
    } else {
        ns = r.y + (r.w & 0x7FFFFFFFu) + 2;
    }
    psz[a] = ps;
    nsz[a] = ns;
}

// n bytes from src to dst by the 16 lanes of a group (any alignment on both sides)
__device__ __forceinline__ void group_copy(uint8_t* dst, const uint8_t* src, uint32_t n, uint32_t sub) {
    for (uint32_t b = 16 * sub; b < n; b += 256) {
        if (b + 16 <= n) {
            *reinterpret_cast<u32x4_unaligned*>(dst + b) = *reinterpret_cast<const u32x4_unaligned*>(src + b);
        } else {
            for (uint32_t i = b; i < n; ++i) dst[i] = src[i];
        }
    }
}

// one 16-lane group per accepted read: its record into the paths stream or the notAligned stream
__global__ void __launch_bounds__(256) bgr_text_write_kernel(const uint8_t* text, const uint2* results, const int32_t* arena, const uint4* rec, const uint32_t* acc_rec,
                                                             uint32_t n_acc, const uint32_t* poff, const uint32_t* noff, uint8_t* pout, uint8_t* nout) {
    const uint32_t a = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, sub = threadIdx.x & 15;
    if (a >= n_acc) return;
    const uint2 res = results[a];
    const uint4 r = rec[acc_rec[a]];
    const uint32_t np = res.y & 0xFFFFFFu, hl = r.y;
    if (np) {  // alignerGreedy.cpp:406-411: header + '\n' + printPath
        uint8_t* d = pout + poff[a];
        group_copy(d, text + r.x, hl, sub);
        if (sub == 0) d[hl] = '\n';
        uint32_t cur = hl + 1;
        for (uint32_t i0 = 0; i0 < np; i0 += 16) {
            const uint32_t i = i0 + sub;
            int32_t v = 0;
            uint32_t len = 0;
            if (i < np) { v = arena[res.x + i]; len = dec_len(v); }
            uint32_t inc = len;  // inclusive scan over the 16 lanes of the group
#pragma unroll
            for (int s = 1; s < 16; s <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)inc, s, 16);
                if ((int)sub >= s) inc += up;
            }
            if (i < np) {  // to_string(v) + '.', written back to front
                uint8_t* e = d + cur + inc;  // one past the '.'
                *--e = '.';
                uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
                do { const uint32_t qd = u / 10; *--e = (uint8_t)('0' + (u - qd * 10)); u = qd; } while (u);
                if (v < 0) *--e = '-';
            }
            cur += (uint32_t)__shfl((int)inc, 15, 16);
        }
        if (sub == 0) d[cur] = '\n';
    } else {   // alignerGreedy.cpp:421-427: header + '\n' + read + '\n'
        uint8_t* d = nout + noff[a];
        const uint32_t L = r.w & 0x7FFFFFFFu;
        group_copy(d, text + r.x, hl, sub);
        if (sub == 0) d[hl] = '\n';
        group_copy(d + hl + 1, text + r.z, L, sub);
        if (sub == 0) d[hl + 1 + L] = '\n';
    }
}

// ---- correction mode (-c) on the device: the read as spelled by its path ----------------------------------------------------------
// recoverPath (aligner.cpp:270-290): walk = getUnitig(path[1]); for every further int: compactionEnd(walk, getUnitig(path[i]), k-1)
// (utils.cpp:171-179) glues the unitig on when the walk's last k-1 characters equal its first k-1 -- as oriented by its sign or,
// failing that, reverse complemented -- else "bug compaction"; then walk.substr(path[0], read size), reverse complemented when the
// path was found on the read's reverse complement (alignerGreedy.cpp:394-404).  Graphs without exception planes only (every unitig
// character is one of ACGT, so the 2-bit store spells it); the caller formats such a batch on the host otherwise.
struct WalkIter {   // the oriented unitigs of a path, one after the other
    const BgrDeviceGraph& g;
    const int32_t* path;
    uint32_t np, K1;
    uint32_t i = 1;          // path int of the current unitig
    u64 base = 0;            // where its oriented bases start in seq
    uint32_t len = 0, skip = 0;  // its length; bases of it that belong to the overlap with its predecessor (0 for the first, else k-1)
    bool bad = false;
    __device__ WalkIter(const BgrDeviceGraph& g_, const int32_t* p, uint32_t n) : g(g_), path(p), np(n), K1(g_.k - 1) {}
    __device__ u64 kmer_at(u64 b) const { return win32(g.seq, b) >> (64 - 2 * K1); }
    __device__ bool load() {  // unitig i as its sign orients it; false when the id is out of range (getUnitig of a bad int)
        const int32_t s = path[i];
        const uint32_t id = (uint32_t)(s < 0 ? -(int64_t)s : (int64_t)s);
        if (id == 0 || id > (uint32_t)g.hdr->n_unitigs) return false;
        const BgrUnitigMeta m = g.meta[id];
        len = m.len;
        base = m.F + (s < 0 ? m.len : 0);
        return true;
    }
    __device__ bool first() { skip = 0; i = 1; if (np < 2 || !load()) { bad = true; return false; } return true; }
    __device__ bool next() {   // false: no further unitig (or bad set: bug compaction)
        if (i + 1 >= np) return false;
        const u64 tail = kmer_at(base + len - K1);
        ++i;
        if (!load()) { bad = true; return false; }
        if (kmer_at(base) != tail) {  // compactionEnd's second try: the reverse complement (the other strand of the store)
            const int32_t s = path[i];
            const uint32_t id = (uint32_t)(s < 0 ? -(int64_t)s : (int64_t)s);
            const BgrUnitigMeta m = g.meta[id];
            const u64 other = m.F + (s < 0 ? 0 : m.len);
            if (kmer_at(other) != tail) { bad = true; return false; }
            base = other;
        }
        skip = K1;
        return true;
    }
};

// sizes: mapped read -> header + '\n' + min(read size, walk size - offset) + '\n' in the paths stream; *bug (atomicMin) = the first
// accepted read whose path does not spell a walk
__global__ void __launch_bounds__(256) bgr_text_correct_sizes_kernel(BgrDeviceGraph g, const uint2* results, const int32_t* arena, const uint4* rec, const uint32_t* acc_rec,
                                                                     uint32_t n_acc, uint32_t* psz, uint32_t* nsz, uint32_t* clen, uint32_t* bug) {
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_acc) return;
    const uint2 res = results[a];
    const uint4 r = rec[acc_rec[a]];
    const uint32_t np = res.y & 0xFFFFFFu, L = r.w & 0x7FFFFFFFu;
    uint32_t ps = 0, ns = 0, cl = 0;
    if (np) {
        WalkIter w(g, arena + res.x, np);
        u64 total = 0;
        if (w.first()) { total = w.len; while (w.next()) total += w.len - w.skip; }
        const int32_t off = arena[res.x];
        if (w.bad || off < 0 || (u64)off > total) atomicMin(bug, a);
        else {
            const u64 left = total - (u64)off;
            cl = left < L ? (uint32_t)left : L;
        }
        ps = r.y + 2 + cl;
    } else {
        ns = r.y + L + 2;
    }
    psz[a] = ps;
    nsz[a] = ns;
    clen[a] = cl;
}

__device__ __forceinline__ uint8_t base_char(uint32_t code) { return (uint8_t)((0x54474341u >> (8 * code)) & 0xFF); }  // "ACGT"

// one 16-lane group per accepted read; a mapped read's record is header + '\n' + the corrected read + '\n'
__global__ void __launch_bounds__(256) bgr_text_correct_write_kernel(BgrDeviceGraph g, const uint8_t* text, const uint2* results, const int32_t* arena, const uint4* rec,
                                                                     const uint32_t* acc_rec, uint32_t n_acc, const uint32_t* poff, const uint32_t* noff, const uint32_t* clen,
                                                                     uint8_t* pout, uint8_t* nout) {
    const uint32_t a = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, sub = threadIdx.x & 15;
    if (a >= n_acc) return;
    const uint2 res = results[a];
    const uint4 r = rec[acc_rec[a]];
    const uint32_t np = res.y & 0xFFFFFFu, hl = r.y;
    if (!np) {   // alignerGreedy.cpp:421-427: header + '\n' + read + '\n'
        uint8_t* d = nout + noff[a];
        const uint32_t L = r.w & 0x7FFFFFFFu;
        group_copy(d, text + r.x, hl, sub);
        if (sub == 0) d[hl] = '\n';
        group_copy(d + hl + 1, text + r.z, L, sub);
        if (sub == 0) d[hl + 1 + L] = '\n';
        return;
    }
    uint8_t* d = pout + poff[a];
    group_copy(d, text + r.x, hl, sub);
    if (sub == 0) d[hl] = '\n';
    const uint32_t cl = clen[a];
    uint8_t* o = d + hl + 1;
    if (sub == 0) o[cl] = '\n';
    const bool rc = ((res.y >> 24) & BGR_ST_RC) != 0;
    // every lane walks the path (a handful of unitigs) and spells the characters j = sub, sub + 16, ... of walk[off, off + cl)
    const uint32_t off = (uint32_t)arena[res.x];
    WalkIter w(g, arena + res.x, np);
    if (!w.first()) return;
    u64 start = 0;  // walk position of the current unitig's first NEW base (behind the overlap)
    for (;;) {
        const u64 new_len = w.len - w.skip, end = start + new_len;   // walk positions [start, end) come from this unitig at oriented offsets skip + (p - start)
        // walk positions wanted: off + j, j < cl
        if (end > off && start < (u64)off + cl) {
            const u64 lo = start > off ? start : off, hi = end < (u64)off + cl ? end : (u64)off + cl;
            // j with off + j in [lo, hi), j = sub mod 16
            uint32_t j = (uint32_t)(lo - off);
            j += (sub + 16 - (j & 15)) & 15;
            for (; (u64)off + j < hi; j += 16) {
                const u64 p = w.base + w.skip + ((u64)off + j - start);
                const uint32_t code = (uint32_t)(g.seq[p >> 5] >> (62 - 2 * (p & 31))) & 3u;
                if (!rc) o[j] = base_char(code);
                else o[cl - 1 - j] = base_char(3u - code);  // reverseComplements(corrected) (utils.cpp:66-73; ACGT only here)
            }
        }
        start = end;
        if (start >= (u64)off + cl) break;
        if (!w.next()) break;
    }
}

}  // namespace

hipError_t launch_scan_u32(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* sums, uint32_t* total_out, hipStream_t stream) {
    const uint32_t nb = std::max<uint32_t>(1, (n + kScanTile - 1) / kScanTile);
    hipLaunchKernelGGL(bgr_scan_block_sums, dim3(nb), dim3(kTxtThreads), 0, stream, in, n, sums);
    hipLaunchKernelGGL(bgr_scan_sums, dim3(1), dim3(kTxtThreads), 0, stream, sums, nb, total_out);
    hipLaunchKernelGGL(bgr_scan_apply, dim3(nb), dim3(kTxtThreads), 0, stream, in, n, sums, out);
    return hipGetLastError();
}
hipError_t launch_scan2_u32(const uint32_t* inA, const uint32_t* inB, uint32_t* outA, uint32_t* outB, uint32_t n, const uint32_t* n_dev, uint32_t* sums, uint32_t* totalA,
                            uint32_t* totalB, hipStream_t stream) {
    const uint32_t nb = std::max<uint32_t>(1, (n + kScanTile - 1) / kScanTile);
    hipLaunchKernelGGL(bgr_scan2_block_sums, dim3(nb), dim3(kTxtThreads), 0, stream, inA, inB, n, n_dev, sums, nb);
    hipLaunchKernelGGL(bgr_scan2_sums, dim3(1), dim3(kTxtThreads), 0, stream, sums, nb, totalA, totalB);
    hipLaunchKernelGGL(bgr_scan2_apply, dim3(nb), dim3(kTxtThreads), 0, stream, inA, inB, n, n_dev, sums, nb, outA, outB);
    return hipGetLastError();
}
uint32_t scan_tiles(uint32_t n) { return std::max<uint32_t>(1, (n + kScanTile - 1) / kScanTile); }
uint32_t text_tiles(uint32_t bytes) { return std::max<uint32_t>(1, (bytes + kTxtTile - 1) / kTxtTile); }

hipError_t launch_text_mark(const uint8_t* text, uint32_t n, uint32_t fastq_lines, uint32_t* sums, uint32_t* rec_start, uint32_t rec_cap, uint32_t* n_rec_out, hipStream_t stream) {
    const uint32_t nb = text_tiles(n);
    if (fastq_lines) {
        if (fastq_lines == 2) hipLaunchKernelGGL((bgr_text_mark_kernel<0, 2>), dim3(nb), dim3(kTxtThreads), 0, stream, text, n, sums, rec_start, rec_cap);
        else hipLaunchKernelGGL((bgr_text_mark_kernel<0, 4>), dim3(nb), dim3(kTxtThreads), 0, stream, text, n, sums, rec_start, rec_cap);
        hipLaunchKernelGGL(bgr_scan_sums, dim3(1), dim3(kTxtThreads), 0, stream, sums, nb, n_rec_out);
        if (fastq_lines == 2) hipLaunchKernelGGL((bgr_text_mark_kernel<1, 2>), dim3(nb), dim3(kTxtThreads), 0, stream, text, n, sums, rec_start, rec_cap);
        else hipLaunchKernelGGL((bgr_text_mark_kernel<1, 4>), dim3(nb), dim3(kTxtThreads), 0, stream, text, n, sums, rec_start, rec_cap);
        hipLaunchKernelGGL(bgr_text_fastq_count_kernel, dim3(1), dim3(1), 0, stream, n_rec_out, fastq_lines);
    } else {
        hipLaunchKernelGGL((bgr_text_mark_kernel<0, 0>), dim3(nb), dim3(kTxtThreads), 0, stream, text, n, sums, rec_start, rec_cap);
        hipLaunchKernelGGL(bgr_scan_sums, dim3(1), dim3(kTxtThreads), 0, stream, sums, nb, n_rec_out);
        hipLaunchKernelGGL((bgr_text_mark_kernel<1, 0>), dim3(nb), dim3(kTxtThreads), 0, stream, text, n, sums, rec_start, rec_cap);
    }
    return hipGetLastError();
}

hipError_t launch_text_records(const uint8_t* text, uint32_t n, bool fastq, const uint32_t* rec_start, const uint32_t* n_rec_p, uint32_t max_rec, uint32_t k, uint4* rec,
                               uint32_t* acc_flag, uint32_t* acc_len, uint32_t* info, hipStream_t stream) {
    if (max_rec == 0) return hipSuccess;
    // (the record count lives on the device: a grid of the device's size, 16-lane groups striding over the records; ~110 bytes of text per record at least)
    const uint32_t blocks = std::max<uint32_t>(1, std::min<uint32_t>((max_rec + 15) / 16, std::min<uint32_t>(256 * 16, n / (16 * 110) + 1)));
    if (fastq) hipLaunchKernelGGL(bgr_text_records_kernel<true>, dim3(blocks), dim3(256), 0, stream, text, n, rec_start, n_rec_p, k, rec, acc_flag, acc_len, info, max_rec);
    else hipLaunchKernelGGL(bgr_text_records_kernel<false>, dim3(blocks), dim3(256), 0, stream, text, n, rec_start, n_rec_p, k, rec, acc_flag, acc_len, info, max_rec);
    return hipGetLastError();
}

hipError_t launch_text_compact(const uint4* rec, const uint32_t* n_rec_p, uint32_t max_rec, const uint32_t* acc_idx, const uint32_t* base_off, uint32_t* acc_rec,
                               uint32_t* acc_src, uint64_t* read_offs, const uint32_t* n_acc_p, const uint32_t* bases_p, hipStream_t stream) {
    hipLaunchKernelGGL(bgr_text_compact_kernel, dim3(std::max<uint32_t>(1, (max_rec + 255) / 256)), dim3(256), 0, stream, rec, n_rec_p, acc_idx, base_off, acc_rec, acc_src,
                       reinterpret_cast<u64*>(read_offs), n_acc_p, bases_p, max_rec);
    return hipGetLastError();
}

hipError_t launch_text_record_info(const uint4* rec, const uint32_t* acc_idx, const uint2* results, uint32_t n_rec, uint32_t* out, hipStream_t stream) {
    if (n_rec == 0) return hipSuccess;
    hipLaunchKernelGGL(bgr_text_record_info_kernel, dim3((n_rec + 255) / 256), dim3(256), 0, stream, rec, acc_idx, results, n_rec, out);
    return hipGetLastError();
}

hipError_t launch_text_sizes(const uint2* results, const int32_t* arena, const uint4* rec, const uint32_t* acc_rec, uint32_t n_acc, uint32_t* psz, uint32_t* nsz,
                             hipStream_t stream) {
    if (n_acc == 0) return hipSuccess;
    hipLaunchKernelGGL(bgr_text_sizes_kernel, dim3((n_acc + 255) / 256), dim3(256), 0, stream, results, arena, rec, acc_rec, n_acc, psz, nsz);
    return hipGetLastError();
}

hipError_t launch_text_write(const uint8_t* text, const uint2* results, const int32_t* arena, const uint4* rec, const uint32_t* acc_rec, uint32_t n_acc,
                             const uint32_t* poff, const uint32_t* noff, uint8_t* pout, uint8_t* nout, hipStream_t stream) {
    if (n_acc == 0) return hipSuccess;
    hipLaunchKernelGGL(bgr_text_write_kernel, dim3((n_acc + 15) / 16), dim3(256), 0, stream, text, results, arena, rec, acc_rec, n_acc, poff, noff, pout, nout);
    return hipGetLastError();
}

hipError_t launch_text_correct_sizes(const BgrDeviceGraph& g, const uint2* results, const int32_t* arena, const uint4* rec, const uint32_t* acc_rec, uint32_t n_acc, uint32_t* psz,
                                     uint32_t* nsz, uint32_t* clen, uint32_t* bug, hipStream_t stream) {
    if (n_acc == 0) return hipSuccess;
    hipLaunchKernelGGL(bgr_text_correct_sizes_kernel, dim3((n_acc + 255) / 256), dim3(256), 0, stream, g, results, arena, rec, acc_rec, n_acc, psz, nsz, clen, bug);
    return hipGetLastError();
}

hipError_t launch_text_correct_write(const BgrDeviceGraph& g, const uint8_t* text, const uint2* results, const int32_t* arena, const uint4* rec, const uint32_t* acc_rec, uint32_t n_acc,
                                     const uint32_t* poff, const uint32_t* noff, const uint32_t* clen, uint8_t* pout, uint8_t* nout, hipStream_t stream) {
    if (n_acc == 0) return hipSuccess;
    hipLaunchKernelGGL(bgr_text_correct_write_kernel, dim3((n_acc + 15) / 16), dim3(256), 0, stream, g, text, results, arena, rec, acc_rec, n_acc, poff, noff, clen, pout, nout);
    return hipGetLastError();
}

}  // namespace bgr
