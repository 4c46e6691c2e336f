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

#ifdef BGR_X_TIMES
__device__ unsigned long long bgr_x_times[4 * 16];
extern "C" int bgr_x_times_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bgr_x_times), sizeof(bgr_x_times)); }
#define BGR_XT(slot, ph) do { if (threadIdx.x == 0 && (vt == 0 || vt == ntiles / 2)) bgr_x_times[((slot) * 2 + (vt ? 1 : 0)) * 16 + (ph)] = wall_clock64(); } while (0)
#else
#define BGR_XT(slot, ph) do {} while (0)
#endif

namespace bgr {
namespace {

typedef uint32_t __attribute__((ext_vector_type(4), aligned(1))) u32x4_unaligned;

constexpr uint32_t kTxtThreads = 1024;

// inclusive scan over the 64 lanes of a wave without LDS: four row_shr steps scan each row of 16, row_bcast:15 carries row 0 into row 1 and row 2 into row 3,
// row_bcast:31 carries rows 0..1 into rows 2..3 (a lane whose source lies outside its row, or whose row is masked out, adds the 0 given as `old`)
#define BGR_DPP0(x, ctrl, rows) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), ctrl, rows, 0xF, false))
__device__ __forceinline__ uint32_t wave_inclusive_scan32(uint32_t x) {
    x += BGR_DPP0(x, 0x111, 0xF); x += BGR_DPP0(x, 0x112, 0xF); x += BGR_DPP0(x, 0x114, 0xF); x += BGR_DPP0(x, 0x118, 0xF);
    x += BGR_DPP0(x, 0x142, 0xA);
    x += BGR_DPP0(x, 0x143, 0xC);
    return x;
}
#define BGR_DPP0_64(x, ctrl, rows) (((u64)BGR_DPP0((uint32_t)((x) >> 32), ctrl, rows) << 32) | BGR_DPP0((uint32_t)(x), ctrl, rows))
__device__ __forceinline__ u64 wave_inclusive_scan64(u64 x) {
    x += BGR_DPP0_64(x, 0x111, 0xF); x += BGR_DPP0_64(x, 0x112, 0xF); x += BGR_DPP0_64(x, 0x114, 0xF); x += BGR_DPP0_64(x, 0x118, 0xF);
    x += BGR_DPP0_64(x, 0x142, 0xA);
    x += BGR_DPP0_64(x, 0x143, 0xC);
    return x;
}

__device__ __forceinline__ uint32_t block_exclusive_scan32(uint32_t v, uint32_t* lds_waves, uint32_t* block_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;   // (up to 16 waves)
    const uint32_t inc = wave_inclusive_scan32(v);
    if (lane == 63) lds_waves[wave] = inc;
    __syncthreads();
    // every wave scans the (at most 16) wave totals itself: one LDS read, four shuffles (a loop over them waits for LDS sixteen times)
    const uint32_t t = lane < nw ? lds_waves[lane] : 0u;
    uint32_t ti = t;
    ti += BGR_DPP0(ti, 0x111, 0xF); ti += BGR_DPP0(ti, 0x112, 0xF); ti += BGR_DPP0(ti, 0x114, 0xF); ti += BGR_DPP0(ti, 0x118, 0xF);   // (row 0 holds them)
    const uint32_t before = (uint32_t)__shfl((int)(ti - t), wave, 64);
    *block_total = (uint32_t)__shfl((int)ti, 15, 64);
    __syncthreads();
    return before + inc - v;
}

constexpr uint32_t kScanItems = 4, kScanTile = kTxtThreads * kScanItems;

// ---- device-wide exclusive scan of TWO u32 arrays of equal length in one go (4096 items per workgroup; sums, one-workgroup scan of the sums, apply),
// over the first min(n, *n_dev) entries (n_dev may be null).  Correction mode's sizes; the default route's totals run down chains inside its two kernels.
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

// bit 7 of every byte of x that equals the byte replicated in `pat`
__device__ __forceinline__ uint32_t eq_bytes(uint32_t x, uint32_t pat) { return bgr_zero_bytes(x ^ pat); }
// bits 7, 15, 23, 31 of h -> bits 0..3 (one multiply: the four shifted copies do not meet)
__device__ __forceinline__ uint32_t nibble_of(uint32_t h) { return (((h >> 7) * 0x00204081u) >> 21) & 0xFu; }

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
    for (int d = 0; d < 4; ++d) m |= nibble_of(eq_bytes(w[d], pat)) << (4 * d);
    return m;
}

// ---- a running total handed from tile to tile inside ONE launch (decoupled look-back) ----------------------------------------------------------
// st[t]: bits 63..62 = 1 tile t's own sum, 2 the sum of tiles 0..t; bits 61..40 the EPOCH of the launch that wrote the word (a word of another
// epoch counts as "nothing yet": the chains are never cleared between launches -- the host hands every launch a fresh epoch and clears them when the
// 22 bits wrap); bits 39..0 the value.  Tiles are numbered by a ticket taken at run time (ticket word - what the host knows earlier launches took), so
// every tile a workgroup waits for has already started.  Called by ALL 64 lanes of one wave; -> the sum of the tiles in front of vt.  The words are
// read and written whole (64-bit, device scope): no fence needed, and nothing else passes between workgroups.
constexpr u64 kChainValue = (1ull << 40) - 1;
__device__ __forceinline__ u64 chain_word(uint32_t flag, uint32_t epoch, u64 value) { return ((u64)flag << 62) | ((u64)epoch << 40) | (value & kChainValue); }
// (in two halves, so that a workgroup can publish its own sum, do other work, and walk the chain when the tiles in front have had time to publish theirs)
__device__ __forceinline__ void chain_publish(u64* st, uint32_t vt, uint32_t epoch, u64 own) {
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(&st[vt], chain_word(vt ? 1 : 2, epoch, own), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 chain_walk(u64* st, uint32_t vt, uint32_t epoch, u64 own) {
    const int lane = threadIdx.x & 63;
    if (vt == 0) return 0;
    u64 before = 0;
    int64_t base = (int64_t)vt - 1;
    // Four windows of 64 tiles per round trip, lane l at tiles base - l, base - 64 - l, ...: 64 consecutive words per load instruction (8 sectors).  All the tiles
    // of a launch start at about the same time, so the tile that knows its running total is far back and every tile reads most of the chain: the number of
    // sector requests on these few cache lines (device-scope loads go past the L2) is what the walk costs -- with lane l at four neighbouring tiles it was
    // four times as many and the chains took 18 us of a 48 us launch.
    for (bool done = false; !done;) {
        u64 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t idx = base - 64 * q - lane;
            v[q] = idx >= 0 ? __hip_atomic_load(&st[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : chain_word(2, epoch, 0);
        }
        u64 sum = 0;
        int taken = 0;   // windows of this round trip that were complete up to a tile that knows its running total / to their end
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (done || taken != q) continue;
            const uint32_t flag = ((uint32_t)(v[q] >> 40) & 0x3FFFFFu) == epoch ? (uint32_t)(v[q] >> 62) : 0u;
            const u64 m_none = __ballot(flag == 0), m_full = __ballot(flag == 2);
            const int stop = m_full ? __ffsll((long long)m_full) - 1 : 63;
            const u64 need = stop == 63 ? ~0ull : ((2ull << stop) - 1ull);
            if (m_none & need) continue;                                      // a tile in it has not published yet: read again from this window on
            sum += lane <= stop ? (v[q] & kChainValue) : 0ull;
            taken = q + 1;
            if (m_full) done = true;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)sum, d, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(sum >> 32), d, 64);
            sum += ((u64)hi << 32) | lo;
        }
        before += sum;
        base -= 64 * taken;
        if (!done && taken < 4) __builtin_amdgcn_s_sleep(20);   // ~0.5 us: whoever is missing is busy publishing; polling faster only adds traffic
    }
    if (lane == 0) __hip_atomic_store(&st[vt], chain_word(2, epoch, before + own), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return before;
}
__device__ __forceinline__ u64 chain_lookback(u64* st, uint32_t vt, uint32_t epoch, u64 own) {
    chain_publish(st, vt, epoch, own);
    return chain_walk(st, vt, epoch, own);
}

__device__ __forceinline__ u64 shfl64(u64 x, int l) { return ((u64)(uint32_t)__shfl((int)(uint32_t)(x >> 32), l, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)x, l, 64); }
// exclusive scans of N 64-bit values per thread over the workgroup, array after array (value n of thread t comes behind value n - 1 of every thread): one pair
// of barriers for all of them.  ex[n] = what lies in front of v[n]; *total = everything.  (fields packed in a value must not carry into each other)
template <int N>
__device__ __forceinline__ void block_exclusive_scan64xN(const u64 (&v)[N], u64* lds_waves /* N * 16 */, u64 (&ex)[N], u64* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    u64 inc[N];
#pragma unroll
    for (int n = 0; n < N; ++n) {
        inc[n] = wave_inclusive_scan64(v[n]);
        if (lane == 63) lds_waves[n * 16 + wave] = inc[n];
    }
    __syncthreads();
    u64 run = 0;
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const u64 t = lane < nw ? lds_waves[n * 16 + lane] : 0ull;
        u64 ti = t;
        ti += BGR_DPP0_64(ti, 0x111, 0xF); ti += BGR_DPP0_64(ti, 0x112, 0xF); ti += BGR_DPP0_64(ti, 0x114, 0xF); ti += BGR_DPP0_64(ti, 0x118, 0xF);
        ex[n] = run + shfl64(ti - t, wave) + inc[n] - v[n];
        run += shfl64(ti, 15);
    }
    *total = run;
    __syncthreads();
}

// ---- the piece in ONE pass: record starts, extents, shape, accept test, accepted records compacted in input order --------------------------------
// (round 4/5: eight launches -- two marking passes and a scan for the record starts, a 16-lane group per record reading the text a third time,
//  three launches of scans, a compaction; 230 us per 44 MB piece.)  A workgroup takes 96 KB of text as three stretches of 32 KB, 32 bytes per thread
// and stretch, and turns each stretch into three bit masks in LDS (newline, '>', "not one of ACGTN"), with the 1 KB behind it seen through the same
// masks.  A record start is a '>' behind a newline (FASTA) or the byte behind every LN-th newline (FASTQ: record j is lines LN*j .. whatever they
// hold, aligner.cpp:51-68; the newline count in front of the tile comes down a chain).  One THREAD per record then reads the record's shape off the
// masks: first newline f, second newline s; the shape this route takes is header line + ONE sequence line, i.e. the byte behind s starts the next
// record or ends the piece (FASTA), anything else raises `irregular` and the caller parses the piece on the host with the exact state machine
// (fastx.cpp).  Accepted: size > 2, ACGTN only, FASTA: size > k (aligner.cpp:54-66, :78-88).  Records that leave the window (long reads) go to
// 16-lane groups that scan the text itself.  Running totals down three chains: record starts / newlines, accepted records, their bases.
// rec[j] = {header offset, header length, sequence offset, sequence length | accepted << 31}
constexpr uint32_t kParseThreads = 1024, kParseStretch = kParseThreads * 32, kParseStretches = 3, kParseTile = kParseStretch * kParseStretches;
constexpr uint32_t kParseHaloWords = 32, kParseWords = kParseThreads + kParseHaloWords, kParseWindow = kParseWords * 32;
constexpr uint32_t kParseMaxLocal = kParseThreads;   // record starts per stretch (a record of 32 bytes is hardly a read: such a piece goes to the host)
constexpr uint32_t kNone = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t next_set(const uint32_t* bits, uint32_t from) {   // first set bit at or behind `from` in the window, or kNone
    if (from >= kParseWindow) return kNone;
    uint32_t w = from >> 5, x = bits[w] & (0xFFFFFFFFu << (from & 31));
    while (!x) { if (++w >= kParseWords) return kNone; x = bits[w]; }
    return (w << 5) + (uint32_t)__ffs((int)x) - 1;
}
__device__ __forceinline__ bool any_set(const uint32_t* bits, uint32_t lo, uint32_t hi) {   // a set bit in [lo, hi), hi inside the window
    if (lo >= hi) return false;
    uint32_t w = lo >> 5;
    const uint32_t wl = (hi - 1) >> 5;
    uint32_t x = bits[w] & (0xFFFFFFFFu << (lo & 31));
    while (w < wl) { if (x) return true; x = bits[++w]; }
    if (hi & 31) x &= (1u << (hi & 31)) - 1u;
    return x != 0;
}
// 32 bytes at pos (a multiple of 32) as masks, bit i = byte pos + i; bytes at or behind n read as "nothing"
__device__ __forceinline__ void masks32(const uint8_t* text, uint32_t pos, uint32_t n, uint32_t& nl, uint32_t& gt, uint32_t& bad) {
    nl = gt = bad = 0;
    if (pos >= n) return;
    const uint4 v0 = *reinterpret_cast<const uint4*>(text + pos), v1 = *reinterpret_cast<const uint4*>(text + pos + 16);  // (the buffer is padded by 64 bytes)
    const uint32_t w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    uint32_t good = 0;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const uint32_t x = w[d];
        nl |= nibble_of(eq_bytes(x, 0x0A0A0A0Au)) << (4 * d);
        gt |= nibble_of(eq_bytes(x, 0x3E3E3E3Eu)) << (4 * d);
        // one of ACGTN: bits 1..3 of a character number it 0..7 (A 0, C 1, T 2, G 3, N 7) -- the character it has to be then, out of an 8-byte table by v_perm_b32
        const uint32_t must_be = __builtin_amdgcn_perm(0x4EFFFFFFu, 0x47544341u, (x >> 1) & 0x07070707u);
        good |= nibble_of(bgr_zero_bytes(x ^ must_be)) << (4 * d);
    }
    const uint32_t valid = n - pos >= 32 ? 0xFFFFFFFFu : (1u << (n - pos)) - 1u;
    nl &= valid; gt &= valid;
    bad = ~good & valid;
}
// a record that leaves the window, by the 16 lanes of a group from the text itself: its first two newlines, and whether a character between them
// is not one of ACGTN
__device__ __forceinline__ void record_from_text(const uint8_t* text, uint32_t p, uint32_t n, uint32_t sub, uint32_t& first, uint32_t& second, bool& bad_char) {
    first = second = kNone;
    for (uint32_t base = p; base < n; base += 256) {
        uint32_t w[4];
        const uint32_t at = base + 16 * sub;
        load16(text, at, n, w);
        uint32_t m = mask16(w, 0x0A0A0A0Au);
        if (first == kNone) {
            const uint32_t mine = m ? at + (uint32_t)__ffs((int)m) - 1 : kNone;
            first = row16_min(mine);
            if (m && mine == first) m &= m - 1;  // the lane that holds it looks past it; every other newline lies behind it anyway
        }
        if (first != kNone) second = row16_min(m ? at + (uint32_t)__ffs((int)m) - 1 : kNone);
        if (second != kNone) break;
    }
    uint32_t bad = 0;
    if (second != kNone) {
        for (uint32_t base = first + 1; base < second; base += 256) {
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
    }
    bad_char = row16_sum(bad ? 1u : 0u) != 0;
}

template <int LN>   // lines per record of a FASTQ piece (4; 2 when the caller has left the '+' and quality lines on the host); 0 = FASTA
__global__ void __launch_bounds__(kParseThreads, 8) bgr_text_parse_kernel(const uint8_t* text, uint32_t n, uint32_t k, uint32_t ntiles, uint32_t* ticket, uint32_t ticket_base,
                                                                       uint32_t epoch, u64* chainA, u64* chainB, u64* chainC, uint4* rec, uint32_t* acc_idx, uint32_t* acc_rec,
                                                                       uint32_t* acc_src, u64* read_offs, uint32_t* info, uint32_t rec_cap, uint32_t* zero_a, uint32_t words_a, uint32_t* zero_b, uint32_t words_b) {
    __shared__ uint32_t s_nl[kParseWords], s_gt[kParseWords], s_bad[kParseWords];
    __shared__ uint32_t s_pos[kParseStretches][kParseMaxLocal], s_hl[kParseStretches][kParseMaxLocal], s_len[kParseStretches][kParseMaxLocal];
    __shared__ uint16_t s_left[kParseStretches * kParseMaxLocal];
    __shared__ u64 lw64[kParseStretches * 16];
    __shared__ uint32_t lw[16];
    __shared__ uint32_t s_vt, s_nleft, s_jfirst, s_have_first, s_maxlen, s_carry;
    __shared__ u64 s_exA, s_exB, s_exC;
    const uint32_t tid = threadIdx.x, wave = tid >> 6;
    if (tid == 0) {
        s_vt = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - ticket_base;
        s_nleft = 0; s_jfirst = 0; s_have_first = 0; s_maxlen = 0; s_exA = 0; s_exB = 0; s_exC = 0;
    }
    __syncthreads();
    const uint32_t vt = s_vt, T0 = vt * kParseTile;
    if (vt >= ntiles) return;   // (cannot happen while the host's ticket base matches the ticket word: no access outside the chains if it ever does not)
    BGR_XT(0, 0);
    if (vt == 0) {   // (words only LATER launches use: the mapping launch's cursor, the next piece's info block)
        if (tid < words_a) zero_a[tid] = 0;
        if (tid < words_b) zero_b[tid] = 0;
    }
    // 1. the masks of the three stretches (all loads in flight together) and of the 1 KB behind the tile
    uint32_t nl[kParseStretches], gt[kParseStretches], bad[kParseStretches], hnl = 0, hgt = 0, hbad = 0;
#pragma unroll
    for (uint32_t s = 0; s < kParseStretches; ++s) masks32(text, T0 + s * kParseStretch + tid * 32, n, nl[s], gt[s], bad[s]);
    if (tid < kParseHaloWords) masks32(text, T0 + kParseTile + tid * 32, n, hnl, hgt, hbad);
    uint32_t front_nl = 1;   // the byte in front of the tile is a newline (or the piece starts here)
    if (tid == 0 && T0) front_nl = text[T0 - 1] == '\n' ? 1u : 0u;
    BGR_XT(0, 1);
    uint32_t lines_before = 0, tile_lines = 0;   // FASTQ: newlines in front of the current stretch; of the tile
    if (LN) {
        uint32_t c = 0;
#pragma unroll
        for (uint32_t s = 0; s < kParseStretches; ++s) c += (uint32_t)__popc(nl[s]);
        (void)block_exclusive_scan32(c, lw, &tile_lines);
        if (wave == 0) { const u64 b4 = chain_lookback(chainA, vt, epoch, tile_lines); if (tid == 0) s_exA = b4; }
        __syncthreads();
        lines_before = (uint32_t)s_exA;
    }
    uint32_t total[kParseStretches];   // record starts per stretch
    BGR_XT(0, 2);
    bool irregular = false, over = false;
    if (!LN && vt == 0 && tid == 0 && n && !((gt[0] & 1u))) irregular = true;   // bytes in front of the piece's first record start: not this route's shape
#pragma unroll
    for (uint32_t s = 0; s < kParseStretches; ++s) {
        const uint32_t t0 = T0 + s * kParseStretch, pos = t0 + tid * 32;
        __syncthreads();   // (the masks of the stretch before this one are no longer read)
        s_nl[tid] = nl[s]; s_gt[tid] = gt[s]; s_bad[tid] = bad[s];
        if (tid < kParseHaloWords) {
            s_nl[kParseThreads + tid] = s + 1 < kParseStretches ? nl[s + 1 < kParseStretches ? s + 1 : s] : hnl;
            s_gt[kParseThreads + tid] = s + 1 < kParseStretches ? gt[s + 1 < kParseStretches ? s + 1 : s] : hgt;
            s_bad[kParseThreads + tid] = s + 1 < kParseStretches ? bad[s + 1 < kParseStretches ? s + 1 : s] : hbad;
        }
        const uint32_t carry = s ? s_carry : front_nl;   // (tid 0 reads it; written behind the barriers of the stretch before)
        __syncthreads();
        const uint32_t valid = pos < n ? (n - pos >= 32 ? 0xFFFFFFFFu : (1u << (n - pos)) - 1u) : 0u;
        const uint32_t after_nl = ((nl[s] << 1) | (tid ? s_nl[tid - 1] >> 31 : carry)) & valid;   // bit i: byte pos + i follows a newline
        uint32_t rs, j_mine = 0;   // rs: bit i = byte pos + i starts a record
        if (LN) {
            uint32_t stretch_lines;
            const uint32_t before = lines_before + block_exclusive_scan32((uint32_t)__popc(nl[s]), lw, &stretch_lines);   // newlines in front of this thread's bytes
            lines_before += stretch_lines;
            rs = 0;
            for (uint32_t m = after_nl; m; m &= m - 1) {
                const uint32_t i = (uint32_t)__ffs((int)m) - 1;
                const uint32_t c = before + (uint32_t)__popc(nl[s] & ((1u << i) - 1u));   // newlines in front of byte pos + i
                if ((c & (LN - 1)) == 0) { if (!rs) j_mine = c / (LN ? LN : 1); rs |= 1u << i; }
            }
        } else {
            rs = gt[s] & after_nl;
        }
        const uint32_t ex = block_exclusive_scan32((uint32_t)__popc(rs), lw, &total[s]);
        const bool over_s = total[s] > kParseMaxLocal;
        if (!over_s) {
            uint32_t at = ex;
            for (uint32_t m = rs; m; m &= m - 1) s_pos[s][at++] = pos + (uint32_t)__ffs((int)m) - 1;
        } else over = true;
        if (LN && rs && ex == 0 && !s_have_first) { s_jfirst = j_mine; s_have_first = 1; }   // (one thread per stretch can be here, stretches in turn: barriers in between)
        if (tid == kParseThreads - 1) s_carry = nl[s] >> 31;
        __syncthreads();
        BGR_XT(0, 3 + 2 * s);
        // 2. one thread per record of the stretch, from the masks
        if (!over_s && tid < total[s]) {
            const uint32_t l = tid, pr = s_pos[s][l] - t0;
            const uint32_t f = next_set(s_nl, pr);
            const uint32_t e = f != kNone ? next_set(s_nl, f + 1) : kNone;
            if (e == kNone || (!LN && e + 1 >= kParseWindow && t0 + e + 1 < n)) s_left[atomicAdd(&s_nleft, 1u)] = (uint16_t)((s << 10) | l);
            else {
                const bool regular = LN ? true : (t0 + e + 1 == n || ((s_gt[(e + 1) >> 5] >> ((e + 1) & 31)) & 1u));
                const uint32_t L = e - f - 1;
                const uint32_t ok = (regular && L > 2 && (LN || L > k) && !any_set(s_bad, f + 1, e)) ? 1u : 0u;   // aligner.cpp:78-88: size > 2, alphabet, size > k (FASTA only)
                s_hl[s][l] = regular ? f - pr : kNone;
                s_len[s][l] = regular ? (L | (ok << 31)) : 0u;
                if (!regular) irregular = true;
            }
        }
        BGR_XT(0, 4 + 2 * s);
    }
    __syncthreads();
    // 3. records that leave their window: a 16-lane group each, from the text
    BGR_XT(0, 9);
    {
        const uint32_t nleft = s_nleft, sub = tid & 15;
        for (uint32_t e = tid >> 4; e < nleft; e += kParseThreads / 16) {
            const uint32_t s = s_left[e] >> 10, l = s_left[e] & 1023u, p = s_pos[s][l];
            uint32_t first, second;
            bool bad_char;
            record_from_text(text, p, n, sub, first, second, bad_char);
            bool regular = second != kNone;
            if (!LN && regular && second + 1 != n) regular = text[second + 1] == '>';
            if (sub == 0) {
                const uint32_t L = regular ? second - first - 1 : 0u;
                const uint32_t ok = (regular && L > 2 && (LN || L > k) && !bad_char) ? 1u : 0u;
                s_hl[s][l] = regular ? first - p : kNone;
                s_len[s][l] = regular ? (L | (ok << 31)) : 0u;
                if (!regular) irregular = true;
            }
        }
    }
    if ((irregular || over) && !__hip_atomic_load(&info[TXT_INFO_IRREGULAR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&info[TXT_INFO_IRREGULAR], 1u);
    __syncthreads();
    // 4. accepted records and their bases in front of each record of the tile (accepted << 40 | bases in one scan); the running totals
    BGR_XT(0, 10);
    uint32_t v[kParseStretches];
    u64 pk[kParseStretches], exs[kParseStretches], t_all;
    uint32_t starts = 0, mx = 0;
#pragma unroll
    for (uint32_t s = 0; s < kParseStretches; ++s) {
        v[s] = (!over && tid < total[s]) ? s_len[s][tid] : 0u;
        const uint32_t L = (v[s] >> 31) ? v[s] & 0x7FFFFFFFu : 0u;
        pk[s] = ((u64)(v[s] >> 31) << 40) | L;
        starts += total[s];
        mx = L > mx ? L : mx;
    }
    block_exclusive_scan64xN<kParseStretches>(pk, lw64, exs, &t_all);
    if (mx) atomicMax(&s_maxlen, mx);
    BGR_XT(0, 11);
    const uint32_t t_acc = (uint32_t)(t_all >> 40);
    const u64 t_bases = t_all & ((1ull << 40) - 1);
    if (!LN && wave == 0) { const u64 b4 = chain_lookback(chainA, vt, epoch, starts); if (tid == 0) s_exA = b4; }
    if (wave == 1) { const u64 b4 = chain_lookback(chainB, vt, epoch, t_acc); if ((tid & 63) == 0) s_exB = b4; }
    if (wave == 2) { const u64 b4 = chain_lookback(chainC, vt, epoch, t_bases); if ((tid & 63) == 0) s_exC = b4; }
    __syncthreads();
    const uint32_t a0 = (uint32_t)s_exB;
    const u64 b0 = s_exC;
    BGR_XT(0, 12);
    // 5. the records; the accepted ones compacted
    if (!over) {
        uint32_t j0 = LN ? s_jfirst : (uint32_t)s_exA;
#pragma unroll
        for (uint32_t s = 0; s < kParseStretches; ++s) {
            const uint32_t j = j0 + tid;
            if (tid < total[s] && j < rec_cap) {
                const uint32_t p = s_pos[s][tid], hl = s_hl[s][tid];
                rec[j] = hl == kNone ? make_uint4(p, 0, 0, 0) : make_uint4(p, hl, p + hl + 1, v[s]);
                if (v[s] >> 31) {
                    const uint32_t a = a0 + (uint32_t)(exs[s] >> 40);
                    acc_idx[j] = a;
                    acc_rec[a] = j;
                    acc_src[a] = p + hl + 1;
                    read_offs[a] = b0 + (exs[s] & ((1ull << 40) - 1));
                }
            }
            j0 += total[s];
        }
    }
    BGR_XT(0, 13);
    if (tid == 0) {
        // (one atomic per record on one word would cap the kernel near 90 M records/s: one per tile, and only when it raises the word)
        if (s_maxlen > __hip_atomic_load(&info[TXT_INFO_MAX_LEN], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&info[TXT_INFO_MAX_LEN], s_maxlen);
        if (vt == ntiles - 1) {   // the piece's last tile: the totals
            const uint32_t n_acc = a0 + t_acc;
            const u64 bases = b0 + t_bases;
            uint32_t n_rec;
            if (LN) {
                const uint32_t lines = (uint32_t)s_exA + tile_lines;
                n_rec = lines / (LN ? LN : 1);   // a piece of whole records ends with a newline: LN per record
                if (n && ((lines & (LN - 1)) || text[n - 1] != '\n')) atomicOr(&info[TXT_INFO_IRREGULAR], 1u);   // (it does not: the host parser decides)
            } else n_rec = (uint32_t)s_exA + starts;
            info[TXT_INFO_N_REC] = n_rec;
            info[TXT_INFO_N_ACC] = n_acc;
            info[TXT_INFO_BASES] = (uint32_t)bases;
            if (n_acc <= rec_cap) read_offs[n_acc] = bases;
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
// the first min(m, 16) bytes of v to dst (any alignment)
__device__ __forceinline__ void store_upto16(uint8_t* dst, const u32x4_unaligned v, uint32_t m) {
    if (m >= 16) { *reinterpret_cast<u32x4_unaligned*>(dst) = v; return; }
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};   // (the last bytes: byte stores out of registers -- a loop of byte loads and stores waits for memory once per byte)
#pragma unroll
    for (uint32_t i = 0; i < 15; ++i) if (i < m) dst[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
}
// n bytes from src to dst by the 16 lanes of a group (any alignment on both sides; the SOURCE may be read up to 15 bytes past its end: the text is padded)
__device__ __forceinline__ void group_copy(uint8_t* dst, const uint8_t* src, uint32_t n, uint32_t sub) {
    for (uint32_t b = 16 * sub; b < n; b += 256) {
        store_upto16(dst + b, *reinterpret_cast<const u32x4_unaligned*>(src + b), n - b);
    }
}

// one accepted read's record by the 16 lanes of a group: into the paths stream (d_p) or the notAligned stream (d_n), whichever is not null
__device__ __forceinline__ void write_record(const uint8_t* text, const uint2 res, const uint4 r, const int32_t* arena, uint8_t* d_p, uint8_t* d_n, uint32_t sub) {
    const uint32_t np = res.y & 0xFFFFFFu, hl = r.y;
    if (np) {  // alignerGreedy.cpp:406-411: header + '\n' + printPath
        if (!d_p) return;
        uint8_t* d = d_p;
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
    } else if (d_n) {   // alignerGreedy.cpp:421-427: header + '\n' + read + '\n'
        uint8_t* d = d_n;
        const uint32_t L = r.w & 0x7FFFFFFFu;
        // (the first 256 bytes of both, loaded before either is stored: one wait for memory instead of two)
        const uint32_t b = 16 * sub;
        u32x4_unaligned vh = {0, 0, 0, 0}, vr = {0, 0, 0, 0};
        if (b < hl) vh = *reinterpret_cast<const u32x4_unaligned*>(text + r.x + b);
        if (b < L) vr = *reinterpret_cast<const u32x4_unaligned*>(text + r.z + b);
        if (b < hl) store_upto16(d + b, vh, hl - b);
        if (b < L) store_upto16(d + hl + 1 + b, vr, L - b);
        if (hl > 256) group_copy(d + 256, text + r.x + 256, hl - 256, sub);
        if (L > 256) group_copy(d + hl + 1 + 256, text + r.z + 256, L - 256, sub);
        if (sub == 0) { d[hl] = '\n'; d[hl + 1 + L] = '\n'; }
    }
}

// one 16-lane group per accepted read, offsets from arrays (bgr_aligner_fetch_text into larger buffers: the format kernel below has left them)
__global__ void __launch_bounds__(256) bgr_text_write_kernel(const uint8_t* text, const uint2* results, const int32_t* arena, const uint4* rec, const uint32_t* acc_rec,
                                                             uint32_t n_acc, const uint32_t* poff, const uint32_t* noff, uint8_t* pout, uint8_t* nout) {
    const uint32_t a = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, sub = threadIdx.x & 15;
    if (a >= n_acc) return;
    write_record(text, results[a], rec[acc_rec[a]], arena, pout + poff[a], nout + noff[a], sub);
}

// ---- out in ONE launch: sizes, stream offsets (running totals down two chains), the bytes -------------------------------------------------------
// (round 4/5: sizes, three launches of scans, a write kernel of 79 us per 274 k reads whose path digits went out as single-byte stores.)
// A workgroup takes 1024 accepted reads, one thread each for the sizes; its stretch of the paths stream (~30 bytes per mapped read) is put together
// in LDS -- header bytes by 16-byte loads, digits as LDS byte stores -- and leaves as aligned 16-byte stores; a stretch that does not fit (long
// headers, long paths) and the notAligned records (header + read: 16-byte copies as they are) go out by 16-lane groups.  Bytes are written only
// where the whole stretch fits below pcap / ncap; the totals say what is needed.  poff/noff are kept for bgr_aligner_fetch_text.
constexpr uint32_t kFmtThreads = 1024, kFmtChunk = 49152, kFmtInts = 6;
__global__ void __launch_bounds__(kFmtThreads, 8) bgr_text_format_kernel(const uint8_t* text, const uint2* results, const int32_t* arena, const uint4* rec, const uint32_t* acc_rec,
                                                                      uint32_t n_acc, uint32_t ntiles, uint32_t* ticket, uint32_t ticket_base, uint32_t epoch, u64* chainP, u64* chainN,
                                                                      uint32_t* poff, uint32_t* noff, uint8_t* pout, uint8_t* nout, u64 pcap, u64 ncap, uint32_t* info) {
    __shared__ __attribute__((aligned(16))) uint8_t s_chunk[kFmtChunk + 16];
    __shared__ uint32_t lw[16];
    __shared__ uint32_t s_rel[kFmtThreads];   // where each read's record starts in the workgroup's stretch of ITS stream
    __shared__ uint16_t s_un[kFmtThreads];    // the reads without a path
    __shared__ uint4 s_unrec[kFmtThreads];    // ... and their records
    __shared__ uint32_t s_vt, s_nun;
    __shared__ u64 s_exP, s_exN;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, sub = tid & 15, grp = tid >> 4;
    if (tid == 0) { s_vt = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - ticket_base; s_exP = 0; s_exN = 0; s_nun = 0; }
    __syncthreads();
    const uint32_t vt = s_vt, a0 = vt * kFmtThreads, a = a0 + tid;
    if (vt >= ntiles) return;   // (see bgr_text_parse_kernel)
    BGR_XT(1, 0);
    uint2 res = make_uint2(0, 0);
    uint4 r = make_uint4(0, 0, 0, 0);
    uint32_t ps = 0, ns = 0, np = 0;
    int32_t pv[kFmtInts];   // the first ints of the path, loaded together (a path is its offset + a handful of unitigs)
    if (a < n_acc) {
        res = results[a];
        const uint32_t j = acc_rec[a];
        np = res.y & 0xFFFFFFu;
#pragma unroll
        for (uint32_t i = 0; i < kFmtInts; ++i) pv[i] = i < np ? arena[res.x + i] : 0;
        r = rec[j];
        if (np) {
            ps = r.y + 2;
#pragma unroll
            for (uint32_t i = 0; i < kFmtInts; ++i) if (i < np) ps += dec_len(pv[i]);
            for (uint32_t i = kFmtInts; i < np; ++i) ps += dec_len(arena[res.x + i]);
        } else {
            ns = r.y + (r.w & 0x7FFFFFFFu) + 2;
            const uint32_t e = atomicAdd(&s_nun, 1u);
            s_un[e] = (uint16_t)tid;
            s_unrec[e] = r;
        }
    }
    uint32_t tp, tn;
    BGR_XT(1, 1);
    const uint32_t exp_ = block_exclusive_scan32(ps, lw, &tp);
    const uint32_t exn = block_exclusive_scan32(ns, lw, &tn);
    s_rel[tid] = ps ? exp_ : exn;
    BGR_XT(1, 2);
    if (wave == 0) chain_publish(chainP, vt, epoch, tp);
    if (wave == 1) chain_publish(chainN, vt, epoch, tn);
    // (the totals are out; before anyone walks the chains the paths stretch is put together -- it does not need to know where it goes: it is laid down at 0 and
    // shifted to the destination's 16-byte units on the way out)
    const bool in_lds = tp && tp <= kFmtChunk;
    BGR_XT(1, 3);
    if (in_lds && ps) {
        uint32_t o = exp_;
        const uint32_t hl = r.y;
        for (uint32_t b = 0; b < hl; b += 16) {   // the header, 16 bytes per load
            const u32x4_unaligned v = *reinterpret_cast<const u32x4_unaligned*>(text + r.x + b);  // (the text is padded: reading past the header is reading the read)
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            const uint32_t m = hl - b < 16 ? hl - b : 16u;
#pragma unroll
            for (uint32_t i = 0; i < 16; ++i) if (i < m) s_chunk[o + b + i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
        }
        o += hl;
        s_chunk[o++] = '\n';
        for (uint32_t i = 0; i < np; ++i) {   // to_string(v) + '.', written back to front (aligner.cpp:600-609)
            int32_t v = 0;
            if (i < kFmtInts) {
#pragma unroll
                for (uint32_t q = 0; q < kFmtInts; ++q) if (q == i) v = pv[q];
            } else v = arena[res.x + i];
            const uint32_t len = dec_len(v);
            uint32_t e = o + len;
            s_chunk[--e] = '.';
            uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
            do { const uint32_t qd = u / 10; s_chunk[--e] = (uint8_t)('0' + (u - qd * 10)); u = qd; } while (u);
            if (v < 0) s_chunk[--e] = '-';
            o += len;
        }
        s_chunk[o] = '\n';
    }
    if (wave == 0) { const u64 b4 = chain_walk(chainP, vt, epoch, tp); if (tid == 0) s_exP = b4; }
    if (wave == 1) { const u64 b4 = chain_walk(chainN, vt, epoch, tn); if ((tid & 63) == 0) s_exN = b4; }
    __syncthreads();
    const u64 P0 = s_exP, N0 = s_exN;
    BGR_XT(1, 4);
    if (a < n_acc) {
        const u64 po = P0 + exp_, no = N0 + exn;
        poff[a] = po > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)po;
        noff[a] = no > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)no;
    }
    const bool p_fits = P0 + tp <= pcap, n_fits = N0 + tn <= ncap;
    if (tp && p_fits) {
        if (in_lds) {
            // byte i of the stretch lies at s_chunk[i] and goes to pout[P0 + i]: 16-byte units of the DESTINATION, each put together from two LDS reads
            const uint32_t shift = (uint32_t)(P0 & 15);
            uint8_t* base = pout + (P0 - shift);   // 16-byte aligned (pout is); unit u covers stretch bytes [16 u - shift, 16 u - shift + 16)
            const uint32_t end = shift + tp;
            for (uint32_t u = tid * 16; u < end; u += kFmtThreads * 16) {
                if (u >= shift && u + 16 <= end) {
                    const uint32_t from = u - shift;
                    uint32_t w[4];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const uint32_t at = from + 4 * d, al = at & ~3u, sh = 8 * (at & 3);
                        const uint32_t lo = *reinterpret_cast<const uint32_t*>(s_chunk + al), hi = *reinterpret_cast<const uint32_t*>(s_chunk + al + 4);
                        w[d] = sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
                    }
                    *reinterpret_cast<uint4*>(base + u) = make_uint4(w[0], w[1], w[2], w[3]);
                } else for (uint32_t i = u < shift ? shift : u; i < u + 16 && i < end; ++i) base[i] = s_chunk[i - shift];
            }
        } else {
            for (uint32_t t = grp; t < kFmtThreads; t += kFmtThreads / 16) {
                const uint32_t b = a0 + t;
                if (b >= n_acc) break;
                const uint2 rs = results[b];
                if (rs.y & 0xFFFFFFu) write_record(text, rs, rec[acc_rec[b]], arena, pout + P0 + s_rel[t], nullptr, sub);
            }
        }
    }
    BGR_XT(1, 5);
    if (tn && n_fits) {   // header + '\n' + read + '\n' of the reads without a path: 16-byte copies by 16-lane groups, straight from the text
        const uint32_t nun = s_nun, b = 16 * sub;
        for (uint32_t e = grp; e < nun; e += 2 * (kFmtThreads / 16)) {   // two records per turn, their (first 256) bytes loaded before any is stored: one wait for memory
            const uint32_t e2 = e + kFmtThreads / 16;
            const bool two = e2 < nun;
            const uint4 r1 = s_unrec[e], r2 = two ? s_unrec[e2] : make_uint4(0, 0, 0, 0);
            const uint32_t L1 = r1.w & 0x7FFFFFFFu, L2 = r2.w & 0x7FFFFFFFu;
            uint8_t* d1 = nout + N0 + s_rel[s_un[e]];
            uint8_t* d2 = nout + N0 + s_rel[s_un[two ? e2 : e]];
            u32x4_unaligned h1 = {0, 0, 0, 0}, q1 = {0, 0, 0, 0}, h2 = {0, 0, 0, 0}, q2 = {0, 0, 0, 0};
            if (b < r1.y) h1 = *reinterpret_cast<const u32x4_unaligned*>(text + r1.x + b);
            if (b < L1) q1 = *reinterpret_cast<const u32x4_unaligned*>(text + r1.z + b);
            if (b < r2.y) h2 = *reinterpret_cast<const u32x4_unaligned*>(text + r2.x + b);
            if (b < L2) q2 = *reinterpret_cast<const u32x4_unaligned*>(text + r2.z + b);
            if (b < r1.y) store_upto16(d1 + b, h1, r1.y - b);
            if (b < L1) store_upto16(d1 + r1.y + 1 + b, q1, L1 - b);
            if (b < r2.y) store_upto16(d2 + b, h2, r2.y - b);
            if (b < L2) store_upto16(d2 + r2.y + 1 + b, q2, L2 - b);
            if (r1.y > 256) group_copy(d1 + 256, text + r1.x + 256, r1.y - 256, sub);
            if (L1 > 256) group_copy(d1 + r1.y + 1 + 256, text + r1.z + 256, L1 - 256, sub);
            if (r2.y > 256) group_copy(d2 + 256, text + r2.x + 256, r2.y - 256, sub);
            if (L2 > 256) group_copy(d2 + r2.y + 1 + 256, text + r2.z + 256, L2 - 256, sub);
            if (sub == 0) { d1[r1.y] = '\n'; d1[r1.y + 1 + L1] = '\n'; if (two) { d2[r2.y] = '\n'; d2[r2.y + 1 + L2] = '\n'; } }
        }
    }
    BGR_XT(1, 6);
    if (tid == 0 && vt == ntiles - 1) {
        const u64 P = P0 + tp, N = N0 + tn;
        info[TXT_INFO_PBYTES] = P > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)P;
        info[TXT_INFO_NBYTES] = N > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)N;
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

hipError_t launch_scan2_u32(const uint32_t* inA, const uint32_t* inB, uint32_t* outA, uint32_t* outB, uint32_t n, const uint32_t* n_dev, uint32_t* sums, uint32_t* totalA,
                            uint32_t* totalB, hipStream_t stream) {
    const uint32_t nb = std::max<uint32_t>(1, (n + kScanTile - 1) / kScanTile);
    hipLaunchKernelGGL(bgr_scan2_block_sums, dim3(nb), dim3(kTxtThreads), 0, stream, inA, inB, n, n_dev, sums, nb);
    hipLaunchKernelGGL(bgr_scan2_sums, dim3(1), dim3(kTxtThreads), 0, stream, sums, nb, totalA, totalB);
    hipLaunchKernelGGL(bgr_scan2_apply, dim3(nb), dim3(kTxtThreads), 0, stream, inA, inB, n, n_dev, sums, nb, outA, outB);
    return hipGetLastError();
}
uint32_t scan_tiles(uint32_t n) { return std::max<uint32_t>(1, (n + kScanTile - 1) / kScanTile); }
uint32_t text_tiles(uint32_t bytes) { return std::max<uint32_t>(1, (bytes + kParseTile - 1) / kParseTile); }
uint32_t format_tiles(uint32_t n_acc) { return std::max<uint32_t>(1, (n_acc + kFmtThreads - 1) / kFmtThreads); }

hipError_t launch_text_parse(const uint8_t* text, uint32_t n, uint32_t fastq_lines, uint32_t k, uint32_t* ticket, uint32_t ticket_base, uint32_t epoch, uint64_t* chains,
                             uint4* rec, uint32_t* acc_idx, uint32_t* acc_rec, uint32_t* acc_src, uint64_t* read_offs, uint32_t* info, uint32_t rec_cap, uint32_t* zero_a,
                             uint32_t words_a, uint32_t* zero_b, uint32_t words_b, hipStream_t stream) {
    const uint32_t nt = text_tiles(n);
    u64* cA = reinterpret_cast<u64*>(chains);
    u64* cB = cA + nt;
    u64* cC = cB + nt;
    u64* offs = reinterpret_cast<u64*>(read_offs);
#define BGR_PARSE_ARGS text, n, k, nt, ticket, ticket_base, epoch, cA, cB, cC, rec, acc_idx, acc_rec, acc_src, offs, info, rec_cap, zero_a, words_a, zero_b, words_b
    if (fastq_lines == 4) hipLaunchKernelGGL(bgr_text_parse_kernel<4>, dim3(nt), dim3(kParseThreads), 0, stream, BGR_PARSE_ARGS);
    else if (fastq_lines == 2) hipLaunchKernelGGL(bgr_text_parse_kernel<2>, dim3(nt), dim3(kParseThreads), 0, stream, BGR_PARSE_ARGS);
    else hipLaunchKernelGGL(bgr_text_parse_kernel<0>, dim3(nt), dim3(kParseThreads), 0, stream, BGR_PARSE_ARGS);
#undef BGR_PARSE_ARGS
    return hipGetLastError();
}

hipError_t launch_text_format(const uint8_t* text, const uint2* results, const int32_t* arena, const uint4* rec, const uint32_t* acc_rec, uint32_t n_acc, uint32_t* ticket,
                              uint32_t ticket_base, uint32_t epoch, uint64_t* chains, uint32_t* poff, uint32_t* noff, uint8_t* pout, uint8_t* nout, uint64_t pcap, uint64_t ncap,
                              uint32_t* info, hipStream_t stream) {
    if (n_acc == 0) return hipSuccess;
    const uint32_t nt = format_tiles(n_acc);
    u64* cP = reinterpret_cast<u64*>(chains);
    hipLaunchKernelGGL(bgr_text_format_kernel, dim3(nt), dim3(kFmtThreads), 0, stream, text, results, arena, rec, acc_rec, n_acc, nt, ticket, ticket_base, epoch, cP, cP + nt, poff, noff,
                       pout, nout, (u64)pcap, (u64)ncap, info);
    return hipGetLastError();
}

hipError_t launch_text_record_info(const uint4* rec, const uint32_t* acc_idx, const uint2* results, uint32_t n_rec, uint32_t* out, hipStream_t stream) {
    if (n_rec == 0) return hipSuccess;
    hipLaunchKernelGGL(bgr_text_record_info_kernel, dim3((n_rec + 255) / 256), dim3(256), 0, stream, rec, acc_idx, results, n_rec, out);
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
