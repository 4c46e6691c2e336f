// device_common.h -- device helpers shared by the mapping kernels of the BGREAT path (gfx950; included by the .hip files of
// this directory only): wave primitives, 2-bit window reads, the MPHF cascade lookup, the read planes, candidate scoring
// (missmatchNumber over 2-bit words), the greedy step / walks, workgroup prologue (cascade staging), path arena.
// Integer/byte work only: no MFMA anywhere (there is no dense contraction on this path).
#ifndef BGREAT_AMD_DEVICE_COMMON_H
#define BGREAT_AMD_DEVICE_COMMON_H

#include <algorithm>

#include "../../include/bgreat_gpu.h"
#include "align_kernels.h"

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
// reverse complement of the n bases in the low 2n bits of x.  Per 32-bit half: v_bfrev, then the bits of every pair swapped AND complemented
// by one v_bitop3 over (y >> 1, y << 1, 0x55555555): "~(mask ? a : b)" = truth table 0x1B -- 9 instructions instead of the 13 of ~rev2_fast(x)
// (two 64-bit shifts, four ands, two ors, two nots); this runs once per read position of every scan
__device__ __forceinline__ uint32_t swap_pairs_not(uint32_t y) { return (uint32_t)__builtin_amdgcn_bitop3_b32((int)(y >> 1), (int)(y << 1), 0x55555555, 0x1B); }
__device__ __forceinline__ u64 rcb_fast(u64 x, uint32_t n) {
    const uint32_t lo = swap_pairs_not(__builtin_bitreverse32((uint32_t)(x >> 32))), hi = swap_pairs_not(__builtin_bitreverse32((uint32_t)x));
    return (((u64)hi << 32) | lo) >> (64 - 2 * n);
}

__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }

#ifndef BGR_DIRECT_BUCKETS
#define BGR_DIRECT_BUCKETS 1  /* behind the minimizer filter: compare a bucket's four keys directly (0 = fingerprints first, as everywhere else) */
#endif
// Key table lookup for one key per lane (graph_layout.h): the slot of `key` -- its index into keys[] and recs[] -- or
// BGR_NONE when the key is no overlap of the graph (the membership test of aligner.cpp:158,219,353,361).  A byte that
// equals the fingerprint is confirmed against keys[].  Most read positions are no overlaps and match no byte, so the loop
// below usually runs once per call for the wave's few candidates (members and ~3 % false matches).
// Table staged in LDS (LAZY2 = false): both buckets are read at once, two independent ds_read_b32.
// Table in L2/HBM (LAZY2 = true): such a launch is bound by the L2's request rate (one request per lane and probe), so
// bucket 2 is only read by the lanes whose bucket 1 is full -- the builder fills bucket 1 first and never empties a slot, so
// a key can sit in bucket 2 only then.  At the sparse fill such graphs are built with (0.55) that is one lane in five:
// 1.2 instead of 2 requests per position (chr1-scale graph: 749 -> 930 Mreads/s).
// (mblock: the key's block of the minimizer filter, from the caller -- the scans work it out across lanes, wave_window_max below;
// callers that cannot pass a graph without filter)
template <bool LAZY2 = false, typename TP>
__device__ __forceinline__ uint32_t find_key(const BgrDeviceGraph& g, TP tab, u64 key, bool active, uint32_t mblock = 0) {
    const u64 m = bgr_mix64(key);
    const uint32_t b1 = __umulhi((uint32_t)m, g.n_buckets), b2 = __umulhi((uint32_t)(m >> 32), g.n_buckets);
    uint32_t res = BGR_NONE;
    if (LAZY2 && BGR_DIRECT_BUCKETS && g.bloom && g.filter_kind == BGR_FILTER_MINIMIZER) {
        // large graph behind the minimizer filter: the few lanes it lets through (members and ~1 % of the rest) compare the four keys of
        // their bucket directly -- keys[4 b .. 4 b + 3] share one 64-byte line -- instead of fingerprints first: one dependent miss
        // less per member (filter block -> bucket keys instead of filter block -> fingerprints -> key entry)
        if (active) {
            const uint32_t bits = bgr_mmx_bits(m);
            active = (g.bloom[(mblock << 4) + bgr_mmx_word(m)] & bits) == bits;
        }
        if (wave_any(active)) {
            if (active) {
                const BgrKeyEntry* e = g.keys + (size_t)b1 * 4;
                u64 k0 = e[0].key, k1 = e[1].key, k2 = e[2].key, k3 = e[3].key;
                uint32_t hit = k0 == key ? 0u : k1 == key ? 1u : k2 == key ? 2u : k3 == key ? 3u : 4u;
                if (hit < 4) res = b1 * 4 + hit;
                else if (((k0 | k1 | k2 | k3) >> 62) == 0) {  // no empty slot (an empty one holds ~0; a key has its top two bits clear): bucket 2
                    e = g.keys + (size_t)b2 * 4;
                    k0 = e[0].key; k1 = e[1].key; k2 = e[2].key; k3 = e[3].key;
                    hit = k0 == key ? 0u : k1 == key ? 1u : k2 == key ? 2u : k3 == key ? 3u : 4u;
                    if (hit < 4) res = b2 * 4 + hit;
                }
            }
        }
    } else {
        uint32_t w1 = 0, w2 = 0;
        if (LAZY2) {
            // large graph: a filter turns most positions (no overlaps) away before the table -- every probe of which is a cache line of
            // its own -- is touched; a member always passes, and the key compare below decides as before
            if (g.bloom && active) {
                if (g.filter_kind == BGR_FILTER_MINIMIZER) {
                    const uint32_t bits = bgr_mmx_bits(m);
                    active = (g.bloom[(mblock << 4) + bgr_mmx_word(m)] & bits) == bits;
                } else {
                    const uint32_t bit = bgr_bloom_bit(m, g.bloom_mask);
                    active = (g.bloom[bit >> 5] >> (bit & 31)) & 1u;
                }
            }
            if (active) w1 = tab[b1];
            if (active && bgr_zero_bytes(w1) == 0) w2 = tab[b2];
        } else if (active) { w1 = tab[b1]; w2 = tab[b2]; }
        const uint32_t f4 = bgr_tab_fp(m) * 0x01010101u;  // (an empty slot is 0 and the fingerprint is not: lanes that sit out match nothing)
        // (b2 == b1, one key in ~n_buckets: the second word then repeats the first one's matches, and the loop below re-checks a slot it has
        // already ruled out -- harmless; testing for it cost six instructions per scan step)
        uint32_t c1 = bgr_zero_bytes(w1 ^ f4), c2 = bgr_zero_bytes(w2 ^ f4);
        while (wave_any((c1 | c2) != 0)) {
            if (c1 | c2) {
                const bool first = c1 != 0;
                const uint32_t c = first ? c1 : c2;
                const uint32_t idx = (first ? b1 : b2) * 4 + ((uint32_t)(__ffs((int)c) - 1) >> 3);
                if (g.keys[idx].key == key) { res = idx; c1 = 0; c2 = 0; }
                else if (first) c1 &= c1 - 1;
                else c2 &= c2 - 1;
            }
        }
    }
    if ((g.flags & BGR_GF_HAS_FALLBACK) && wave_any(active && res == BGR_NONE)) {
        if (active && res == BGR_NONE) {  // bisection in the (tiny) sorted fallback list; its location comes from the blob header
            const uint32_t nfb = (uint32_t)g.hdr->n_fallback;
            const u64* fb = reinterpret_cast<const u64*>(reinterpret_cast<const char*>(g.hdr) + g.hdr->off_fallback);
            uint32_t lo = 0, hi = nfb;
            while (lo < hi) {
                uint32_t mid = (lo + hi) >> 1;
                if (fb[mid] < key) lo = mid + 1; else hi = mid;
            }
            if (lo < nfb && fb[lo] == key) res = 4u * g.n_buckets + lo;
        }
    }
    return res;
}

// Minimizer filter, scan side.  h = bgr_mmx_hash of the 16-mer that starts at the lane's read position (0 where the read has none).
// Returns, for the (k-1)-mer that starts there, the largest h over its W = k-16 16-mers -- bgr_mmx_of_key of it -- in lanes
// 0 .. 64-W (the lanes behind lack their right neighbours: a scan advances by 65-W positions per step).  Whole-wave DPP shifts
// (wave_shl:1: lane i takes lane i+1, lane 63 takes 0 -- neutral for a maximum): a shift by d is d moves, the last one folded into
// the v_max; k = 31: 14 VALU instructions, no LDS (profiles/r03_wave_shift_dpp.txt: 4.2 cycles each, a ds_bpermute is 24).
__device__ __forceinline__ uint32_t wave_shl1(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130, 0xF, 0xF, true); }
template <int D>
__device__ __forceinline__ uint32_t wave_shl(uint32_t x) {
#pragma unroll
    for (int s = 0; s < D; ++s) x = wave_shl1(x);
    return x;
}
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t wave_window_max(uint32_t h, uint32_t W) {
    if (W == 15 || W == 16) {  // k = 31 / 32, straight-line (the loops below cost three scalar instructions per shift)
        uint32_t r = umax(h, wave_shl<1>(h));
        r = umax(r, wave_shl<2>(r));
        r = umax(r, wave_shl<4>(r));
        return W == 15 ? umax(r, wave_shl<7>(r)) : umax(r, wave_shl<8>(r));
    }
    uint32_t r = h, w = 1;
    while (2 * w <= W) {  // r: maximum over w positions -> over 2w
        uint32_t t = r;
        for (uint32_t s = 0; s < w; ++s) t = wave_shl1(t);
        r = t > r ? t : r;
        w *= 2;
    }
    if (w < W) {          // two windows of w that overlap
        uint32_t t = r;
        for (uint32_t s = 0; s < W - w; ++s) t = wave_shl1(t);
        r = t > r ? t : r;
    }
    return r;
}

// the filter block of the (k-1)-mer at the lane's position: win = the 32 bases from there (left aligned), has16 = a 16-mer starts there
__device__ __forceinline__ uint32_t scan_mblock(const BgrDeviceGraph& g, u64 win, bool has16, uint32_t W) {
    const uint32_t h = has16 ? bgr_mmx_hash((uint32_t)(win >> 32)) : 0u;
    return bgr_mmx_block(wave_window_max(h, W), g.bloom_mask);
}
// ... in ALL 64 lanes for W = 15 / 16 (k = 31 / 32), so that a scan step covers 64 positions instead of 65 - W = 50 / 49 (a 150-base read: two steps instead of
// three).  Lanes 65 - W .. 63 lack the 16-mers that start at positions 64 .. 62 + W of the step -- which lie in the LOW halves of the windows of lanes 48 .. 46 + W:
// their hashes, a running maximum from lane 48 upwards (row_shr 1, 2, 4, 8 inside row 3) and lane i takes the one of lane i - (17 - W).
// has16_hi = a 16-mer starts 16 bases behind the lane's position.
__device__ __forceinline__ bool scan_mblock_is_wide(uint32_t W) { return W == 15 || W == 16; }
__device__ __forceinline__ uint32_t scan_mblock_wide(const BgrDeviceGraph& g, u64 win, bool has16, bool has16_hi, uint32_t W) {
    const uint32_t h = has16 ? bgr_mmx_hash((uint32_t)(win >> 32)) : 0u;
    uint32_t r = wave_window_max(h, W);
    uint32_t q = has16_hi ? bgr_mmx_hash((uint32_t)win) : 0u;
    q = umax(q, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)q, 0x111, 0xF, 0xF, false));   // row_shr:1 (a lane without a source inside its row takes the 0)
    q = umax(q, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)q, 0x112, 0xF, 0xF, false));
    q = umax(q, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)q, 0x114, 0xF, 0xF, false));
    q = umax(q, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)q, 0x118, 0xF, 0xF, false));
    const uint32_t up = W == 15 ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)q, 0x112, 0x8, 0xF, false)    // row 3 only: the other rows take the 0
                                : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)q, 0x111, 0x8, 0xF, false);
    r = umax(r, up);
    return bgr_mmx_block(r, g.bloom_mask);
}

// The half of key entry `idx` a walk step reads (graph_layout.h: handles): getEnd(bin) -- a step to the LEFT -- reads the right table
// when bin is canonical, else the left one; getBegin(bin) -- a step to the right -- the other way round (aligner.cpp:147-267).
__device__ __forceinline__ uint32_t half_handle(const BgrDeviceGraph& g, uint32_t idx, bool canon, bool left) {
    if (idx == BGR_NONE) return BGR_HNONE;
    const uint2 h = *reinterpret_cast<const uint2*>(&g.keys[idx].hL);
    return canon == left ? h.y : h.x;
}

// ---- ASCII -> 2-bit codes (the pre-pass bgr_pack_reads_kernel, and -- greedy mode -- the mapping kernels themselves, which stage their reads
// straight from the characters: no planes written and read back) --------------------------------------------------------------------------
// 4 ASCII bases (one dword, first base in the low byte; a zero byte = past the end) -> their 2-bit codes, one per byte.
// On the alphabet the parser admits (ACGTN, aligner.cpp:56-61; either case): high bit = bit 2 of the character (G T N),
// low bit = bit 4 (T) | bit 3 (N) | bit 1 & ~bit 2 (C): A0 C1 G2 T3 N3, and 0 for a zero byte.
// (round 5: as a table lookup -- bits 1..3 of a character number it 0..7 (A 0, C 1, T 2, G 3, N 7, either case) and v_perm_b32 picks the four codes out
// of an 8-byte table in one instruction: 3 instructions per dword instead of 7 of bit logic; a zero byte picks entry 0 = code 0, as before)
__device__ __forceinline__ uint32_t codes4(uint32_t x) {
    const uint32_t sel = (x >> 1) & 0x07070707u;
    return __builtin_amdgcn_perm(0x03030303u, 0x02030100u, sel);  // bytes 0..3 of the second operand = entries 0..3 (A C T G), 4..7 of the first (3: N and the rest)
}
// the four 2-bit fields of c (bytes 0..3, values 0..3) as one byte in bits 24..31, first base in the top two bits: the
// partial products c << 30, c << 20, c << 10, c put b0 b1 b2 b3 at bits 30 28 26 24 and nothing else at or above bit 24
__device__ __forceinline__ uint32_t gather4(uint32_t c) { return c * 0x40100401u; }
// top bytes of four such products -> one dword, p0's first
__device__ __forceinline__ uint32_t top_bytes(uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3) {
    const uint32_t a = __builtin_amdgcn_perm(p0, p1, 0x07030000u);  // [p0.b3, p1.b3, -, -]
    const uint32_t b = __builtin_amdgcn_perm(p2, p3, 0x07030000u);
    return __builtin_amdgcn_perm(a, b, 0x07060302u);                // [a.b3, a.b2, b.b3, b.b2]
}

// 32 bases [32j, 32j+32) of a read as 8 dwords of ASCII (first base in the low byte of xs[0]), zero beyond the read's end
__device__ __forceinline__ void load32(const uint8_t* rd, uint32_t L, uint32_t j, bool whole_in_buffer, uint32_t xs[8]) {
    const uint32_t valid = L - 32 * j;  // >= 1
    if (whole_in_buffer) {  // all 32 bytes lie inside the batch buffer: two (unaligned) 16-byte loads, bytes past the read masked off
        typedef uint32_t __attribute__((ext_vector_type(4), aligned(1))) u32x4_unaligned;
        const u32x4_unaligned v0 = *reinterpret_cast<const u32x4_unaligned*>(rd + 32 * j);
        const u32x4_unaligned v1 = *reinterpret_cast<const u32x4_unaligned*>(rd + 32 * j + 16);
        xs[0] = v0.x; xs[1] = v0.y; xs[2] = v0.z; xs[3] = v0.w; xs[4] = v1.x; xs[5] = v1.y; xs[6] = v1.z; xs[7] = v1.w;
        if (valid < 32) {
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const uint32_t lo = 4u * d;
                if (valid <= lo) xs[d] = 0;
                else if (valid < lo + 4) xs[d] &= 0xFFFFFFFFu >> (8 * (lo + 4 - valid));
            }
        }
    } else {  // the batch's last bytes: never touch a byte past the buffer
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            uint32_t x = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4u * d + e < valid) x |= (uint32_t)rd[32 * j + 4 * d + e] << (8 * e);
            xs[d] = x;
        }
    }
}
__device__ __forceinline__ u64 pack_codes(const uint32_t c[8]) {
    const uint32_t hi = top_bytes(gather4(c[0]), gather4(c[1]), gather4(c[2]), gather4(c[3]));
    const uint32_t lo = top_bytes(gather4(c[4]), gather4(c[5]), gather4(c[6]), gather4(c[7]));
    return (u64)hi << 32 | lo;
}

// 32 bases from byte `start` of the batch's characters (`valid` <= 32 of them belong to the read) as one word, first base most significant, zero
// beyond `valid`; *chars |= every character seen (bit 3 of a character is set for N only, of the alphabet getReads admits)
// loose: the characters behind the read's end are not masked off one by one -- the word is cut to `valid` bases after packing, and *chars may see
// up to 31 characters that follow the read.  Good enough where those are the next read's bases (reads end to end: an N there sends THIS read to the
// general kernel too, which maps any read); not in a FASTA text, where a newline and a header follow (bit 3 of '\n' is set).
template <bool LOOSE = false>
__device__ __forceinline__ u64 ascii_word(const uint8_t* reads, u64 start, uint32_t valid, u64 total_bytes, uint32_t* chars) {
    uint32_t xs[8], c[8];
    const bool whole = start + 32 <= total_bytes;
    load32(reads + start, (LOOSE && whole) ? 32u : valid, 0, whole, xs);
#pragma unroll
    for (int d = 0; d < 8; ++d) { c[d] = codes4(xs[d]); *chars |= xs[d]; }
    u64 f = pack_codes(c);
    if (LOOSE && valid < 32) f &= ~0ULL << (64 - 2 * valid);
    return f;
}
// ... and their N mask (3 on every N)
__device__ __forceinline__ u64 ascii_nmask_word(const uint8_t* reads, u64 start, uint32_t valid, u64 total_bytes) {
    uint32_t xs[8], c[8];
    load32(reads + start, valid, 0, start + 32 <= total_bytes, xs);
#pragma unroll
    for (int d = 0; d < 8; ++d) c[d] = ((xs[d] >> 3) & 0x01010101u) * 3u;
    return pack_codes(c);
}
// where read r's characters start: the reads of a batch lie end to end (read_offs), or scattered in a FASTA text (ascii_src, text_kernels.hip)
__device__ __forceinline__ u64 ascii_start(const BatchIO& io, uint32_t r, u64 off) { return io.ascii_src ? (u64)io.ascii_src[r] : off; }

// ---- stage A: the read's 2-bit words, from the planes the pre-pass (bgr_pack_reads_kernel) or the host packer wrote ----
// FW3: str2num codes (N->3).  NM: 3 on every N.  RCW: reverseComplements(read) (utils.cpp:66-73, non-ACG -> 'A').
// FWQ: what the rolling `num` of getNOverlap/getListOverlap holds: str2num codes inside the first window,
//      nuc2int codes (N->0) for bases entered by update() (aligner.cpp:305-309, utils.cpp:132-140).
// Read r of a batch owns words [woff, woff + ceil(L/32)) of both planes, woff = (read_offs[r] >> 5) + r: computable from the
// ASCII offsets alone (no scan), never overlapping, at most one spare word per read.  The N plane of a read is valid
// only if its bit in `hasn` is set.
__device__ __forceinline__ uint32_t packed_word_offset(u64 off, uint32_t r) { return (uint32_t)(off >> 5) + r; }

__device__ __forceinline__ bool load_packed(const BatchIO& io, uint32_t r, u64 off, uint32_t L, uint32_t W, u64* FW3, u64* NM, int lane) {
    if (io.ascii) {  // no planes: the words straight from the read's characters (greedy mode: the launch has no pre-pass)
        const uint32_t Wr = (L + 31) >> 5;
        const u64 s0 = ascii_start(io, r, off);
        uint32_t chars = 0;
        for (uint32_t j = lane; j < W; j += 64) FW3[j] = j < Wr ? ascii_word(io.ascii, s0 + 32ull * j, L - 32 * j, io.ascii_bytes, &chars) : 0;
        const bool hasN = wave_any((chars & 0x08080808u) != 0);
        for (uint32_t j = lane; j < W; j += 64) NM[j] = (hasN && j < Wr) ? ascii_nmask_word(io.ascii, s0 + 32ull * j, L - 32 * j, io.ascii_bytes) : 0;
        wave_sync();
        return hasN;
    }
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

// The GL lanes of a read's group stage its words straight from the characters (several-reads-per-wave kernels; `on`: the group has a read it maps);
// -> nonzero in every lane of the group when the read holds an N (bit 3 of a character: such a read goes to the one-read-per-wave kernels)
__device__ __forceinline__ uint32_t quad_xor1(uint32_t x);
__device__ __forceinline__ uint32_t quad_xor2(uint32_t x);
__device__ __forceinline__ uint32_t half_row_mirror(uint32_t x);
template <int GL>
__device__ __forceinline__ uint32_t stage_group_ascii(const BatchIO& io, u64 a0, uint32_t L, uint32_t W, uint32_t on, u64* F, uint32_t sub) {
    uint32_t chars = 0;
    const uint32_t Wr = (L + 31) >> 5;
    for (uint32_t j = sub; j < W; j += GL) {
        u64 f = 0;
        if (on && j < Wr) {
            if (io.ascii_src) f = ascii_word<false>(io.ascii, a0 + 32ull * j, L - 32 * j, io.ascii_bytes, &chars);
            else f = ascii_word<true>(io.ascii, a0 + 32ull * j, L - 32 * j, io.ascii_bytes, &chars);
        }
        F[j] = f;
    }
    uint32_t n8 = chars & 0x08080808u;
    n8 |= quad_xor1(n8);
    n8 |= quad_xor2(n8);
    if (GL >= 8) n8 |= half_row_mirror(n8);
    if (GL == 16) n8 |= (uint32_t)__builtin_amdgcn_mov_dpp((int)n8, 0x140, 0xF, 0xF, true);  // row_mirror
    return n8;
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

// mismatches between n (1..32) read bases from base rb on and the unitig bases next to the overlap, taken from the slot
// (`near`: read away from the overlap; a walk to the left sees their reverse complement).  Reads without N only.
__device__ __forceinline__ uint32_t ham_near(const u64* CMP, u64 near, bool left, uint32_t n, uint32_t rb) {
    const u64 u = left ? ~rev2_fast(near >> (64 - 2 * n)) : near;
    const u64 x = u ^ lds_win32(CMP, rb);
    u64 mm = (x | (x >> 1)) & EVEN_BITS;
    if (n < 32) mm &= ~(~0ULL >> (2 * n));
    return __popcll(mm);
}

// `hnd` = handle of the half to read (half_handle / a slot's nx word without its flag bit); the half's slots follow each other,
// the last one flagged (candidate lanes behind it see whatever comes next in the array and are masked off)
template <int DIR>
__device__ __forceinline__ Scored score_candidates(const BgrDeviceGraph& g, const u64* CMP, const u64* NM, bool useN, uint32_t L,
                                                   uint32_t K1, uint32_t hnd, bool canon, uint32_t pos, int lane) {
    Scored sc;
    const int c = lane >> 4, sub = lane & 15;
    const uint32_t fbit = canon ? BGR_SLOT_F0 : BGR_SLOT_F1;
    // one 32-byte slot per candidate (graph_layout.h BgrSlot): id + orientation bits, length, sequence address |
    // where the walk goes on behind the unitig (what the NEXT step needs)
    const uint4* sp = reinterpret_cast<const uint4*>(g.recs) + (size_t)(hnd + (uint32_t)c) * 2;
    const uint4 sl = sp[0];
    const uint4 m0 = sp[1];  // y = nx0, z = nx1: next half | canonical << 28, reached by a canonical / non-canonical query
    const uint32_t id = sl.x & BGR_SLOT_ID_MASK;
    const u64 lmask = __ballot((sl.w & BGR_SLOT_LAST) != 0) & 0x0001000100010001ULL;  // lane 0 of each candidate's 16
    sc.first_zero = ((__ffsll((long long)lmask) - 1) >> 4) + 1;  // candidates = slots up to and including the first flagged one
    const bool valid = c < sc.first_zero;
    const bool fwd = (sl.x & fbit) != 0;
    const uint32_t len = valid ? sl.y : 0;
    // oriented strand start: forward at (Fw, Fo), reverse complement `len` bases further
    const uint32_t fw = sl.z, fo = (sl.w & BGR_SLOT_FO_MASK) + (fwd ? 0u : len);
    sc.sid = fwd ? (int32_t)id : -(int32_t)id;
    sc.ext = len - K1;
    bool fits;
    uint32_t n, ustart, rstart;
    if (DIR == 0) {
        fits = sc.ext >= pos;
        n = fits ? pos : sc.ext;
        ustart = fits ? sc.ext - pos : 0;
        rstart = fits ? 0 : pos - sc.ext;
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
    }
    {
        const uint32_t nx = canon ? m0.y : m0.z;
        sc.nrec = nx & BGR_HNONE;
        sc.info = (fits ? 1u : 0u) | ((nx & BGR_H_CANON) ? 2u : 0u);
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

__device__ __forceinline__ uint32_t quad_xor1(uint32_t x) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t quad_xor2(uint32_t x) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t row_ror4(uint32_t x) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x124, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t half_row_mirror(uint32_t x) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x141, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t row_ror8(uint32_t x) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x128, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t lane_get(uint32_t v, uint32_t src_lane) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v); }

// record index | canonical << 28 (| fits << 29 | found << 30 in a step result) as the several-reads-per-wave kernels pass it around
#define G4_REC_MASK 0x0FFFFFFFu
#define G4_CANON (1u << 28)
#define G4_FITS (1u << 29)
#define G4_FOUND (1u << 30)
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
    if (rec == BGR_HNONE) return out;  // no such half: getBegin/getEnd return an empty list  (`rec` = the HANDLE of the half to read)
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
    rec = half_handle(g, rec, canon, true);  // (the caller names the key entry; the steps pass handles on)
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
    rec = half_handle(g, rec, canon, false);
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
    uint32_t pos = a_pos, rec = half_handle(g, a_rec, a_canon, true);  // (a_rec: the anchor's key entry; the walks pass handles on)
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
    pos = a_pos; rec = half_handle(g, a_rec, a_canon, false); canon = a_canon;
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

// ---- the extension step of the several-reads-per-wave kernels (greedy mode, anchors mode) ----------------------------------------
// One extension step for up to 64 / GL walks, one per GL-lane group (GL = 16, 8 or 4: four, two or one lane per candidate slot).  `phase` (uniform within a group): 0 = the group sits
// out, 1 = left step (checkBeginGreedy / mapOnLeftEndGreedy), 2 = first right step (checkEndGreedy: the read slice starts
// behind the k-1 overlap), 3 = later right step (mapOnRightEndGreedy: the slice includes the overlap).  alignerGreedy.cpp:167-364.
// Result, uniform within a group: next record | next canonical << 28 | fits << 29 | found << 30; miss; ext; sid.
template <int GL = 16>
__device__ __forceinline__ uint32_t g4_step(const BgrDeviceGraph& g, const u64* FW, uint32_t L, uint32_t K1, uint32_t phase, uint32_t rec, uint32_t canon,
                                            uint32_t pos, uint32_t budget, int lane, uint32_t* miss, uint32_t* ext_o, int32_t* sid_o) {
    constexpr uint32_t QL = GL / 4;  // lanes per candidate slot: each takes 32 bases per round of the compare
    uint32_t c = ((uint32_t)lane / QL) & 3u, q = (uint32_t)lane % QL, ql = QL;
    const uint32_t left = phase == 1 ? 1u : 0u;
    // `rec` = the HANDLE of the half to read (G4_REC_MASK == BGR_HNONE: none): its slots follow each other, the last one flagged; lanes
    // of candidates behind it see whatever comes next in the array and are masked off
    uint4 sl = make_uint4(0, 0, 0, BGR_SLOT_LAST), m0 = make_uint4(0, 0, 0, 0);
    if (phase != 0 && rec != G4_REC_MASK) {
        const uint4* sp = reinterpret_cast<const uint4*>(g.recs) + (size_t)(rec + c) * 2;
        sl = sp[0];
        m0 = sp[1];
    }
    const u64 lmask = __ballot((sl.w & BGR_SLOT_LAST) != 0);  // (all lanes of a candidate agree)
    const uint32_t nb = (uint32_t)(lmask >> ((uint32_t)lane & (64u - GL))) & (GL == 16 ? 0x1111u : GL == 8 ? 0x55u : 0xFu);
    // candidates = the slots up to and including the first flagged one (a group that sits out, or whose half is empty: none)
    const uint32_t n_cand = (phase != 0 && rec != G4_REC_MASK && nb) ? (uint32_t)(__ffs((int)nb) - 1) / QL + 1u : 0u;
    // four lanes per read: a half with one or two candidates (nearly every half of a graph of 2-allele sites) gives the lanes of candidates
    // 2 and 3 to candidates 0 and 1 -- they take the slot over (DPP, lanes 0 1 0 1 of the quad) and compare every other 32-base chunk, so
    // the compare loop below runs half the rounds (each round costs the whole wave ~100 instructions)
    uint32_t two = 0;
    if (GL == 4) {
        two = (n_cand - 1u < 2u) ? 1u : 0u;
        const bool take = two && ((uint32_t)lane & 2u);
        const uint32_t t0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)sl.x, 0x44, 0xF, 0xF, true), t1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)sl.y, 0x44, 0xF, 0xF, true),
                       t2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)sl.z, 0x44, 0xF, 0xF, true), t3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)sl.w, 0x44, 0xF, 0xF, true);
        if (take) { sl.x = t0; sl.y = t1; sl.z = t2; sl.w = t3; }
        if (two) { c = (uint32_t)lane & 1u; q = ((uint32_t)lane >> 1) & 1u; ql = 2; }
    }
    const uint32_t id = (phase != 0 && rec != G4_REC_MASK) ? sl.x & BGR_SLOT_ID_MASK : 0u;
    const uint32_t fwd = (sl.x & (canon ? BGR_SLOT_F0 : BGR_SLOT_F1)) ? 1u : 0u;
    const uint32_t len = sl.y;
    const uint32_t fw = sl.z, fo = (sl.w & BGR_SLOT_FO_MASK) + (fwd ? 0u : len);
    const uint32_t ext = len - K1;
    // left: `rl` bases of the read lie left of the overlap; right: behind it (first step) / from its start (later steps)
    const uint32_t kk = phase == 2 ? K1 : 0u;
    const uint32_t rl = left ? pos : L - pos - kk;
    const uint32_t fits = ext >= rl ? 1u : 0u;
    const uint32_t span = left ? ext : ext + K1 - kk;  // what is compared when the walk goes on: the unitig beyond the overlap, or all of it
    uint32_t n = fits ? rl : (span < rl ? span : rl);  // (later right steps: read.substr(pos, |u|) is clipped at |read|)
    const uint32_t ustart = left ? ext - n : kk;
    const uint32_t rstart = left ? rl - n : pos + kk;
    const uint32_t nx = canon ? m0.y : m0.z;  // next half | canonical << 28 (graph_layout.h nx0 / nx1)
    if (c >= n_cand) n = 0;
    // (a slot also carries the <= 32 bases next to its overlap, graph_layout.h `near`: the exhaustive several-reads-per-wave kernel compares
    // against those when the graph lives in L2/HBM.  Here, with sixteen walks per wave waiting on memory at once, the launch is bound by
    // instruction issue and the second compare path cost more than the load it saves -- chr1-scale 1 331 with, 1 359 Mreads/s without;
    // E. coli scale 1 206 / 1 286 -- so this step always reads the bases from `seq`)
    uint32_t cnt = 0;
    for (uint32_t b = q * 32; wave_any(b < n); b += 32 * ql)
        if (b < n) cnt += ham_chunk(g, FW, nullptr, false, fw, fo + ustart + b, rstart + b, n - b);
    if (QL >= 2) cnt += quad_xor1(cnt);
    if (QL == 4) cnt += quad_xor2(cnt);
    if (GL == 4) { const uint32_t o2 = quad_xor2(cnt); if (two) cnt += o2; }  // (the two lanes of a candidate: l and l ^ 2)
    // best = smallest miss, lowest slot on ties, only if miss <= budget (== "first zero wins, else strict min")
    uint32_t key = c < n_cand ? ((cnt > 0x0FFFFFFFu ? 0x0FFFFFFFu : cnt) << 2) | c : 0xFFFFFFFFu;
    uint32_t o = GL == 16 ? row_ror4(key) : GL == 8 ? quad_xor2(key) : quad_xor1(key);   // 16 lanes: slots sit 4 lanes apart; 8 lanes: 2 apart; 4: neighbours
    key = o < key ? o : key;
    o = GL == 16 ? row_ror8(key) : GL == 8 ? half_row_mirror(key) : quad_xor2(key);       // (8: lane 7 - l of the half row: the other two slots)
    key = o < key ? o : key;
    const uint32_t src = ((uint32_t)lane & (64u - GL)) | ((key & 3u) * QL);
    const uint32_t pk = (nx & (G4_REC_MASK | G4_CANON)) | (fits ? G4_FITS : 0u);
    const uint32_t w1 = lane_get(pk, src);
    *ext_o = lane_get(ext, src);
    *sid_o = (int32_t)lane_get(fwd ? id : 0u - id, src);
    *miss = key >> 2;
    return (key >> 2) <= budget ? w1 | G4_FOUND : 0u;  // an empty record gives key 0xFFFFFFFF: not found
}

// waves per SIMD the kernels are compiled for (__launch_bounds__)
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
#ifndef BGR_EXH_DEEP_OCC
#define BGR_EXH_DEEP_OCC 4 /* ... its variant with the search state in HBM: it carries the level search as well (128 VGPRs: the last pass sees few reads, its occupancy matters little) */
#endif

// ================================================ kernels ===============================================
template <bool STAGE>
__device__ __forceinline__ const uint32_t* block_prologue(const BgrDeviceGraph& g, u64* lds, uint32_t* tab_words) {
    // LDS: [512 B reserved][optional copy of the key table, rounded up to 16 B][per-wave regions]
    *tab_words = STAGE ? ((g.table_bytes + 15) / 16) * 2 : 0;
    const uint32_t* table = g.table;
    if (STAGE) {
        const uint4* src = reinterpret_cast<const uint4*>(g.table);  // (the blob keeps 16 spare bytes behind the table)
        uint4* dst = reinterpret_cast<uint4*>(lds + 64);
        for (uint32_t i = threadIdx.x; i < (g.table_bytes + 15) / 16; i += blockDim.x) dst[i] = src[i];
        table = reinterpret_cast<const uint32_t*>(lds + 64);
    }
    __syncthreads();
    return table;
}

// The aligner.h:68 counters of a launch ([0] readNumber [1] noOverlapRead [2] alignedRead [3] notAligned [4] overlaps) are summed per
// workgroup in the first words of its LDS and added to HBM by one thread: one global atomic per wave and counter, all on the same
// few addresses, cost a launch of 131 k reads a third of its time.  wg_counts_init before block_prologue (its barrier publishes the
// zeroes); wg_counts_flush at the very end of the kernel, reached by every wave of the workgroup.
__device__ __forceinline__ unsigned long long* wg_counts_init(u64* lds) {
    unsigned long long* c = reinterpret_cast<unsigned long long*>(lds);
    if (threadIdx.x < 6) c[threadIdx.x] = 0;
    return c;
}
__device__ __forceinline__ void wg_counts_flush(const BatchIO& io, unsigned long long* c, int lane, unsigned long long reads, unsigned long long noov, unsigned long long al,
                                                unsigned long long na, unsigned long long ov) {
    if (lane == 0) {
        if (reads) atomicAdd(&c[0], reads);
        if (noov) atomicAdd(&c[1], noov);
        if (al) atomicAdd(&c[2], al);
        if (na) atomicAdd(&c[3], na);
        if (ov) atomicAdd(&c[4], ov);
    }
    __syncthreads();
    if (threadIdx.x < 5 && c[threadIdx.x]) atomicAdd(reinterpret_cast<unsigned long long*>(io.cursor + 16) + threadIdx.x, c[threadIdx.x]);
}

// ---- which reads a wave maps next: claimed at run time, not dealt out beforehand ------------------------------------------------
// A TASK is one wave-load of reads (64 / lanes-per-read of them, consecutive in the batch).  Dealing the tasks out by wave number
// leaves a third of the wave slots idle: the waves of a SIMD do not advance at one rate (the issue arbiter prefers the oldest), the
// first ones end after 55 % of the launch and nothing takes their place (profiles/r04_wave_times_static_split_*.txt: a wave of the
// E. coli-scale launch lives 0.74 of its duration, 0.64 at configs[1], 0.83 at chr1 scale).  So a wave CLAIMS its next task: from its
// workgroup's stock in LDS (one 64-bit word {end, next}, one LDS atomic per task), which the wave that finds it used up refills with
// up to kTaskRefill tasks from the launch's counter in HBM (one global atomic per refill: ~20-40 k per launch).  The end of a launch
// is then ragged by one task per wave, not by a third of the batch.
#ifndef BGR_TASK_SHARE_DIV
#define BGR_TASK_SHARE_DIV 2u
#endif
#ifndef BGR_TASK_REFILL
#define BGR_TASK_REFILL 16
#endif
#ifndef BGR_TASK_REFILL_MIN
#define BGR_TASK_REFILL_MIN 1
#endif
constexpr uint32_t kTaskRefill = BGR_TASK_REFILL;        // tasks a workgroup takes from the launch's counter at a time, at most (round 4: 32; same box, interleaved, round 5:
constexpr uint32_t kTaskRefillMin = BGR_TASK_REFILL_MIN;  // ... and at least (4)        16 / 1 gives the default line +0.5 %, configs[1] +3 %; 64 / 2 and 8 / 1 lose)
constexpr uint32_t kTaskDone = 0xC0000000u;   // `next` of a stock whose launch has no task left
constexpr uint32_t kLdsTaskWord = 8;          // the stock: u64 number 8 of the workgroup's reserved LDS header (bytes 64 .. 71)
// A workgroup starts with one task per wave, its own by number (tasks [b W, (b + 1) W) for workgroup b of W waves): no atomic at the start of a
// launch, when every workgroup would ask at once (512 of them queue ~6 us on one address: a fifth of a 131 k-read launch, whose tasks are as many
// as its waves); the launch's counter hands out the tasks behind those, so it counts from gridDim.x * W (task_stock_base).
__device__ __forceinline__ uint32_t task_stock_base() { return gridDim.x * (blockDim.x >> 6); }
__device__ __forceinline__ void task_stock_init(u64* lds, uint32_t n_tasks) {  // (before the prologue's barrier)
    if (threadIdx.x == 0) {
        const uint32_t w = blockDim.x >> 6, first = blockIdx.x * w, end = n_tasks - first < w ? n_tasks : first + w;
        lds[kLdsTaskWord] = first < n_tasks ? ((u64)end << 32) | first : 0;  // (an empty stock {0, 0}: the first claim goes to the counter and finds the launch used up)
    }
}
// -> the task claimed, or BGR_NONE when the launch has none left; wave-uniform.  `ctr`: the launch's counter (zero at launch).
__device__ __forceinline__ uint32_t claim_task(u64* lds, uint32_t* ctr, uint32_t n_tasks, int lane) {
    const uint32_t base = task_stock_base();
    unsigned long long* stock = reinterpret_cast<unsigned long long*>(lds + kLdsTaskWord);
    for (;;) {
        unsigned long long old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(stock, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t next = rl32((uint32_t)old, 0), end = rl32((uint32_t)(old >> 32), 0);
        if (next < end) return next;
        if (next >= kTaskDone) return BGR_NONE;
        if (next == end) {  // exactly one wave sees the stock run out (or finds it empty at the start): it refills, and takes the first task itself
            // (guided: what is left of the batch -- judged by where the stock just used up ended -- shared out twice over all workgroups, at most
            // kTaskRefill and at least kTaskRefillMin tasks: towards the end of a launch the workgroups take small bites)
            const uint32_t seen = end > base ? end : base;  // (tasks known to be handed out: the static first ones, and what this workgroup took last)
            const uint32_t left = n_tasks > seen ? n_tasks - seen : 0u, share = left / (BGR_TASK_SHARE_DIV * gridDim.x);
            const uint32_t want = share > kTaskRefill ? kTaskRefill : share < kTaskRefillMin ? kTaskRefillMin : share;
            uint32_t g = 0;
            if (lane == 0) g = atomicAdd(ctr, want);
            g = rl32(g, 0) + base;
            const bool none = g >= n_tasks;
            const uint32_t e = n_tasks - g < want ? n_tasks : g + want;
            if (lane == 0) __hip_atomic_store(stock, none ? (unsigned long long)kTaskDone : ((unsigned long long)e << 32) | (g + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return none ? BGR_NONE : g;
        }
        __builtin_amdgcn_s_sleep(4);  // next > end: another wave of the workgroup is refilling the stock; it never waits for this one
    }
}

// Arena space comes in per-wave chunks: ONE global atomic per ~50 reads instead of one per read (a
// single-address atomic saturates near 90 M/s chip-wide, MI355X_MICROARCH.md "dequeue").
__device__ __forceinline__ uint32_t publish_path(const BatchIO& io, const int32_t* PATH, uint32_t p_lo, uint32_t p_n,
                                                 uint32_t* chunk_pos, uint32_t* chunk_end, int lane) {
    if (p_n > *chunk_end - *chunk_pos) {
        const uint32_t want = p_n > io.arena_chunk ? p_n : io.arena_chunk;
        uint32_t got = 0;
        if (lane == 0) got = io.arena_own + atomicAdd(io.cursor, want);
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
}  // namespace bgr

#endif
