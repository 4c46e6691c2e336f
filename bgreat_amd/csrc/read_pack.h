// read_pack.h -- host twin of the device pre-pass (bgr_pack_reads_kernel, batch_kernels.hip): reads as the 2-bit planes the
// mapping kernels read, so that a batch crosses PCIe at ~0.3 byte per base instead of 1.  Host only, header only.
//
// Layout (same as on the device): str2num codes (utils.cpp:117-129: A0 C1 G2, anything else 3 -- the parser admits only
// ACGTN, aligner.cpp:56-61, so "anything else" is N), 32 bases per u64, first base most significant, zero beyond the
// read's end.  Read r of a batch owns words [w, w + ceil(len/32)) of the plane, w = (read_offsets[r] >> 5) + r.  Reads
// that hold an N additionally get an N-mask word (3 on every N) for each of their words, kept as a sparse (index, value)
// list, and a bit in the `hasn` bitmap.
#ifndef BGREAT_AMD_READ_PACK_H
#define BGREAT_AMD_READ_PACK_H

#include <stdint.h>
#include <string.h>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace bgr {

inline uint64_t packed_plane_words(uint64_t n_reads, uint64_t total_bases) { return (total_bases >> 5) + n_reads + 4; }
inline uint64_t packed_word_offset(uint64_t base_offset, uint64_t r) { return (base_offset >> 5) + r; }

// 16 characters -> 32 bits (first base in the top two bits); *nmask16: bit i set = character i is 'N'
inline uint32_t pack16_scalar(const unsigned char* s, uint32_t* nmask16) {
    uint32_t w = 0, nm = 0;
    for (int i = 0; i < 16; ++i) {
        const unsigned c = s[i];
        unsigned code = ((c >> 1) ^ (c >> 2)) & 3u;
        if (c == 'N') { code = 3; nm |= 1u << i; }
        w = (w << 2) | code;
    }
    *nmask16 = nm;
    return w;
}

#if defined(__x86_64__)
__attribute__((target("ssse3"))) inline uint32_t pack16_ssse3(const unsigned char* s, uint32_t* nmask16) {
    const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s));
    const __m128i three = _mm_set1_epi8(3);
    const __m128i isn = _mm_cmpeq_epi8(v, _mm_set1_epi8('N'));
    // ((c >> 1) ^ (c >> 2)) & 3 per byte (16-bit shifts: what crosses a byte boundary lands in bits the mask drops)
    __m128i c = _mm_and_si128(_mm_xor_si128(_mm_srli_epi16(v, 1), _mm_srli_epi16(v, 2)), three);
    c = _mm_or_si128(c, _mm_and_si128(isn, three));
    const __m128i nib = _mm_maddubs_epi16(c, _mm_set1_epi16(0x0104));        // byte pairs -> c0*4 + c1
    const __m128i byt = _mm_madd_epi16(nib, _mm_set1_epi32(0x00010010));     // nibble pairs -> n0*16 + n1 (4 bases per 32-bit lane)
    const __m128i out = _mm_shuffle_epi8(byt, _mm_set_epi8(-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12));
    *nmask16 = (uint32_t)_mm_movemask_epi8(isn);
    return (uint32_t)_mm_cvtsi128_si32(out);
}
inline bool have_ssse3() {
    static const bool ok = __builtin_cpu_supports("ssse3");
    return ok;
}
// 32 characters -> one plane word (first base in the top two bits); *nmask32: bit i set = character i is 'N'
__attribute__((target("avx2"))) inline uint64_t pack32_avx2(const unsigned char* s, uint32_t* nmask32) {
    const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s));
    const __m256i three = _mm256_set1_epi8(3);
    const __m256i isn = _mm256_cmpeq_epi8(v, _mm256_set1_epi8('N'));
    __m256i c = _mm256_and_si256(_mm256_xor_si256(_mm256_srli_epi16(v, 1), _mm256_srli_epi16(v, 2)), three);
    c = _mm256_or_si256(c, _mm256_and_si256(isn, three));
    const __m256i nib = _mm256_maddubs_epi16(c, _mm256_set1_epi16(0x0104));     // byte pairs -> c0*4 + c1
    const __m256i byt = _mm256_madd_epi16(nib, _mm256_set1_epi32(0x00010010));  // nibble pairs -> n0*16 + n1 (4 bases per 32-bit lane)
    const __m256i sel = _mm256_setr_epi8(12, 8, 4, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 12, 8, 4, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    const __m256i out = _mm256_shuffle_epi8(byt, sel);  // per 128-bit half: its four bytes, first base's byte on top of the low dword
    *nmask32 = (uint32_t)_mm256_movemask_epi8(isn);
    const uint32_t hi = (uint32_t)_mm_cvtsi128_si32(_mm256_castsi256_si128(out));
    const uint32_t lo = (uint32_t)_mm_cvtsi128_si32(_mm256_extracti128_si256(out, 1));
    return (uint64_t)hi << 32 | lo;
}
inline bool have_avx2() {
    static const bool ok = __builtin_cpu_supports("avx2");
    return ok;
}
#endif

inline uint32_t pack16(const unsigned char* s, uint32_t* nmask16) {
#if defined(__x86_64__)
    if (have_ssse3()) return pack16_ssse3(s, nmask16);
#endif
    return pack16_scalar(s, nmask16);
}

// One read -> ceil(len/32) words at fw.  Returns true when the read holds an N; nm (may be null when the caller does not
// want the mask) then receives the same number of N-mask words.  nm must not alias fw.
inline bool pack_read(const char* seq, uint32_t len, uint64_t* fw, uint64_t* nm) {
    const unsigned char* s = reinterpret_cast<const unsigned char*>(seq);
    const uint32_t words = (len + 31) >> 5;
    bool any = false;
    uint32_t done = 0;
    for (uint32_t w = 0; w < words; ++w) {
        uint32_t half[2] = {0, 0}, nmh[2] = {0, 0};
#if defined(__x86_64__)
        if (len - done >= 32 && have_avx2()) {  // a whole word at once
            uint32_t nm32;
            fw[w] = pack32_avx2(s + done, &nm32);
            done += 32;
            nmh[0] = nm32 & 0xFFFFu;
            nmh[1] = nm32 >> 16;
            goto mask;
        }
#endif
        for (int h = 0; h < 2; ++h) {
            if (done >= len) break;
            if (len - done >= 16) {
                half[h] = pack16(s + done, &nmh[h]);
                done += 16;
            } else {  // the read's tail: zero padded (a zero byte packs to 0 and is no N)
                unsigned char tmp[16];
                memset(tmp, 0, sizeof(tmp));
                memcpy(tmp, s + done, len - done);
                half[h] = pack16(tmp, &nmh[h]);
                done = len;
            }
        }
        fw[w] = (uint64_t)half[0] << 32 | half[1];
#if defined(__x86_64__)
    mask:
#endif
        if (nmh[0] | nmh[1]) {
            if (!any && nm) for (uint32_t j = 0; j < w; ++j) nm[j] = 0;
            any = true;
        }
        if (any && nm) {
            uint64_t m = 0;
            const uint32_t bits = nmh[0] | nmh[1] << 16;  // bit i = base 32w + i is N
            for (uint32_t b = bits; b; b &= b - 1) m |= 3ull << (62 - 2 * (uint32_t)__builtin_ctz(b));
            nm[w] = m;
        }
    }
    return any;
}

}  // namespace bgr
#endif
