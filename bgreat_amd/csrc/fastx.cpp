// fastx.cpp -- FASTA/FASTQ reader reproducing what Aligner::getReads (aligner.cpp:46-117) accepts, over a
// memory image of the file instead of an ifstream.  The reference's behaviour is defined by the iostream
// calls it makes (getline / peek / eof in a fixed order, 10000 records per call); `Cursor` models exactly
// the stream state those calls observe (eofbit, failbit, "getline on a failed stream leaves the string
// untouched"), and the two loops below are the same state machines driven through it.
#include "fastx.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <thread>
#if defined(__SSE2__)
#include <immintrin.h>
#endif

namespace bgr {

namespace {

struct Slice {
    const char* p = "";
    uint64_t n = 0;
};

struct Cursor {
    const char* d;
    uint64_t n, pos;
    bool eofbit = false, failbit = false;
    Cursor(const char* data, uint64_t begin, uint64_t end) : d(data), n(end), pos(begin) {}
    bool good() const { return !eofbit && !failbit; }
    // std::getline(stream, s): returns false when s is left untouched (stream was not good()).
    bool getline(Slice& s) {
        if (!good()) { failbit = true; return false; }
        if (pos == n) { eofbit = failbit = true; s.p = d + pos; s.n = 0; return true; }  // s erased, nothing extracted
        const char* b = d + pos;
        const char* q = static_cast<const char*>(memchr(b, '\n', n - pos));
        if (q) { s.p = b; s.n = (uint64_t)(q - b); pos = (uint64_t)(q - d) + 1; }
        else { s.p = b; s.n = n - pos; pos = n; eofbit = true; }
        return true;
    }
    int peek() {
        if (!good()) { failbit = true; return -1; }
        if (pos == n) { eofbit = true; return -1; }
        return (unsigned char)d[pos];
    }
};

struct ValidTable {
    bool ok[256];
    ValidTable() { memset(ok, 0, sizeof(ok)); ok['A'] = ok['C'] = ok['G'] = ok['T'] = ok['N'] = true; }
};
const ValidTable kValid;

inline bool valid_chars(const char* p, uint64_t n) {  // aligner.cpp:56-61
    for (uint64_t i = 0; i < n; ++i)
        if (!kValid.ok[(unsigned char)p[i]]) return false;
    return true;
}

// One pass over a line: returns the position of its '\n' (or `end`), and whether every byte before it is one of
// ACGTN (aligner.cpp:56-61).  Replaces memchr + valid_chars on the sequence line of a record (the bulk of a file).
#if defined(__x86_64__)
// The same with 32 bytes per step, when the CPU has AVX2 (checked once at run time: the binary is built without -march).
// Validity by two nibble look-ups instead of five compares: the admitted characters are 0x41 0x43 0x47 0x4E (high nibble 4,
// low nibble 1 3 7 E) and 0x54 (high nibble 5, low nibble 4); a byte passes when its two table entries share a bit.
__attribute__((target("avx2"))) const char* scan_line_avx2(const char* p, const char* end, unsigned& bad_io, bool& found) {
    const __m256i lut_lo = _mm256_setr_epi8(0, 1, 0, 1, 2, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0, 1, 2, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0);
    const __m256i lut_hi = _mm256_setr_epi8(0, 0, 0, 0, 1, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i nib = _mm256_set1_epi8(0x0F), vNL = _mm256_set1_epi8('\n'), zero = _mm256_setzero_si256();
    unsigned bad = bad_io;
    found = false;
    while (p + 32 <= end) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p));
        const __m256i lo = _mm256_shuffle_epi8(lut_lo, _mm256_and_si256(v, nib));
        const __m256i hi = _mm256_shuffle_epi8(lut_hi, _mm256_and_si256(_mm256_srli_epi16(v, 4), nib));
        const unsigned m_bad = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_and_si256(lo, hi), zero));
        const unsigned m_nl = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, vNL));
        if (m_nl) {
            const unsigned idx = (unsigned)__builtin_ctz(m_nl);
            bad |= m_bad & (idx ? (0xFFFFFFFFu >> (32 - idx)) : 0u);
            bad_io = bad;
            found = true;
            return p + idx;
        }
        bad |= m_bad;
        p += 32;
    }
    bad_io = bad;
    return p;
}
static const bool kHaveAvx2 = __builtin_cpu_supports("avx2");
#endif

inline const char* scan_line(const char* p, const char* end, bool& valid) {
    unsigned bad = 0;
#if defined(__x86_64__)
    if (kHaveAvx2) {
        bool found;
        p = scan_line_avx2(p, end, bad, found);
        if (found) { valid = bad == 0; return p; }
    }
#endif
#if defined(__SSE2__)
    const __m128i vA = _mm_set1_epi8('A'), vC = _mm_set1_epi8('C'), vG = _mm_set1_epi8('G'), vT = _mm_set1_epi8('T'),
                  vN = _mm_set1_epi8('N'), vNL = _mm_set1_epi8('\n');
    while (p + 16 <= end) {
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p));
        const __m128i ok = _mm_or_si128(_mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(v, vA), _mm_cmpeq_epi8(v, vC)),
                                                     _mm_or_si128(_mm_cmpeq_epi8(v, vG), _mm_cmpeq_epi8(v, vT))), _mm_cmpeq_epi8(v, vN));
        const unsigned m_nl = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(v, vNL));
        const unsigned m_bad = ~(unsigned)_mm_movemask_epi8(ok) & 0xFFFFu;
        if (m_nl) {
            const unsigned idx = (unsigned)__builtin_ctz(m_nl);
            bad |= m_bad & ((1u << idx) - 1u);
            valid = bad == 0;
            return p + idx;
        }
        bad |= m_bad;
        p += 16;
    }
#endif
    for (; p < end && *p != '\n'; ++p) bad |= kValid.ok[(unsigned char)*p] ? 0u : 1u;
    valid = bad == 0;
    return p;
}

inline void push(ParsedChunk& out, const Slice& h, const char* s, uint64_t sn, bool from_joined, const std::string& tmp) {
    RecSlice r;
    r.h = h.p;
    r.hl = (uint32_t)h.n;
    r.sl = (uint32_t)sn;
    if (from_joined) {
        r.s = reinterpret_cast<const char*>((uintptr_t)out.joined.size());
        out.joined.append(tmp);
        out.joined_idx.push_back((uint32_t)out.recs.size());
    } else {
        r.s = s;
    }
    out.recs.push_back(r);
    if (out.track_iters) out.rec_iter.push_back((uint32_t)(out.iters - 1));  // the iteration in progress
    out.seq_bytes += sn;
    out.hdr_bytes += h.n;
}

const unsigned kBatch = 10000;  // alignerGreedy.cpp:375

void to_readset(const ParsedChunk& c, ReadSet& out) {
    for (const RecSlice& r : c.recs) {
        out.headers.insert(out.headers.end(), r.h, r.h + r.hl);
        out.header_offs.push_back(out.headers.size());
        out.reads.insert(out.reads.end(), r.s, r.s + r.sl);
        out.read_offs.push_back(out.reads.size());
    }
}

}  // namespace

void parse_fasta_chunk(const char* data, uint64_t begin, uint64_t end, uint32_t k, ParsedChunk& out) {
    Cursor cur(data, begin, end);
    std::string joined;
    out.recs.reserve(out.recs.size() + (size_t)((end - begin) / 96) + 16);
    while (!cur.eofbit) {  // alignerGreedy.cpp:372  while(!readFile.eof())
        Slice header, read, inter;  // one getReads() call: its locals start empty
        bool returned = false;
        for (unsigned i = 0; i < kBatch && !returned; ++i) {
            ++out.iters;
            cur.getline(header);
            // Fast path for the shape nearly every record has: the stream is good, the sequence line ends with a
            // newline and the next line starts with '>'.  Exactly what the general machine below does for it
            // (getline, peek == '>', the three tests, read = ""), with the newline search and the character test fused.
            if (cur.good() && cur.pos < cur.n) {
                bool ok;
                const char* b = data + cur.pos;
                const char* q = scan_line(b, data + cur.n, ok);
                if (q + 1 < data + cur.n && q[1] == '>') {
                    const uint64_t rn = (uint64_t)(q - b);
                    if (rn > 2 && ok && rn > k) push(out, header, b, rn, false, joined);
                    cur.pos = (uint64_t)(q - data) + 1;
                    continue;
                }
            }
            cur.getline(read);
            bool multi = false;
            for (;;) {
                int c = cur.peek();
                const char* rp = multi ? joined.data() : read.p;
                uint64_t rn = multi ? joined.size() : read.n;
                if (c == '>') {
                    if (rn > 2 && valid_chars(rp, rn) && rn > k) push(out, header, rp, rn, multi, joined);
                    read = Slice();  // aligner.cpp:91  read=""
                    break;
                }
                if (!cur.eofbit) {
                    cur.getline(inter);
                    if (inter.n) {
                        if (!multi) { joined.assign(read.p, read.n); multi = true; }
                        joined.append(inter.p, inter.n);
                    }
                } else {
                    if (rn > 2 && valid_chars(rp, rn) && rn > k) push(out, header, rp, rn, multi, joined);
                    returned = true;
                    break;
                }
            }
        }
    }
    out.fixup();
}

static void parse_fastq_range(const char* data, uint64_t begin, uint64_t size, ParsedChunk& out);

void parse_fastq_image(const char* data, uint64_t size, ParsedChunk& out) { parse_fastq_range(data, 0, size, out); }

FastqPlan::FastqPlan(const char* data, uint64_t size, uint64_t chunk_bytes)
    : data_(data), size_(size), chunk_bytes_(chunk_bytes ? chunk_bytes : 1) {
    nc_ = (size_t)((size_ + chunk_bytes_ - 1) / chunk_bytes_);
    nl_.assign(nc_ + 1, 0);
    tail_start_.assign(nc_, UINT64_MAX);
}

static uint64_t count_newlines(const char* p, const char* end);
static void collect_newlines(const char* p, const char* end, std::vector<uint32_t>& out);

void FastqPlan::count_chunk(size_t c) {  // pass 1: newlines in chunk c
    const uint64_t b = c * chunk_bytes_, e = std::min<uint64_t>(size_, b + chunk_bytes_);
    if (keep_nl_ && chunk_bytes_ <= 0xFFFFFFFFull) {  // (one pass: where the newlines are, for the gather behind; their number follows)
        auto v = std::make_shared<std::vector<uint32_t>>();
        v->reserve((size_t)((e - b) / 64) + 16);
        collect_newlines(data_ + b, data_ + e, *v);
        nl_[c + 1] = v->size();
        nlpos_[c] = std::move(v);
        return;
    }
    nl_[c + 1] = count_newlines(data_ + b, data_ + e);
}

void FastqPlan::extend_counts(size_t c_end) {  // chunks [0, c_end) have been counted: their prefix sums
    if (c_end > nc_) c_end = nc_;
    for (size_t c = summed_; c < c_end; ++c) nl_[c + 1] += nl_[c];  // nl_[c] = newlines before chunk c = index of the line containing its first byte
    if (c_end > summed_) summed_ = c_end;
}

void FastqPlan::finish_counts() {
    extend_counts(nc_);
    complete_ = nc_ ? nl_[nc_] / 4 : 0;                      // records whose four lines all end with a newline
    par_records_ = complete_ ? ((complete_ - 1) / kBatch) * kBatch : 0;  // a getReads() call boundary
}

bool FastqPlan::parse_chunk(size_t c, ParsedChunk& out) {  // pass 2: chunk c owns the records whose header line STARTS inside it
    const char* data = data_;
    const uint64_t size = size_;
    const uint64_t b = c * chunk_bytes_, e = std::min<uint64_t>(size, b + chunk_bytes_);
    uint64_t line = nl_[c];  // index of the line containing byte b
    uint64_t pos = b;
    if (b > 0 && data[b - 1] != '\n') {  // b is inside a line: the first line starting in this chunk is the next one
        const char* q = static_cast<const char*>(memchr(data + b, '\n', (size_t)(e - b)));
        if (!q) return true;
        pos = (uint64_t)(q - data) + 1;
        ++line;
    }
    std::string none;
    out.recs.reserve(out.recs.size() + (size_t)((e - b) / 200) + 16);
    while (pos < e) {
        if (line % 4 == 0) {
            const uint64_t rec = line / 4;
            if (rec >= par_records_) { tail_start_[c] = pos; return false; }
            ++out.iters;  // record `rec` is this chunk's iteration number out.iters - 1
            const char* h = data + pos;
            const char* hq = static_cast<const char*>(memchr(h, '\n', (size_t)(size - pos)));
            const char* s = hq + 1;  // complete record: all four newlines exist
            bool ok;
            const char* sq = scan_line(s, data + size, ok);
            Slice hs; hs.p = h; hs.n = (uint64_t)(hq - h);
            const uint64_t sn = (uint64_t)(sq - s);
            if (sn > 2 && ok) push(out, hs, s, sn, false, none);
            pos = (uint64_t)(sq - data) + 1;
            line += 2;
        } else {
            const char* q = static_cast<const char*>(memchr(data + pos, '\n', (size_t)(size - pos)));
            if (!q) return true;
            pos = (uint64_t)(q - data) + 1;
            ++line;
        }
    }
    return true;
}

uint64_t FastqPlan::record_offset(uint64_t rec) const {
    const uint64_t T = 4 * rec;  // line T starts behind the T-th newline
    if (T == 0) return 0;
    // the chunk that holds the T-th newline: nl_[c] < T <= nl_[c + 1] (among the chunks summed so far)
    if (T > nl_[summed_]) return size_;
    size_t lo = 0, hi = summed_;
    while (lo + 1 < hi) { const size_t mid = (lo + hi) / 2; if (nl_[mid] < T) lo = mid; else hi = mid; }
    const uint64_t b = lo * chunk_bytes_, e = std::min<uint64_t>(size_, b + chunk_bytes_);
    uint64_t need = T - nl_[lo];
    const char* p = data_ + b;
    const char* end = data_ + e;
    while (p < end && (p = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p))))) { ++p; if (--need == 0) return (uint64_t)(p - data_); }
    return size_;
}

void FastqPlan::piece_parts(uint64_t o0, uint64_t o1, uint64_t r0, std::vector<std::pair<uint64_t, uint64_t>>& out) const {
    out.clear();
    out.push_back({o0, 4 * r0});
    for (size_t c = (size_t)(o0 / chunk_bytes_) + 1; c < nc_ && c <= summed_ && c * chunk_bytes_ < o1; ++c)
        out.push_back({c * chunk_bytes_, nl_[c]});  // nl_[c] = newlines in front of the chunk = number of the line its first byte lies in
}

// the newlines of [p, end), in order: fn(position).  32 bytes per step with AVX2 (a memchr call per ~80-byte line costs more than the line)
#if defined(__x86_64__)
template <typename F>
__attribute__((target("avx2"))) static inline void each_newline_avx2(const char* p, const char* end, F&& fn) {
    const __m256i vNL = _mm256_set1_epi8('\n');
    while (p + 32 <= end) {
        uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(p)), vNL));
        while (m) { fn(p + __builtin_ctz(m)); m &= m - 1; }
        p += 32;
    }
    for (; p < end; ++p) if (*p == '\n') fn(p);
}
__attribute__((target("avx2"))) static uint64_t count_newlines_avx2(const char* p, const char* end) {
    const __m256i vNL = _mm256_set1_epi8('\n');
    uint64_t n = 0;
    while (p + 32 <= end) {
        n += (uint64_t)__builtin_popcount((uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(p)), vNL)));
        p += 32;
    }
    for (; p < end; ++p) n += *p == '\n';
    return n;
}
#endif
template <typename F>
static inline void each_newline_plain(const char* p, const char* end, F&& fn) {
    while (p < end && (p = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p))))) { fn(p); ++p; }
}
static uint64_t count_newlines(const char* p, const char* end) {
#if defined(__x86_64__)
    if (kHaveAvx2) return count_newlines_avx2(p, end);
#endif
    uint64_t n = 0;
    each_newline_plain(p, end, [&](const char*) { ++n; });
    return n;
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) static void collect_newlines_avx2(const char* p, const char* end, std::vector<uint32_t>& out) {
    const char* const p0 = p;
    each_newline_avx2(p, end, [&](const char* q) { out.push_back((uint32_t)(q - p0)); });
}
#endif
static void collect_newlines(const char* p, const char* end, std::vector<uint32_t>& out) {
#if defined(__x86_64__)
    if (kHaveAvx2) { collect_newlines_avx2(p, end, out); return; }
#endif
    const char* const p0 = p;
    each_newline_plain(p, end, [&](const char* q) { out.push_back((uint32_t)(q - p0)); });
}

namespace {
struct GatherState {  // the kept lines (4j, 4j+1) of a FASTQ range, copied run by run: a run = header line + read line = one memcpy
    const char* run;  // start of the kept run the cursor is in, or null
    uint64_t line;
    char* o;
    inline void newline(const char* q) {
        const uint64_t ph = line++ & 3;
        if (ph == 1) { memcpy(o, run, (size_t)(q + 1 - run)); o += q + 1 - run; run = nullptr; }
        else if (ph == 3) run = q + 1;
    }
};
#if defined(__x86_64__)
__attribute__((target("avx2"))) void gather_avx2(const char* p, const char* end, GatherState& st) { each_newline_avx2(p, end, [&](const char* q) { st.newline(q); }); }
#endif
}  // namespace

uint64_t fastq_gather_lines(const char* data, uint64_t b, uint64_t e, uint64_t line, char* dst) {
    GatherState st{(line & 3) < 2 ? data + b : nullptr, line, dst};
#if defined(__x86_64__)
    if (kHaveAvx2) gather_avx2(data + b, data + e, st);
    else
#endif
    each_newline_plain(data + b, data + e, [&](const char* q) { st.newline(q); });
    if (st.run && st.run < data + e) { memcpy(st.o, st.run, (size_t)(data + e - st.run)); st.o += data + e - st.run; }  // the range ends inside a kept line
    return (uint64_t)(st.o - dst);
}

uint64_t fastq_gather_lines_at(const char* data, uint64_t b, uint64_t e, uint64_t line, char* dst, const uint32_t* nl, size_t n, uint64_t nl_base) {
    GatherState st{(line & 3) < 2 ? data + b : nullptr, line, dst};
    const uint32_t* p = std::lower_bound(nl, nl + n, (uint32_t)(b - nl_base));
    for (const uint32_t* end = nl + n; p < end && nl_base + *p < e; ++p) st.newline(data + nl_base + *p);
    if (st.run && st.run < data + e) { memcpy(st.o, st.run, (size_t)(data + e - st.run)); st.o += data + e - st.run; }
    return (uint64_t)(st.o - dst);
}

void parse_fastq_records(const char* data, uint64_t begin, uint64_t end, ParsedChunk& out) {
    std::string none;
    uint64_t pos = begin;
    while (pos < end) {
        ++out.iters;
        const char* h = data + pos;
        const char* hq = static_cast<const char*>(memchr(h, '\n', (size_t)(end - pos)));
        if (!hq) break;
        const char* s = hq + 1;
        bool ok;
        const char* sq = scan_line(s, data + end, ok);
        Slice hs; hs.p = h; hs.n = (uint64_t)(hq - h);
        const uint64_t sn = (uint64_t)(sq - s);
        if (sn > 2 && ok) push(out, hs, s, sn, false, none);
        pos = (uint64_t)(sq - data) + 1;
        for (int l = 0; l < 2 && pos < end; ++l) {  // the '+' line and the quality line
            const char* q = static_cast<const char*>(memchr(data + pos, '\n', (size_t)(end - pos)));
            if (!q) { pos = end; break; }
            pos = (uint64_t)(q - data) + 1;
        }
    }
}

void parse_fastq_from(const char* data, uint64_t begin, uint64_t size, ParsedChunk& out) { parse_fastq_range(data, begin, size, out); }

bool FastqPlan::parse_tail(ParsedChunk& out) {
    uint64_t tstart = size_;
    for (size_t c = 0; c < nc_; ++c) if (tail_start_[c] != UINT64_MAX) { tstart = tail_start_[c]; break; }
    if (par_records_ == 0) tstart = 0;
    if (complete_ == 0 || tstart < size_ || par_records_ == 0) { parse_fastq_range(data_, tstart, size_, out); return true; }
    return false;
}

void parse_fastq_parallel(const char* data, uint64_t size, unsigned threads, uint64_t chunk_bytes, std::vector<ParsedChunk>& chunks) {
    if (threads < 1) threads = 1;
    FastqPlan plan(data, size, chunk_bytes);
    const size_t nc = plan.chunks();
    chunks.clear();
    if (threads == 1 || nc < 2) { chunks.resize(1); parse_fastq_range(data, 0, size, chunks[0]); return; }
    auto run = [&](auto fn) {
        std::vector<std::thread> ts;
        for (unsigned t = 0; t < threads; ++t) ts.emplace_back([&, t]() { for (size_t c = t; c < nc; c += threads) fn(c); });
        for (auto& t : ts) t.join();
    };
    run([&](size_t c) { plan.count_chunk(c); });
    plan.finish_counts();
    chunks.resize(nc + 1);
    if (plan.sequential_only()) { plan.parse_tail(chunks[nc]); return; }
    run([&](size_t c) { plan.parse_chunk(c, chunks[c]); });
    plan.parse_tail(chunks[nc]);
}

static void parse_fastq_range(const char* data, uint64_t begin, uint64_t size, ParsedChunk& out) {
    Cursor cur(data, begin, size);
    std::string none;
    while (!cur.eofbit) {
        Slice header, read;  // one getReads() call: its locals start empty
        for (unsigned i = 0; i < kBatch; ++i) {
            ++out.iters;
            cur.getline(header);
            cur.getline(read);  // untouched (== previous record's sequence) once the stream has failed
            if (read.n > 2 && valid_chars(read.p, read.n)) push(out, header, read.p, read.n, false, none);
            cur.getline(header);
            cur.getline(header);
            if (cur.eofbit) break;
        }
    }
}

uint64_t fasta_cut_at(const char* data, uint64_t size, uint64_t from) {
    // Line 0 is a header whatever it starts with, so the line after it is sequence even if it starts with '>':
    // a start point must have a previous line that is neither a '>' line nor line 0.
    const char* nl0 = static_cast<const char*>(memchr(data, '\n', size));
    const uint64_t line1 = nl0 ? (uint64_t)(nl0 - data) + 1 : size;
    uint64_t p = from;
    while (p < size) {
        const char* q = static_cast<const char*>(memchr(data + p, '\n', size - p));
        if (!q) break;
        const uint64_t ls = (uint64_t)(q - data) + 1;  // start of the next line
        if (ls >= size) break;
        if (data[ls] == '>') {
            uint64_t ps = (uint64_t)(q - data);  // walk back to the start of the line that ends at q
            while (ps > 0 && data[ps - 1] != '\n') --ps;
            if (data[ps] != '>' && ps >= line1) return ls;
        }
        p = ls;
    }
    return size;
}

std::vector<uint64_t> split_fasta(const char* data, uint64_t size, uint64_t chunk_bytes) {
    std::vector<uint64_t> starts(1, 0);
    if (chunk_bytes == 0) chunk_bytes = 1;
    uint64_t target = chunk_bytes;
    while (target < size) {
        const uint64_t found = fasta_cut_at(data, size, target);
        if (found >= size) break;
        starts.push_back(found);
        target = found + chunk_bytes;
    }
    return starts;
}

void parse_reads(const char* data, uint64_t size, bool fastq, uint32_t k, ReadSet& out) {
    if (out.read_offs.empty()) out.clear();
    ParsedChunk c;
    if (fastq) parse_fastq_image(data, size, c);
    else parse_fasta_chunk(data, 0, size, k, c);
    to_readset(c, out);
}

void parse_reads_parallel(const char* data, uint64_t size, bool fastq, uint32_t k, unsigned threads, uint64_t chunk_bytes, ReadSet& out) {
    if (out.read_offs.empty()) out.clear();
    if (threads <= 1) { parse_reads(data, size, fastq, k, out); return; }
    if (fastq) {
        std::vector<ParsedChunk> chunks;
        parse_fastq_parallel(data, size, threads, chunk_bytes, chunks);
        for (const ParsedChunk& c : chunks) to_readset(c, out);
        return;
    }
    std::vector<uint64_t> starts = split_fasta(data, size, chunk_bytes);
    std::vector<ParsedChunk> chunks(starts.size());
    std::vector<std::thread> ts;
    for (unsigned t = 0; t < threads; ++t) {
        ts.emplace_back([&, t]() {
            for (size_t c = t; c < starts.size(); c += threads) {
                uint64_t e = c + 1 < starts.size() ? starts[c + 1] : size;
                parse_fasta_chunk(data, starts[c], e, k, chunks[c]);
            }
        });
    }
    for (auto& t : ts) t.join();
    for (const ParsedChunk& c : chunks) to_readset(c, out);
}

bool parse_reads_file(const std::string& path, bool fastq, uint32_t k, ReadSet& out, std::string& err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open read file " + path; return false; }
    std::vector<char> buf;
    char tmp[1 << 16];
    size_t got;
    while ((got = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + got);
    fclose(f);
    parse_reads(buf.data(), buf.size(), fastq, k, out);
    return true;
}

}  // namespace bgr
