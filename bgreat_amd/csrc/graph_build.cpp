// graph_build.cpp -- host construction of the device graph blob.  See graph_layout.h for the layout and
// graph_build.h for the contract.  Own code throughout: the reference's index (aligner.cpp:407-534) is
// two boomphf::mphf + two vector<unitigIndices>; here it is one cascade MPHF over the union of both key
// sets plus flat arrays, because only membership and the slot order are observable.
#include "graph_build.h"
#include "anchor_index.h"
#include "host_parallel.h"
#include "file_image.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include "options.h"

namespace bgr {

namespace {

inline uint32_t code_of(char c) { return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : 3u; }  // utils.cpp:117-129

std::atomic<unsigned> g_build_threads{0};

// ORs `len` bases of one strand into the packed store at base position `pos`, 32 bases per word; only the first
// and last word of the run can be shared with a neighbouring run (another thread), so those go through atomics.
// rc == false: codes of s[0..len); rc == true: 3 - code of s[len-1..0] (utils.cpp:52-59: non-ACG -> 'A').
inline void pack_strand(uint64_t* seq, uint64_t pos, const char* s, uint32_t len, bool rc) {
    uint32_t j = 0;
    while (j < len) {
        uint64_t w = (pos + j) >> 5;
        uint32_t o = (uint32_t)((pos + j) & 31), n = std::min<uint32_t>(32 - o, len - j);
        uint64_t x = 0;
        for (uint32_t i = 0; i < n; ++i) {
            uint32_t c = rc ? 3u - code_of(s[len - 1 - (j + i)]) : code_of(s[j + i]);
            x |= (uint64_t)c << (62 - 2 * (o + i));
        }
        if (n == 32) seq[w] = x;
        else __atomic_fetch_or(&seq[w], x, __ATOMIC_RELAXED);
        j += n;
    }
}
inline uint64_t window(const uint64_t* seq, uint64_t pos, uint32_t n) {  // n in 1..32 bases starting at base pos
    uint64_t w = pos >> 5;
    uint32_t s = (uint32_t)(pos & 31) * 2;
    uint64_t x = s ? (seq[w] << s) | (seq[w + 1] >> (64 - s)) : seq[w];
    return x >> (64 - 2 * n);
}
inline uint64_t align256(uint64_t x) { return (x + 255) & ~(uint64_t)255; }

struct KeyTable {
    std::vector<uint32_t> buckets;   // 4 one-byte fingerprints per dword, 0 = empty
    std::vector<uint32_t> who;       // per slot: which key (index into the sorted key list) lives there, BGR_NONE = empty
    std::vector<uint64_t> fallback;  // keys that found no slot, sorted
    uint64_t n_placed = 0;
};

// Two-choice bucketed table over `keys` (sorted, unique), graph_layout.h: a key lives in one of the four slots of
// bucket 1 or bucket 2 of its hash; when both are full a resident is evicted to its other bucket (random walk, at most
// kMaxKicks moves).  Keys are inserted in sorted order by one thread with a fixed pseudo-random sequence, so the table
// -- and with it every index in the blob -- does not depend on the thread count.
void build_key_table(const std::vector<uint64_t>& keys, double slots_per_key, unsigned T, bool evictions, KeyTable& t) {
    const uint64_t n = keys.size();
    const uint32_t nb = (uint32_t)std::max<uint64_t>(1, (uint64_t)std::ceil(slots_per_key * (double)n / 4.0));
    t.buckets.assign(nb, 0);
    std::vector<uint64_t> mix(n);
    parallel_ranges(T, n, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t i = b; i < e; ++i) mix[i] = bgr_mix64(keys[i]);
    });
    std::vector<uint32_t>& who = t.who;
    who.assign((size_t)nb * 4, BGR_NONE);
    uint8_t* fp = reinterpret_cast<uint8_t*>(t.buckets.data());
    auto try_place = [&](uint32_t bucket, uint32_t i) {
        for (uint32_t s = 0; s < 4; ++s) {
            const size_t slot = (size_t)bucket * 4 + s;
            if (!fp[slot]) { fp[slot] = (uint8_t)bgr_tab_fp(mix[i]); who[slot] = i; return true; }
        }
        return false;
    };
    const int kMaxKicks = evictions ? 4000 : 0;
    uint64_t rng = 0x9E3779B97F4A7C15ULL;
    for (uint64_t i0 = 0; i0 < n; ++i0) {
        uint32_t cur = (uint32_t)i0;
        uint32_t from = BGR_NONE;  // the bucket `cur` was just evicted from
        bool done = false;
        for (int kick = 0; kick <= kMaxKicks; ++kick) {
            const uint32_t b1 = bgr_tab_bucket((uint32_t)mix[cur], nb), b2 = bgr_tab_bucket((uint32_t)(mix[cur] >> 32), nb);
            if (try_place(b1, cur) || (b2 != b1 && try_place(b2, cur))) { done = true; break; }
            if (kick == kMaxKicks) break;
            rng = rng * 6364136223846793005ULL + 1442695040888963407ULL;
            uint32_t vb = (rng >> 33) & 1u ? b1 : b2;
            if (vb == from && b1 != b2) vb = vb == b1 ? b2 : b1;  // do not walk straight back
            const size_t slot = (size_t)vb * 4 + ((rng >> 40) & 3u);
            const uint32_t victim = who[slot];
            fp[slot] = (uint8_t)bgr_tab_fp(mix[cur]);
            who[slot] = cur;
            cur = victim;
            from = vb;
        }
        if (!done) t.fallback.push_back(keys[cur]);
    }
    std::sort(t.fallback.begin(), t.fallback.end());
    t.n_placed = n - t.fallback.size();
}

// `near_from` = base offset in seq of the strand on which the unitig begins with the slot's key (or its reverse complement)
inline void fill_slot(BgrSlot* s, uint32_t idf, const BgrUnitigMeta& m, const uint64_t* seq, uint64_t near_from, uint32_t K1) {  // aligner.cpp:481-489: first free of 1..3, else overwrite 4
    int j = 3;
    if (s[0].idf == 0) j = 0;
    else if (s[1].idf == 0) j = 1;
    else if (s[2].idf == 0) j = 2;
    const uint32_t ext = m.len - K1, nb = ext < 32 ? ext : 32;
    uint64_t near = nb ? window(seq, near_from + K1, nb) << (64 - 2 * nb) : 0;  // the bases behind the overlap, first one on top
    s[j].idf = idf;
    s[j].len = m.len;
    s[j].Fw = (uint32_t)(m.F >> 5);
    s[j].Fo_x = (uint32_t)(m.F & 31) | ((uint32_t)((near >> 32) & 15u) << 8);
    s[j].mflags_x = (m.flags & 15u) | (uint32_t)((near >> 32) & 0xFFFFFFF0u);
    s[j].nx0 = m.rec_beg;   // (temporary: the unitig's end KEYS; compact_slots() turns them into the handles of the next halves)
    s[j].nx1 = m.rec_end;
    s[j].near_lo = (uint32_t)near;
}

}  // namespace

uint32_t host_lookup(const BgrBlobHeader* h, const uint8_t* base, uint64_t key) {
    const uint32_t* table = reinterpret_cast<const uint32_t*>(base + h->off_table);
    const BgrKeyEntry* keys = reinterpret_cast<const BgrKeyEntry*>(base + h->off_keys);
    const uint32_t nb = (uint32_t)h->n_buckets;
    const uint64_t m = bgr_mix64(key);
    const uint32_t bk[2] = {bgr_tab_bucket((uint32_t)m, nb), bgr_tab_bucket((uint32_t)(m >> 32), nb)};
    const uint32_t f4 = bgr_tab_fp(m) * 0x01010101u;
    for (int c = 0; c < 2; ++c) {
        if (c == 1 && bk[1] == bk[0]) break;
        for (uint32_t z = bgr_zero_bytes(table[bk[c]] ^ f4); z; z &= z - 1) {
            const uint32_t idx = bk[c] * 4 + ((uint32_t)__builtin_ctz(z) >> 3);
            if (keys[idx].key == key) return idx;
        }
    }
    if (h->n_fallback) {
        const uint64_t* fb = reinterpret_cast<const uint64_t*>(base + h->off_fallback);
        const uint64_t* e = fb + h->n_fallback;
        const uint64_t* it = std::lower_bound(fb, e, key);
        if (it != e && *it == key) return (uint32_t)(4 * h->n_buckets + (it - fb));
    }
    return BGR_NONE;
}

void resolve_device_graph(const BgrBlobHeader* h, const void* basev, BgrDeviceGraph& dg) {
    const uint8_t* base = static_cast<const uint8_t*>(basev);
    memset(&dg, 0, sizeof(dg));
    dg.table = reinterpret_cast<const uint32_t*>(base + h->off_table);
    dg.keys = reinterpret_cast<const BgrKeyEntry*>(base + h->off_keys);
    dg.recs = reinterpret_cast<const BgrSlot*>(base + h->off_recs);
    dg.meta = reinterpret_cast<const BgrUnitigMeta*>(base + h->off_meta);
    dg.seq = reinterpret_cast<const uint64_t*>(base + h->off_seq);
    dg.hdr = reinterpret_cast<const BgrBlobHeader*>(base);
    dg.k = h->k;
    dg.n_buckets = (uint32_t)h->n_buckets;
    dg.flags = (h->has_exc ? BGR_GF_HAS_EXC : 0u) | (h->n_fallback ? BGR_GF_HAS_FALLBACK : 0u);
    dg.table_bytes = (uint32_t)(h->n_buckets * 4);
    dg.bloom = h->bloom_bits ? reinterpret_cast<const uint32_t*>(base + h->off_bloom) : nullptr;
    dg.filter_kind = h->bloom_bits ? h->filter_kind : BGR_FILTER_NONE;
    dg.bloom_mask = 0;
    if (dg.filter_kind == BGR_FILTER_FLAT) dg.bloom_mask = (uint32_t)(h->bloom_bits - 1);
    if (dg.filter_kind == BGR_FILTER_MINIMIZER) { uint32_t lg = 0; while ((512ull << lg) < h->bloom_bits) ++lg; dg.bloom_mask = 32 - lg; }
}

// Everything the kernels later trust about a blob can be checked on its header alone (section extents, level table):
// a truncated or foreign blob -- read from a file, or received from a rank running another version -- must be refused
// here, because the kernels index HBM with these numbers unchecked.
bool validate_blob_header(const BgrBlobHeader* h, uint64_t bytes, std::string& err) {
    if (bytes < sizeof(BgrBlobHeader)) { err = "blob smaller than its header"; return false; }
    if (h->magic != BGR_MAGIC || h->version != BGR_BLOB_VERSION) { err = "not a bgreat graph blob (magic/version)"; return false; }
    if (h->blob_bytes != bytes) { err = "blob size does not match its header"; return false; }
    if (h->k < 2 || h->k > 32) { err = "corrupt blob header"; return false; }
    // off + count * size <= bytes without overflow; sections start behind the header, 256-byte aligned
    auto inside = [&](uint64_t off, uint64_t count, uint64_t size) {
        if (off < sizeof(BgrBlobHeader) || off > bytes || (off & 255u)) return false;
        return count <= (bytes - off) / size;
    };
    if (h->n_keys >= 0x0FFFFFFFull || h->n_unitigs > 0x40000000ull || h->n_buckets == 0 || h->n_buckets >= (1ull << 26)) { err = "corrupt blob header (counts)"; return false; }
    if (h->n_slots >= BGR_HNONE - 8) { err = "corrupt blob header (slots)"; return false; }
    if (!inside(h->off_table, h->n_buckets, 4) || !inside(h->off_keys, h->n_keys, sizeof(BgrKeyEntry)) || !inside(h->off_recs, h->n_slots + 4, sizeof(BgrSlot)) ||
        !inside(h->off_meta, h->n_unitigs + 1, sizeof(BgrUnitigMeta)) || !inside(h->off_seq, h->seq_words, 8)) { err = "blob section outside the blob"; return false; }
    if (h->n_fallback && !inside(h->off_fallback, h->n_fallback, 8)) { err = "blob section outside the blob"; return false; }
    if (h->bloom_bits && ((h->bloom_bits & (h->bloom_bits - 1)) || h->bloom_bits < 64 || !inside(h->off_bloom, h->bloom_bits / 32, 4))) { err = "corrupt blob header (filter)"; return false; }
    if (h->bloom_bits && h->filter_kind == BGR_FILTER_FLAT && h->bloom_bits > (1ull << 32)) { err = "corrupt blob header (flat filter size)"; return false; }
    // (minimizer blocks: at least 2 of them, so that the block shift stays below 32; at most 2^27 = 8 GiB)
    if (h->bloom_bits && h->filter_kind == BGR_FILTER_MINIMIZER && (h->bloom_bits < 1024 || h->bloom_bits > (512ull << 27) || h->k - 1 < BGR_MMX_MIN_K1)) { err = "corrupt blob header (minimizer filter)"; return false; }
    if (h->bloom_bits && h->filter_kind != BGR_FILTER_FLAT && h->filter_kind != BGR_FILTER_MINIMIZER) { err = "corrupt blob header (filter kind)"; return false; }
    if (h->n_keys != 4 * h->n_buckets + h->n_fallback || h->n_placed > 4 * h->n_buckets) { err = "corrupt blob header (key counts)"; return false; }
    if (h->seq_words < 2 || h->total_bases > (h->seq_words - 2) * 32 || h->seq_words * 8 >= (1ull << 32)) { err = "corrupt blob header (sequence store)"; return false; }
    if (h->has_exc) {  // one bit per base, read 64 bits at a time one word past the addressed one
        const uint64_t plane_words = (h->total_bases + 63) / 64 + 2;
        if (!inside(h->off_exc, plane_words, 8) || !inside(h->off_excn, plane_words, 8)) { err = "blob exception planes outside the blob"; return false; }
    }
    if (h->anc_n) {
        if (!inside(h->off_anc_bits, h->anc_words, 8) || !inside(h->off_anc_ranks, h->anc_rank_words, 8) ||
            (h->anc_n_final && !inside(h->off_anc_final, h->anc_n_final, 16)) || !inside(h->off_anc_pos, h->anc_n, 8)) { err = "blob section outside the blob"; return false; }
        for (int i = 0; i < BGR_ANC_LEVELS; ++i) {
            const BgrAncLevel& lv = h->anc_levels[i];
            if (lv.domain == 0 || lv.word_base > h->anc_words || 1 + lv.domain / 64 > h->anc_words - lv.word_base) { err = "corrupt anchors index"; return false; }
            if (lv.rank_base > h->anc_rank_words || (1 + lv.domain / 64 + 7) / 8 > h->anc_rank_words - lv.rank_base) { err = "corrupt anchors index"; return false; }
        }
        if (h->anc_active_levels >= BGR_ANC_LEVELS) { err = "corrupt anchors index"; return false; }
    }
    return true;
}

bool validate_blob(const void* blob, uint64_t bytes, std::string& err) {
    if (bytes < sizeof(BgrBlobHeader)) { err = "blob smaller than its header"; return false; }
    return validate_blob_header(static_cast<const BgrBlobHeader*>(blob), bytes, err);
}

bool ZeroPages::reset(uint64_t words) {
    release();
    if (words == 0) return true;
    void* p = mmap(nullptr, words * 8, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) return false;
    p_ = static_cast<uint64_t*>(p);
    words_ = words;
    return true;
}
void ZeroPages::release() {
    if (p_) munmap(p_, words_ * 8);
    p_ = nullptr;
    words_ = 0;
}

void set_build_threads(unsigned t) { g_build_threads.store(t); }
unsigned build_threads() {
    unsigned t = g_build_threads.load();
    if (t == 0) t = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    return t;
}

bool read_unitig_fasta(const std::string& path, uint32_t k, std::vector<char>& seqs, std::vector<uint64_t>& offs, std::string& err) {
    FileImage img;  // mmap, or read-until-EOF for a FIFO / process substitution
    if (!img.open(path, err)) { err = "unitig file: " + err; return false; }
    seqs.clear();
    offs.assign(1, 0);
    const uint64_t size = img.size;
    if (size == 0) return true;
    const char* d = img.data;
    // aligner.cpp:415-420: two getline per record (header ignored, a missing line reads as empty), stop at the
    // first sequence shorter than k.  Pass 1 finds the record extents, pass 2 copies them in parallel.
    struct Ext { uint64_t b; uint32_t len; };
    std::vector<Ext> ext;
    uint64_t p = 0, total = 0;
    while (p < size) {
        const char* nl = static_cast<const char*>(memchr(d + p, '\n', size - p));
        if (!nl) break;  // header without a sequence line: the sequence reads as "" -> stop
        uint64_t sb = (uint64_t)(nl - d) + 1;
        const char* nl2 = sb < size ? static_cast<const char*>(memchr(d + sb, '\n', size - sb)) : nullptr;
        uint64_t se = nl2 ? (uint64_t)(nl2 - d) : size;
        if (se - sb < k) break;
        if (se - sb > 0xFFFFFFFFull) { err = "unitig longer than 2^32-1 bases"; return false; }
        ext.push_back({sb, (uint32_t)(se - sb)});
        total += se - sb;
        p = se + 1;
    }
    seqs.resize(total);
    offs.resize(ext.size() + 1);
    uint64_t o = 0;
    for (size_t i = 0; i < ext.size(); ++i) { offs[i] = o; o += ext[i].len; }
    offs[ext.size()] = o;
    parallel_ranges(build_threads(), ext.size(), [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t i = b; i < e; ++i) memcpy(seqs.data() + offs[i], d + ext[i].b, ext[i].len);
    });
    return true;
}

// LDS a CU has for key table copies next to 16 waves of sixteen 150-bp reads each (160 KB - 64, 512 B fixed per workgroup, 768 B per wave)
static const double kStageTwice = (163776.0 / 2 - 512 - 16 * 768), kStageOnce = (163776.0 - 512 - 16 * 768);

bool build_graph(uint32_t k, uint64_t n_in, const char* seqs, const uint64_t* offs, double gamma, uint32_t flags, HostGraph& out, std::string& err) {
    if (k < 2 || k > 32) { err = "k must be in [2,32] (kmer is uint64_t, utils.h:27)"; return false; }
    if (gamma != 0.0 && !(gamma >= 1.03 && gamma <= 64.0)) { err = "gamma (key table slots per key) must be in [1.03,64] (0 = choose)"; return false; }
    const uint32_t K1 = k - 1;
    // aligner.cpp:418-420: stop at the first sequence shorter than k
    uint64_t n = 0;
    while (n < n_in && offs[n + 1] - offs[n] >= k) ++n;
    if (n >= BGR_SLOT_ID_MASK) { err = "too many unitigs (limit 2^30-1)"; return false; }
    uint64_t sum = 0, maxlen = 0;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t len = offs[i + 1] - offs[i];
        if (len > 0xFFFFFFFFull) { err = "unitig longer than 2^32-1 bases"; return false; }
        sum += len;
        maxlen = std::max(maxlen, len);
    }
    const uint64_t total = 2 * sum, seq_words = (total + 31) / 32 + 2;
    if (seq_words * 8 >= (1ull << 32)) { err = "graph too large: the packed sequence store must stay below 4 GiB (2^34 bases over both strands)"; return false; }

    const unsigned T = build_threads();
    PhaseTimer tm;
    // ---- unitig extents + key sets (aligner.cpp:422-433), straight from the characters ----------------
    std::vector<BgrUnitigMeta> meta(n + 1);
    memset(meta.data(), 0, meta.size() * sizeof(BgrUnitigMeta));
    {
        uint64_t F = 0;
        for (uint64_t i = 0; i < n; ++i) {
            uint32_t len = (uint32_t)(offs[i + 1] - offs[i]);
            meta[i + 1].F = F;
            meta[i + 1].len = len;
            F += 2ull * len;
        }
    }
    std::vector<uint64_t> begs(n + 1), ends(n + 1);
    std::vector<uint64_t> lr(2 * n);       // per unitig: its two canonical end keys
    std::vector<uint8_t> side(2 * n);      // 0 = goes to the left key set, 1 = right
    std::atomic<bool> any_exc{false};
    parallel_ranges(T, n, [&](uint64_t b, uint64_t e, unsigned) {
        bool mine = false;
        for (uint64_t i = b + 1; i <= e; ++i) {
            const char* s = seqs + offs[i - 1];
            const uint32_t len = meta[i].len;
            uint64_t beg = 0, end = 0;  // str2num of the first / last k-1 characters (utils.cpp:117-129)
            for (uint32_t j = 0; j < K1; ++j) {
                beg = beg << 2 | code_of(s[j]);
                end = end << 2 | code_of(s[len - K1 + j]);
            }
            uint64_t rcBeg = bgr_rcb(beg, K1), rcEnd = bgr_rcb(end, K1);
            begs[i] = beg;
            ends[i] = end;
            if (beg <= rcBeg) { lr[2 * i - 2] = beg; side[2 * i - 2] = 0; } else { lr[2 * i - 2] = rcBeg; side[2 * i - 2] = 1; }
            if (end <= rcEnd) { lr[2 * i - 1] = end; side[2 * i - 1] = 1; } else { lr[2 * i - 1] = rcEnd; side[2 * i - 1] = 0; }
            if (!mine)
                for (uint32_t j = 0; j < len; ++j) {
                    char ch = s[j];
                    if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T') { mine = true; break; }
                }
        }
        if (mine) any_exc.store(true);
    });
    const bool has_exc = any_exc.load();
    std::vector<uint64_t> left, right;
    left.reserve(n);
    right.reserve(n);
    for (uint64_t j = 0; j < 2 * n; ++j) (side[j] ? right : left).push_back(lr[j]);
    {
        unsigned Th = std::max(1u, T / 2);
        std::thread tl([&] { parallel_sort(left, Th); left.erase(std::unique(left.begin(), left.end()), left.end()); });
        parallel_sort(right, Th);
        right.erase(std::unique(right.begin(), right.end()), right.end());
        tl.join();
    }
    std::vector<uint64_t> keys(left.size() + right.size());
    std::merge(left.begin(), left.end(), right.begin(), right.end(), keys.begin());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    if (keys.size() >= 0x0FFFFFFFull) { err = "too many overlap keys (limit 2^28-1)"; return false; }
    tm.lap("keys");

    // gamma 0 = choose.  A table that can be staged in LDS twice per CU (<= ~72 KB: one byte per slot) is built tight, 1.07
    // slots per key (fill 0.935, below the two-choice/four-slot threshold of 0.977, so the eviction walks stay short).  A
    // larger one is probed in L2, where the request rate is the limit: built sparse (1.8 slots per key) few first buckets
    // are full and few lanes need the second probe (find_key, device_common.h; chr1-scale graph, slots per key 1.07 / 1.4 /
    // 1.8 / 2.5: 800 / 900 / 930 / 870 Mreads/s -- beyond 2 the table outgrows the L2).
    // Staging has two break points (capi.hip geometry; sixteen 150-bp reads per wave = 768 B of LDS per wave): a table of at most
    // kStageTwice bytes fits twice per CU next to 16 waves each (32 resident waves), one of at most kStageOnce bytes once (16 waves).
    // A key count just above a break point at 1.07 is built tighter, down to 1.03 slots per key (E. coli-scale graph, 66 k keys:
    // 2 x 12 waves at 1.07 -> 1 881 Mreads/s, 2 x 16 at 1.03 -> 2 030).
    if (gamma == 0.0) {
        const double n = (double)keys.size();
        if (n * 1.07 <= kStageTwice) gamma = 1.07;
        else if (n * 1.03 <= kStageTwice) gamma = std::max(1.03, (kStageTwice - 64) / n);  // (64: the bucket count and the staged copy round up)
        else if (n * 1.07 <= kStageOnce) gamma = 1.07;
        else if (n * 1.03 <= kStageOnce) gamma = std::max(1.03, (kStageOnce - 64) / n);
        else gamma = 1.8;
    }
    KeyTable tab;
    build_key_table(keys, gamma, T, !(flags & BGR_BUILD_NO_EVICTIONS), tab);
    tm.lap("keytable");

    // ---- anchors index of -G (aligner.cpp:434-442,457-462): canonical k-mers j = 0 .. len-k-1 of every unitig ----
    AnchorMphf anc;
    std::vector<uint64_t> anc_first;  // index of each unitig's first anchor in the key sequence
    if (flags & BGR_BUILD_ANCHORS) {
        anc_first.assign(n + 2, 0);
        for (uint64_t i = 1; i <= n; ++i) anc_first[i + 1] = anc_first[i] + (meta[i].len > k ? meta[i].len - k : 0);
        std::vector<uint64_t> akeys(anc_first[n + 1]);
        const uint64_t kmask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
        parallel_ranges(T, n, [&](uint64_t b, uint64_t e, unsigned) {
            for (uint64_t i = b + 1; i <= e; ++i) {
                const char* s = seqs + offs[i - 1];
                const uint32_t len = meta[i].len;
                if (len <= k) continue;
                uint64_t x = 0;
                for (uint32_t j = 0; j < k - 1; ++j) x = x << 2 | code_of(s[j]);
                uint64_t* o = akeys.data() + anc_first[i];
                for (uint32_t j = 0; j + k < len; ++j) {
                    x = (x << 2 | code_of(s[j + k - 1])) & kmask;
                    const uint64_t rc = bgr_rcb(x, k);
                    o[j] = x < rc ? x : rc;
                }
            }
        });
        build_anchor_mphf(akeys, T, anc);
        tm.lap("anchors");
    }

    // ---- blob layout ------------------------------------------------------------------------------
    const uint64_t exc_words = has_exc ? (total + 63) / 64 + 2 : 0;
    BgrBlobHeader h;
    memset(&h, 0, sizeof(h));
    h.magic = BGR_MAGIC;
    h.version = BGR_BLOB_VERSION;
    h.k = k;
    h.n_unitigs = n;
    h.n_buckets = tab.buckets.size();
    h.n_placed = tab.n_placed;
    h.n_fallback = tab.fallback.size();
    h.n_keys = 4 * h.n_buckets + h.n_fallback;
    if (h.n_keys >= 0x0FFFFFFFull) { err = "too many overlap keys (limit 2^28-1 table slots)"; return false; }
    h.seq_words = seq_words;
    h.total_bases = total;
    h.has_exc = has_exc ? 1 : 0;
    h.max_unitig_len = maxlen;
    h.n_left_keys = left.size();
    h.n_right_keys = right.size();
    h.gamma = gamma;
    uint64_t off = align256(4096);
    static_assert(sizeof(BgrBlobHeader) <= 4096, "header must fit its 4 KiB slot");
    h.off_table = off;    off = align256(off + h.n_buckets * 4 + 16);
    h.off_keys = off;     off = align256(off + h.n_keys * sizeof(BgrKeyEntry) + 16);
    h.off_meta = off;     off = align256(off + (n + 1) * sizeof(BgrUnitigMeta));
    h.off_seq = off;      off = align256(off + seq_words * 8);
    if (has_exc) {
        h.off_exc = off;  off = align256(off + exc_words * 8);
        h.off_excn = off; off = align256(off + exc_words * 8);
    }
    h.off_fallback = off; off = align256(off + h.n_fallback * 8 + 8);
    // a table too large for LDS staging gets a filter in front: minimizer-blocked when k-1 >= 20 (24-48 bits per key), else one hash
    // (4-8 bits per key).  BGREAT_BLOOM=0 builds without, =1 the one-hash kind, =2 the minimizer kind whatever the table size (tests)
    const int filter_env = (int)opt("build_filter");  // (bgr_set_option: tests force a filter kind onto small graphs)
    const bool large_table = (double)tab.buckets.size() * 4.0 > kStageTwice;  // (may be probed in memory: always beyond kStageOnce, below it with long reads)
    if (filter_env != 0 && !keys.empty() && (large_table || filter_env == 2)) {
        const bool minimizer = k - 1 >= BGR_MMX_MIN_K1 && filter_env != 1;
        uint64_t bits = minimizer ? 1024 : 64;
        while (bits < (minimizer ? 24 : 4) * keys.size()) bits <<= 1;
        if (minimizer && bits > (512ull << 27)) bits = 512ull << 27;
        h.filter_kind = minimizer ? BGR_FILTER_MINIMIZER : BGR_FILTER_FLAT;
        h.bloom_bits = bits;
        h.off_bloom = off; off = align256(off + bits / 8 + 16);
    }
    if (flags & BGR_BUILD_ANCHORS) {
        h.anc_n = anc.n;
        h.anc_last_rank = anc.last_rank;
        h.anc_n_final = anc.final_kv.size() / 2;
        h.anc_words = anc.bits.size();
        h.anc_rank_words = anc.ranks.size();
        h.anc_active_levels = anc.active_levels;
        memcpy(h.anc_levels, anc.levels, sizeof(h.anc_levels));
        h.off_anc_bits = off;  off = align256(off + h.anc_words * 8 + 8);
        h.off_anc_ranks = off; off = align256(off + h.anc_rank_words * 8 + 8);
        h.off_anc_final = off; off = align256(off + h.anc_n_final * 16 + 16);
        h.off_anc_pos = off;   off = align256(off + h.anc_n * 8 + 8);
    }
    // the compact slots come last: their number is known once the records are filled (at most two per unitig: one per end); the
    // blob is allocated for that bound (untouched zero pages cost nothing) and blob_bytes set to what is used
    h.off_recs = off;
    const uint64_t slots_bound = 2 * n + 4;
    if (slots_bound >= BGR_HNONE - 8) { err = "too many unitigs for the slot handles (limit 2^27-8)"; return false; }
    off = align256(off + slots_bound * sizeof(BgrSlot));
    h.blob_bytes = off;

    if (!out.blob.reset(off / 8)) { err = "out of memory for the graph blob"; return false; }  // zero pages, touched below in parallel
    uint8_t* base = reinterpret_cast<uint8_t*>(out.blob.data());
    memcpy(base, &h, sizeof(h));
    memcpy(base + h.off_table, tab.buckets.data(), tab.buckets.size() * 4);
    if (h.n_fallback) memcpy(base + h.off_fallback, tab.fallback.data(), h.n_fallback * 8);
    if (h.bloom_bits) {
        uint32_t* bl = reinterpret_cast<uint32_t*>(base + h.off_bloom);
        const uint32_t mask = (uint32_t)(h.bloom_bits - 1);
        uint32_t lg = 0;
        while ((512ull << lg) < h.bloom_bits) ++lg;
        const bool minimizer = h.filter_kind == BGR_FILTER_MINIMIZER;
        parallel_ranges(T, keys.size(), [&](uint64_t b, uint64_t e, unsigned) {
            for (uint64_t i = b; i < e; ++i) {
                const uint64_t m = bgr_mix64(keys[i]);
                if (minimizer) {
                    const uint64_t w = ((uint64_t)bgr_mmx_block(bgr_mmx_of_key(keys[i], k - 1), 32 - lg) << 4) + bgr_mmx_word(m);
                    __atomic_fetch_or(&bl[w], bgr_mmx_bits(m), __ATOMIC_RELAXED);
                } else {
                    const uint32_t bit = bgr_bloom_bit(m, mask);
                    __atomic_fetch_or(&bl[bit >> 5], 1u << (bit & 31), __ATOMIC_RELAXED);
                }
            }
        });
    }

    // ---- pack both strands into the blob; non-ACGT exceptions of the forward strand -----------------
    uint64_t* seq = reinterpret_cast<uint64_t*>(base + h.off_seq);
    parallel_ranges(T, n, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t i = b + 1; i <= e; ++i) {
            const char* s = seqs + offs[i - 1];
            pack_strand(seq, meta[i].F, s, meta[i].len, false);
            pack_strand(seq, meta[i].F + meta[i].len, s, meta[i].len, true);
        }
    });
    if (has_exc) {  // rare: bit set = this forward-strand base is not A/C/G/T (exc); it is 'N' (excn)
        uint64_t* exc = reinterpret_cast<uint64_t*>(base + h.off_exc);
        uint64_t* excn = reinterpret_cast<uint64_t*>(base + h.off_excn);
        for (uint64_t i = 1; i <= n; ++i) {
            const char* s = seqs + offs[i - 1];
            uint64_t F = meta[i].F;
            for (uint32_t j = 0; j < meta[i].len; ++j) {
                char ch = s[j];
                if (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T') continue;
                exc[(F + j) >> 6] |= 1ULL << (63 - ((F + j) & 63));
                if (ch == 'N') excn[(F + j) >> 6] |= 1ULL << (63 - ((F + j) & 63));
            }
        }
    }
    tm.lap("pack");

    // keys by table slot (~0 = empty slot), then the fallback list's; every key must now be found where it was put
    BgrKeyEntry* kout = reinterpret_cast<BgrKeyEntry*>(base + h.off_keys);
    parallel_ranges(T, 4 * h.n_buckets, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t j = b; j < e; ++j) { kout[j].key = tab.who[j] == BGR_NONE ? BGR_EMPTY_KEY : keys[tab.who[j]]; kout[j].hL = kout[j].hR = BGR_HNONE; }
    });
    for (uint64_t j = 0; j < h.n_fallback; ++j) { kout[4 * h.n_buckets + j].key = tab.fallback[j]; kout[4 * h.n_buckets + j].hL = kout[4 * h.n_buckets + j].hR = BGR_HNONE; }
    const BgrBlobHeader* hp = reinterpret_cast<const BgrBlobHeader*>(base);
    std::atomic<bool> bad{false};
    parallel_ranges(T, keys.size(), [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t j = b; j < e; ++j) {
            const uint32_t idx = host_lookup(hp, base, keys[j]);
            if (idx == BGR_NONE || idx >= h.n_keys || kout[idx].key != keys[j]) { bad.store(true); return; }
            if (idx < 4 * h.n_buckets) {  // what find_key<LAZY2> relies on: a key sits in its bucket 2 only when its bucket 1 is full
                const uint32_t b1 = bgr_tab_bucket((uint32_t)bgr_mix64(keys[j]), (uint32_t)h.n_buckets);
                if (idx / 4 != b1 && bgr_zero_bytes(tab.buckets[b1]) != 0) { bad.store(true); return; }
            }
        }
    });
    if (bad.load()) { err = "internal: key table inconsistent"; return false; }
    tm.lap("keycheck");

    // ---- slot fill in unitig order (aligner.cpp:466-533) + orientation bits ---------------------
    // Pass 1 (parallel over unitigs): record indices and flags.  Pass 2: the fill order within a record is the
    // unitig order, so every thread walks all unitigs in order and fills only the records of its own index range.
    // the records are filled the round-2 way -- 8 slots per key (left-table slots 0..3, right-table slots 4..7), in a scratch buffer
    // -- and compacted into the blob afterwards (compact_slots below)
    ZeroPages dense;
    if (!dense.reset(h.n_keys * 8 * sizeof(BgrSlot) / 8 + 8)) { err = "out of memory for the record scratch"; return false; }
    BgrSlot* recs = reinterpret_cast<BgrSlot*>(dense.data());
    BgrUnitigMeta* mout = reinterpret_cast<BgrUnitigMeta*>(base + h.off_meta);
    parallel_ranges(T, n, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t i = b + 1; i <= e; ++i) {
            uint64_t beg = begs[i], rcBeg = bgr_rcb(beg, K1), end = ends[i], rcEnd = bgr_rcb(end, K1);
            BgrUnitigMeta m = meta[i];
            m.rec_beg = host_lookup(hp, base, std::min(beg, rcBeg));
            m.rec_end = host_lookup(hp, base, std::min(end, rcEnd));
            m.flags = (beg <= rcBeg ? BGR_META_CANON_BEG : 0) | (end <= rcEnd ? BGR_META_CANON_END : 0) |
                      (rcBeg <= beg ? BGR_META_CANON_RCBEG : 0) | (rcEnd <= end ? BGR_META_CANON_RCEND : 0);
            mout[i] = m;
        }
    });
    tm.lap("lookup");
    const uint64_t nk = h.n_keys;
    parallel_ranges(T, T, [&](uint64_t tb, uint64_t te, unsigned) {
        const uint64_t lo = nk * tb / T, hi = nk * te / T;
        for (uint64_t i = 1; i <= n; ++i) {
            const BgrUnitigMeta& m = mout[i];
            const bool mb = m.rec_beg >= lo && m.rec_beg < hi, me = m.rec_end >= lo && m.rec_end < hi;
            if (!mb && !me) continue;
            uint64_t beg = begs[i], rcBeg = bgr_rcb(beg, K1), end = ends[i], rcEnd = bgr_rcb(end, K1);
            uint32_t id = (uint32_t)i;
            uint32_t ib = m.rec_beg, ie = m.rec_end;
            // left-table slot of key x : F0 = (beg == x), F1 = (end == rc(x)); right-table slot of key y: F0 = (end == y), F1 = (beg == rc(y))
            if (mb) {
                if (beg <= rcBeg) fill_slot(recs + (size_t)ib * 8, id | BGR_SLOT_F0 /* beg == key */ | (end == rcBeg ? BGR_SLOT_F1 : 0), m, seq, m.F, K1);  // forward begins with the key
                else fill_slot(recs + (size_t)ib * 8 + 4, id | (end == rcBeg ? BGR_SLOT_F0 : 0) | BGR_SLOT_F1 /* beg == rc(key) */, m, seq, m.F, K1);  // forward begins with rc(key)
            }
            if (me) {
                if (end <= rcEnd) fill_slot(recs + (size_t)ie * 8 + 4, id | BGR_SLOT_F0 /* end == key */ | (beg == rcEnd ? BGR_SLOT_F1 : 0), m, seq, m.F + m.len, K1);  // the reverse strand begins with rc(key)
                else fill_slot(recs + (size_t)ie * 8, id | (beg == rcEnd ? BGR_SLOT_F0 : 0) | BGR_SLOT_F1 /* end == rc(key) */, m, seq, m.F + m.len, K1);  // the reverse strand begins with the key
            }
        }
    });
    {   // how branchy the graph is: filled slots per non-empty half record (decides the exhaustive search formulation)
        std::vector<uint64_t> filled(T, 0), halves(T, 0);
        parallel_ranges(T, 2 * nk, [&](uint64_t b, uint64_t e, unsigned t) {
            uint64_t f = 0, hh = 0;
            for (uint64_t j = b; j < e; ++j) {
                const BgrSlot* sl4 = recs + j * 4;
                if (!sl4[0].idf) continue;
                ++hh;
                for (int q = 0; q < 4; ++q) f += sl4[q].idf != 0;
            }
            filled[t] = f;
            halves[t] = hh;
        });
        uint64_t f = 0, hh = 0;
        for (unsigned t = 0; t < T; ++t) { f += filled[t]; hh += halves[t]; }
        BgrBlobHeader* hw = reinterpret_cast<BgrBlobHeader*>(base);
        hw->slot_fill_x100 = hh ? (uint32_t)(100 * f / hh) : 100;
        if (opt("timing")) fprintf(stderr, "[build] slot fill %.2f per non-empty half record\n", hw->slot_fill_x100 / 100.0);
    }
    {   // ---- compaction: the filled slots of every half next to each other, handles in the key entries and in the slots ----
        const uint64_t nh = 2 * nk;
        std::vector<uint32_t> hoff(nh + 1);
        parallel_ranges(T, nh, [&](uint64_t b, uint64_t e, unsigned) {
            for (uint64_t j = b; j < e; ++j) {  // the reference's nested ifs stop at the first empty slot (aligner.cpp:160-203)
                const BgrSlot* sl4 = recs + j * 4;
                uint32_t c = 0;
                while (c < 4 && sl4[c].idf) ++c;
                hoff[j + 1] = c;
            }
        });
        hoff[0] = 0;
        for (uint64_t j = 0; j < nh; ++j) hoff[j + 1] += hoff[j];
        const uint64_t n_slots = hoff[nh];
        if (n_slots > slots_bound - 4) { err = "internal: more slots than unitig ends"; return false; }
        auto handle = [&](uint64_t half) { return hoff[half + 1] == hoff[half] ? (uint32_t)BGR_HNONE : hoff[half]; };
        parallel_ranges(T, nk, [&](uint64_t b, uint64_t e, unsigned) {
            for (uint64_t j = b; j < e; ++j) { kout[j].hL = handle(2 * j); kout[j].hR = handle(2 * j + 1); }
        });
        BgrSlot* cs = reinterpret_cast<BgrSlot*>(base + h.off_recs);
        parallel_ranges(T, nh, [&](uint64_t b, uint64_t e, unsigned) {
            for (uint64_t j = b; j < e; ++j) {
                const uint32_t cnt = hoff[j + 1] - hoff[j];
                const uint32_t side = (uint32_t)(j & 1);  // 0 = the key's left-table slots, 1 = its right-table slots
                for (uint32_t q = 0; q < cnt; ++q) {
                    BgrSlot sl = recs[j * 4 + q];
                    const uint32_t rec_beg = sl.nx0, rec_end = sl.nx1, mflags = sl.mflags_x & 15u;
                    // Where the walk goes on behind this unitig, for the two ways the slot can be reached.  A query for key x on side
                    // `side` that is canonical (c = 1) or not (c = 0): getBegin(bin) reads the left table when bin <= rc(bin), else the
                    // right one; getEnd(bin) the other way round (aligner.cpp:147-267) -- so the walk goes LEFT exactly when c == side.
                    // The unitig's orientation is bit F0 (c = 1) / F1 (c = 0); its far end in walking direction names the next key
                    // (forward unitig walking left: its beg key; ...), the canonical flag of that far-end (k-1)-mer the side of the NEXT
                    // query: again left walk <-> side == canonical.
                    uint32_t nx[2];
                    for (uint32_t c = 0; c < 2; ++c) {
                        const bool left = c == side;
                        const bool fwd = (sl.idf & (c ? BGR_SLOT_F0 : BGR_SLOT_F1)) != 0;
                        const uint32_t nrec = fwd == left ? rec_beg : rec_end;
                        const uint32_t cbit = left ? (fwd ? BGR_META_CANON_BEG : BGR_META_CANON_RCEND) : (fwd ? BGR_META_CANON_END : BGR_META_CANON_RCBEG);
                        const bool cn = (mflags & cbit) != 0;
                        const bool next_right_table = cn == left;
                        const uint32_t hd = nrec < nk ? handle(2 * (uint64_t)nrec + (next_right_table ? 1 : 0)) : (uint32_t)BGR_HNONE;
                        nx[c] = hd | (cn ? BGR_H_CANON : 0u);
                    }
                    sl.nx0 = nx[1];
                    sl.nx1 = nx[0];
                    if (q + 1 == cnt) sl.Fo_x |= BGR_SLOT_LAST;
                    cs[hoff[j] + q] = sl;
                }
            }
        });
        BgrBlobHeader* hw = reinterpret_cast<BgrBlobHeader*>(base);
        hw->n_slots = n_slots;
        hw->blob_bytes = align256(h.off_recs + (n_slots + 4) * sizeof(BgrSlot));
        h.n_slots = n_slots;
        h.blob_bytes = hw->blob_bytes;
        dense.release();
    }
    tm.lap("slots");

    if (h.anc_n) {  // aligner.cpp:465-476: anchorsPosition[lookup(canon)] = {i, j} in unitig order, i.e. the LAST one wins
        memcpy(base + h.off_anc_bits, anc.bits.data(), anc.bits.size() * 8);
        memcpy(base + h.off_anc_ranks, anc.ranks.data(), anc.ranks.size() * 8);
        if (!anc.final_kv.empty()) memcpy(base + h.off_anc_final, anc.final_kv.data(), anc.final_kv.size() * 8);
        uint64_t* apos = reinterpret_cast<uint64_t*>(base + h.off_anc_pos);
        const uint64_t kmask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
        std::atomic<bool> oob{false};
        parallel_ranges(T, n, [&](uint64_t b, uint64_t e, unsigned) {
            for (uint64_t i = b + 1; i <= e; ++i) {
                const char* s = seqs + offs[i - 1];
                const uint32_t len = meta[i].len;
                if (len <= k) continue;
                uint64_t x = 0;
                for (uint32_t j = 0; j < k - 1; ++j) x = x << 2 | code_of(s[j]);
                for (uint32_t j = 0; j + k < len; ++j) {
                    x = (x << 2 | code_of(s[j + k - 1])) & kmask;
                    const uint64_t rc = bgr_rcb(x, k);
                    const uint64_t idx = anchor_lookup(hp, base, x < rc ? x : rc);
                    if (idx >= h.anc_n) { oob.store(true); return; }
                    const uint64_t v = i << 32 | j;  // later (i, j) is larger: "last wins" == maximum
                    uint64_t cur = __atomic_load_n(&apos[idx], __ATOMIC_RELAXED);
                    while (cur < v && !__atomic_compare_exchange_n(&apos[idx], &cur, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
                }
            }
        });
        if (oob.load()) { err = "internal: anchors index out of range"; return false; }
        tm.lap("anc_pos");
    }
    return true;
}

}  // namespace bgr
