// graph_build.cpp -- host construction of the device graph blob.  See graph_layout.h for the layout and
// graph_build.h for the contract.  Own code throughout: the reference's index (aligner.cpp:407-534) is
// two boomphf::mphf + two vector<unitigIndices>; here it is one cascade MPHF over the union of both key
// sets plus flat arrays, because only membership and the slot order are observable.
#include "graph_build.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>

namespace bgr {

namespace {

inline uint32_t code_of(char c) { return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : 3u; }  // utils.cpp:117-129

inline void put_base(uint64_t* seq, uint64_t pos, uint32_t code) {
    seq[pos >> 5] |= (uint64_t)code << (62 - 2 * (pos & 31));
}
inline uint64_t window(const uint64_t* seq, uint64_t pos, uint32_t n) {  // n in 1..32 bases starting at base pos
    uint64_t w = pos >> 5;
    uint32_t s = (uint32_t)(pos & 31) * 2;
    uint64_t x = s ? (seq[w] << s) | (seq[w + 1] >> (64 - s)) : seq[w];
    return x >> (64 - 2 * n);
}
inline uint64_t align256(uint64_t x) { return (x + 255) & ~(uint64_t)255; }

struct Cascade {
    std::vector<BgrLevel> levels;
    std::vector<uint32_t> units;  // 4 u32 per unit
    std::vector<uint64_t> fallback;
    uint64_t n_placed = 0;
};

// BBHash-style cascade over `keys` (sorted, unique) with 2-bit position states (graph_layout.h).  Level l
// places every remaining key that is alone on its position (state 1); positions hit by several keys get
// state 3 and those keys move on.  What is left after the last level (or once only a handful remain) goes
// to a sorted fallback list searched by bisection.
void build_cascade(const std::vector<uint64_t>& keys, double gamma, Cascade& c) {
    struct Rem { uint64_t key; uint32_t h, hb; };
    std::vector<Rem> rem(keys.size()), next;
    for (size_t i = 0; i < keys.size(); ++i) {
        uint64_t m = bgr_mix64(keys[i]);
        rem[i] = {keys[i], (uint32_t)m, (uint32_t)(m >> 32) | 1u};
    }
    uint32_t base = 0;
    for (int l = 0; l < BGR_MAX_LEVELS && !rem.empty(); ++l) {
        if (l > 0 && rem.size() <= 4) break;  // a handful left: cheaper in the fallback list than more levels
        uint64_t want = (uint64_t)std::ceil(gamma * (double)rem.size() / BGR_UNIT_POS);
        uint32_t nu = (uint32_t)std::max<uint64_t>(1, want);
        size_t ubase = c.units.size();
        c.units.resize(ubase + (size_t)nu * 4, 0);
        uint32_t* U = c.units.data() + ubase;
        for (const Rem& r : rem) {  // 0 -> 1 -> 3
            uint32_t u = bgr_level_unit(r.h, nu), p = bgr_level_pos(r.h);
            uint32_t& w = U[(size_t)u * 4 + (p >> 4)];
            uint32_t sh = 2 * (p & 15), st = (w >> sh) & 3u;
            w |= (st == 0 ? 1u : 3u) << sh;
        }
        next.clear();
        for (const Rem& r : rem) {
            uint32_t u = bgr_level_unit(r.h, nu), p = bgr_level_pos(r.h);
            uint32_t st = (U[(size_t)u * 4 + (p >> 4)] >> (2 * (p & 15))) & 3u;
            if (st == 3u) next.push_back({r.key, r.h + r.hb, r.hb}); else ++c.n_placed;
        }
        c.levels.push_back({nu, base});
        base += nu;
        rem.swap(next);
    }
    for (const Rem& r : rem) c.fallback.push_back(r.key);
    std::sort(c.fallback.begin(), c.fallback.end());
    // rank of every unit = placed keys in all units before it (over all levels)
    uint32_t run = 0;
    for (size_t u = 0; u < c.units.size() / 4; ++u) {
        c.units[u * 4 + 3] = run;
        for (int w = 0; w < 3; ++w) run += __builtin_popcount(bgr_unique_mask(c.units[u * 4 + w]));
    }
}

inline void fill_slot(BgrSlot* s, uint32_t idf, const BgrUnitigMeta& m) {  // aligner.cpp:481-489: first free of 1..3, else overwrite 4
    int j = 3;
    if (s[0].idf == 0) j = 0;
    else if (s[1].idf == 0) j = 1;
    else if (s[2].idf == 0) j = 2;
    s[j].idf = idf;
    s[j].len = m.len;
    s[j].Fw = (uint32_t)(m.F >> 5);
    s[j].Fo = (uint32_t)(m.F & 31);
}

}  // namespace

uint32_t host_lookup(const BgrBlobHeader* h, const uint8_t* base, uint64_t key) {
    const uint32_t* units = reinterpret_cast<const uint32_t*>(base + h->off_units);
    uint64_t m = bgr_mix64(key);
    uint32_t hl = (uint32_t)m, hb = (uint32_t)(m >> 32) | 1u;
    for (uint32_t l = 0; l < h->n_levels; ++l, hl += hb) {
        uint32_t u = h->levels[l].base + bgr_level_unit(hl, h->levels[l].units), p = bgr_level_pos(hl);
        const uint32_t* q = units + (size_t)u * 4;
        uint32_t sh = 2 * (p & 15), st = (q[p >> 4] >> sh) & 3u;
        if (st == 0) return BGR_NONE;  // no key hashes here
        if (st == 1) {
            uint32_t r = q[3];
            for (uint32_t w = 0; w < (p >> 4); ++w) r += __builtin_popcount(bgr_unique_mask(q[w]));
            r += __builtin_popcount(bgr_unique_mask(q[p >> 4]) & ((1u << sh) - 1u));
            return r;
        }
    }
    if (h->n_fallback) {
        const uint64_t* fb = reinterpret_cast<const uint64_t*>(base + h->off_fallback);
        const uint64_t* e = fb + h->n_fallback;
        const uint64_t* it = std::lower_bound(fb, e, key);
        if (it != e && *it == key) return (uint32_t)(h->n_placed + (it - fb));
    }
    return BGR_NONE;
}

void resolve_device_graph(const BgrBlobHeader* h, const void* basev, BgrDeviceGraph& dg) {
    const uint8_t* base = static_cast<const uint8_t*>(basev);
    memset(&dg, 0, sizeof(dg));
    dg.units = reinterpret_cast<const uint32_t*>(base + h->off_units);
    dg.keys = reinterpret_cast<const uint64_t*>(base + h->off_keys);
    dg.recs = reinterpret_cast<const BgrSlot*>(base + h->off_recs);
    dg.meta = reinterpret_cast<const BgrUnitigMeta*>(base + h->off_meta);
    dg.seq = reinterpret_cast<const uint64_t*>(base + h->off_seq);
    dg.hdr = reinterpret_cast<const BgrBlobHeader*>(base);
    dg.k = h->k;
    dg.n_levels = h->n_levels;
    dg.flags = (h->has_exc ? BGR_GF_HAS_EXC : 0u) | (h->n_fallback ? BGR_GF_HAS_FALLBACK : 0u);
    dg.units_bytes = (uint32_t)(h->n_units * 16);
}

bool validate_blob(const void* blob, uint64_t bytes, std::string& err) {
    if (bytes < sizeof(BgrBlobHeader)) { err = "blob smaller than its header"; return false; }
    const BgrBlobHeader* h = static_cast<const BgrBlobHeader*>(blob);
    if (h->magic != BGR_MAGIC || h->version != BGR_BLOB_VERSION) { err = "not a bgreat graph blob (magic/version)"; return false; }
    if (h->blob_bytes != bytes) { err = "blob size does not match its header"; return false; }
    if (h->n_levels > BGR_MAX_LEVELS || h->k < 2 || h->k > 32) { err = "corrupt blob header"; return false; }
    uint64_t ends[] = {h->off_units + h->n_units * 16, h->off_keys + h->n_keys * 8, h->off_recs + h->n_keys * 128,
                       h->off_meta + (h->n_unitigs + 1) * sizeof(BgrUnitigMeta), h->off_seq + h->seq_words * 8,
                       h->off_fallback + h->n_fallback * 8};
    for (uint64_t e : ends) if (e > bytes) { err = "blob section outside the blob"; return false; }
    return true;
}

bool read_unitig_fasta(const std::string& path, uint32_t k, std::vector<char>& seqs, std::vector<uint64_t>& offs, std::string& err) {
    std::ifstream in(path, std::ios::binary);
    if (!in) { err = "cannot open unitig file " + path; return false; }
    seqs.clear();
    offs.assign(1, 0);
    std::string line;
    while (!in.eof()) {  // aligner.cpp:415-420
        std::getline(in, line);
        std::getline(in, line);
        if (line.size() < k) break;
        seqs.insert(seqs.end(), line.begin(), line.end());
        offs.push_back(seqs.size());
    }
    return true;
}

bool build_graph(uint32_t k, uint64_t n_in, const char* seqs, const uint64_t* offs, double gamma, HostGraph& out, std::string& err) {
    if (k < 2 || k > 32) { err = "k must be in [2,32] (kmer is uint64_t, utils.h:27)"; return false; }
    if (!(gamma >= 0.5 && gamma <= 64.0)) { err = "gamma must be in [0.5,64]"; return false; }
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

    // ---- pack both strands; collect non-ACGT exceptions of the forward strand -------------------
    std::vector<uint64_t> seq(seq_words, 0), exc, excn;
    std::vector<BgrUnitigMeta> meta(n + 1);
    memset(meta.data(), 0, meta.size() * sizeof(BgrUnitigMeta));
    bool has_exc = false;
    uint64_t F = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const char* s = seqs + offs[i];
        uint32_t len = (uint32_t)(offs[i + 1] - offs[i]);
        for (uint32_t j = 0; j < len; ++j) {
            char ch = s[j];
            uint32_t c = code_of(ch);
            put_base(seq.data(), F + j, c);
            put_base(seq.data(), F + len + (len - 1 - j), 3 - c);  // utils.cpp:52-59: non-ACG -> 'A' == 3 - 3
            if (c == 3 && ch != 'T') {
                if (!has_exc) { exc.assign((total + 63) / 64 + 2, 0); excn.assign((total + 63) / 64 + 2, 0); has_exc = true; }
                exc[(F + j) >> 6] |= 1ULL << (63 - ((F + j) & 63));
                if (ch == 'N') excn[(F + j) >> 6] |= 1ULL << (63 - ((F + j) & 63));
            }
        }
        meta[i + 1].F = F;
        meta[i + 1].len = len;
        F += 2ull * len;
    }

    // ---- key sets (aligner.cpp:422-433) -----------------------------------------------------------
    std::vector<uint64_t> left, right;
    left.reserve(n);
    right.reserve(n);
    std::vector<uint64_t> begs(n + 1), ends(n + 1);
    for (uint64_t i = 1; i <= n; ++i) {
        uint64_t beg = window(seq.data(), meta[i].F, K1), rcBeg = bgr_rcb(beg, K1);
        uint64_t end = window(seq.data(), meta[i].F + meta[i].len - K1, K1), rcEnd = bgr_rcb(end, K1);
        begs[i] = beg;
        ends[i] = end;
        if (beg <= rcBeg) left.push_back(beg); else right.push_back(rcBeg);
        if (end <= rcEnd) right.push_back(end); else left.push_back(rcEnd);
    }
    std::sort(left.begin(), left.end());
    left.erase(std::unique(left.begin(), left.end()), left.end());
    std::sort(right.begin(), right.end());
    right.erase(std::unique(right.begin(), right.end()), right.end());
    std::vector<uint64_t> keys(left.size() + right.size());
    std::merge(left.begin(), left.end(), right.begin(), right.end(), keys.begin());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    if (keys.size() >= 0x3FFFFFFFull) { err = "too many overlap keys (limit 2^30-1)"; return false; }

    Cascade cas;
    build_cascade(keys, gamma, cas);

    // ---- blob assembly ----------------------------------------------------------------------------
    BgrBlobHeader h;
    memset(&h, 0, sizeof(h));
    h.magic = BGR_MAGIC;
    h.version = BGR_BLOB_VERSION;
    h.k = k;
    h.n_unitigs = n;
    h.n_keys = keys.size();
    h.n_placed = cas.n_placed;
    h.n_fallback = cas.fallback.size();
    h.seq_words = seq_words;
    h.total_bases = total;
    h.n_units = cas.units.size() / 4;
    h.n_levels = (uint32_t)cas.levels.size();
    h.has_exc = has_exc ? 1 : 0;
    h.max_unitig_len = maxlen;
    h.n_left_keys = left.size();
    h.n_right_keys = right.size();
    h.gamma = gamma;
    for (size_t l = 0; l < cas.levels.size(); ++l) h.levels[l] = cas.levels[l];
    uint64_t off = align256(4096);
    static_assert(sizeof(BgrBlobHeader) <= 4096, "header must fit its 4 KiB slot");
    h.off_units = off;    off = align256(off + h.n_units * 16 + 16);
    h.off_keys = off;     off = align256(off + h.n_keys * 8 + 8);
    h.off_recs = off;     off = align256(off + h.n_keys * 128 + 128);
    h.off_meta = off;     off = align256(off + (n + 1) * sizeof(BgrUnitigMeta));
    h.off_seq = off;      off = align256(off + seq_words * 8);
    if (has_exc) {
        h.off_exc = off;  off = align256(off + exc.size() * 8);
        h.off_excn = off; off = align256(off + excn.size() * 8);
    }
    h.off_fallback = off; off = align256(off + h.n_fallback * 8 + 8);
    h.blob_bytes = off;

    out.blob.assign(off / 8, 0);
    uint8_t* base = reinterpret_cast<uint8_t*>(out.blob.data());
    memcpy(base, &h, sizeof(h));
    memcpy(base + h.off_units, cas.units.data(), cas.units.size() * 4);
    memcpy(base + h.off_seq, seq.data(), seq_words * 8);
    if (has_exc) {
        memcpy(base + h.off_exc, exc.data(), exc.size() * 8);
        memcpy(base + h.off_excn, excn.data(), excn.size() * 8);
    }
    if (h.n_fallback) memcpy(base + h.off_fallback, cas.fallback.data(), h.n_fallback * 8);

    // keys by MPHF index (also proves the hash is a bijection onto [0, n_keys))
    uint64_t* kout = reinterpret_cast<uint64_t*>(base + h.off_keys);
    std::vector<uint8_t> taken(keys.size(), 0);
    const BgrBlobHeader* hp = reinterpret_cast<const BgrBlobHeader*>(base);
    for (uint64_t key : keys) {
        uint32_t idx = host_lookup(hp, base, key);
        if (idx == BGR_NONE || idx >= keys.size() || taken[idx]) { err = "internal: MPHF is not a bijection"; return false; }
        taken[idx] = 1;
        kout[idx] = key;
    }

    // ---- slot fill in unitig order (aligner.cpp:466-533) + orientation bits ---------------------
    BgrSlot* recs = reinterpret_cast<BgrSlot*>(base + h.off_recs);
    BgrUnitigMeta* mout = reinterpret_cast<BgrUnitigMeta*>(base + h.off_meta);
    for (uint64_t i = 1; i <= n; ++i) {
        uint64_t beg = begs[i], rcBeg = bgr_rcb(beg, K1), end = ends[i], rcEnd = bgr_rcb(end, K1);
        uint32_t id = (uint32_t)i;
        uint32_t ib = host_lookup(hp, base, std::min(beg, rcBeg)), ie = host_lookup(hp, base, std::min(end, rcEnd));
        // left-table slot of key x : F0 = (beg == x), F1 = (end == rc(x)); right-table slot of key y: F0 = (end == y), F1 = (beg == rc(y))
        if (beg <= rcBeg) fill_slot(recs + (size_t)ib * 8, id | BGR_SLOT_F0 /* beg == key */ | (end == rcBeg ? BGR_SLOT_F1 : 0), meta[i]);
        else fill_slot(recs + (size_t)ib * 8 + 4, id | (end == rcBeg ? BGR_SLOT_F0 : 0) | BGR_SLOT_F1 /* beg == rc(key) */, meta[i]);
        if (end <= rcEnd) fill_slot(recs + (size_t)ie * 8 + 4, id | BGR_SLOT_F0 /* end == key */ | (beg == rcEnd ? BGR_SLOT_F1 : 0), meta[i]);
        else fill_slot(recs + (size_t)ie * 8, id | (beg == rcEnd ? BGR_SLOT_F0 : 0) | BGR_SLOT_F1 /* end == rc(key) */, meta[i]);
        BgrUnitigMeta m = meta[i];
        m.rec_beg = ib;
        m.rec_end = ie;
        m.flags = (beg <= rcBeg ? BGR_META_CANON_BEG : 0) | (end <= rcEnd ? BGR_META_CANON_END : 0) |
                  (rcBeg <= beg ? BGR_META_CANON_RCBEG : 0) | (rcEnd <= end ? BGR_META_CANON_RCEND : 0);
        mout[i] = m;
    }
    return true;
}

}  // namespace bgr
