// anchor_index.cpp -- see anchor_index.h.  Own code: BooPHF's published construction (level cascade with a collision
// bit array per level, rank samples every 512 bits, exact map for what is left after 24 levels) restated so that
// the resulting structure is the one boomphf::mphf builds for the same key sequence (BooPHF.h line references inline).
#include "anchor_index.h"

#include <algorithm>
#include <cmath>
#include <cstring>

#include "host_parallel.h"

namespace bgr {

namespace {

struct Rem { uint64_t key, s0, s1, h; };

inline bool test_and_set(uint64_t* w, uint64_t pos) {
    const uint64_t m = 1ULL << (pos & 63);
    return (__atomic_fetch_or(&w[pos >> 6], m, __ATOMIC_RELAXED) & m) != 0;
}

}  // namespace

void build_anchor_mphf(const std::vector<uint64_t>& keys, unsigned T, AnchorMphf& out) {
    const uint64_t n = keys.size();
    out = AnchorMphf();
    memset(out.levels, 0, sizeof(out.levels));
    out.n = n;
    if (n == 0) return;  // BooPHF.h:736: an mphf over nothing is never "built"; lookups answer ULLONG_MAX
    const double gamma = 10.0;  // aligner.h:94 gammaFactor
    const uint64_t hash_domain = (uint64_t)std::ceil((double)n * gamma);                              // BooPHF.h:733
    const double p = 1.0 - std::pow((gamma * (double)n - 1) / (gamma * (double)n), (double)(n - 1));  // BooPHF.h:1018
    uint64_t words = 0, rwords = 0;
    for (int i = 0; i < BGR_ANC_LEVELS; ++i) {  // BooPHF.h:1034-1035
        uint64_t d = (((uint64_t)((double)hash_domain * std::pow(p, (double)i)) + 63) / 64) * 64;
        if (d == 0) d = 64;
        out.levels[i].domain = d;
        out.levels[i].magic = ~0ULL / d + ((d & (d - 1)) == 0 ? 1 : 0);  // floor(2^64 / d); (2^64-1)/d falls one short only when d | 2^64
        out.levels[i].word_base = words;
        out.levels[i].rank_base = rwords;
        const uint64_t nw = 1 + d / 64;  // BooPHF.h:425-429 bitVector(n): 1 + n/64 words
        words += nw;
        rwords += (nw + 7) / 8;          // BooPHF.h:594-607: one sample per 512 bits
    }
    out.bits.assign(words, 0);
    out.ranks.assign(rwords, 0);

    std::vector<Rem> rem, next;
    std::vector<std::vector<Rem>> part(T);
    std::vector<uint64_t> coll;
    uint64_t offset = 0;
    for (int i = 0; i < BGR_ANC_LEVELS - 1; ++i) {
        const uint64_t d = out.levels[i].domain, nw = 1 + d / 64;
        uint64_t* B = out.bits.data() + out.levels[i].word_base;
        coll.assign(nw, 0);
        const uint64_t items = i == 0 ? n : rem.size();
        const unsigned Tl = items < 4096 ? 1 : T;
        // BooPHF.h:1091-1100: first key on a position sets its bit, any further one marks the collision array
        parallel_ranges(Tl, items, [&](uint64_t b, uint64_t e, unsigned) {
            for (uint64_t j = b; j < e; ++j) {
                uint64_t h;
                if (i == 0) {
                    h = bgr_boo_hash64(keys[j], BGR_BOO_SEED0);
                } else {
                    Rem& r = rem[j];
                    if (i == 1) { h = bgr_boo_hash64(r.key, BGR_BOO_SEED1); r.s1 = h; }
                    else h = bgr_boo_next(&r.s0, &r.s1);
                    r.h = h;
                }
                const uint64_t pos = h % d;
                if (test_and_set(B, pos)) test_and_set(coll.data(), pos);
            }
        });
        for (uint64_t w = 0; w < nw; ++w) B[w] &= ~coll[w];  // BooPHF.h:509-520 clearCollisions
        // keys whose bit was cleared go on to the next level, in input order
        for (auto& v : part) v.clear();
        parallel_ranges(Tl, items, [&](uint64_t b, uint64_t e, unsigned t) {
            std::vector<Rem>& o = part[t];
            for (uint64_t j = b; j < e; ++j) {
                const uint64_t h = i == 0 ? bgr_boo_hash64(keys[j], BGR_BOO_SEED0) : rem[j].h;
                const uint64_t pos = h % d;
                if ((B[pos >> 6] >> (pos & 63)) & 1) continue;  // placed here
                if (i == 0) o.push_back({keys[j], h, 0, 0}); else o.push_back(rem[j]);
            }
        });
        next.clear();
        for (unsigned t = 0; t < Tl; ++t) next.insert(next.end(), part[t].begin(), part[t].end());
        rem.swap(next);
        // BooPHF.h:594-607 build_ranks
        uint64_t* R = out.ranks.data() + out.levels[i].rank_base;
        const uint64_t before = offset;
        for (uint64_t w = 0; w < nw; ++w) {
            if ((w & 7) == 0) R[w >> 3] = offset;
            offset += (uint64_t)__builtin_popcountll(B[w]);
        }
        if (offset != before) out.active_levels = (uint32_t)i + 1;
    }
    {   // level 24 holds no bits (BooPHF.h:891-899: its keys go to the exact map) but is still ranked
        const int i = BGR_ANC_LEVELS - 1;
        const uint64_t nw = 1 + out.levels[i].domain / 64;
        uint64_t* R = out.ranks.data() + out.levels[i].rank_base;
        for (uint64_t w = 0; w < nw; w += 8) R[w >> 3] = offset;
    }
    out.last_rank = offset;
    // _final_hash[key] = index++ in input order: a repeated key keeps its LAST index
    std::vector<std::pair<uint64_t, uint64_t>> kv(rem.size());
    for (uint64_t j = 0; j < rem.size(); ++j) kv[j] = {rem[j].key, j};
    std::sort(kv.begin(), kv.end());
    for (size_t j = 0; j < kv.size(); ++j) {
        if (j + 1 < kv.size() && kv[j + 1].first == kv[j].first) continue;
        out.final_kv.push_back(kv[j].first);
        out.final_kv.push_back(kv[j].second);
    }
}

uint64_t anchor_lookup(const BgrBlobHeader* h, const uint8_t* base, uint64_t key) {
    if (h->anc_n == 0) return ~0ULL;
    const uint64_t* bits = reinterpret_cast<const uint64_t*>(base + h->off_anc_bits);
    uint64_t s0 = 0, s1 = 0;
    const int active = (int)h->anc_active_levels;  // the levels above hold no set bit: probing them cannot answer
    for (int i = 0; i < active; ++i) {  // BooPHF.h:1058-1087 getLevel
        uint64_t hh;
        if (i == 0) hh = s0 = bgr_boo_hash64(key, BGR_BOO_SEED0);
        else if (i == 1) hh = s1 = bgr_boo_hash64(key, BGR_BOO_SEED1);
        else hh = bgr_boo_next(&s0, &s1);
        const BgrAncLevel& lv = h->anc_levels[i];
        const uint64_t pos = bgr_mod_magic(hh, lv.domain, lv.magic);
        const uint64_t* B = bits + lv.word_base;
        if ((B[pos >> 6] >> (pos & 63)) & 1) {  // BooPHF.h:609-622 rank
            const uint64_t* R = reinterpret_cast<const uint64_t*>(base + h->off_anc_ranks) + lv.rank_base;
            const uint64_t widx = pos >> 6, blk = pos >> 9;
            uint64_t r = R[blk];
            for (uint64_t w = blk * 8; w < widx; ++w) r += (uint64_t)__builtin_popcountll(B[w]);
            r += (uint64_t)__builtin_popcountll(B[widx] & ((1ULL << (pos & 63)) - 1));
            return r;
        }
    }
    const uint64_t* kv = reinterpret_cast<const uint64_t*>(base + h->off_anc_final);
    uint64_t lo = 0, hi = h->anc_n_final;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (kv[2 * mid] < key) lo = mid + 1; else hi = mid;
    }
    if (lo < h->anc_n_final && kv[2 * lo] == key) return h->anc_last_rank + kv[2 * lo + 1];
    return ~0ULL;
}

}  // namespace bgr
