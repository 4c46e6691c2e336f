// tools/synth.cpp -- seeded synthetic de Bruijn graph + read generator.
//
// Not part of the product path and not part of the oracle: it only manufactures inputs for tests/ and
// bench.py.  Everything is a pure function of (seed, index) through SplitMix64, so any slice of the
// read set can be regenerated anywhere (host threads here; rank r of an N-GPU bench generates its shard).
//
// Graph by direct construction: random genome g of length G; variant sites every ~d bases (spacing
// >= k+1 so bubbles are isolated), `alleles` alleles per site.  Unitigs = shared segments
// g[p_i+1, p_{i+1}) plus one (2k-1)-mer per allele [p-k+1, p+k), emitted in genome order with a random
// strand per unitig, 2 lines per FASTA record.  Reads: uniform start, one allele per site chosen per
// read, e ~ U{0..max_sub} substitutions at uniform positions, 50 % reverse complemented, header >r<i>.

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

inline uint64_t splitmix(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
inline uint64_t mix2(uint64_t a, uint64_t b) {
    uint64_t s = a * 0xD1342543DE82EF95ULL + b;
    splitmix(s);
    return splitmix(s);
}
const char kNuc[4] = {'A', 'C', 'G', 'T'};
inline char comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A'; }
void revcomp_inplace(std::string& s) {
    std::reverse(s.begin(), s.end());
    for (char& c : s) c = comp(c);
}

struct Synth {
    uint32_t k = 0, alleles = 2;
    std::string genome;
    std::vector<uint64_t> sites;          // sorted variant positions
    std::vector<uint8_t> site_alt;        // alleles-1 alternative base codes per site (packed 2 bits each, up to 3)
    std::vector<std::string> unitigs;     // as emitted (random strand)
};

inline char allele_base(const Synth& s, size_t site, uint32_t a) {
    // allele 0 = genome base; allele j>0 = (genome base + rot_j) mod 4 where rot_j are distinct non-zero
    char g = s.genome[s.sites[site]];
    if (a == 0) return g;
    int gi = g == 'A' ? 0 : g == 'C' ? 1 : g == 'G' ? 2 : 3;
    int rot = (s.site_alt[site] >> (2 * (a - 1))) & 3;
    return kNuc[(gi + rot) & 3];
}

}  // namespace

extern "C" {

void* syn_create(uint64_t G, uint32_t d, uint32_t alleles, uint32_t k, uint64_t seed) {
    Synth* s = new Synth();
    s->k = k;
    s->alleles = alleles < 1 ? 1 : (alleles > 4 ? 4 : alleles);
    s->genome.resize(G);
    uint64_t st = mix2(seed, 1);
    for (uint64_t i = 0; i < G; i += 32) {
        uint64_t r = splitmix(st);
        for (uint64_t j = i; j < G && j < i + 32; ++j) { s->genome[j] = kNuc[r & 3]; r >>= 2; }
    }
    // sites: spacing uniform in [max(k+1, d/2), d + d/2), first site >= k, last <= G-k-1
    uint64_t st2 = mix2(seed, 2);
    uint64_t lo = std::max<uint64_t>(k + 1, d / 2), hi = std::max<uint64_t>(lo + 1, (uint64_t)d + d / 2);
    uint64_t p = k + splitmix(st2) % (hi - lo);
    while (s->alleles > 1 && p + k + 1 < G) {
        s->sites.push_back(p);
        // distinct non-zero rotations 1,2,3 in a random order
        uint64_t r = splitmix(st2);
        int perm[3] = {1, 2, 3};
        int a0 = r % 3; std::swap(perm[0], perm[a0]);
        int a1 = 1 + (r >> 8) % 2; std::swap(perm[1], perm[a1]);
        s->site_alt.push_back((uint8_t)(perm[0] | (perm[1] << 2) | (perm[2] << 4)));
        p += lo + splitmix(st2) % (hi - lo);
    }
    // unitigs in genome order
    uint64_t st3 = mix2(seed, 3);
    auto emit = [&](std::string u) {
        if (splitmix(st3) & 1) revcomp_inplace(u);
        s->unitigs.push_back(std::move(u));
    };
    uint64_t prev = 0;  // start of the current shared segment
    for (size_t i = 0; i < s->sites.size(); ++i) {
        uint64_t sp = s->sites[i];
        if (sp - prev >= k) emit(s->genome.substr(prev, sp - prev));
        for (uint32_t a = 0; a < s->alleles; ++a) {
            std::string u = s->genome.substr(sp - k + 1, 2 * k - 1);
            u[k - 1] = allele_base(*s, i, a);
            emit(std::move(u));
        }
        prev = sp + 1;
    }
    if (G - prev >= k) emit(s->genome.substr(prev, G - prev));
    return s;
}
void syn_destroy(void* h) { delete static_cast<Synth*>(h); }
uint64_t syn_unitig_count(void* h) { return static_cast<Synth*>(h)->unitigs.size(); }
uint64_t syn_unitig_bases(void* h) {
    uint64_t t = 0;
    for (auto& u : static_cast<Synth*>(h)->unitigs) t += u.size();
    return t;
}
// seqs: syn_unitig_bases bytes; offs: count+1
void syn_unitigs(void* h, char* seqs, uint64_t* offs) {
    Synth* s = static_cast<Synth*>(h);
    uint64_t w = 0;
    for (size_t i = 0; i < s->unitigs.size(); ++i) {
        offs[i] = w;
        memcpy(seqs + w, s->unitigs[i].data(), s->unitigs[i].size());
        w += s->unitigs[i].size();
    }
    offs[s->unitigs.size()] = w;
}
int syn_write_unitigs(void* h, const char* path) {
    Synth* s = static_cast<Synth*>(h);
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    for (size_t i = 0; i < s->unitigs.size(); ++i) fprintf(f, ">%zu\n%s\n", i + 1, s->unitigs[i].c_str());
    fclose(f);
    return 0;
}

static void make_read(const Synth& s, uint64_t idx, uint32_t L, uint32_t max_sub, uint64_t seed, char* out) {
    uint64_t st = mix2(seed, idx);
    uint64_t G = s.genome.size();
    uint64_t start = splitmix(st) % (G - L + 1);
    memcpy(out, s.genome.data() + start, L);
    // alleles for the sites inside [start, start+L)
    uint64_t hap = splitmix(st);
    auto it = std::lower_bound(s.sites.begin(), s.sites.end(), start);
    for (; it != s.sites.end() && *it < start + L; ++it) {
        size_t si = it - s.sites.begin();
        uint32_t a = (uint32_t)(mix2(hap, si) % s.alleles);
        out[*it - start] = allele_base(s, si, a);
    }
    uint32_t e = (uint32_t)(splitmix(st) % (max_sub + 1));
    for (uint32_t j = 0; j < e; ++j) {
        uint64_t r = splitmix(st);
        uint32_t pos = (uint32_t)(r % L);
        int cur = out[pos] == 'A' ? 0 : out[pos] == 'C' ? 1 : out[pos] == 'G' ? 2 : 3;
        out[pos] = kNuc[(cur + 1 + (r >> 32) % 3) & 3];
    }
    if (splitmix(st) & 1) {
        std::reverse(out, out + L);
        for (uint32_t j = 0; j < L; ++j) out[j] = comp(out[j]);
    }
}

// genome start positions of reads [first, first+n) (what make_read draws first): lets a bench order a batch by locus
void syn_read_starts(void* h, uint64_t first, uint64_t n, uint32_t L, uint64_t seed, uint64_t* out) {
    const Synth& s = *static_cast<Synth*>(h);
    const uint64_t G = s.genome.size();
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t st = mix2(seed, first + i);
        out[i] = splitmix(st) % (G - L + 1);
    }
}

// reads [first, first+n) of fixed length L into out (n*L bytes, no separators)
void syn_reads(void* h, uint64_t first, uint64_t n, uint32_t L, uint32_t max_sub, uint64_t seed, char* out, int threads) {
    const Synth& s = *static_cast<Synth*>(h);
    if (threads < 1) threads = 1;
    std::vector<std::thread> ts;
    for (int t = 0; t < threads; ++t) {
        ts.emplace_back([&, t]() {
            uint64_t lo = n * t / threads, hi = n * (t + 1) / threads;
            for (uint64_t i = lo; i < hi; ++i) make_read(s, first + i, L, max_sub, seed, out + i * L);
        });
    }
    for (auto& t : ts) t.join();
}
// same file as syn_write_reads, formatted by `threads` threads in blocks (for multi-GB benchmark inputs)
int syn_write_reads_mt(void* h, const char* path, uint64_t first, uint64_t n, uint32_t L, uint32_t max_sub, uint64_t seed, int fastq, int threads) {
    const Synth& s = *static_cast<Synth*>(h);
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    if (threads < 1) threads = 1;
    const uint64_t block = 1 << 18;
    std::string qual(L, 'I');
    for (uint64_t b0 = 0; b0 < n; b0 += block * threads) {
        std::vector<std::string> bufs(threads);
        std::vector<std::thread> ts;
        for (int t = 0; t < threads; ++t) {
            ts.emplace_back([&, t]() {
                uint64_t lo = b0 + block * t, hi = std::min<uint64_t>(n, lo + block);
                if (lo >= hi) return;
                std::string& o = bufs[t];
                o.reserve((hi - lo) * (L + 16) * (fastq ? 2 : 1));
                std::vector<char> rd(L);
                char hd[32];
                for (uint64_t i = lo; i < hi; ++i) {
                    make_read(s, first + i, L, max_sub, seed, rd.data());
                    int hl = snprintf(hd, sizeof(hd), "%cr%llu\n", fastq ? '@' : '>', (unsigned long long)(first + i));
                    o.append(hd, hl);
                    o.append(rd.data(), L);
                    o.push_back('\n');
                    if (fastq) { o.append("+\n"); o.append(qual); o.push_back('\n'); }
                }
            });
        }
        for (auto& t : ts) t.join();
        for (auto& o : bufs) if (!o.empty()) fwrite(o.data(), 1, o.size(), f);
    }
    fclose(f);
    return 0;
}

int syn_write_reads(void* h, const char* path, uint64_t first, uint64_t n, uint32_t L, uint32_t max_sub, uint64_t seed, int fastq) {
    const Synth& s = *static_cast<Synth*>(h);
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    std::vector<char> buf(L + 1);
    std::string qual(L, 'I');
    for (uint64_t i = 0; i < n; ++i) {
        make_read(s, first + i, L, max_sub, seed, buf.data());
        buf[L] = 0;
        if (fastq) fprintf(f, "@r%llu\n%s\n+\n%s\n", (unsigned long long)(first + i), buf.data(), qual.c_str());
        else fprintf(f, ">r%llu\n%s\n", (unsigned long long)(first + i), buf.data());
    }
    fclose(f);
    return 0;
}

}  // extern "C"
