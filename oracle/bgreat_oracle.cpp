// oracle/bgreat_oracle.cpp -- CPU restatement of BGREAT's per-read mapping path.
//
// *** TEST INFRASTRUCTURE.  This file is the parity CHECKER for the HIP path; it is never the thing
// *** shipped or measured.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// *** load it.  The product library (bgreat_amd/csrc) does not link, include or call anything here.
//
// Parity status: PINNED.  tests/test_oracle_vs_ref.py runs this restatement and the compiled reference
// (oracle/_ref/bgreat, built from /root/reference by oracle/Makefile) on the same inputs and requires
// byte-identical `paths` / `notAligned.fa` and identical counters; tests/golden/ holds the committed
// inputs + reference outputs so the pin also holds where /root/reference is absent (the GPU box).
//
// Every function states the reference location it restates (file:line under /root/reference).
// It is written from the behaviour, with std::string sequences like the reference so that the
// reference's corner cases (N handling, clipped substr, first-zero-wins selection, the overlap
// double count of the greedy right walk, FASTQ phantom record ...) fall out of the same arithmetic.
//
// Build: see oracle/Makefile (liboracle.so for ctypes, bgreat_oracle as a CLI twin of `bgreat`).

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

using std::string;
using std::vector;
typedef uint64_t kmer_t;   // utils.h:27  (#define kmer uint64_t)
typedef int32_t unum_t;    // utils.h:26  (#define uNumber int32_t)

// ------------------------------------------------------------------------------------------------
// Work counters (SURVEY.md section 8d "ALGORITHMIC bytes per read").  Thread-local, summed on demand.
// ------------------------------------------------------------------------------------------------
struct Work {
    uint64_t reads = 0, read_bases = 0;
    uint64_t lookups = 0;          // mphf::lookup calls
    uint64_t probes_all = 0;       // level bit probes incl. empty tail levels (what the reference executes)
    uint64_t probes_nonempty = 0;  // level bit probes on levels that hold at least one key
    uint64_t level_hits = 0;       // lookups that stop on a level (then rank)
    uint64_t rank_words = 0;       // words popcounted by rank (excl. the rank sample)
    uint64_t final_finds = 0;      // unordered_map finds
    uint64_t tab_records = 0;      // unitigIndices records fetched
    uint64_t unitig_fetch = 0;     // candidate unitigs materialised
    uint64_t mm_calls = 0, mm_bases = 0;  // missmatchNumber calls / characters actually compared
    uint64_t path_ints = 0;
    void add(const Work& o) {
        const uint64_t* s = reinterpret_cast<const uint64_t*>(&o);
        uint64_t* d = reinterpret_cast<uint64_t*>(this);
        for (size_t i = 0; i < sizeof(Work) / 8; ++i) d[i] += s[i];
    }
};
static thread_local Work tl_work;

// ------------------------------------------------------------------------------------------------
// L0 sequence primitives                                                       utils.cpp:52-192
// ------------------------------------------------------------------------------------------------
// utils.cpp:52-59  revCompChar: A->T C->G G->C, anything else (T, N, ...) -> 'A'
static inline char rev_comp_char(char c) {
    if (c == 'A') return 'T';
    if (c == 'C') return 'G';
    if (c == 'G') return 'C';
    return 'A';
}
// utils.cpp:66-73  reverseComplements
static string reverse_complements(const string& s) {
    string out(s.size(), '\0');
    for (size_t i = 0; i < s.size(); ++i) out[i] = rev_comp_char(s[s.size() - 1 - i]);
    return out;
}
// utils.cpp:117-129  str2num: 2 bits per base, first base most significant; A0 C1 G2, everything else 3
static kmer_t str2num(const string& s) {
    kmer_t v = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        v <<= 2;
        char c = s[i];
        v += (c == 'A') ? 0 : (c == 'C') ? 1 : (c == 'G') ? 2 : 3;
    }
    return v;
}
// utils.cpp:132-140  nuc2int: C1 G2 T3, everything else (A, N) 0
static inline kmer_t nuc2int(char c) { return c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 0; }
// utils.cpp:143-151  nuc2intrc: A3 C2 G1, everything else (T, N) 0
static inline kmer_t nuc2intrc(char c) { return c == 'A' ? 3 : c == 'C' ? 2 : c == 'G' ? 1 : 0; }
// utils.cpp:154-168  missmatchNumber: loop bound is seq2.size(); returns as soon as the count exceeds n
static unsigned mismatch_number(const string& a, const string& b, unsigned n) {
    unsigned miss = 0;
    ++tl_work.mm_calls;
    for (size_t i = 0; i < b.size(); ++i) {
        ++tl_work.mm_bases;
        if (b[i] != a[i]) {
            if (++miss > n) return miss;
        }
    }
    return miss;
}
// utils.cpp:182-192  rcb: reverse complement of an n-digit base-4 number
static kmer_t rcb(kmer_t v, unsigned n) {
    kmer_t res = 0, offset = 1;
    offset <<= (2 * n - 2);
    for (unsigned i = 0; i < n; ++i) {
        res += (3 - (v % 4)) * offset;
        v >>= 2;
        offset >>= 2;
    }
    return res;
}

// ------------------------------------------------------------------------------------------------
// BooPHF restatement (query side + a single-threaded build that yields the same level structure).
// The index VALUES are not observable in BGREAT's output -- the structure is kept
// faithful (gamma, 25 levels, level sizing, hashes, rank sampling) so that the work counters above
// describe what the reference executes.
// ------------------------------------------------------------------------------------------------
namespace boo {

// BooPHF.h:251-264  HashFunctors::hash64.  NB precedence on the first line: `key * (hash >> 3)` binds
// tighter than the surrounding ^.
static inline uint64_t hash64(uint64_t key, uint64_t seed) {
    uint64_t h = seed;
    h ^= (h << 7) ^ (key * (h >> 3)) ^ (~((h << 11) + (key ^ (h >> 5))));
    h = (~h) + (h << 21);
    h = h ^ (h >> 24);
    h = (h + (h << 3)) + (h << 8);
    h = h ^ (h >> 14);
    h = (h + (h << 2)) + (h << 4);
    h = h ^ (h >> 28);
    h = h + (h << 31);
    return h;
}

struct BitVec {  // BooPHF.h:425-660 bitVector
    vector<uint64_t> words;
    uint64_t size = 0;
    vector<uint64_t> ranks;
    static const uint64_t kSample = 512;  // BooPHF.h:657
    void init(uint64_t n) {               // BooPHF.h:425-429: 1 + n/64 words
        size = n;
        words.assign(1 + n / 64, 0);
        ranks.clear();
    }
    uint64_t get(uint64_t pos) const { return (words[pos >> 6] >> (pos & 63)) & 1; }  // BooPHF.h:553-556
    bool test_and_set(uint64_t pos) {                                                  // BooPHF.h:559-565
        uint64_t m = 1ULL << (pos & 63);
        bool old = words[pos >> 6] & m;
        words[pos >> 6] |= m;
        return old;
    }
    uint64_t build_ranks(uint64_t offset) {  // BooPHF.h:594-607
        uint64_t cur = offset;
        for (size_t i = 0; i < words.size(); ++i) {
            if ((i * 64) % kSample == 0) ranks.push_back(cur);
            cur += __builtin_popcountll(words[i]);
        }
        return cur;
    }
    uint64_t rank(uint64_t pos) const {  // BooPHF.h:609-622
        uint64_t word_idx = pos / 64, word_off = pos % 64, block = pos / kSample;
        uint64_t r = ranks[block];
        for (uint64_t w = block * kSample / 64; w < word_idx; ++w) {
            r += __builtin_popcountll(words[w]);
            ++tl_work.rank_words;
        }
        r += __builtin_popcountll(words[word_idx] & ((1ULL << word_off) - 1));
        ++tl_work.rank_words;
        return r;
    }
};

struct Level {  // BooPHF.h:665-682
    uint64_t hash_domain = 0;
    BitVec bits;
    bool nonempty = false;
};

struct Mphf {  // BooPHF.h:711-1216, instantiated as mphf<u64, SingleHashFunctor<u64>> (aligner.h:40-41)
    static const int kLevels = 25;  // BooPHF.h:1023
    vector<Level> levels;
    std::unordered_map<uint64_t, uint64_t> final_hash;
    uint64_t last_rank = 0, nelem = 0;
    bool built = false;

    // BooPHF.h:336-356: h0 / h1 are hash64 with two literal seeds (SingleHashFunctor passes the seed
    // straight through, BooPHF.h:296); further levels come from xorshift128+ seeded with (h0, h1).
    static inline uint64_t next_hash(uint64_t s[2], uint64_t key, int ii) {
        if (ii == 0) return s[0] = hash64(key, 0xAAAAAAAA55555555ULL);
        if (ii == 1) return s[1] = hash64(key, 0x33333333CCCCCCCCULL);
        uint64_t s1 = s[0];
        const uint64_t s0 = s[1];
        s[0] = s0;
        s1 ^= s1 << 23;
        return (s[1] = (s1 ^ s0 ^ (s1 >> 17) ^ (s0 >> 26))) + s0;
    }

    // BooPHF.h:1058-1087 getLevel
    uint64_t get_level(uint64_t s[2], uint64_t key, int* res_level, int maxlevel, bool count) const {
        int level = 0;
        uint64_t h = 0;
        for (int ii = 0; ii < kLevels - 1 && ii < maxlevel; ++ii) {
            h = next_hash(s, key, ii);
            if (count) {
                ++tl_work.probes_all;
                if (levels[ii].nonempty) ++tl_work.probes_nonempty;
            }
            if (levels[ii].bits.get(h % levels[ii].hash_domain)) break;  // BooPHF.h:673-677
            ++level;
        }
        *res_level = level;
        return h;
    }

    // BooPHF.h:732-780 constructor + :1010-1054 setup + :845-924/:1091-1155 level processing,
    // executed by one thread (the reference's pthreads only change the arrival order inside the final
    // map, which is not observable).
    void build(const vector<uint64_t>& keys, double gamma) {
        nelem = keys.size();
        if (nelem == 0) return;  // BooPHF.h:736
        double n = (double)nelem;
        uint64_t hash_domain = (uint64_t)std::ceil(n * gamma);                       // BooPHF.h:733
        double p = 1.0 - std::pow((gamma * n - 1) / (gamma * n), (double)(nelem - 1));  // BooPHF.h:1018
        levels.assign(kLevels, Level());
        for (int ii = 0; ii < kLevels; ++ii) {  // BooPHF.h:1034-1035
            uint64_t d = (((uint64_t)(hash_domain * std::pow(p, ii)) + 63) / 64) * 64;
            if (d == 0) d = 64;
            levels[ii].hash_domain = d;
        }
        uint64_t offset = 0, final_idx = 0;
        for (int i = 0; i < kLevels; ++i) {
            levels[i].bits.init(levels[i].hash_domain);
            BitVec coll;
            coll.init(levels[i].hash_domain);
            for (uint64_t key : keys) {
                uint64_t s[2] = {0, 0};
                int level;
                get_level(s, key, &level, i, false);
                if (level != i) continue;
                if (i == kLevels - 1) {
                    final_hash[key] = final_idx++;  // BooPHF.h:891-899
                } else {
                    uint64_t h = next_hash(s, key, i);
                    uint64_t pos = h % levels[i].hash_domain;
                    if (levels[i].bits.test_and_set(pos)) coll.test_and_set(pos);  // BooPHF.h:1091-1100
                }
            }
            for (size_t w = 0; w < levels[i].hash_domain / 64; ++w)  // BooPHF.h:509-520 clearCollisions
                levels[i].bits.words[w] &= ~coll.words[w];
            for (uint64_t w : levels[i].bits.words) if (w) levels[i].nonempty = true;
            offset = levels[i].bits.build_ranks(offset);  // BooPHF.h:761
        }
        last_rank = offset;  // BooPHF.h:770
        built = true;
    }

    // BooPHF.h:783-818 lookup
    uint64_t lookup(uint64_t key) const {
        ++tl_work.lookups;
        if (!built) return ULLONG_MAX;
        uint64_t s[2] = {0, 0};
        int level;
        uint64_t h = get_level(s, key, &level, 100, true);
        if (level == kLevels - 1) {
            ++tl_work.final_finds;
            auto it = final_hash.find(key);
            if (it == final_hash.end()) return ULLONG_MAX;
            return it->second + last_rank;
        }
        ++tl_work.level_hits;
        return levels[level].bits.rank(h % levels[level].hash_domain);
    }
};
}  // namespace boo

// ------------------------------------------------------------------------------------------------
// Aligner restatement
// ------------------------------------------------------------------------------------------------
struct UnitigIndices {  // aligner.h:49-55
    kmer_t overlap;
    uint32_t indice[4];
};

struct Cand {  // one element of the vector<pair<string,uNumber>> of getBegin/getEnd
    string seq;
    unum_t id;
};

struct Oracle {
    unsigned k = 0;
    vector<string> unitigs;  // index 0 is "" (aligner.cpp:408)
    boo::Mphf leftMPHF, rightMPHF;
    vector<UnitigIndices> leftIndices, rightIndices;
    kmer_t offsetUpdate = 0;  // aligner.h:101-102
    // anchors ("dog") mode, -G: an MPHF over the canonical k-mers of every unitig and their (unitig, offset)
    bool dogMode = false;                                         // aligner.h:60,82
    unsigned fracKmer = 1;                                        // aligner.h:95
    boo::Mphf anchorsMPHF;                                        // aligner.h:65
    vector<std::pair<uint32_t, uint32_t>> anchorsPosition;        // aligner.h:67
    // run parameters (aligner.h:90-104)
    unsigned errorsMax = 2, tryNumber = 2;
    bool partial = false;
    // counters (aligner.h:68)
    std::atomic<uint64_t> alignedRead{0}, readNumber{0}, noOverlapRead{0}, notAligned{0}, overlaps{0};
    std::mutex work_mutex;
    Work work_total;

    // ---- index build ------------------------------------------------------- aligner.cpp:407-534
    static void fill_slot(UnitigIndices& r, kmer_t key, uint32_t i) {  // aligner.cpp:479-490 (x4)
        r.overlap = key;
        if (r.indice[0] == 0) r.indice[0] = i;
        else if (r.indice[1] == 0) r.indice[1] = i;
        else if (r.indice[2] == 0) r.indice[2] = i;
        else r.indice[3] = i;  // slot 4 is overwritten
    }
    void index_unitigs(const vector<string>& seqs) {
        offsetUpdate = 1;
        offsetUpdate <<= (2 * (k - 1));
        unitigs.clear();
        unitigs.push_back("");
        vector<kmer_t> leftOver, rightOver, anchors;
        for (const string& line : seqs) {
            if (line.size() < k) break;  // aligner.cpp:418-420: loading stops at the first short sequence
            unitigs.push_back(line);
            kmer_t beg = str2num(line.substr(0, k - 1)), rcBeg = rcb(beg, k - 1);
            if (beg <= rcBeg) leftOver.push_back(beg); else rightOver.push_back(rcBeg);
            kmer_t end = str2num(line.substr(line.size() - k + 1, k - 1)), rcEnd = rcb(end, k - 1);
            if (end <= rcEnd) rightOver.push_back(end); else leftOver.push_back(rcEnd);
            if (dogMode) {  // aligner.cpp:434-442: every k-mer but the last (j + k < size), NOT deduplicated
                for (unsigned j = 0; j + k < line.size(); ++j) {
                    if (j % fracKmer == 0) {
                        kmer_t seq = str2num(line.substr(j, k)), rcSeq = rcb(seq, k);
                        anchors.push_back(std::min(seq, rcSeq));
                    }
                }
            }
        }
        std::sort(leftOver.begin(), leftOver.end());
        leftOver.erase(std::unique(leftOver.begin(), leftOver.end()), leftOver.end());
        std::sort(rightOver.begin(), rightOver.end());
        rightOver.erase(std::unique(rightOver.begin(), rightOver.end()), rightOver.end());
        leftMPHF = boo::Mphf();
        rightMPHF = boo::Mphf();
        leftMPHF.build(leftOver, 10.0);   // aligner.cpp:450, gammaFactor=10 (aligner.h:94)
        rightMPHF.build(rightOver, 10.0);  // aligner.cpp:454
        leftIndices.assign(leftOver.size(), UnitigIndices{0, {0, 0, 0, 0}});
        rightIndices.assign(rightOver.size(), UnitigIndices{0, {0, 0, 0, 0}});
        anchorsMPHF = boo::Mphf();
        if (dogMode) anchorsMPHF.build(anchors, 10.0);  // aligner.cpp:457-460
        anchorsPosition.assign(anchors.size(), {0u, 0u});  // aligner.cpp:461,465
        Work saved = tl_work;  // build-time lookups are not per-read work
        for (uint32_t i = 1; i < unitigs.size(); ++i) {  // aligner.cpp:466-533
            const string& line = unitigs[i];
            if (dogMode) {  // aligner.cpp:468-476: a repeated k-mer keeps its LAST (unitig, offset)
                for (unsigned j = 0; j + k < line.size(); ++j) {
                    if (j % fracKmer == 0) {
                        kmer_t seq = str2num(line.substr(j, k)), rcSeq = rcb(seq, k);
                        anchorsPosition[anchorsMPHF.lookup(std::min(seq, rcSeq))] = {i, j};
                    }
                }
            }
            kmer_t beg = str2num(line.substr(0, k - 1)), rcBeg = rcb(beg, k - 1);
            if (beg <= rcBeg) fill_slot(leftIndices[leftMPHF.lookup(beg)], beg, i);
            else fill_slot(rightIndices[rightMPHF.lookup(rcBeg)], rcBeg, i);
            kmer_t end = str2num(line.substr(line.size() - k + 1, k - 1)), rcEnd = rcb(end, k - 1);
            if (end <= rcEnd) fill_slot(rightIndices[rightMPHF.lookup(end)], end, i);
            else fill_slot(leftIndices[leftMPHF.lookup(rcEnd)], rcEnd, i);
        }
        tl_work = saved;
    }
    // aligner.cpp:415-417: two getline per record, header ignored, no validation of the sequence.
    bool load_unitig_file(const string& path) {
        std::ifstream in(path);
        if (!in) return false;
        vector<string> seqs;
        string line;
        while (!in.eof()) {
            std::getline(in, line);
            std::getline(in, line);
            if (line.size() < k) break;
            seqs.push_back(line);
        }
        index_unitigs(seqs);
        return true;
    }

    // ---- neighbour fetch --------------------------------------------------- aligner.cpp:147-267
    bool fetch_record(bool useLeft, kmer_t key, UnitigIndices& out) const {
        const boo::Mphf& m = useLeft ? leftMPHF : rightMPHF;
        const vector<UnitigIndices>& tab = useLeft ? leftIndices : rightIndices;
        uint64_t h = m.lookup(key);
        if (h == ULLONG_MAX) return false;
        ++tl_work.tab_records;
        out = tab[h];
        return out.overlap == key;
    }
    // aligner.cpp:147-206 getEnd: unitigs that (in some orientation) END with bin.
    vector<Cand> get_end(kmer_t bin) const {
        vector<Cand> res;
        kmer_t rc = rcb(bin, k - 1);
        UnitigIndices ind;
        bool go = (bin <= rc) ? fetch_record(false, bin, ind) : fetch_record(true, rc, ind);
        if (!go) return res;
        for (int s = 0; s < 4 && ind.indice[s] != 0; ++s) {  // nested ifs of :172-203 == stop at first 0
            const string& u = unitigs[ind.indice[s]];
            ++tl_work.unitig_fetch;
            if (str2num(u.substr(u.size() - k + 1, k - 1)) == bin) res.push_back({u, (unum_t)ind.indice[s]});
            else res.push_back({reverse_complements(u), -(unum_t)ind.indice[s]});
        }
        return res;
    }
    // aligner.cpp:209-267 getBegin: unitigs that (in some orientation) BEGIN with bin.
    vector<Cand> get_begin(kmer_t bin) const {
        vector<Cand> res;
        kmer_t rc = rcb(bin, k - 1);
        UnitigIndices ind;
        bool go = (bin <= rc) ? fetch_record(true, bin, ind) : fetch_record(false, rc, ind);
        if (!go) return res;
        for (int s = 0; s < 4 && ind.indice[s] != 0; ++s) {
            const string& u = unitigs[ind.indice[s]];
            ++tl_work.unitig_fetch;
            if (str2num(u.substr(0, k - 1)) == bin) res.push_back({u, (unum_t)ind.indice[s]});
            else res.push_back({reverse_complements(u), -(unum_t)ind.indice[s]});
        }
        return res;
    }

    // ---- anchors ----------------------------------------------------------- aligner.cpp:305-378
    void update(kmer_t& v, char c) const { v <<= 2; v += nuc2int(c); v %= offsetUpdate; }       // :305-309
    void update_rc(kmer_t& v, char c) const { v >>= 2; v += (nuc2intrc(c) << (2 * k - 4)); }   // :312-315
    bool is_overlap(kmer_t rep) const {  // aligner.cpp:351-365: left table first, then right
        uint64_t h = leftMPHF.lookup(rep);
        if (h != ULLONG_MAX) {
            ++tl_work.tab_records;
            if (leftIndices[h].overlap == rep) return true;
        }
        h = rightMPHF.lookup(rep);
        if (h != ULLONG_MAX) {
            ++tl_work.tab_records;
            if (rightIndices[h].overlap == rep) return true;
        }
        return false;
    }
    // aligner.cpp:345-378 getNOverlap
    vector<std::pair<kmer_t, unsigned>> get_n_overlap(const string& read, unsigned n) const {
        vector<std::pair<kmer_t, unsigned>> list;
        kmer_t num = str2num(read.substr(0, k - 1)), rcnum = rcb(num, k - 1), rep = std::min(num, rcnum);
        for (unsigned i = 0;; ++i) {
            if (is_overlap(rep)) list.push_back({num, i});
            if (list.size() >= n) return list;
            if (i + k - 1 < read.size()) {
                update(num, read[i + k - 1]);
                update_rc(rcnum, read[i + k - 1]);
                rep = std::min(num, rcnum);
            } else {
                return list;
            }
        }
    }
    // aligner.cpp:318-342 getListOverlap: every position; the MPHF result is discarded (:324-326)
    vector<std::pair<kmer_t, unsigned>> get_list_overlap(const string& read) const {
        vector<std::pair<kmer_t, unsigned>> list;
        kmer_t num = str2num(read.substr(0, k - 1)), rcnum = rcb(num, k - 1), rep = std::min(num, rcnum);
        for (unsigned i = 0;; ++i) {
            (void)leftMPHF.lookup(rep);
            list.push_back({num, i});
            if (i + k - 1 < read.size()) {
                update(num, read[i + k - 1]);
                update_rc(rcnum, read[i + k - 1]);
                rep = std::min(num, rcnum);
            } else {
                return list;
            }
        }
    }

    // ---- greedy ------------------------------------------------------------ alignerGreedy.cpp
    // alignerGreedy.cpp:167-218 mapOnLeftEndGreedy and :268-319 checkBeginGreedy have the same body;
    // both are this function (the first call is checkBeginGreedy, the recursion mapOnLeftEndGreedy).
    unsigned left_greedy(const string& read, vector<unum_t>& path, kmer_t ov, unsigned pos, unsigned errors) const {
        if (pos == 0) { path.push_back(0); return 0; }
        string readLeft(read.substr(0, pos));
        vector<Cand> range(get_end(ov));
        unsigned minMiss = errors + 1, idxMin = 9;
        bool ended = false;
        int offset = 0;
        kmer_t nextOverlap = 0;
        string nextUnitig;
        for (unsigned i = 0; i < range.size(); ++i) {
            const string& u = range[i].seq;
            if (u.size() - k + 1 >= readLeft.size()) {  // the unitig covers the rest of the read
                unsigned miss = mismatch_number(u.substr(u.size() - readLeft.size() - k + 1, readLeft.size()), readLeft, errors);
                if (miss == 0) {
                    path.push_back(range[i].id);
                    path.push_back((unum_t)(u.size() - readLeft.size() - k + 1));
                    return 0;
                } else if (miss < minMiss) {
                    minMiss = miss; idxMin = i; ended = true;
                    offset = (int)(u.size() - readLeft.size() - k + 1);
                }
            } else {
                unsigned miss = mismatch_number(u.substr(0, u.size() - k + 1), readLeft.substr(readLeft.size() + k - 1 - u.size()), errors);
                if (miss == 0) {
                    path.push_back(range[i].id);
                    return left_greedy(read, path, str2num(u.substr(0, k - 1)), pos - (unsigned)(u.size() - k + 1), errors);
                } else if (miss < minMiss) {
                    ended = false; minMiss = miss; idxMin = i;
                    nextUnitig = u; nextOverlap = str2num(u.substr(0, k - 1));
                }
            }
        }
        if (minMiss <= errors) {
            path.push_back(range[idxMin].id);
            if (ended) { path.push_back(offset); return minMiss; }
            return minMiss + left_greedy(read, path, nextOverlap, pos - (unsigned)(nextUnitig.size() - k + 1), errors - minMiss);
        }
        return minMiss;
    }
    // alignerGreedy.cpp:221-265 mapOnRightEndGreedy (readLeft INCLUDES the k-1 overlap here)
    unsigned right_greedy_later(const string& read, vector<unum_t>& path, kmer_t ov, unsigned pos, unsigned errors) const {
        string readLeft(read.substr(pos));
        if (readLeft.size() < k) return 0;
        vector<Cand> range(get_begin(ov));
        unsigned minMiss = errors + 1, idxMin = 9;
        bool ended = false;
        kmer_t nextOverlap = 0;
        string nextUnitig;
        for (unsigned i = 0; i < range.size(); ++i) {
            const string& u = range[i].seq;
            if (u.size() - k + 1 >= readLeft.size()) {
                unsigned miss = mismatch_number(u.substr(0, readLeft.size()), readLeft, errors);
                if (miss == 0) { path.push_back(range[i].id); return 0; }
                else if (miss < minMiss) { minMiss = miss; idxMin = i; ended = true; }
            } else {
                unsigned miss = mismatch_number(u, read.substr(pos, u.size()), errors);  // slice clipped at |read|
                if (miss == 0) {
                    path.push_back(range[i].id);
                    return right_greedy_later(read, path, str2num(u.substr(u.size() - k + 1, k - 1)), pos + (unsigned)(u.size() - k + 1), errors);
                } else if (miss < minMiss) {
                    ended = false; minMiss = miss; idxMin = i;
                    nextUnitig = u; nextOverlap = str2num(u.substr(u.size() - k + 1, k - 1));
                }
            }
        }
        if (minMiss <= errors) {
            path.push_back(range[idxMin].id);
            if (ended) return minMiss;
            return minMiss + right_greedy_later(read, path, nextOverlap, pos + (unsigned)(nextUnitig.size() - k + 1), errors - minMiss);
        }
        return minMiss;
    }
    // alignerGreedy.cpp:322-364 checkEndGreedy (first right step: readLeft EXCLUDES the overlap)
    unsigned right_greedy_first(const string& read, vector<unum_t>& path, kmer_t ov, unsigned pos, unsigned errors) const {
        string readLeft(read.substr(pos + k - 1));
        if (readLeft.empty()) return 0;
        vector<Cand> range(get_begin(ov));
        unsigned minMiss = errors + 1, idxMin = 9;
        bool ended = false;
        kmer_t nextOverlap = 0;
        string nextUnitig;
        for (unsigned i = 0; i < range.size(); ++i) {
            const string& u = range[i].seq;
            if (u.size() - k + 1 >= readLeft.size()) {
                unsigned miss = mismatch_number(u.substr(k - 1, readLeft.size()), readLeft, errors);
                if (miss == 0) { path.push_back(range[i].id); return 0; }
                else if (miss < minMiss) { minMiss = miss; idxMin = i; ended = true; }
            } else {
                unsigned miss = mismatch_number(u.substr(k - 1), readLeft.substr(0, u.size() - k + 1), errors);
                if (miss == 0) {
                    path.push_back(range[i].id);
                    return right_greedy_later(read, path, str2num(u.substr(u.size() - k + 1, k - 1)), pos + (unsigned)(u.size() - k + 1), errors);
                } else if (miss < minMiss) {
                    minMiss = miss; idxMin = i; ended = false;
                    nextUnitig = u; nextOverlap = str2num(u.substr(u.size() - k + 1, k - 1));
                }
            }
        }
        if (minMiss <= errors) {
            path.push_back(range[idxMin].id);
            if (ended) return minMiss;
            return minMiss + right_greedy_later(read, path, nextOverlap, pos + (unsigned)(nextUnitig.size() - k + 1), errors - minMiss);
        }
        return minMiss;
    }
    // alignerGreedy.cpp:35-57 alignReadGreedy.  status: 0 no anchor, 1 anchored but failed, 2 aligned;
    // +4 when the answer came from the reverse-complement retry (the `rc` out-parameter).
    vector<unum_t> align_read_greedy(const string& read, bool& overlapFound, unsigned errors, bool& rc) {
        auto list = get_n_overlap(read, tryNumber);
        if (list.empty()) { ++noOverlapRead; return {}; }
        overlapFound = true;
        vector<unum_t> pathBegin, pathEnd;
        for (size_t s = 0; s < list.size(); ++s) {
            pathBegin.clear();
            unsigned eb = left_greedy(read, pathBegin, list[s].first, list[s].second, errors);
            if (eb <= errors) {
                pathEnd.clear();
                unsigned ee = right_greedy_first(read, pathEnd, list[s].first, list[s].second, errors - eb);
                if (ee + eb <= errors) {
                    ++alignedRead;
                    std::reverse(pathBegin.begin(), pathBegin.end());
                    pathBegin.insert(pathBegin.end(), pathEnd.begin(), pathEnd.end());
                    return pathBegin;
                }
            }
        }
        if (!rc) { rc = true; return align_read_greedy(reverse_complements(read), overlapFound, errors, rc); }
        ++notAligned;
        return {};
    }

    // ---- anchors ("dog") mode, -G ------------------------------------------- aligner.cpp:381-405
    // getNAnchors.  NB the rolling update() / updateRC() are the (k-1)-mer ones (aligner.cpp:305-315) applied to
    // k-mers, and the MPHF answer is used WITHOUT a key check: past position 0 the looked-up value is not the
    // read's k-mer, and any non-ULLONG_MAX answer (BooPHF returns a rank for many non-keys) becomes an anchor.
    vector<std::pair<std::pair<uint32_t, uint32_t>, unsigned>> get_n_anchors(const string& read, unsigned n) const {
        vector<std::pair<std::pair<uint32_t, uint32_t>, unsigned>> list;
        kmer_t num = str2num(read.substr(0, k)), rcnum = rcb(num, k), rep = std::min(num, rcnum);
        for (unsigned i = 0;; ++i) {
            uint64_t hash = anchorsMPHF.lookup(rep);
            if (hash != ULLONG_MAX) list.push_back({anchorsPosition[hash], i});
            if (list.size() >= n) return list;
            if (i + k < read.size()) {
                update(num, read[i + k]);
                update_rc(rcnum, read[i + k]);
                rep = std::min(num, rcnum);
            } else {
                return list;
            }
        }
    }
    // alignerGreedy.cpp:60-164 alignReadGreedyAnchors: place the anchoring unitig on the read (4 cases), then the
    // usual greedy walks from its two ends.
    vector<unum_t> align_read_greedy_anchors(const string& read, bool& overlapFound, unsigned errorMax, bool& rc) {
        auto listAnchors = get_n_anchors(read, tryNumber);
        if (listAnchors.empty()) { ++noOverlapRead; return {}; }
        overlapFound = true;
        vector<unum_t> pathBegin, pathEnd;
        string unitig;
        bool returned = false;
        for (unsigned start = 0; start < (unsigned)listAnchors.size(); ++start) {
            unsigned unitigNumber = listAnchors[start].first.first, positionUnitig = listAnchors[start].first.second,
                     positionRead = listAnchors[start].second;
            unitig = unitigs[unitigNumber];
            ++tl_work.unitig_fetch;
            if (unitig.size() < k) continue;  // :72-75 (entry never written: unitig 0 is "")
            if (str2num(unitig.substr(positionUnitig, k)) != str2num(read.substr(positionRead, k))) {  // :76-83
                unitig = reverse_complements(unitig);
                positionUnitig = (unsigned)unitig.size() - k - positionUnitig;
                returned = true;
            } else {
                returned = false;
            }
            const unum_t uid = returned ? -(unum_t)unitigNumber : (unum_t)unitigNumber;
            if (positionRead >= positionUnitig) {
                if (read.size() - positionRead >= unitig.size() - positionUnitig) {
                    // CASE 1: unitig included in read (:87-110)
                    unsigned errors = mismatch_number(read.substr(positionRead - positionUnitig, unitig.size()), unitig, errorMax);
                    if (errors <= errorMax) {
                        pathBegin = {};
                        unsigned errorBegin = left_greedy(read, pathBegin, str2num(unitig.substr(0, k - 1)), positionRead - positionUnitig, errorMax - errors);
                        if (errorBegin + errors <= errorMax) {
                            pathEnd = {uid};
                            unsigned errorsEnd = right_greedy_first(read, pathEnd, str2num(unitig.substr(unitig.size() - k + 1, k - 1)),
                                                                    positionRead - positionUnitig + (unsigned)unitig.size() - k + 1, errorMax - errors - errorBegin);
                            if (errorBegin + errors + errorsEnd <= errorMax) {
                                ++alignedRead;
                                std::reverse(pathBegin.begin(), pathBegin.end());
                                pathBegin.insert(pathBegin.end(), pathEnd.begin(), pathEnd.end());
                                return pathBegin;
                            }
                        }
                    }
                } else {
                    // CASE 2: unitig overlaps the read's end (:111-130)
                    unsigned errors = mismatch_number(read.substr(positionRead - positionUnitig),
                                                      unitig.substr(0, read.size() - positionRead + positionUnitig), errorMax);
                    if (errors <= errorMax) {
                        pathBegin = {};
                        unsigned errorBegin = left_greedy(read, pathBegin, str2num(unitig.substr(0, k - 1)), positionRead - positionUnitig, errorMax - errors);
                        if (errorBegin + errors <= errorMax) {
                            ++alignedRead;
                            std::reverse(pathBegin.begin(), pathBegin.end());
                            pathBegin.push_back(uid);
                            return pathBegin;
                        }
                    }
                }
            } else {
                if (read.size() - positionRead >= unitig.size() - positionUnitig) {
                    // CASE 3: the read starts inside the unitig and runs past its end (:133-148)
                    unsigned errors = mismatch_number(unitig.substr(positionUnitig - positionRead),
                                                      read.substr(0, unitig.size() + positionRead - positionUnitig), errorMax);
                    if (errors <= errorMax) {
                        pathEnd = {(unum_t)((int)positionUnitig - (int)positionRead), uid};
                        unsigned errorsEnd = right_greedy_first(read, pathEnd, str2num(unitig.substr(unitig.size() - k + 1, k - 1)),
                                                                positionRead - positionUnitig + (unsigned)unitig.size() - k + 1, errorMax - errors);
                        if (errors + errorsEnd <= errorMax) {
                            ++alignedRead;
                            return pathEnd;
                        }
                    }
                } else {
                    // CASE 4: read included in the unitig (:149-160)
                    unsigned errors = mismatch_number(unitig.substr(positionUnitig - positionRead, read.size()), read, errorMax);
                    if (errors <= errorMax) {
                        ++alignedRead;
                        return {(unum_t)((int)positionUnitig - (int)positionRead), uid};
                    }
                }
            }
        }
        if (!rc) { rc = true; return align_read_greedy_anchors(reverse_complements(read), overlapFound, errorMax, rc); }
        ++notAligned;
        return {};
    }

    // ---- exhaustive -------------------------------------------------------- alignerExhaustive.cpp
    // alignerExhaustive.cpp:61-106 mapOnRightEndExhaustive / :206-259 checkEndExhaustive.  Same shape at
    // every depth; `top` adds only the partial (-i) early return of :217-221.
    // (exh_memo, off by default: the literal recursion above is what pins this file to the reference.  On unitig sets that duplicate their own k-mers
    // the literal recursion is exponential -- 150 s for one 79-base read, in the reference too -- and cannot serve as a checker; with exh_memo every
    // non-top call (ov, pos, errors) is remembered per read: an answer within the budget is exact whatever the budget was (the children get
    // errors - miss, :85,:134, and the winner is the first strict minimum), a failure holds for every budget up to the one it was found with.  The
    // memoised form must return what the literal form returns: tools/fuzz_soup.py cpu compares the two, and both with the compiled reference, on
    // every soup the reference finishes.)
    struct MemoVal { bool exact = false; unsigned val = 0, bmax = 0; vector<unum_t> path; };
    struct MemoKey {
        kmer_t ov; unsigned pos; bool right;
        bool operator<(const MemoKey& o) const { return ov != o.ov ? ov < o.ov : pos != o.pos ? pos < o.pos : right < o.right; }
    };
    bool exh_memo = false;
    static std::map<MemoKey, MemoVal>& memo() { static thread_local std::map<MemoKey, MemoVal> m; return m; }
    // -> true: answered from the table (*out = what the call returns, path appended)
    bool memo_lookup(bool right, kmer_t ov, unsigned pos, unsigned errors, vector<unum_t>& path, unsigned* out) const {
        auto it = memo().find(MemoKey{ov, pos, right});
        if (it == memo().end()) return false;
        const MemoVal& v = it->second;
        if (v.exact) {
            if (v.val <= errors) { path.insert(path.end(), v.path.begin(), v.path.end()); *out = v.val; }
            else *out = errors + 1;
            return true;
        }
        if (errors <= v.bmax) { *out = errors + 1; return true; }
        return false;
    }
    void memo_store(bool right, kmer_t ov, unsigned pos, unsigned errors, unsigned minMiss, const vector<unum_t>& path, size_t path_from) const {
        MemoVal& v = memo()[MemoKey{ov, pos, right}];
        if (minMiss <= errors) { v.exact = true; v.val = minMiss; v.path.assign(path.begin() + (long)path_from, path.end()); }
        else if (!v.exact) v.bmax = std::max(v.bmax, errors);
    }
    unsigned right_exh(const string& read, vector<unum_t>& path, kmer_t ov, unsigned pos, unsigned errors, bool top) const {
        const size_t path_from = path.size();
        if (exh_memo && !top) { unsigned r; if (memo_lookup(true, ov, pos, errors, path, &r)) return r; }
        const unsigned r = right_exh_body(read, path, ov, pos, errors, top);
        if (exh_memo && !top) memo_store(true, ov, pos, errors, r, path, path_from);
        return r;
    }
    unsigned left_exh(const string& read, vector<unum_t>& path, kmer_t ov, unsigned pos, unsigned errors, bool top) const {
        const size_t path_from = path.size();
        if (exh_memo && !top) { unsigned r; if (memo_lookup(false, ov, pos, errors, path, &r)) return r; }
        const unsigned r = left_exh_body(read, path, ov, pos, errors, top);
        if (exh_memo && !top) memo_store(false, ov, pos, errors, r, path, path_from);
        return r;
    }
    unsigned right_exh_body(const string& read, vector<unum_t>& path, kmer_t ov, unsigned pos, unsigned errors, bool top) const {
        string readLeft(read.substr(pos + k - 1));
        vector<unum_t> path2keep;
        if (readLeft.empty()) { path.push_back(0); return 0; }
        vector<Cand> range(get_begin(ov));
        unsigned minMiss = errors + 1, idxMin = 9;
        bool ended = false;
        if (top && partial && range.empty()) return 0;
        for (unsigned i = 0; i < range.size(); ++i) {
            const string& u = range[i].seq;
            if (u.size() - k + 1 >= readLeft.size()) {
                unsigned miss = mismatch_number(u.substr(k - 1, readLeft.size()), readLeft, errors);
                if (miss < minMiss) { minMiss = miss; idxMin = i; ended = true; }
            } else {
                unsigned miss = mismatch_number(u.substr(k - 1), readLeft.substr(0, u.size() - k + 1), errors);
                if (miss < minMiss) {
                    vector<unum_t> possible;
                    miss += right_exh(read, possible, str2num(u.substr(u.size() - k + 1, k - 1)), pos + (unsigned)(u.size() - k + 1), errors - miss, false);
                    if (miss < minMiss) { path2keep = possible; minMiss = miss; idxMin = i; ended = false; }
                }
            }
        }
        if (minMiss <= errors) {
            path.push_back(range[idxMin].id);
            if (ended) path.push_back((unum_t)(readLeft.size() + k - 1));
            else path.insert(path.end(), path2keep.begin(), path2keep.end());
        }
        return minMiss;
    }
    // alignerExhaustive.cpp:109-155 mapOnLeftEndExhaustive / :158-203 checkBeginExhaustive.
    unsigned left_exh_body(const string& read, vector<unum_t>& path, kmer_t ov, unsigned pos, unsigned errors, bool top) const {
        if (pos == 0) { if (top) path.push_back(0); return 0; }  // :159 pushes 0, :112 does not
        string readLeft(read.substr(0, pos));
        vector<unum_t> path2keep;
        vector<Cand> range(get_end(ov));
        unsigned minMiss = errors + 1, idxMin = 0;
        int offset = -2;
        bool ended = false;
        for (unsigned i = 0; i < range.size(); ++i) {
            const string& u = range[i].seq;
            if (u.size() - k + 1 >= readLeft.size()) {
                unsigned miss = mismatch_number(u.substr(u.size() - readLeft.size() - k + 1, readLeft.size()), readLeft, errors);
                if (miss < minMiss) { minMiss = miss; idxMin = i; ended = true; offset = (int)(u.size() - readLeft.size() - k + 1); }
            } else {
                unsigned miss = mismatch_number(u.substr(0, u.size() - k + 1), readLeft.substr(readLeft.size() + k - 1 - u.size()), errors);
                if (miss < minMiss) {
                    vector<unum_t> possible;
                    miss += left_exh(read, possible, str2num(u.substr(0, k - 1)), pos - (unsigned)(u.size() - k + 1), errors - miss, false);
                    if (miss < minMiss) { minMiss = miss; idxMin = i; path2keep = possible; ended = false; }
                }
            }
        }
        if (minMiss <= errors) {
            if (ended) path.push_back(offset);
            else path.insert(path.end(), path2keep.begin(), path2keep.end());
            path.push_back(range[idxMin].id);
        }
        return minMiss;
    }
    // alignerExhaustive.cpp:35-58 alignReadExhaustive (readNumber is counted in here, not by the worker)
    vector<unum_t> align_read_exhaustive(const string& read, bool& overlapFound, unsigned errors) {
        auto list = get_list_overlap(read);
        if (exh_memo) memo().clear();  // (what is remembered holds for one read)
        if (list.empty()) { ++noOverlapRead; ++readNumber; return {}; }
        overlaps += list.size();
        overlapFound = true;
        vector<unum_t> pathBegin, pathEnd;
        for (size_t s = 0; s < list.size(); ++s) {
            pathBegin.clear();
            unsigned eb = left_exh(read, pathBegin, list[s].first, list[s].second, errors, true);
            if (eb <= errors) {
                pathEnd.clear();
                unsigned ee = right_exh(read, pathEnd, list[s].first, list[s].second, errors - eb, true);
                if (ee + eb <= errors) {
                    ++alignedRead; ++readNumber;
                    pathBegin.insert(pathBegin.end(), pathEnd.begin(), pathEnd.end());
                    return pathBegin;
                }
            }
        }
        ++notAligned; ++readNumber;
        return {};
    }

    // One read through the selected mode; fills the work counters.  mode 0 greedy, 1 exhaustive, 2 greedy from
    // k-mer anchors (-G; needs dogMode at index time).
    vector<unum_t> align_one(int mode, const string& read, uint8_t& status) {
        bool overlapFound = false, rc = false;
        vector<unum_t> path;
        uint64_t no0 = noOverlapRead.load();
        if (mode == 0) {
            ++readNumber;  // alignerGreedy.cpp:382
            path = align_read_greedy(read, overlapFound, errorsMax, rc);
        } else if (mode == 2) {
            ++readNumber;  // alignerGreedy.cpp:382
            path = align_read_greedy_anchors(read, overlapFound, errorsMax, rc);  // alignerGreedy.cpp:387-388
        } else {
            path = align_read_exhaustive(read, overlapFound, errorsMax);
        }
        bool noAnchorCounted = noOverlapRead.load() != no0;  // single-threaded callers only use this
        status = !path.empty() ? 2 : (noAnchorCounted ? 0 : 1);
        if (rc) status |= 4;
        ++tl_work.reads;
        tl_work.read_bases += read.size();
        tl_work.path_ints += path.size();
        return path;
    }
    void flush_work() {
        std::lock_guard<std::mutex> g(work_mutex);
        work_total.add(tl_work);
        tl_work = Work();
    }
};

// utils.cpp:171-179 compactionEnd: seq1 + (seq2 or its reverse complement) minus the k overlapping characters
static string compaction_end(const string& seq1, const string& seq2, unsigned k) {
    size_t s1 = seq1.size(), s2 = seq2.size();
    if (s1 == 0 || s2 == 0) return "";
    string rc2(reverse_complements(seq2)), end1(seq1.substr(s1 - k, k)), beg2(seq2.substr(0, k));
    if (end1 == beg2) return seq1 + seq2.substr(k);
    string begrc2(rc2.substr(0, k));
    if (end1 == begrc2) return seq1 + rc2.substr(k);
    return "";
}
// aligner.cpp:293-302 getUnitig + aligner.cpp:270-290 recoverPath (correction mode -c): the read as spelled by its path
static bool recover_path(const Oracle& o, const vector<unum_t>& numbers, unsigned size, string& out) {
    auto get_unitig = [&](int position) { return position > 0 ? o.unitigs[position] : reverse_complements(o.unitigs[-position]); };
    int offset = numbers[0];
    string path(get_unitig(numbers[1]));
    for (size_t i = 2; i < numbers.size(); ++i) {
        string inter(compaction_end(path, get_unitig(numbers[i]), o.k - 1));
        if (inter.empty()) return false;  // the reference prints "bug compaction" and exits (aligner.cpp:280-283)
        path = inter;
    }
    out = path.substr(offset, size);
    return true;
}

// aligner.cpp:600-609 printPath
static string print_path(const vector<unum_t>& path) {
    string res;
    for (size_t i = 0; i < path.size(); ++i) { res += std::to_string(path[i]); res += '.'; }
    res += '\n';
    return res;
}

// ------------------------------------------------------------------------------------------------
// Read parser + worker loop + CLI twin                 aligner.cpp:46-117, alignerGreedy.cpp:367-431,
//                                                      alignerExhaustive.cpp:262-318, bgreat.cpp:54-130
// ------------------------------------------------------------------------------------------------
struct Runner {
    Oracle& o;
    bool fastq = false, exhaustive_writes = false, correction = false;
    std::ifstream readFile;
    FILE* pathF = nullptr;
    FILE* notMappedF = nullptr;
    std::mutex readMutex, pathMutex, notMappedMutex;
    explicit Runner(Oracle& oo) : o(oo) {}

    static bool valid_chars(const string& r) {  // aligner.cpp:56-61 / :79-84 / :100-105
        for (char c : r) if (c != 'A' && c != 'C' && c != 'T' && c != 'G' && c != 'N') return false;
        return true;
    }
    // aligner.cpp:46-117 getReads; the same iostream calls in the same order, so getline/peek/eof
    // corner cases (phantom FASTQ record, missing trailing newline, multi-line FASTA) are inherited.
    void get_reads(vector<std::pair<string, string>>& reads, unsigned n) {
        reads.clear();
        string read, header, inter;
        if (fastq) {
            for (unsigned i = 0; i < n; ++i) {
                std::getline(readFile, header);
                std::getline(readFile, read);
                if (read.size() > 2 && valid_chars(read)) reads.push_back({header, read});
                std::getline(readFile, header);
                std::getline(readFile, header);
                if (readFile.eof()) return;
            }
        } else {
            for (unsigned i = 0; i < n; ++i) {
                std::getline(readFile, header);
                std::getline(readFile, read);
                for (;;) {
                    char c = (char)readFile.peek();
                    if (c == '>') {
                        if (read.size() > 2 && valid_chars(read) && read.size() > o.k) reads.push_back({header, read});
                        read = "";
                        break;
                    }
                    if (!readFile.eof()) {
                        std::getline(readFile, inter);
                        read += inter;
                    } else {
                        if (read.size() > 2 && valid_chars(read) && read.size() > o.k) reads.push_back({header, read});
                        return;
                    }
                }
            }
        }
    }
    // alignerGreedy.cpp:367-431 / alignerExhaustive.cpp:262-318
    void worker(int mode) {
        vector<std::pair<string, string>> batch;
        while (!readFile.eof()) {
            {
                std::lock_guard<std::mutex> g(readMutex);
                get_reads(batch, 10000);
            }
            for (auto& hr : batch) {
                uint8_t st;
                vector<unum_t> path = o.align_one(mode, hr.second, st);
                if (mode == 1 && !exhaustive_writes) continue;  // -b writes nothing
                if (!path.empty() && correction && mode != 1) {  // alignerGreedy.cpp:394-404
                    string corrected;
                    if (!recover_path(o, path, (unsigned)hr.second.size(), corrected)) { std::cout << "bug compaction" << std::endl; exit(0); }
                    if (st & 4) corrected = reverse_complements(corrected);
                    string rec = hr.first + '\n' + corrected + '\n';
                    std::lock_guard<std::mutex> g(pathMutex);
                    fwrite(rec.data(), 1, rec.size(), pathF);
                } else if (!path.empty()) {
                    string rec = hr.first + '\n' + print_path(path);
                    std::lock_guard<std::mutex> g(pathMutex);
                    fwrite(rec.data(), 1, rec.size(), pathF);
                } else {
                    string rec = hr.first + '\n' + hr.second + '\n';
                    std::lock_guard<std::mutex> g(notMappedMutex);
                    fwrite(rec.data(), 1, rec.size(), notMappedF);
                }
            }
        }
        o.flush_work();
    }
    bool run_file(const string& file, int mode, unsigned threads) {
        readFile.close();
        readFile.clear();
        readFile.open(file);
        if (!readFile) { fprintf(stderr, "bgreat_oracle: cannot open %s\n", file.c_str()); return false; }
        vector<std::thread> ts;
        for (unsigned t = 0; t < threads; ++t) ts.emplace_back(&Runner::worker, this, mode);
        for (auto& t : ts) t.join();
        return true;
    }
};

// ------------------------------------------------------------------------------------------------
// C interface for ctypes (tests, smoke, bench cpu_baseline).
// ------------------------------------------------------------------------------------------------
extern "C" {

void* orc_create_from_file(const char* unitig_path, int k) {
    Oracle* o = new Oracle();
    o->k = (unsigned)k;
    if (!o->load_unitig_file(unitig_path)) { delete o; return nullptr; }
    return o;
}
// seqs: concatenated unitig sequences, offs[n+1]
void* orc_create(int k, const char* seqs, const uint64_t* offs, uint64_t n) {
    Oracle* o = new Oracle();
    o->k = (unsigned)k;
    vector<string> v;
    v.reserve(n);
    for (uint64_t i = 0; i < n; ++i) v.emplace_back(seqs + offs[i], seqs + offs[i + 1]);
    o->index_unitigs(v);
    return o;
}
// Same with the anchors index of -G (dog != 0): mode 2 of orc_align needs it.
void* orc_create_from_file2(const char* unitig_path, int k, int dog) {
    Oracle* o = new Oracle();
    o->k = (unsigned)k;
    o->dogMode = dog != 0;
    if (!o->load_unitig_file(unitig_path)) { delete o; return nullptr; }
    return o;
}
void* orc_create2(int k, const char* seqs, const uint64_t* offs, uint64_t n, int dog) {
    Oracle* o = new Oracle();
    o->k = (unsigned)k;
    o->dogMode = dog != 0;
    vector<string> v;
    v.reserve(n);
    for (uint64_t i = 0; i < n; ++i) v.emplace_back(seqs + offs[i], seqs + offs[i + 1]);
    o->index_unitigs(v);
    return o;
}
void orc_destroy(void* h) { delete static_cast<Oracle*>(h); }
uint64_t orc_unitig_count(void* h) { return static_cast<Oracle*>(h)->unitigs.size() - 1; }

// Aligns reads [0,n) (reads_concat + offsets[n+1]); writes the flattened paths and CSR offsets.
// status[i]: 0 no anchor, 1 anchored-not-aligned, 2 aligned; bit 2 (value 4) = reverse-complement retry ran.
// Returns the number of path ints, or -1 if paths_cap is too small.  Single-threaded per call; callers
// may split the batch over threads themselves (orc_align is re-entrant on one handle).
int64_t orc_align(void* h, int mode, int m, int effort, int partial, const char* reads, const uint64_t* offs,
                  uint64_t n, int32_t* paths_out, uint64_t paths_cap, uint64_t* path_offs, uint8_t* status) {
    Oracle* o = static_cast<Oracle*>(h);
    o->errorsMax = (unsigned)m;
    o->tryNumber = (unsigned)effort;
    o->partial = partial != 0;
    o->exh_memo = mode == 3;  // mode 3: exhaustive with remembered calls (the checker for unitig sets on which the literal recursion is exponential)
    if (mode == 3) mode = 1;
    uint64_t w = 0;
    for (uint64_t i = 0; i < n; ++i) {
        string r(reads + offs[i], reads + offs[i + 1]);
        uint8_t st;
        vector<unum_t> p = o->align_one(mode, r, st);
        path_offs[i] = w;
        status[i] = st;
        if (w + p.size() > paths_cap) return -1;
        for (unum_t v : p) paths_out[w++] = v;
    }
    path_offs[n] = w;
    o->flush_work();
    return (int64_t)w;
}
// anchors of one read as getNOverlap returns them (kmer, pos); returns count (<= cap)
int orc_anchors(void* h, const char* read, uint64_t len, int effort, uint64_t* kmers, uint32_t* pos, int cap) {
    Oracle* o = static_cast<Oracle*>(h);
    auto l = o->get_n_overlap(string(read, read + len), (unsigned)effort);
    int c = 0;
    for (auto& a : l) { if (c < cap) { kmers[c] = a.first; pos[c] = a.second; ++c; } }
    return c;
}
// anchorsMPHF.lookup(kmer) and the anchorsPosition entry behind it (needs dog != 0 at creation).  Returns 0 and sets
// *index = ULLONG_MAX when the lookup answers "not a key".
int orc_anchor_lookup(void* h, uint64_t kmer, uint64_t* index, uint32_t* unitig, uint32_t* offset) {
    Oracle* o = static_cast<Oracle*>(h);
    Work saved = tl_work;
    uint64_t idx = o->anchorsMPHF.lookup(kmer);
    tl_work = saved;
    *index = idx;
    *unitig = *offset = 0;
    if (idx == ULLONG_MAX) return 0;
    if (idx >= o->anchorsPosition.size()) return -1;
    *unitig = o->anchorsPosition[idx].first;
    *offset = o->anchorsPosition[idx].second;
    return 1;
}
// counters: readNumber, noOverlapRead, alignedRead, notAligned, overlaps   (aligner.h:68)
void orc_counters(void* h, uint64_t out[5]) {
    Oracle* o = static_cast<Oracle*>(h);
    out[0] = o->readNumber; out[1] = o->noOverlapRead; out[2] = o->alignedRead; out[3] = o->notAligned; out[4] = o->overlaps;
}
void orc_reset_counters(void* h) {
    Oracle* o = static_cast<Oracle*>(h);
    o->readNumber = 0; o->noOverlapRead = 0; o->alignedRead = 0; o->notAligned = 0; o->overlaps = 0;
    o->work_total = Work();
}
// Work counters in declaration order of struct Work (13 values).
int orc_work(void* h, uint64_t* out, int cap) {
    Oracle* o = static_cast<Oracle*>(h);
    o->flush_work();
    int n = (int)(sizeof(Work) / 8);
    const uint64_t* s = reinterpret_cast<const uint64_t*>(&o->work_total);
    for (int i = 0; i < n && i < cap; ++i) out[i] = s[i];
    return n;
}
// ALGORITHMIC bytes summed over everything aligned since the last reset.
//   B_read = ceil(L/4); B_probe = 8*probes on non-empty levels; B_rank = 8*(1 + words) per level hit;
//   B_tab = 24*records; B_seq = ceil(bases/4) per Hamming call (+1 byte rounding per call, bounded by
//   calls) + 16 per candidate; B_out = 4*path ints + 8 per read.
double orc_alg_bytes(void* h) {
    Oracle* o = static_cast<Oracle*>(h);
    o->flush_work();
    const Work& w = o->work_total;
    double b = 0;
    b += (double)(w.read_bases + 3 * w.reads) / 4.0;  // sum ceil(L/4) <= (L+3)/4
    b += 8.0 * w.probes_nonempty;
    b += 8.0 * (w.level_hits + w.rank_words);
    b += 24.0 * w.tab_records;
    b += (double)w.mm_bases / 4.0 + 16.0 * w.unitig_fetch;
    b += 4.0 * w.path_ints + 8.0 * w.reads;
    return b;
}

// getReads (aligner.cpp:46-117) over a whole file, batches of 10000 like the worker loop: the accepted
// (header, read) records in order.  Two-call protocol: pass null buffers to obtain the sizes.
// out_sizes: [0]=records, [1]=read bytes, [2]=header bytes.
int orc_parse_file(const char* path, int fastq, int k, uint64_t* out_sizes, char* reads, uint64_t* read_offs, char* headers, uint64_t* header_offs) {
    Oracle o;
    o.k = (unsigned)k;
    Runner run(o);
    run.fastq = fastq != 0;
    run.readFile.open(path);
    if (!run.readFile) return -1;
    uint64_t n = 0, rb = 0, hb = 0;
    vector<std::pair<string, string>> batch;
    while (!run.readFile.eof()) {
        run.get_reads(batch, 10000);
        for (auto& hr : batch) {
            if (reads) {
                memcpy(headers + hb, hr.first.data(), hr.first.size());
                memcpy(reads + rb, hr.second.data(), hr.second.size());
                header_offs[n] = hb;
                read_offs[n] = rb;
            }
            hb += hr.first.size();
            rb += hr.second.size();
            ++n;
        }
    }
    if (reads) { header_offs[n] = hb; read_offs[n] = rb; }
    out_sizes[0] = n; out_sizes[1] = rb; out_sizes[2] = hb;
    return 0;
}

// CLI twin of bgreat.cpp:54-130.  Returns 0.  `exh_writes` != 0 makes -b write the paths it computes
// (the reference computes and discards them; compare with _ref/bgreat_exh).
int orc_main(int argc, char** argv, int exh_writes) {
    string reads, unitigs("unitig.fa"), pathFile("paths"), notAlignedFile("notAligned.fa");
    int errors = 2, threads = 1, ka = 30, effort = 2;
    bool brute = false, incomplete = false, fastq = false, correction = false, dog = false;
    for (int i = 1; i < argc; ++i) {  // same single-letter flags as getopt "r:k:g:m:t:e:f:o:a:biqpcG"
        string a = argv[i];
        auto val = [&](void) -> string { return (i + 1 < argc) ? string(argv[++i]) : string(); };
        if (a == "-r") reads = val();
        else if (a == "-k") ka = std::stoi(val());
        else if (a == "-g") unitigs = val();
        else if (a == "-m") errors = std::stoi(val());
        else if (a == "-t") threads = std::stoi(val());
        else if (a == "-e") effort = std::stoi(val());
        else if (a == "-f") pathFile = val();
        else if (a == "-a") notAlignedFile = val();
        else if (a == "-o") (void)val();
        else if (a == "-b") brute = true;
        else if (a == "-i") incomplete = true;
        else if (a == "-q") fastq = true;
        else if (a == "-c") correction = true;
        else if (a == "-G") dog = true;
    }
    if (reads.empty()) { printf("-r read_file\n"); return 0; }
    Oracle o;
    o.k = (unsigned)ka;
    o.errorsMax = (unsigned)errors;
    o.tryNumber = (unsigned)effort;
    o.partial = incomplete;
    o.dogMode = dog;
    o.exh_memo = getenv("ORACLE_EXH_MEMO") != nullptr;  // (-b: remembered calls, see Oracle::exh_memo)
    Runner run(o);
    run.fastq = fastq;
    run.correction = correction;
    run.exhaustive_writes = exh_writes != 0;
    run.pathF = fopen(pathFile.c_str(), "wb");          // aligner.h:85
    run.notMappedF = fopen(notAlignedFile.c_str(), "wb");  // aligner.h:86
    auto t0 = std::chrono::system_clock::now();
    o.load_unitig_file(unitigs);
    auto t1 = std::chrono::system_clock::now();
    std::cout << "Indexing in seconds : " << std::chrono::duration_cast<std::chrono::seconds>(t1 - t0).count() << std::endl;
    auto start = std::chrono::system_clock::now();
    size_t last = 0;
    unsigned nth = (unsigned char)threads;  // aligner.h:80 `unsigned char cores`
    for (size_t i = 0; i <= reads.size(); ++i) {
        if (i == reads.size() || reads[i] == ',') {
            string f = reads.substr(last, i - last);
            std::cout << f << std::endl;
            run.run_file(f, brute ? 1 : (dog ? 2 : 0), nth);  // -b selects alignPartExhaustive (aligner.cpp:563-567), where -G has no effect
            last = i + 1;
        }
    }
    uint64_t rn = o.readNumber, no = o.noOverlapRead, al = o.alignedRead, na = o.notAligned;
    std::cout << "The End" << std::endl;  // aligner.cpp:588-596
    std::cout << "Reads : " << rn << std::endl;
    std::cout << "No overlap : " << no << " Percent : " << (100 * float(no)) / rn << std::endl;
    std::cout << "Got overlap : " << al + na << " Percent : " << (100 * float(al + na)) / rn << std::endl;
    std::cout << "Overlap and aligned : " << al << " Percent : " << (100 * float(al)) / (al + na) << std::endl;
    std::cout << "Overlap but not aligned : " << na << " Percent : " << (100 * float(na)) / (al + na) << std::endl;
    auto end = std::chrono::system_clock::now();
    auto secs = std::chrono::duration_cast<std::chrono::seconds>(end - start).count();
    std::cout << "Reads/seconds : " << rn / (secs + 1) << std::endl;
    std::cout << "Mapping in seconds : " << secs << std::endl;
    fclose(run.pathF);
    fclose(run.notMappedF);
    if (getenv("ORACLE_WORK")) {
        double ab = orc_alg_bytes(&o);
        const Work& w = o.work_total;
        fprintf(stderr, "work: reads %llu lookups %llu probes_all %llu probes_nonempty %llu level_hits %llu rank_words %llu "
                "final_finds %llu tab_records %llu unitig_fetch %llu mm_calls %llu mm_bases %llu path_ints %llu alg_bytes_per_read %.1f\n",
                (unsigned long long)w.reads, (unsigned long long)w.lookups, (unsigned long long)w.probes_all,
                (unsigned long long)w.probes_nonempty, (unsigned long long)w.level_hits, (unsigned long long)w.rank_words,
                (unsigned long long)w.final_finds, (unsigned long long)w.tab_records, (unsigned long long)w.unitig_fetch,
                (unsigned long long)w.mm_calls, (unsigned long long)w.mm_bases, (unsigned long long)w.path_ints,
                w.reads ? ab / w.reads : 0.0);
    }
    return 0;
}
}  // extern "C"

#ifdef ORACLE_MAIN
int main(int argc, char** argv) {
    int exh = getenv("ORACLE_EXH_WRITES") ? 1 : 0;
    return orc_main(argc, argv, exh);
}
#endif
