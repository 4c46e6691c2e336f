// graph_layout.h -- the immutable de Bruijn graph index as ONE position-independent blob, laid out for
// gfx950.  Built on the host (graph_build.cpp), copied to HBM once, broadcast between GPUs as raw bytes.
// Shared by host code (g++) and device code (hipcc); no HIP types in here.
//
// What the reference keeps in std:: containers (aligner.h:65-70: vector<string> unitigs, two boomphf::mphf,
// two vector<unitigIndices>) becomes:
//
//   seq    2-bit packed bases (A0 C1 G2 other 3 == str2num, utils.cpp:117-129), 32 bases per u64, FIRST
//          base in the MOST significant bits so a (k-1)-window read out of the stream IS the reference's
//          k-mer integer.  Every unitig is stored twice, forward strand at base offset F and its
//          reverse complement (utils.cpp:66-73: non-ACG -> 'A') right behind it at F+len, so an oriented
//          unitig is just (base offset, length) and no kernel ever reverses bits.
//   exc/excN  optional 1-bit-per-base planes over the same base index space: forward-strand bases that
//          are not ACGT (they can never equal a read's ACGT; an 'N' can only equal a read's 'N').  Absent
//          (offset 0) for ordinary graphs.
//   meta   32 B per unitig id: len, and the record index + canonical flag of BOTH end (k-1)-mers, so
//          a walk step never hashes: the next neighbour record is a direct index.
//   table  ONE key table over the union of the reference's left and right overlap key sets (index values are
//          unobservable): two-choice bucketed hashing with one-byte fingerprints.  A bucket is one
//          dword = 4 slots; a slot holds 0 (empty) or the fingerprint 1..255 of the key that lives there, and the slot
//          number (4 * bucket + slot) IS the key's index into `keys` and `recs`.  A query reads its two buckets (two
//          independent dword loads, no dependent second access), compares the four bytes of each with its fingerprint
//          and confirms a match against `keys`; a non-member is rejected by the fingerprints alone 97 times in 100 --
//          most read positions are not overlaps -- so the 8-byte key load is left to the members and a few false matches.
//          About 8.6 bits per key at the default fill (0.935): the table of an E. coli-scale graph fits LDS twice per CU.
//          Keys that found no slot after the eviction budget (in practice none) go to a sorted fallback list.
//   keys   16 B per table slot: the u64 key (~0 for an empty slot: membership check, aligner.cpp:158,219,353,361) and the
//          HANDLES of the key's two halves -- where its "left table" slots and its "right table" slots start in `slots`.
//   slots  the neighbour records, COMPACT (round 3; round 2 kept 256 B per table slot, 8 x 32 B of which 1-3 were filled
//          and 45 % of the table slots of a large graph hold no key at all: 1.2 of the chr1-scale blob's 1.6 GB): the
//          filled "left table" slots of a key (aligner.h:49-55 indice1..4, filled in unitig order with slot-4 overwrite,
//          aligner.cpp:466-533) stand next to each other, the last one flagged; likewise its "right table" slots; a half
//          without any slot has no storage (handle BGR_HNONE).  A slot is 32 B: {id | orientation bits, len, F as seq
//          word + base-in-word} -- everything a walk step needs to start streaming the candidate's bases -- plus where
//          the walk goes NEXT: the handle (and canonical flag) of the half the following step reads, one for each of
//          the two ways a slot can be reached (by a canonical or a non-canonical query), so a step is slot -> bases
//          (two dependent loads) and never goes back to the key table.  Bits 30/31 of the id word carry
//          the orientation the reference recomputes by string compare at query time (aligner.cpp:174,235).
#ifndef BGREAT_AMD_GRAPH_LAYOUT_H
#define BGREAT_AMD_GRAPH_LAYOUT_H

#include <stdint.h>

#if defined(__HIP__)
#define BGR_HD __host__ __device__ __forceinline__
#else
#define BGR_HD inline
#endif

#define BGR_MAGIC 0x3130484752474742ULL /* "BGGRGH01" */
#define BGR_BLOB_VERSION 13u /* 13: minimizer-blocked filter in front of large key tables; 12: one-hash Bloom filter there; 11: compact slots + half handles; 10: slots carry the 32 bases next to the overlap; 9: fingerprint key table instead of the MPHF cascade; 8: slot_fill_x100; 7: anchors levels with division magic */
#define BGR_EMPTY_KEY 0xFFFFFFFFFFFFFFFFULL /* keys[] of an empty table slot: no (k-1)-mer, k <= 32, has bit 62 or 63 set */
#define BGR_NONE 0xFFFFFFFFu
#define BGR_SLOT_ID_MASK 0x3FFFFFFFu
#define BGR_HNONE 0x0FFFFFFFu      /* handle of a half without slots (28 bits; bit 28 of a handle word = "the query is canonical") */
#define BGR_H_CANON (1u << 28)
// slot flag bits (see graph_build.cpp fill_records):
//   left-table slot : bit30 = unitig forward when asked "who BEGINS with key"      (getBegin(key))
//                     bit31 = unitig forward when asked "who ENDS with rc(key)"    (getEnd(rc key))
//   right-table slot: bit30 = unitig forward when asked "who ENDS with key"        (getEnd(key))
//                     bit31 = unitig forward when asked "who BEGINS with rc(key)"  (getBegin(rc key))
#define BGR_SLOT_F0 0x40000000u
#define BGR_SLOT_F1 0x80000000u

// meta flags
#define BGR_META_CANON_BEG 1u    /* beg <= rc(beg): forward unitig, walking left, next getEnd key is canonical   */
#define BGR_META_CANON_END 2u    /* end <= rc(end): forward unitig, walking right, next getBegin key is canonical */
#define BGR_META_CANON_RCBEG 4u  /* rc(beg) <= beg: reversed unitig, walking right                                */
#define BGR_META_CANON_RCEND 8u  /* rc(end) <= end: reversed unitig, walking left                                 */

typedef struct {
    uint32_t len;     // bases
    uint32_t flags;   // BGR_META_*
    uint32_t rec_beg; // key index (table slot) of canonical(first k-1 bases)
    uint32_t rec_end; // key index of canonical(last k-1 bases)
    uint64_t F;       // base offset of the forward strand in `seq` (reverse complement at F + len)
    uint32_t hw[2];   // unused (0)
} BgrUnitigMeta;      // 32 B (anchors mode and the correction formatter read it; a walk step does not)

typedef struct {
    uint32_t idf;     // unitig id | BGR_SLOT_F0 | BGR_SLOT_F1 ; 0 = empty slot
    uint32_t len;
    uint32_t Fw;      // forward strand starts at base Fo of seq word Fw  (F = 32*Fw + Fo): 32-bit address arithmetic
    uint32_t Fo_x;    //   in the kernels (seq must stay below 4 GiB = 2^34 bases, checked at build time).  Bits 0..4 = Fo,
                      //   bit 5 = BGR_SLOT_LAST (the last slot of its half), bits 8..11 = bits 0..3 of `near` (below)
    uint32_t mflags_x;// bits 0..3 = BGR_META_* of this unitig; bits 4..31 = bits 4..31 of `near`'s high word
    uint32_t nx0;     // where the walk goes on behind this unitig: handle | BGR_H_CANON of the half the NEXT step reads, when this
    uint32_t nx1;     //   slot was reached by a canonical query (nx0) / by a non-canonical one (nx1) -- the direction of the walk and the
                      //   unitig's orientation follow from (table side, canonical or not), see graph_build.cpp next_half()
    uint32_t near_lo; // `near` (64 bits, bgr_slot_near): the <= 32 unitig bases NEXT TO the overlap this slot hangs on, outside it,
                      //   read away from the overlap in the orientation in which the unitig BEGINS with the key (or with the
                      //   key's reverse complement), first base most significant, zero beyond the unitig's end.  A walk to the
                      //   right compares exactly these bases, a walk to the left their reverse complement, so a step over
                      //   a short unitig (or a short rest of the read) never touches `seq`.  Valid when exactly one of F0/F1
                      //   is set (a hairpin unitig hangs on its overlap both ways) and the graph has no exception planes.
} BgrSlot;            // 32 B
#define BGR_SLOT_FO_MASK 31u
#define BGR_SLOT_LAST 32u
typedef struct {
    uint64_t key;     // BGR_EMPTY_KEY for an empty table slot
    uint32_t hL, hR;  // handles of the key's left-table / right-table slots in `slots` (BGR_HNONE: none)
} BgrKeyEntry;        // 16 B per table slot (and per fallback key)
BGR_HD uint64_t bgr_slot_near(uint32_t Fo_x, uint32_t mflags_x, uint32_t near_lo) {
    return ((uint64_t)((mflags_x & 0xFFFFFFF0u) | ((Fo_x >> 8) & 15u)) << 32) | near_lo;
}

// ---- anchors index (-G, optional) --------------------------------------------------------------------
// The reference's anchors mode looks read k-mers up in a boomphf::mphf over the canonical k-mers of all unitigs
// and uses the answer WITHOUT a key check (aligner.cpp:387-389): a non-key that lands on a set bit gets the rank
// of that bit as its "index".  Those indices are therefore observable, and this section is BooPHF's structure bit
// for bit (BooPHF.h:425-660 bitVector, :732-780 constructor, :1010-1054 setup, gamma 10, 25 levels):
//   anc_bits   per level 1 + domain/64 u64 words, bit p of a level = word p>>6, bit p&63 (LSB first)
//   anc_ranks  per level one u64 per 8 words: set bits before that block, counted over all levels so far
//   anc_final  keys left after 24 levels (in practice: repeated k-mers), sorted {key, index} pairs
//   anc_pos    per index (unitig id << 32 | offset of the k-mer in that unitig), 0 = never written
#define BGR_ANC_LEVELS 25
typedef struct {
    uint64_t domain;     // hash domain of the level (multiple of 64)
    uint64_t word_base;  // first u64 of the level in anc_bits
    uint64_t rank_base;  // first u64 of the level in anc_ranks
    uint64_t magic;      // floor(2^64 / domain): h % domain by one 64x64 high multiply and at most two subtractions
} BgrAncLevel;

// Blob header (first 4096 bytes of the blob).  All section offsets are bytes from the blob start and
// multiples of 256.
typedef struct {
    uint64_t magic;
    uint32_t version, k;
    uint64_t blob_bytes;
    uint64_t n_unitigs;     // ids 1..n_unitigs (meta has n_unitigs+1 entries, entry 0 unused)
    uint64_t n_keys;        // index space of keys[] and recs[]: 4 * n_buckets table slots, then the fallback list's entries
    uint64_t n_placed;      // keys that live in the table; the rest (normally none) sit in the sorted fallback list
    uint64_t n_fallback;
    uint64_t seq_words;     // u64 words in seq (incl. 2 trailing pad words)
    uint64_t total_bases;   // 2 * sum(len)
    uint64_t n_buckets;     // 4-slot buckets (one dword each) of the key table
    uint64_t off_table, off_keys, off_recs, off_meta, off_seq, off_exc, off_excn, off_fallback;  // off_recs: the compact `slots`
    uint32_t filter_kind, has_exc;       // filter_kind: BGR_FILTER_* of the section at off_bloom
    uint64_t max_unitig_len;
    uint64_t n_left_keys, n_right_keys;  // sizes of the reference's two key sets (informational)
    uint32_t slot_fill_x100, pad0;       // 100 x mean number of filled slots per non-empty half record (how branchy the graph is)
    uint64_t n_slots;       // entries of `slots` (filled slots of all halves; 4 zero entries follow them)
    uint64_t off_bloom;     // large graphs (key table probed in L2/HBM, not staged in LDS): a filter over the keys in front of the table, so
    uint64_t bloom_bits;    //   that most read positions (no overlaps) are turned away before a table probe; bits = a power of two, 0 = none.
                            //   BGR_FILTER_MINIMIZER (k-1 >= 20): 64-byte blocks chosen by the (k-1)-mer's minimizer -- consecutive read
                            //   positions share it, so the lanes of a scan read a few lines instead of one line each (see bgr_mmx_*);
                            //   BGR_FILTER_FLAT (shorter k): one hash, 4-8 bits per key
    double gamma;           // table slots per key
    // anchors index (all zero when the graph was built without it)
    uint64_t anc_n;          // anchors = k-mers of all unitigs but each unitig's last, repeats included (aligner.cpp:434-442)
    uint64_t anc_last_rank;  // set bits over all levels; indices of anc_final entries start here
    uint64_t anc_n_final;
    uint64_t anc_words, anc_rank_words;
    uint64_t anc_active_levels;  // levels 0 .. anc_active_levels-1 hold set bits; the rest are all zero and cannot answer
    uint64_t off_anc_bits, off_anc_ranks, off_anc_final, off_anc_pos;
    BgrAncLevel anc_levels[BGR_ANC_LEVELS];
} BgrBlobHeader;

// What a kernel receives by value (pointers resolved against the device copy of the blob).  Kept small on
// purpose: everything here lives in SGPRs for the whole kernel; what only rare paths need (exception planes,
// fallback list, level table) is reached through `hdr`, the blob header in HBM.
#define BGR_GF_HAS_EXC 1u
#define BGR_GF_HAS_FALLBACK 2u
typedef struct {
    const uint32_t* table;   // n_buckets dwords: 4 one-byte fingerprints each (slot s = byte s), 0 = empty
    const BgrKeyEntry* keys; // n_keys entries
    const BgrSlot* recs;     // n_slots compact slots
    const BgrUnitigMeta* meta;
    const uint64_t* seq;
    const BgrBlobHeader* hdr;
    uint32_t k, n_buckets, flags, table_bytes;  // table_bytes = 4 * n_buckets (LDS staging)
    const uint32_t* bloom;   // null = none
    uint32_t bloom_mask;     // BGR_FILTER_FLAT: bloom_bits - 1; BGR_FILTER_MINIMIZER: 32 - log2(number of blocks)
    uint32_t filter_kind;
} BgrDeviceGraph;

// ---- hashing shared by the host builder and the device lookup ---------------------------------------
// Key -> 64 hash bits: fold the high word into the low one, one 64-bit multiply.  Of m = bgr_mix64(key) the table uses
// bucket 1 = mulhi32(low word, n_buckets) and bucket 2 = mulhi32(high word, n_buckets) -- decided by the words' HIGH bits,
// which a multiplication mixes best -- and the fingerprint = the low byte of the HIGH word, 0 mapped to 1.  (Membership is
// decided by the key compare, never by the hash: a weak hash could only cost table fill or false fingerprint matches.)
BGR_HD uint64_t bgr_mix64(uint64_t x) {
    x ^= x >> 32;
    return x * 0x9E3779B97F4A7C15ULL;
}
BGR_HD uint32_t bgr_tab_bucket(uint32_t h, uint32_t n_buckets) { return (uint32_t)(((uint64_t)h * (uint64_t)n_buckets) >> 32); }
BGR_HD uint32_t bgr_bloom_bit(uint64_t m, uint32_t mask) { return (uint32_t)(m >> 20) & mask; }  // (bits of both words of the hash)
// ---- the minimizer-blocked filter (BGR_FILTER_MINIMIZER) ------------------------------------------------------------------
// A scan asks for the (k-1)-mers at ~120 consecutive positions of a read and nearly all of them are no keys.  With one hashed bit per
// key every one of those probes is its own cache line of a table far larger than the L2 (52-80 G probes/s on an MI355X whatever
// the kernel does, profiles/r03_probe_locality.txt).  Here the 64-byte block of a key is chosen by the key's MINIMIZER: of the
// k-16 16-mers of the (k-1)-mer (canonical form: the smaller of the 16-mer and its reverse complement, so both strands agree), the one
// with the largest hash.  Neighbouring positions share their minimizer (runs of ~(k-14)/2 positions), so the lanes of a scan touch a
// handful of lines (x5-10 probes/s).  Inside the block: one of 16 words and two of its 32 bits, from the key's own hash.
#define BGR_FILTER_NONE 0u
#define BGR_FILTER_FLAT 1u
#define BGR_FILTER_MINIMIZER 2u
#define BGR_MMX_BASES 16u      // bases of a minimizer
#define BGR_MMX_MIN_K1 20u     // shortest (k-1)-mer the filter is built for (window of at least 5 16-mers)
BGR_HD uint32_t bgr_rc16(uint32_t x) {  // reverse complement of 16 bases (2-bit codes A0 C1 G2 T3: complement = ~code)
#if defined(__HIP_DEVICE_COMPILE__)
    x = __builtin_bitreverse32(x);
    return (uint32_t)__builtin_amdgcn_bitop3_b32((int)(x >> 1), (int)(x << 1), 0x55555555, 0x1B);  // pairs swapped and complemented: ~(mask ? a : b)
#else
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = (x >> 24) | ((x >> 8) & 0xFF00u) | ((x << 8) & 0xFF0000u) | (x << 24);
#endif
    return ~x;
}
BGR_HD uint32_t bgr_mmx_hash(uint32_t x) {  // of the 16-mer x; never 0 (0 = "no 16-mer here" in the device's window maximum)
    const uint32_t r = bgr_rc16(x);
    return ((x < r ? x : r) * 0x9E3779B1u) | 1u;
}
BGR_HD uint32_t bgr_mmx_of_key(uint64_t key, uint32_t K1) {  // key: K1 bases, right aligned (either strand gives the same value)
    uint32_t best = 0;
    for (uint32_t j = 0; j + BGR_MMX_BASES <= K1; ++j) {
        const uint32_t h = bgr_mmx_hash((uint32_t)(key >> (2 * (K1 - BGR_MMX_BASES - j))));
        best = h > best ? h : best;
    }
    return best;
}
BGR_HD uint32_t bgr_mmx_block(uint32_t mh, uint32_t shift) { return (mh * 0x85EBCA6Bu) >> shift; }  // 2^(32 - shift) blocks
BGR_HD uint32_t bgr_mmx_word(uint64_t m) { return (uint32_t)(m >> 8) & 15u; }                       // m = bgr_mix64(key)
BGR_HD uint32_t bgr_mmx_bits(uint64_t m) { return (1u << ((uint32_t)(m >> 12) & 31u)) | (1u << ((uint32_t)(m >> 17) & 31u)); }
BGR_HD uint32_t bgr_tab_fp(uint64_t m) { const uint32_t f = (uint32_t)(m >> 32) & 0xFFu; return f ? f : 1u; }
// bit 7 of every byte of x that is zero (exact: no borrow between bytes)
BGR_HD uint32_t bgr_zero_bytes(uint32_t x) { return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u; }

// ---- BooPHF's hashing, needed bit-exact by the anchors index (BooPHF.h:251-264 hash64, :336-356 the level hashes)
BGR_HD uint64_t bgr_boo_hash64(uint64_t key, uint64_t seed) {
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
#define BGR_BOO_SEED0 0xAAAAAAAA55555555ULL
#define BGR_BOO_SEED1 0x33333333CCCCCCCCULL
// level 0 and 1 hash with the two seeds; level >= 2 is xorshift128+ on the state (s0, s1) = (h0, h1)
BGR_HD uint64_t bgr_boo_next(uint64_t* s0, uint64_t* s1) {
    uint64_t a = *s0;
    const uint64_t b = *s1;
    *s0 = b;
    a ^= a << 23;
    *s1 = a ^ b ^ (a >> 17) ^ (b >> 26);
    return *s1 + b;
}

// h % d for d >= 2 with magic = floor(2^64 / d): q = mulhi(h, magic) underestimates floor(h / d) by at most 2
BGR_HD uint64_t bgr_mod_magic(uint64_t h, uint64_t d, uint64_t magic) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t q = __umul64hi(h, magic);
#else
    const uint64_t q = (uint64_t)(((unsigned __int128)h * magic) >> 64);
#endif
    uint64_t r = h - q * d;
    if (r >= d) r -= d;
    if (r >= d) r -= d;
    return r;
}

// reverse complement of a (k-1)-digit base-4 number == utils.cpp:182-192 rcb(), by bit tricks
BGR_HD uint64_t bgr_rev2(uint64_t x) {  // reverse the order of the 32 2-bit digits of x
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFULL) | ((x & 0x00FF00FF00FF00FFULL) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFULL) | ((x & 0x0000FFFF0000FFFFULL) << 16);
    return (x >> 32) | (x << 32);
}
BGR_HD uint64_t bgr_rcb(uint64_t x, uint32_t n) { return (~bgr_rev2(x)) >> (64 - 2 * n); }

#endif
