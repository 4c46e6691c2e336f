// anchor_index.h -- host construction and lookup of the anchors index of -G mode (graph_layout.h, "anchors index").
// Replaces the `anchorsMPHF` / `anchorsPosition` pair of the reference (aligner.h:65-67, aligner.cpp:434-476).
// The MPHF is BooPHF's structure bit for bit, because the reference uses its answers for NON-keys too.
#ifndef BGREAT_AMD_ANCHOR_INDEX_H
#define BGREAT_AMD_ANCHOR_INDEX_H

#include <cstdint>
#include <vector>

#include "graph_layout.h"

namespace bgr {

struct AnchorMphf {
    uint64_t n = 0, last_rank = 0;
    uint32_t active_levels = 0;  // levels 0 .. active_levels-1 hold set bits
    BgrAncLevel levels[BGR_ANC_LEVELS];
    std::vector<uint64_t> bits, ranks;
    std::vector<uint64_t> final_kv;  // {key, index} pairs sorted by key (index excludes last_rank)
};

// boomphf::mphf<u64, SingleHashFunctor<u64>>(n, keys, threads, gamma = 10, ...) over `keys` in this order, repeats
// allowed (BooPHF.h:732-780): the same bit arrays, ranks and final-map indices as a single-threaded reference build.
void build_anchor_mphf(const std::vector<uint64_t>& keys, unsigned threads, AnchorMphf& out);

// boomphf::mphf::lookup (BooPHF.h:783-818) against the blob sections: the index of `key`, a false index for many
// non-keys, or ~0 (ULLONG_MAX).
uint64_t anchor_lookup(const BgrBlobHeader* h, const uint8_t* base, uint64_t key);

}  // namespace bgr
#endif
