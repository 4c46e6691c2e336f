// graph_build.h -- host-side construction of the graph blob (graph_layout.h) from unitig sequences.
// Replaces Aligner::indexUnitigsAux (aligner.cpp:407-534): same key sets, same slot fill order.
#ifndef BGREAT_AMD_GRAPH_BUILD_H
#define BGREAT_AMD_GRAPH_BUILD_H

#include <cstdint>
#include <string>
#include <vector>

#include "graph_layout.h"

namespace bgr {

// Anonymous-mmap storage: pages arrive zeroed and are first touched by whichever build thread fills them.
class ZeroPages {
public:
    ZeroPages() = default;
    ZeroPages(const ZeroPages&) = delete;
    ZeroPages& operator=(const ZeroPages&) = delete;
    ~ZeroPages() { release(); }
    bool reset(uint64_t words);  // releases, then maps `words` zeroed uint64 (false on failure)
    void release();
    uint64_t* data() { return p_; }
    const uint64_t* data() const { return p_; }
    bool empty() const { return words_ == 0; }
    uint64_t size() const { return words_; }
private:
    uint64_t* p_ = nullptr;
    uint64_t words_ = 0;
};

struct HostGraph {
    ZeroPages blob;  // page-aligned storage; header at blob[0]
    const BgrBlobHeader* header() const { return reinterpret_cast<const BgrBlobHeader*>(blob.data()); }
    uint64_t bytes() const { return header()->blob_bytes; }
    const uint8_t* base() const { return reinterpret_cast<const uint8_t*>(blob.data()); }
};

// seqs/offs: n unitig sequences in file order (offs[n+1]).  Loading stops at the first sequence shorter
// than k, as aligner.cpp:418-420 does.  gamma: key table slots per key (0 = choose: 1.07 when the table fits LDS staging, else 1.8).
// Returns false (and sets err) on invalid arguments / limits.
// flags: BGR_BUILD_ANCHORS adds the anchors index of -G mode (graph_layout.h).
#define BGR_BUILD_ANCHORS 1u
#define BGR_BUILD_NO_EVICTIONS 2u  // test hook (include/bgreat_gpu.h)
bool build_graph(uint32_t k, uint64_t n, const char* seqs, const uint64_t* offs, double gamma, uint32_t flags, HostGraph& out, std::string& err);

// Host threads used by build_graph / read_unitig_fasta (0 = default: the machine's cores, at most 16).  The blob
// does not depend on the thread count.
void set_build_threads(unsigned t);
unsigned build_threads();

// Reads a unitig FASTA the way aligner.cpp:415-417 does: two lines per record, header ignored.
bool read_unitig_fasta(const std::string& path, uint32_t k, std::vector<char>& seqs, std::vector<uint64_t>& offs, std::string& err);

// Host twin of the device lookup (same arithmetic): MPHF index of `key`, or BGR_NONE.  The caller still has
// to compare keys[idx] with key for membership.
uint32_t host_lookup(const BgrBlobHeader* h, const uint8_t* base, uint64_t key);

// Fill a BgrDeviceGraph whose pointers are `base` + section offsets (base may be a device address).
void resolve_device_graph(const BgrBlobHeader* h, const void* base, BgrDeviceGraph& dg);

// A blob is trusted only after this: magic/version, every section inside `bytes` (overflow-checked), the MPHF level table
// tiling the unit array, key counts consistent.  validate_blob_header needs the header only (a blob that lives in HBM).
bool validate_blob_header(const BgrBlobHeader* h, uint64_t bytes, std::string& err);
bool validate_blob(const void* blob, uint64_t bytes, std::string& err);

}  // namespace bgr

#endif
