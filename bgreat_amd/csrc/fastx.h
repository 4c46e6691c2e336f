// fastx.h -- host read parser with the reference's accept/drop behaviour (Aligner::getReads, aligner.cpp:46-117).
#ifndef BGREAT_AMD_FASTX_H
#define BGREAT_AMD_FASTX_H

#include <cstdint>
#include <string>
#include <vector>

namespace bgr {

struct ReadSet {
    std::vector<char> reads;       // concatenated accepted read sequences
    std::vector<uint64_t> read_offs;   // n+1
    std::vector<char> headers;     // concatenated header lines (verbatim, incl. '>' / '@')
    std::vector<uint64_t> header_offs;  // n+1
    uint64_t count() const { return read_offs.size() - 1; }
    void clear() { reads.clear(); headers.clear(); read_offs.assign(1, 0); header_offs.assign(1, 0); }
};

// Parses a whole file (memory image) and appends the accepted records to `out`.
// fastq=false: FASTA (multi-line sequences joined; record kept iff size>2, all chars in ACGTN, size>k).
// fastq=true : 4-line records (kept iff size>2 and all chars in ACGTN; no size>k test), including the
//              phantom record the reference emits at EOF (empty header, last sequence repeated) when the
//              file ends with a newline and the record count is not a multiple of the 10000-read batch.
void parse_reads(const char* data, uint64_t size, bool fastq, uint32_t k, ReadSet& out);
bool parse_reads_file(const std::string& path, bool fastq, uint32_t k, ReadSet& out, std::string& err);

}  // namespace bgr
#endif
