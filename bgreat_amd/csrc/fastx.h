// fastx.h -- host read parser with the reference's accept/drop behaviour (Aligner::getReads, aligner.cpp:46-117).
#ifndef BGREAT_AMD_FASTX_H
#define BGREAT_AMD_FASTX_H

#include <cstdint>
#include <string>
#include <memory>
#include <utility>
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

// One accepted record as slices: the header always points into the file image; the sequence points into the file
// image (single-line record) or into the owning chunk's `joined` storage (multi-line FASTA record).
struct RecSlice {
    const char* h;
    const char* s;
    uint32_t hl, sl;
};

struct ParsedChunk {
    std::vector<RecSlice> recs;
    std::string joined;          // storage of multi-line sequences; recs[].s of those are OFFSETS until fixup()
    std::vector<uint32_t> joined_idx;  // indices of recs whose s is an offset into `joined`
    uint64_t seq_bytes = 0, hdr_bytes = 0;
    // getReads() loop iterations (aligner.cpp:51,70: one per record ATTEMPT, accepted or not) this chunk consumed, and,
    // when track_iters is set, the iteration each accepted record came from (chunk-relative).  The reference's
    // 10000-iteration calls are what its exhaustive worker counts for the periodic stdout block
    // (alignerExhaustive.cpp:306-316); calls of a file = ceil(iterations / 10000).
    uint64_t iters = 0;
    bool track_iters = false;
    std::vector<uint32_t> rec_iter;
    void fixup() {  // turn offsets into pointers once `joined` no longer grows
        for (uint32_t i : joined_idx) recs[i].s = joined.data() + reinterpret_cast<uintptr_t>(recs[i].s);
        joined_idx.clear();
    }
};

// Parses a whole file (memory image) and appends the accepted records to `out`.
// fastq=false: FASTA (multi-line sequences joined; record kept iff size>2, all chars in ACGTN, size>k).
// fastq=true : 4-line records (kept iff size>2 and all chars in ACGTN; no size>k test), including the
//              phantom record the reference emits at EOF (empty header, last sequence repeated) when the
//              file ends with a newline and the record count is not a multiple of the 10000-read batch.
void parse_reads(const char* data, uint64_t size, bool fastq, uint32_t k, ReadSet& out);
bool parse_reads_file(const std::string& path, bool fastq, uint32_t k, ReadSet& out, std::string& err);

// ---- chunk-parallel form (same records, same order) --------------------------------------------------------
// FASTA: the sequential reader's state after a record is "next line is a header", and a line that starts with
// '>' ends the current record exactly when the line before it does not start with '>' (a '>' line right after a
// header is consumed as sequence, aligner.cpp:72-73).  So every such position, except one whose previous line is
// line 0 of the file, is a point where an independent reader can start.  split_fasta returns chunk start
// offsets (first = 0) about `chunk_bytes` apart at such positions.
std::vector<uint64_t> split_fasta(const char* data, uint64_t size, uint64_t chunk_bytes);
// the first such position at or behind `from` (size: none)
uint64_t fasta_cut_at(const char* data, uint64_t size, uint64_t from);
// The reference's FASTA state machine over [begin, end) of the image, `end` acting as end-of-file.
void parse_fasta_chunk(const char* data, uint64_t begin, uint64_t end, uint32_t k, ParsedChunk& out);
// Whole-image FASTQ, sequential.
void parse_fastq_image(const char* data, uint64_t size, ParsedChunk& out);
// FASTQ in parallel.  Record j is lines 4j..4j+3 whatever they contain, so a first pass counts the newlines of
// every chunk (record starts follow from the prefix sums); all records before the last 10000-record boundary are
// plain "header line + sequence line" (the stream is good there), parsed by `threads` threads into chunks[0..n-2];
// the rest of the file, which starts exactly where a fresh getReads() call starts, goes through the sequential
// state machine (phantom record, truncated tails) into the last chunk.
void parse_fastq_parallel(const char* data, uint64_t size, unsigned threads, uint64_t chunk_bytes, std::vector<ParsedChunk>& chunks);
// The same in steps, for a caller that overlaps parsing with what follows: count_chunk(c) for every chunk (any order,
// any threads), finish_counts(), then parse_chunk(c, out) in any order -- it returns false once chunk c reaches the
// sequential tail (no later chunk needs parsing then) -- and parse_tail(out) last (false: nothing was left for it).
class FastqPlan {
public:
    FastqPlan(const char* data, uint64_t size, uint64_t chunk_bytes);
    size_t chunks() const { return nc_; }
    void count_chunk(size_t c);
    void finish_counts();
    // (streaming: once chunks 0 .. c_end-1 are counted, extend_counts(c_end) makes record_offset() usable for the records that end in them;
    // records_counted() of them are complete so far, and every getReads() call boundary at or below records_counted() - 1 is final)
    void extend_counts(size_t c_end);
    uint64_t records_counted() const { return nl_[summed_] / 4; }
    bool sequential_only() const { return complete_ == 0 || par_records_ == 0; }  // everything goes through parse_tail
    bool parse_chunk(size_t c, ParsedChunk& out);
    bool parse_tail(ParsedChunk& out);
    // (text route) records 0 .. par_records()-1 are plain four-line records in front of the file's last getReads() call boundary;
    // record_offset(r), r <= par_records(): the byte where line 4r starts (finish_counts() must have run)
    uint64_t par_records() const { return par_records_; }
    uint64_t record_offset(uint64_t rec) const;
    // (text route) the piece [o0, o1) of whole records, o0 = record_offset(r0), cut where the plan's chunks begin: {first byte, number of
    // the line that byte lies in}, ascending -- what fastq_gather_lines needs to work on the parts independently
    void piece_parts(uint64_t o0, uint64_t o1, uint64_t r0, std::vector<std::pair<uint64_t, uint64_t>>& out) const;
    // keep_newlines(): count_chunk also KEEPS where the newlines of its chunk are (offsets from the chunk's first byte), so that the gather
    // behind it copies the kept lines without scanning the text a second time; newlines_of(c) hands the list of chunk c to a batch,
    // release_newlines(c_end) drops the plan's own hold on the chunks in front of c_end (the batches in flight keep theirs alive)
    void keep_newlines() { keep_nl_ = true; nlpos_.resize(nc_); }
    std::shared_ptr<const std::vector<uint32_t>> newlines_of(size_t c) const { return c < nlpos_.size() ? nlpos_[c] : nullptr; }
    void release_newlines(size_t c_end) { for (; released_ < c_end && released_ < nlpos_.size(); ++released_) nlpos_[released_].reset(); }
    uint64_t chunk_bytes() const { return chunk_bytes_; }
private:
    const char* data_;
    uint64_t size_, chunk_bytes_;
    size_t nc_ = 0, summed_ = 0;
    std::vector<uint64_t> nl_, tail_start_;
    uint64_t complete_ = 0, par_records_ = 0;
    bool keep_nl_ = false;
    size_t released_ = 0;
    std::vector<std::shared_ptr<const std::vector<uint32_t>>> nlpos_;
};
// Bytes [b, e) of a FASTQ image, byte b lying in line number `line` of the file (record j = lines 4j .. 4j+3, aligner.cpp:51-68): those
// that belong to header lines (4j) and read lines (4j+1), newlines included, copied to dst in order -> bytes written (<= e - b).
// The '+' and quality lines never reach the device this way: a 150 bp record crosses PCIe as ~165 instead of ~317 bytes.
uint64_t fastq_gather_lines(const char* data, uint64_t b, uint64_t e, uint64_t line, char* dst);
// the same from the newline positions a FastqPlan kept (nl[0 .. n): ascending offsets from `nl_base`, covering at least [b, e)): no second scan
uint64_t fastq_gather_lines_at(const char* data, uint64_t b, uint64_t e, uint64_t line, char* dst, const uint32_t* nl, size_t n, uint64_t nl_base);
// Whole four-line records in [begin, end) (end = the start of a record): header line + sequence line, kept iff size > 2 and ACGTN
// (what every getReads() call but a file's last one does, aligner.cpp:51-68).
void parse_fastq_records(const char* data, uint64_t begin, uint64_t end, ParsedChunk& out);
// The sequential state machine from `begin` to the end of the image (a getReads() call boundary): phantom record, truncated tails.
void parse_fastq_from(const char* data, uint64_t begin, uint64_t size, ParsedChunk& out);
// Parallel whole-file parse into a ReadSet (threads >= 1); identical result to parse_reads().
void parse_reads_parallel(const char* data, uint64_t size, bool fastq, uint32_t k, unsigned threads, uint64_t chunk_bytes, ReadSet& out);

}  // namespace bgr
#endif
