// text_kernels.h -- launch interface of text_kernels.hip (FASTA text in, formatted records out; see there).
#ifndef BGREAT_AMD_TEXT_KERNELS_H
#define BGREAT_AMD_TEXT_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "graph_layout.h"

// words of the small device `info` block of a text batch
#define TXT_INFO_N_REC 0      /* record starts found */
#define TXT_INFO_N_ACC 1      /* records accepted (aligner.cpp:78-88) */
#define TXT_INFO_BASES 2      /* bases of the accepted records */
#define TXT_INFO_MAX_LEN 3
#define TXT_INFO_IRREGULAR 4  /* the piece is not of the shape this route takes: the caller parses it on the host */
#define TXT_INFO_PBYTES 5     /* bytes of the paths stream */
#define TXT_INFO_NBYTES 6     /* bytes of the notAligned stream */
#define TXT_INFO_BUG 7        /* correction mode: the first accepted read whose path does not spell a walk ("bug compaction"), else ~0 */
#define TXT_INFO_WORDS 8
#define BGR_TEXT_EPOCH_MAX 0x3FFFFFu /* epochs of the chains: 22 bits, 0 = never written */

namespace bgr {

// exclusive scan of two u32 arrays of equal length at once, over their first min(n, *n_dev) entries (n_dev: a count on the device, or null); sums: 2 * scan_tiles(n) words
hipError_t launch_scan2_u32(const uint32_t* inA, const uint32_t* inB, uint32_t* outA, uint32_t* outB, uint32_t n, const uint32_t* n_dev, uint32_t* sums, uint32_t* totalA,
                            uint32_t* totalB, hipStream_t stream);
uint32_t scan_tiles(uint32_t n);
uint32_t text_tiles(uint32_t bytes);     // workgroups (= chain entries) of the parse launch
uint32_t format_tiles(uint32_t n_acc);   // ... of the format launch
// The piece in one launch: info[N_REC, N_ACC, BASES, MAX_LEN, IRREGULAR]; rec[0 .. min(N_REC, rec_cap)); the accepted records compacted in input order: acc_rec (which
// record), acc_src (where its read starts in the text), read_offs (base offsets, N_ACC + 1 of them); acc_idx[j] = a for accepted record j.
// fastq_lines: 0 FASTA, else lines per FASTQ record (4, 2).  Running totals pass between the launch's workgroups through `chains` (3 * text_tiles(n) u64), tagged
// with `epoch` (1 .. kTextEpochMax, a fresh one per launch; the chains are zero before epoch 1 is used), the workgroups numbered by *ticket - ticket_base (*ticket
// grows by text_tiles(n) per launch).  words_a words at zero_a and words_b at zero_b (at most 1024 each) are cleared: words only later launches read.
hipError_t launch_text_parse(const uint8_t* text, uint32_t n, uint32_t fastq_lines, uint32_t k, uint32_t* ticket, uint32_t ticket_base, uint32_t epoch, uint64_t* chains,
                             uint4* rec, uint32_t* acc_idx, uint32_t* acc_rec, uint32_t* acc_src, uint64_t* read_offs, uint32_t* info, uint32_t rec_cap, uint32_t* zero_a,
                             uint32_t words_a, uint32_t* zero_b, uint32_t words_b, hipStream_t stream);
// Sizes, stream offsets and the bytes of the records in one launch: info[PBYTES, NBYTES] (saturating at 2^32 - 1); poff/noff as launch_text_write takes them; the bytes
// where a workgroup's whole stretch ends below pcap / ncap.  chains: 2 * format_tiles(n_acc) u64; ticket / epoch as above.
hipError_t launch_text_format(const uint8_t* text, const uint2* results, const int32_t* arena, const uint4* rec, const uint32_t* acc_rec, uint32_t n_acc, uint32_t* ticket,
                              uint32_t ticket_base, uint32_t epoch, uint64_t* chains, uint32_t* poff, uint32_t* noff, uint8_t* pout, uint8_t* nout, uint64_t pcap, uint64_t ncap,
                              uint32_t* info, hipStream_t stream);
hipError_t launch_text_write(const uint8_t* text, const uint2* results, const int32_t* arena, const uint4* rec, const uint32_t* acc_rec, uint32_t n_acc,
                             const uint32_t* poff, const uint32_t* noff, uint8_t* pout, uint8_t* nout, hipStream_t stream);
// one word per record, record order: kept << 31 | mapped << 30 | read length (bgr_text_batch.record_info_out)
hipError_t launch_text_record_info(const uint4* rec, const uint32_t* acc_idx, const uint2* results, uint32_t n_rec, uint32_t* out, hipStream_t stream);
// correction mode (-c): mapped reads are written as header + the read spelled by its path (recoverPath, aligner.cpp:270-290)
hipError_t launch_text_correct_sizes(const BgrDeviceGraph& g, const uint2* results, const int32_t* arena, const uint4* rec, const uint32_t* acc_rec, uint32_t n_acc, uint32_t* psz,
                                     uint32_t* nsz, uint32_t* clen, uint32_t* bug, hipStream_t stream);
hipError_t launch_text_correct_write(const BgrDeviceGraph& g, const uint8_t* text, const uint2* results, const int32_t* arena, const uint4* rec, const uint32_t* acc_rec, uint32_t n_acc,
                                     const uint32_t* poff, const uint32_t* noff, const uint32_t* clen, uint8_t* pout, uint8_t* nout, hipStream_t stream);
// the pre-pass (batch_kernels.hip) over reads that lie scattered in a text: read r's characters start at reads + src_off[r]
hipError_t launch_pack_reads_at(const uint8_t* text, const uint32_t* src_off, const uint64_t* read_offs, uint32_t n, uint64_t text_bytes, uint64_t total_bases,
                                uint64_t* fw3, uint64_t* nmw, uint32_t* hasn, hipStream_t stream);

}  // namespace bgr

#endif
