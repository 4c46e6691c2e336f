/* bgreat_gpu.h -- C-ABI of libbgreat_gpu.so: the MI355X (gfx950) implementation of BGREAT's per-read
 * de Bruijn-graph mapping path.  Plain pointers and sizes only; no C++/HIP/torch types cross this boundary.
 *
 * The reference (Malfoy/BGREAT) has no plugin/FFI interface.  Its in-process boundary for this path is the
 * worker body alignerGreedy.cpp:378-392 / alignerExhaustive.cpp:273-281, i.e. the member functions
 *
 *     vector<uNumber> Aligner::alignReadGreedy(const string& read, bool& overlapFound, uint errors, bool& rc)
 *                                                       (aligner.h:127, alignerGreedy.cpp:35-57)
 *     vector<uNumber> Aligner::alignReadExhaustive(const string& read, bool& overlapFound, uint errors)
 *                                                       (aligner.h:135, alignerExhaustive.cpp:35-58)
 *
 * called once per read on an immutable Aligner built by Aligner::Aligner + indexUnitigs()
 * (aligner.h:80-105, aligner.cpp:407-547), with side effects only on the counters of aligner.h:68.
 * One-read calls cannot feed a GPU, so every entry point below is the batch form of one of those; each
 * states the reference interface it replaces.  Results are bit-identical to the reference for the same
 * reads in the same order.
 *
 * Conventions: every function returns 0 on success, a negative BGR_E* code otherwise; bgr_last_error()
 * gives the text (thread-local).  Handles are opaque.  Nothing here falls back to a CPU implementation:
 * without a usable HIP device the device functions fail with BGR_E_HIP.
 */
#ifndef BGREAT_GPU_H
#define BGREAT_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BGR_OK 0
#define BGR_E_ARG (-1)      /* invalid argument / limit exceeded */
#define BGR_E_HIP (-2)      /* HIP runtime or device failure */
#define BGR_E_IO (-3)       /* file could not be opened / read */
#define BGR_E_CAPACITY (-4) /* caller-provided output buffer too small */
#define BGR_E_INTERNAL (-5)
#define BGR_E_NOMEM (-7)    /* device memory exhausted (exhaustive mode on a unitig set that duplicates its own k-mers: the table of remembered
                               calls of one read's search outgrew what the device can hold -- see BGR_KNOB_EXH_MEMO_CAP) */
#define BGR_E_COMPACTION (-6) /* correction mode: a path does not spell a walk -- the reference's "bug compaction" exit
                                (aligner.cpp:280-283); bgr_last_error() = "bug compaction\n<walk> <unitig>", outputs hold the
                                records before the offending read, as the reference leaves them */

/* status byte of one read (alignReadGreedy's return + overlapFound + rc; alignerGreedy.cpp:35-57):
 *   low 2 bits: 0 = no anchor (noOverlapRead++), 1 = anchored, not aligned (notAligned++), 2 = aligned
 *   bit 2 (4) : the answer was produced on the reverse-complement retry (`rc` out-parameter)           */
#define BGR_ST_NOANCHOR 0
#define BGR_ST_FAILED 1
#define BGR_ST_ALIGNED 2
#define BGR_ST_MASK 3
#define BGR_ST_RC 4

#define BGR_MODE_GREEDY 0      /* alignAll(true, ...)  -> alignReadGreedy      (bgreat.cpp:115, default) */
#define BGR_MODE_EXHAUSTIVE 1  /* alignAll(false, ...) -> alignReadExhaustive  (bgreat.cpp -b)           */
#define BGR_MODE_ANCHORS 2     /* dogMode (bgreat.cpp -G) -> alignReadGreedyAnchors (alignerGreedy.cpp:60-164); the graph
                                  must have been built with BGR_BUILD_ANCHORS */

typedef struct bgr_graph bgr_graph;     /* immutable index: replaces the Aligner's unitigs/MPHF/indices members */
typedef struct bgr_aligner bgr_aligner; /* per-device mapping context: stream, workspaces, counters            */

typedef struct {
    uint32_t mode;         /* BGR_MODE_*                                            */
    uint32_t max_mismatch; /* -m  errorsMax  (bgreat.cpp:78-80, default 2)          */
    uint32_t effort;       /* -e  tryNumber  (bgreat.cpp:84-86, default 2)          */
    uint32_t partial;      /* -i  partial    (bgreat.cpp:96-98; exhaustive only)    */
} bgr_params;

typedef struct {
    uint32_t k, n_levels;  /* n_levels: buckets a key table lookup reads at most (2) */
    uint64_t n_unitigs, n_keys, n_left_keys, n_right_keys, n_fallback; /* n_keys: distinct overlap (k-1)-mers */
    uint64_t total_bases, blob_bytes, mphf_bytes, max_unitig_len;      /* mphf_bytes: size of the overlap key table */
    uint32_t has_exceptions, has_anchors; /* has_anchors: built with BGR_BUILD_ANCHORS (needed by BGR_MODE_ANCHORS) */
    double gamma;  /* key table slots per key */
} bgr_graph_info_t;

const char* bgr_last_error(void);

/* Process-wide tuning / diagnostic options.  The library never reads the environment: a host sets what it wants changed here (the CLI:
 * --set name=value) before it creates the objects that look at it (graph build, aligners, bgr_align_all).  Names, defaults and what they
 * replace: INTEGRATION.md section 5; bgr_option_name(i, &what) enumerates them (nullptr behind the last).  Unknown name / value out of
 * range: BGR_E_ARG.  Options named test.* are test hooks. */
int bgr_set_option(const char* name, int64_t value);
int bgr_get_option(const char* name, int64_t* value);
const char* bgr_option_name(uint32_t index, const char** what);
int bgr_device_count(void); /* number of HIP devices visible, 0 if none / no driver */

/* ---- index ----------------------------------------------------------------------------------------
 * Replaces Aligner::Aligner + Aligner::indexUnitigs/indexUnitigsAux (aligner.h:80-105, aligner.cpp:407-547).
 * `seqs`/`offsets[n+1]`: the unitig sequences in file order (ids are 1-based ordinals, as in the reference).
 * Loading stops at the first sequence shorter than k (aligner.cpp:418-420).  gamma = slots per key of the overlap key
 * table that stands in for leftMPHF/rightMPHF + the key compare (1.03 .. 64; <= 0 selects the default: 1.07 when the table can be
 * staged in LDS, else 1.8).
 * The graph is built on the host; inputs are only read during the call.
 * Limits (narrower than the reference's int32 unitig ids, utils.h:26; a graph beyond them is refused with BGR_E_ARG and a message that
 * names the limit, never truncated): k <= 32 (as the reference); fewer than 2^30 unitigs (a slot's id field has 30 bits beside its two
 * orientation bits); fewer than 2^28 overlap keys and fewer than 2^27 - 8 filled neighbour slots (a handle is 28 bits, one of them the
 * "query is canonical" flag travelling with it); the packed sequence of both strands below 4 GiB = 2^34 bases (the kernels address it
 * with 32-bit byte offsets).  The BASELINE configs use 2 % / 0.4 % / 5 % of these (4 M unitigs, 2.6 M keys, 0.8 G bases at chr1 scale). */
/* Host threads of the index build (the reference: BooPHF's `coreNumber` threads, aligner.cpp:450,458); 0 = default
 * (all cores, at most 16).  The built graph does not depend on it.  Process-wide. */
void bgr_set_build_threads(uint32_t threads);
int bgr_graph_build(uint32_t k, uint64_t n_unitigs, const char* seqs, const uint64_t* offsets, double gamma, bgr_graph** out);
/* Same, reading the unitig FASTA exactly as aligner.cpp:415-417 does (2 lines per record, header ignored). */
int bgr_graph_build_from_fasta(const char* unitig_fasta_path, uint32_t k, double gamma, bgr_graph** out);
/* The same two with build flags.  BGR_BUILD_ANCHORS = the reference's dogMode index (`-G`, aligner.cpp:434-442,
 * 457-476): an MPHF over the canonical k-mers of all unitigs plus their (unitig, offset) table, the structure of
 * boomphf::mphf bit for bit because the reference consumes its answers for non-keys too (aligner.cpp:387-389).
 * Costs ~9.5 bytes per unitig base on the host and in HBM. */
#define BGR_BUILD_ANCHORS 1u
#define BGR_BUILD_NO_EVICTIONS 2u /* test hook: the key table is built without eviction walks, so a key that finds both of
                                      its buckets full goes to the sorted fallback list (normally empty); same results, slower */
int bgr_graph_build_ex(uint32_t k, uint64_t n_unitigs, const char* seqs, const uint64_t* offsets, double gamma, uint32_t flags, bgr_graph** out);
int bgr_graph_build_from_fasta_ex(const char* unitig_fasta_path, uint32_t k, double gamma, uint32_t flags, bgr_graph** out);
/* boomphf::mphf::lookup on the anchors index, host side (BooPHF.h:783-818): the index of a canonical k-mer, a false
 * index for many non-keys, UINT64_MAX otherwise; *position_out (may be NULL) = unitig id << 32 | offset stored
 * there.  Used by the CPU tests to pin the index against the oracle's. */
int bgr_graph_anchor_lookup(const bgr_graph* g, uint64_t kmer, uint64_t* index_out, uint64_t* position_out);
/* The overlap key table, host side: the membership test of aligner.cpp:158,219,353,361 ("is this canonical (k-1)-mer an
 * overlap of the graph") as the kernels make it.  *slot_out = the key's slot (its index into the blob's keys / records)
 * or UINT32_MAX for a non-member.  Used by the CPU tests. */
int bgr_graph_key_lookup(const bgr_graph* g, uint64_t canonical_k1mer, uint32_t* slot_out);
/* The graph as one position-independent byte blob (what is copied to HBM / broadcast between GPUs). */
const void* bgr_graph_blob(const bgr_graph* g, uint64_t* bytes);
int bgr_graph_from_blob(const void* blob, uint64_t bytes, bgr_graph** out); /* copies the blob */
int bgr_graph_info(const bgr_graph* g, bgr_graph_info_t* out);
/* The unitig characters as they were handed to bgr_graph_build (offsets[n+1]; unitig id i is row i-1), the
 * reference's `vector<string> unitigs` (aligner.h:70).  Only graphs built from sequences carry them (not blobs). */
int bgr_graph_unitigs(const bgr_graph* g, const char** seqs, const uint64_t** offsets, uint64_t* n);
void bgr_graph_destroy(bgr_graph* g);

/* Device residency.  upload: hipMalloc + H2D of the blob on `device` (idempotent per device).
 * adopt: the blob already sits in that device's HBM at `dev_blob` (e.g. it was received by an RCCL
 * broadcast from rank 0); the memory stays owned by the caller and must outlive the graph. */
int bgr_graph_upload(bgr_graph* g, int device);
const void* bgr_graph_device_blob(const bgr_graph* g, int device); /* NULL if not resident there */
/* Several GPUs in one process (replaces the thread fan-out of aligner.cpp:577-586 on the device side): makes the graph
 * resident on devices first_device .. first_device + n_devices - 1.  One host -> device copy to the first, then device to
 * device over xGMI: `how` = BGR_FANOUT_AUTO (one RCCL broadcast when librccl can be loaded at run time, else peer copies),
 * BGR_FANOUT_RCCL (ncclCommInitAll + ncclBroadcast, fail if unavailable), BGR_FANOUT_PEER (hipMemcpyPeerAsync in a doubling
 * schedule 1 -> 2 -> 4 -> 8 holders).  bgr_devices_method: what the last call used (0 = nothing to distribute). */
#define BGR_FANOUT_AUTO 0u
#define BGR_FANOUT_RCCL 1u
#define BGR_FANOUT_PEER 2u
int bgr_devices_init(bgr_graph* g, int first_device, uint32_t n_devices, uint32_t how);
uint32_t bgr_devices_method(const bgr_graph* g);
int bgr_graph_adopt_device_blob(int device, const void* dev_blob, uint64_t bytes, bgr_graph** out);

/* ---- mapping --------------------------------------------------------------------------------------
 * bgr_aligner_create replaces the per-thread worker state of alignPartGreedy/alignPartExhaustive
 * (alignerGreedy.cpp:367-371); it uploads the graph to `device` if needed and owns a HIP stream. */
int bgr_aligner_create(bgr_graph* g, int device, bgr_aligner** out);
void bgr_aligner_destroy(bgr_aligner* a);

/* Batch form of alignReadGreedy / alignReadExhaustive over host buffers (H2D, kernel, D2H; blocking).
 *   reads / read_offsets[n+1] : concatenated read sequences exactly as getReads (aligner.cpp:46-117) hands
 *                               them over (characters ACGTN), borrowed for the call.
 *   paths_out[paths_cap], path_offsets[n+1] : CSR of the returned vector<uNumber> per read, INPUT ORDER;
 *                               an empty row means "not mapped" (the reference's empty vector).
 *   status[n] : BGR_ST_* per read.
 * Counters of aligner.h:68 are accumulated in the aligner (bgr_aligner_counters).  Any batch size: one launch
 * addresses its path arena with 32 bits (about 13 M reads of 150 bp); a larger batch is mapped in pieces.     */
int bgr_align_batch(bgr_aligner* a, const bgr_params* p, const char* reads, const uint64_t* read_offsets, uint64_t n_reads,
                    int32_t* paths_out, uint64_t paths_cap, uint64_t* path_offsets, uint8_t* status);

/* The same with the reads already packed on the host into the 2-bit planes the kernels read (what bgr_align_device's pre-pass
 * makes on the device): a batch then crosses PCIe at ~0.3 byte per base instead of 1.  Layout: str2num codes (utils.cpp:117-129:
 * A0 C1 G2, else 3), 32 bases per uint64, first base most significant, zero beyond a read's end; read r owns words
 * [(read_offsets[r] >> 5) + r, ... + ceil(len/32)) of `fw3` (bgr_packed_plane_words() words in all; read_offsets[0] must be 0).
 * Reads that hold an N have their bit set in `hasn` and one N-mask word (3 on every N) per word of theirs in the sparse list
 * nm_index[] (plane word index) / nm_value[].  bgr_pack_reads fills all of it from ASCII reads (single thread; ranges of a
 * batch can be packed concurrently with the helpers of bgreat_amd/csrc/read_pack.h, as the CLI's pipeline does). */
typedef struct {
    const uint64_t* read_offsets; /* n+1 base offsets, starting at 0 */
    const uint64_t* fw3;
    const uint32_t* hasn;         /* (n + 31) / 32 words */
    const uint32_t* nm_index;
    const uint64_t* nm_value;
    uint64_t nm_count;
    uint32_t max_read_len;        /* longest read of the batch (0 = have it computed) */
} bgr_packed_reads;
uint64_t bgr_packed_plane_words(uint64_t n_reads, uint64_t total_bases);
int bgr_pack_reads(const char* reads, const uint64_t* read_offsets, uint64_t n_reads, uint64_t* fw3, uint32_t* hasn, uint32_t* nm_index,
                   uint64_t* nm_value, uint64_t nm_cap, uint64_t* nm_count, uint32_t* max_read_len);
int bgr_align_batch_packed(bgr_aligner* a, const bgr_params* p, const bgr_packed_reads* reads, uint64_t n_reads, int32_t* paths_out,
                           uint64_t paths_cap, uint64_t* path_offsets, uint8_t* status);

/* Asynchronous form over host buffers ("asynchronous variant with a stream/ticket for double-buffering"):
 * bgr_align_batch_begin starts the copy of the batch to the device and enqueues the mapping launch on the aligner's stream, then
 * returns a ticket WITHOUT waiting; bgr_align_batch_wait blocks until that launch has finished and delivers its results (same
 * output contract as bgr_align_batch); bgr_align_batch_test polls (1 = finished, 0 = still running).  One batch in flight per
 * aligner: one host thread double-buffers with two aligners -- begin(A, b0); begin(B, b1); wait(A); begin(A, b2); wait(B); ... --
 * where the blocking calls need two threads.  `reads` / `read_offsets` (page-locked memory recommended) must stay untouched until
 * the wait has returned.  The batch must fit one launch (2 * (bases + 16 * reads) < 2^32 - 2^28): larger ones go through
 * bgr_align_batch, which cuts them. */
typedef struct {
    bgr_aligner* aligner;
    uint64_t n_reads;
    uint64_t serial;      /* which begin of that aligner this ticket belongs to */
} bgr_ticket;
int bgr_align_batch_begin(bgr_aligner* a, const bgr_params* p, const char* reads, const uint64_t* read_offsets, uint64_t n_reads, bgr_ticket* ticket);
int bgr_align_batch_test(const bgr_ticket* ticket);
int bgr_align_batch_wait(const bgr_ticket* ticket, int32_t* paths_out, uint64_t paths_cap, uint64_t* path_offsets, uint8_t* status);

/* Text form: one piece of a FASTA file in, the bytes to append to `paths` / `notAligned.fa` out -- the whole per-batch body of
 * Aligner::alignPartGreedy (alignerGreedy.cpp:367-431: getReads, alignReadGreedy per read, the fwrite of a record) on the device,
 * so a batch crosses PCIe as the file's own bytes (text_kernels.hip).  The piece must start at a header line and end behind the
 * newline of a sequence line (or at the end of the file).  The device takes the shape nearly every piece has (header line, ONE
 * sequence line, next header ...); for any other piece (multi-line sequences, blank lines, a last record without its newline;
 * records of fewer than 32 bytes on average over 32 KB of the piece -- reads of two dozen bases: more than the device's tables hold)
 * the call returns BGR_OK with `irregular` = 1 and NOTHING mapped: the caller then parses that piece on the host (the exact
 * getReads state machine) and uses bgr_align_batch*, so the records are the reference's either way.
 * BGR_E_CAPACITY: an output buffer is too small; paths_bytes / notaligned_bytes say what is needed, the mapping is done, and
 * bgr_aligner_fetch_text delivers the same bytes into larger buffers.  Blocking; page-locked buffers recommended. */
typedef struct bgr_text_stage bgr_text_stage;
typedef struct {
    uint64_t struct_size;         /* in: sizeof(bgr_text_batch) of the header the caller was built with.  The struct has grown over the rounds and will again: a
                                     caller built against another layout is refused (BGR_E_ARG) instead of having fields read past the end of its struct.
                                     Zero the whole struct, then set this first. */
    const char* text;             /* in: the piece (may be NULL when `stage` holds it) */
    uint64_t text_bytes;          /*     < 2^31 */
    uint32_t want_output;         /*     0 = map and count only (-b without --write-exhaustive writes nothing), 1 = the reference's records,
                                         2 = correction mode (-c, alignerGreedy.cpp:394-404): a mapped read's record is header + the read as
                                         spelled by its path (recoverPath, aligner.cpp:270-290); greedy modes, ACGT-only unitigs */
    uint32_t irregular;           /* out: 1 = piece left to the host parser; 2 = (want_output 2) a path of this piece does not spell a walk --
                                         the reference's "bug compaction" exit, which the caller reproduces on the host */
    char* paths_out;              /* in: where the records of mapped reads go */
    uint64_t paths_cap;
    char* notaligned_out;         /*     and those of the others (never more than text_bytes) */
    uint64_t notaligned_cap;
    uint64_t n_records, n_accepted, paths_bytes, notaligned_bytes;   /* out */
    bgr_text_stage* stage;        /* in, optional: the piece was sent ahead with bgr_text_stage_upload (same bytes, same device): the call
                                     waits for that copy on the device instead of making its own */
    uint32_t fastq;               /* in: 1 = the piece is FASTQ (-q): whole four-line records (record j = lines 4j .. 4j+3 whatever they hold,
                                     aligner.cpp:51-68), ending with a newline; the reference's end-of-file behaviour (its phantom record)
                                     stays with the caller: bgr_align_all maps the file's last getReads() call through the host parser.
                                     2 = the same records with their '+' and quality lines left out by the caller: record j = lines 2j, 2j+1
                                     (header line, read line) -- half the bytes over PCIe, the same records out */
    uint32_t reserved;
    uint32_t* record_info_out;    /* in, optional: one word per RECORD of the piece, in the piece's order -- bit 31: getReads keeps it, bit 30: it was mapped,
                                     bits 0..29: its read's length (0 unless kept) -- what the reference's -b progress blocks count between two getReads()
                                     calls (alignerExhaustive.cpp:306-316: a record, kept or dropped, is one iteration of the call).  Needs room for
                                     text_bytes / 24 + 1024 words (record_info_cap); n_records of them are written.  Not with want_output = 2. */
    uint64_t record_info_cap;
} bgr_text_batch;
/* A stage = a device buffer for one piece + a copy stream: bgr_text_stage_upload starts the host -> device copy and returns; the
 * bgr_align_fasta_text call that names the stage orders itself behind it (hipStreamWaitEvent), so the copy of the next piece runs
 * under the kernels of this one.  The host bytes must stay untouched until that call has returned, and the stage keeps the piece
 * (no new upload, no destroy) until the records have been fetched: after BGR_E_CAPACITY, bgr_aligner_fetch_text still cuts them from it. */
int bgr_text_stage_create(int device, bgr_text_stage** out);
void bgr_text_stage_destroy(bgr_text_stage* s);
int bgr_text_stage_upload(bgr_text_stage* s, const char* text, uint64_t text_bytes);
/* ... a piece that lies in several host ranges (the header and read lines gathered out of a FASTQ file part by part): the device receives
 * them back to back, in order; the piece's size is the sum. */
int bgr_text_stage_upload_parts(bgr_text_stage* s, uint32_t n_parts, const char* const* parts, const uint64_t* part_bytes);
int bgr_text_stage_device(const bgr_text_stage* s);  /* the device the stage was created on (-1: null) */
int bgr_align_fasta_text(bgr_aligner* a, const bgr_params* p, bgr_text_batch* b);
int bgr_aligner_fetch_text(bgr_aligner* a, bgr_text_batch* b);

/* Device-resident form: inputs already in this device's HBM (d_reads bytes, d_read_offsets uint64[n+1]);
 * results stay in aligner-owned device buffers (bgr_aligner_device_results).  Asynchronous on the
 * aligner's stream; max_read_len = longest read in the batch (< 2^24), total_bases = read_offsets[n].
 * Read characters must be from ACGTN (what getReads lets through).                                      */
int bgr_align_device(bgr_aligner* a, const bgr_params* p, const void* d_reads, const void* d_read_offsets, uint64_t n_reads,
                     uint64_t total_bases, uint32_t max_read_len);
/* Waits for the aligner's stream.  In exhaustive mode it also SETTLES the launch: reads whose search outgrew the last pass's table of remembered calls
 * are mapped again with a larger one (BGR_KNOB_EXH_MEMO_CAP) before the results count as final -- every fetch / counters call does the same; a caller
 * that reads the device results below by itself calls this first. */
int bgr_aligner_sync(bgr_aligner* a);
/* Device pointers of the last bgr_align_device results: results uint32[n][2] = {path offset in the arena,
 * path length | status << 24}, arena int32[], cursor u32[1] (ints used in the arena).
 * Row i of the result = arena[results[i][0] .. + (results[i][1] & 0xFFFFFF)]. */
int bgr_aligner_device_results(bgr_aligner* a, void** d_results, void** d_arena, void** d_cursor);
/* Copy the last device results to the host in input order (same output contract as bgr_align_batch). */
int bgr_aligner_fetch(bgr_aligner* a, uint64_t n_reads, int32_t* paths_out, uint64_t paths_cap, uint64_t* path_offsets, uint8_t* status);

/* aligner.h:68 counters since creation/reset: out[0]=readNumber, [1]=noOverlapRead, [2]=alignedRead,
 * [3]=notAligned, [4]=overlaps (exhaustive only).  Synchronises the stream. */
int bgr_aligner_counters(bgr_aligner* a, uint64_t out[5]);
int bgr_aligner_reset_counters(bgr_aligner* a);

/* Kernel timing by HIP events recorded on the aligner's stream around every mapping-kernel launch since the
 * last reset: number of launches and their summed duration in milliseconds.  Synchronises the stream. */
int bgr_aligner_kernel_time(bgr_aligner* a, uint64_t* launches, double* total_ms);
int bgr_aligner_reset_kernel_time(bgr_aligner* a);
/* The same per kernel of a launch, in launch order (a mapping launch is a short sequence on one stream: the pre-pass that
 * packs the reads, then the passes of the mode): summed milliseconds per position and the kernel's name (static strings;
 * NULL behind the last).  Launches of different modes since the last reset share positions. */
int bgr_aligner_kernel_times(bgr_aligner* a, uint64_t* launches, double slot_ms[8], const char* slot_names[8]);
/* Launch geometry of the last mapping kernel (for logs): blocks, threads per block, dynamic LDS bytes, and flags:
 * bit 0 = the overlap key table was staged in LDS, bit 1 = exhaustive mode ran its level search (else depth-first),
 * bit 2 = the mode ran its several-reads-per-wave first pass (sixteen in greedy mode, eight or four in the others; the numbers then
 * describe that launch). */
int bgr_aligner_launch_info(bgr_aligner* a, uint32_t out[4]);
/* How the last mapping launch went through its passes.  Greedy mode: out[0..2] = reads the first / second / third launch of
 * the sixteen-reads-per-wave kernel handed on to the next one, counted in list entries (lists are written in per-wave slices,
 * so the figure includes a few unused entries; out[2] is 0: the third launch hands everything to the general kernel),
 * out[3] = reads mapped by the general kernel.  Exhaustive mode: out[2] = reads the four-reads-per-wave pass left to the level
 * / depth-first search, out[0] = reads that search listed for its second pass, out[1] = for its third.  Synchronises the stream. */
int bgr_aligner_pass_counts(bgr_aligner* a, uint32_t out[4]);
/* Tuning knobs (0 keeps the default): waves per workgroup, workgroups per CU, LDS staging of the overlap
 * key table (0 auto, 1 off: probed in L2, 2 on -- where it can be had: anchors mode stages nothing, a table beyond a CU's LDS is probed in L2;
 * bgr_aligner_launch_info says what the launch did). */
int bgr_aligner_configure(bgr_aligner* a, uint32_t waves_per_block, uint32_t blocks_per_cu, uint32_t lds_mphf);

/* Test / diagnostic hooks, set once per aligner (they used to be environment variables read on every launch):
 *   BGR_KNOB_EXH_FRAME_CAP     exhaustive pass 1: search frames (depth-first) or levels (level search) per wave, 0 = default;
 *                              a tiny cap pushes most reads through the later passes
 *   BGR_KNOB_EXH_SEARCH        0 = choose per graph and budget, 1 = depth-first search, 2 = level search
 *   BGR_KNOB_BATCH_SPLIT_LIMIT bgr_align_batch maps a batch in pieces beyond this many path-arena ints, 0 = default (~2^32)
 *   BGR_KNOB_DEBUG_STOP        diagnostic builds (-DBGR_PHASE_TIMING) only: 1 = stop after packing, 2 = after the position scan */
#define BGR_KNOB_EXH_FRAME_CAP 1u
#define BGR_KNOB_EXH_SEARCH 2u
#define BGR_KNOB_BATCH_SPLIT_LIMIT 3u
#define BGR_KNOB_DEBUG_STOP 4u
#define BGR_KNOB_BATCH_OVERLAP 8u /* bgr_align_batch of >= 512 k reads: 0 = in four pieces on two streams, copies under kernels (default), 1 = one launch */
#define BGR_KNOB_ANCHORS_FAST 7u /* anchors mode: 0 = four-reads-per-wave first pass + the one-read-per-wave kernel for the rest (default), 1 = without it */
#define BGR_KNOB_EXH_FAST 6u    /* exhaustive mode: 0 = four-reads-per-wave first pass + the level / depth-first passes for the rest (default), 1 = without it */
#define BGR_KNOB_EXH_MEMO_CAP 9u /* exhaustive mode, last pass (the reference's recursion memoised on (overlap, position): polynomial on any unitig set): entries per wave of
                                   its table of remembered calls in the FIRST run, 0 = from the read length (>= 1024).  A read whose search fills the table is run again with a table
                                   16 times as large, until it fits (bgr_aligner_last_pass_runs); tests set 8 to walk that path with small inputs */
#define BGR_KNOB_GREEDY_PREPASS 10u /* a launch handed ASCII reads, greedy / exhaustive mode: 0 = the mapping kernels stage their reads straight from the characters (default), 1 = a pre-pass writes 2-bit planes first (rounds 2-4; what anchors mode does) */
#define BGR_KNOB_KERNEL_EVENTS 11u /* 1 = a HIP event in front of a mapping launch and behind each of its kernels (default: bgr_aligner_kernel_times reports them), 0 = none (a caller that never asks for the times: bgr_align_all without its timing option) */
#define BGR_KNOB_GREEDY_FAST 5u /* greedy mode: 0 = sixteen-reads-per-wave pass + general kernel for the rest (default), 1 = general kernel only */
int bgr_aligner_set_knob(bgr_aligner* a, uint32_t knob, uint64_t value);
/* The launch geometry by itself (bgreat_amd/csrc/launch_plan.h: a pure function of these numbers; no device, no graph object needed -- CPU tests sweep
 * it over synthetic graph headers).  Zero fields of the device part mean "an MI355X" (256 CUs, 160 KB of LDS per CU, the shipped kernels' occupancies).
 * pass[]: 0 the mode's general kernel, 1 greedy sixteen-reads-per-wave, 2 exhaustive eight-reads-per-wave, 3 anchors four-reads-per-wave,
 * 4 exhaustive depth-first over what the level search listed, 5 exhaustive last pass.  BGR_E_ARG (message: bgr_last_error) when the batch cannot
 * be mapped in one launch (a read too long for the LDS layouts, a batch beyond the 32-bit path arena). */
typedef struct bgr_plan_input {
    uint32_t k, slot_fill_x100, table_bytes, has_exceptions, anchors, anchor_levels;   /* graph header */
    uint64_t graph_bases, n_unitigs, max_unitig_len;
    uint32_t num_cus, resident_waves[7];                                                /* device (0: MI355X defaults) */
    uint64_t lds_per_cu;
    uint32_t cfg_waves, cfg_blocks_per_cu, cfg_lds_mphf;                                /* bgr_aligner_configure */
    uint32_t mode, max_mismatch, partial, max_read_len;                                 /* batch */
    uint64_t n_reads, total_bases;
} bgr_plan_input;
typedef struct bgr_plan_pass { uint32_t used, blocks, waves_per_block, lds_bytes, table_staged; } bgr_plan_pass;
typedef struct bgr_plan_output {
    bgr_plan_pass pass[6];
    uint32_t level_search, deep_only, x4_levels, memo_cap;
    uint64_t deep_scratch_bytes, arena_ints;
} bgr_plan_output;
int bgr_plan_launch(const bgr_plan_input* in, bgr_plan_output* out);

/* Exhaustive mode: how often the last pass ran for the launch last settled (1 = once, as enqueued; more: reads whose table of remembered calls filled
 * up were run again) and the table size (entries per wave) of its final run.  0 / 0 when the last launch had no last pass (greedy, anchors mode). */
int bgr_aligner_last_pass_runs(const bgr_aligner* a, uint32_t* runs, uint32_t* memo_cap);

/* ---- read files (host) ----------------------------------------------------------------------------
 * Replaces Aligner::getReads (aligner.cpp:46-117) for a whole file: the accepted (header, read) records
 * in file order with the reference's drop rules (characters outside ACGTN, size <= 2, FASTA size <= k,
 * multi-line FASTA, the FASTQ phantom record).  Buffers are owned by the returned object. */
typedef struct bgr_readset bgr_readset;
int bgr_readset_load(const char* path, int fastq, uint32_t k, bgr_readset** out);
/* Same records in the same order, parsed by `threads` host threads over chunks of ~chunk_bytes (0 = default). */
int bgr_readset_load_parallel(const char* path, int fastq, uint32_t k, uint32_t threads, uint64_t chunk_bytes, bgr_readset** out);
uint64_t bgr_readset_count(const bgr_readset* rs);
/* reads_concat / read_offsets[n+1] / headers_concat / header_offsets[n+1] */
int bgr_readset_view(const bgr_readset* rs, const char** reads, const uint64_t** read_offsets, const char** headers, const uint64_t** header_offsets);
void bgr_readset_destroy(bgr_readset* rs);

/* Output formatting of printPath + the fwrite sites (aligner.cpp:600-609, alignerGreedy.cpp:406-427):
 * appends "header\n" + "int." * n + "\n" records for mapped reads to `paths_file` and "header\nread\n" for the
 * others to `notaligned_file` (both FILE* opened by the caller, passed as void*). */
int bgr_write_records(void* paths_file, void* notaligned_file, uint64_t n_reads, const char* headers, const uint64_t* header_offsets,
                      const char* reads, const uint64_t* read_offsets, const int32_t* paths, const uint64_t* path_offsets);

/* ---- whole run -------------------------------------------------------------------------------------
 * Batch form of Aligner::alignAll (aligner.cpp:550-597): maps every file of the comma-separated list `reads_csv`
 * and writes `paths_file` / `notaligned_file` (opened "wb" like aligner.h:85-86) with the bytes the reference
 * produces at -t 1, whatever the thread / GPU count.  Two routes per batch, same bytes: the device takes FASTA text and
 * returns the bytes to write (`route`), or the host pipeline parses chunk-parallel, packs into pinned batches and formats
 * range-parallel; two streams per device, one ordered writer.  counters_out as bgr_aligner_counters. */
typedef struct {
    uint64_t struct_size;      /* sizeof(bgr_run_options) of the header the caller was built with (see bgr_text_batch.struct_size): zero the struct, set this */
    uint32_t n_gpus;           /* devices 0..n_gpus-1 (0 = 1)                                                  */
    uint32_t threads;          /* host threads for parsing / gathering / formatting (-t; 0 = 1)                */
    uint64_t batch_reads;      /* target reads per device batch (0 = default: 128k on the host route, 256k as text)  */
    uint64_t chunk_bytes;      /* parser chunk size (0 = default 8 MiB)                                         */
    uint32_t fastq;            /* -q                                                                            */
    uint32_t write_exhaustive; /* exhaustive mode writes nothing in the reference; 1 = write  */
    uint32_t echo_files;       /* print what the reference's workers print to stdout while mapping, in its -t 1 order: each file
                                  name (aligner.cpp:559,576) and, in exhaustive mode, the block after every tenth getReads()
                                  call (alignerExhaustive.cpp:306-316)                                              */
    uint32_t correction;       /* -c: write header + the read as spelled by its path (recoverPath, aligner.cpp:270-290,
                                  alignerGreedy.cpp:394-404) instead of the path; greedy mode only                 */
    const char* no_overlap_file; /* optional third output (NULL = the reference's behaviour): reads WITHOUT any anchor go
                                  here instead of notAligned.fa -- the split README.md:47-52 documents and
                                  alignerGreedy.cpp:414-419 disables                                               */
    uint32_t first_device;     /* devices first_device .. first_device + n_gpus - 1 (one process per GPU under a launcher
                                  that does not hide the others: first_device = the rank's local index, n_gpus = 1)   */
    uint32_t route;            /* 0 = automatic: FASTA input without -c / --no-overlap / -b progress blocks goes through the device as
                                  text (bgr_align_fasta_text: parsing, packing, mapping and record formatting on the GPU; pieces of an
                                  irregular shape fall back to the host parser one by one); 1 = host parser + host formatter always */
    uint32_t numa;             /* 0 = while the run lasts, its threads (the caller's included) and the page-locked memory they allocate are kept
                                  on the CPUs next to the run's devices when all of them share one NUMA node, and the pool of host threads is
                                  clamped to that CPU set; 1 = the caller's affinity is left alone                          */
    uint32_t split_output;     /* 0 = the reference's two files.  1 = one pipeline PER DEVICE: device d of the run maps the d-th of n_gpus
                                  contiguous shares of the input (cut at record starts, over the files of the list taken together) and writes
                                  its own pair `<paths_file>.<d>` / `<notaligned_file>.<d>`; the pairs concatenated in device order are the
                                  reference's -t 1 bytes.  One ordered stream into ONE file is bound by what a single writer gets out of the
                                  file system (~6-13 GB/s: 125-250 Mreads/s at ~50 bytes per read) whatever the number of GPUs; this form has
                                  no stage shared between devices.  FASTA input without -c, --no-overlap and -b progress blocks; any
                                  other run ignores the flag.                                                                  */
} bgr_run_options;
/* bgr_align_all keeps its page-locked staging buffers for the next call of the process (they cost ~0.2 s per GB to allocate);
 * this frees them. */
void bgr_host_cache_release(void);
int bgr_align_all(bgr_graph* g, const bgr_params* p, const bgr_run_options* o, const char* reads_csv, const char* paths_file,
                  const char* notaligned_file, uint64_t counters_out[5], double* mapping_seconds);

/* The CPUs next to a device (the `local_cpulist` of its PCI function in sysfs, e.g. "0-63,128-191"): threads that feed a GPU and the
 * page-locked memory they allocate belong on its NUMA node.  BGR_E_IO when the platform does not say. */
int bgr_device_local_cpus(int device, char* cpulist_out, uint64_t cap);

/* Device memory for callers of bgr_align_device that have no HIP runtime of their own (a cgo / ctypes host parks its batches in HBM
 * through these three); any other device pointer of the same process works as well.  upload / download are blocking copies. */
int bgr_device_alloc(int device, uint64_t bytes, void** out);
int bgr_device_free(int device, void* p);
int bgr_device_upload(int device, void* dst_device, const void* src_host, uint64_t bytes);
int bgr_device_download(int device, void* dst_host, const void* src_device, uint64_t bytes);

/* Page-locked host memory for batches handed to bgr_align_batch (faster H2D/D2H); plain memory works too. */
int bgr_host_alloc(uint64_t bytes, void** out);
int bgr_host_free(void* p);

#ifdef __cplusplus
}
#endif
#endif
