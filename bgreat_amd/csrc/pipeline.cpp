// pipeline.cpp -- bgr_align_all: the batch form of Aligner::alignAll (aligner.cpp:550-597) as a host pipeline.
//
//   reference:  N threads, each: lock; getReads(10000); unlock; per read alignRead*; lock; fwrite; unlock
//   here     :  parse (chunk-parallel, exact getReads semantics, fastx.cpp)
//                 -> batches of reads gathered into pinned memory
//                 -> GPU workers (2 per device, each with its own bgr_aligner/stream: H2D, kernel, D2H of
//                    consecutive batches overlap)
//                 -> formatter (range-parallel printPath/record formatting) -> ONE writer, batches in input order
//
// so the bytes written are the reference's `-t 1` stream whatever the thread/GPU count.  Pure host C++: talks to
// the GPU only through the C-ABI (bgr_aligner_create, bgr_align_batch, bgr_host_alloc ...).
#include <fcntl.h>
#include <sched.h>
#include <functional>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bgreat_gpu.h"
#include "fastx.h"
#include "file_image.h"
#include "read_pack.h"
#include "options.h"

namespace bgr {
int set_error(int code, const std::string& msg);  // capi.hip
}

namespace {

using bgr::ParsedChunk;
using bgr::RecSlice;

using MappedFile = bgr::FileImage;  // mmap for regular files, read-until-EOF for FIFOs and other streams

// Persistent workers shared by the parse, gather and format stages (a batch is a few hundred thousand reads: spawning
// threads per stage and batch cost more than the work).  run(n, fn): fn(0..n-1) on the workers and the calling
// thread, returns when all are done; concurrent run() calls from different stage threads interleave.
class WorkerPool {
public:
    explicit WorkerPool(unsigned n) {
        for (unsigned i = 0; i + 1 < n; ++i) ts_.emplace_back([this]() { loop(); });
    }
    ~WorkerPool() {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : ts_) t.join();
    }
    // tag: which stage the work belongs to (BGREAT_TIMING: thread CPU seconds per stage, cpu_us)
    template <typename F>
    void run(size_t n, F fn, int tag = 0) {
        if (n == 0) return;
        if (n == 1 || ts_.empty()) { const uint64_t c0 = cpu_now(); for (size_t i = 0; i < n; ++i) fn(i); cpu_us[tag] += cpu_now() - c0; return; }
        Job job;
        job.n = n;
        job.tag = tag;
        job.fn = [&fn](size_t i) { fn(i); };
        const size_t helpers = std::min<size_t>(ts_.size(), n - 1);
        job.pending = helpers;
        {
            std::lock_guard<std::mutex> l(m_);
            for (size_t i = 0; i < helpers; ++i) q_.push_back(&job);
        }
        cv_.notify_all();
        work(job);
        std::unique_lock<std::mutex> l(job.m);
        job.cv.wait(l, [&] { return job.pending == 0; });
    }
private:
    struct Job {
        size_t n = 0;
        std::atomic<size_t> next{0};
        std::function<void(size_t)> fn;
        std::mutex m;
        std::condition_variable cv;
        size_t pending = 0;
        int tag = 0;
    };
    void work(Job& j) {
        const uint64_t c0 = cpu_now();
        for (size_t i; (i = j.next.fetch_add(1)) < j.n;) j.fn(i);
        cpu_us[j.tag] += cpu_now() - c0;
    }
    void loop() {
        for (;;) {
            Job* j = nullptr;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                j = q_.front();
                q_.pop_front();
            }
            work(*j);
            std::lock_guard<std::mutex> l(j->m);  // notify under the lock: the job lives on the caller's stack
            if (--j->pending == 0) j->cv.notify_one();
        }
    }
    std::vector<std::thread> ts_;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Job*> q_;
    bool stop_ = false;
public:
    bool timing = false;
    std::atomic<uint64_t> cpu_us[8] = {};
    uint64_t cpu_now() const {  // CPU time of the calling thread, microseconds (0 unless timing)
        if (!timing) return 0;
        timespec ts;
        clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
        return (uint64_t)ts.tv_sec * 1000000ull + (uint64_t)ts.tv_nsec / 1000;
    }
};

struct HostBuf {  // grow-only host buffer; `pinned` = page-locked (20 GB/s to allocate on an idle process, several times slower
                  // while parser threads are faulting the input file in: the pool is sized and allocated before they start)
    void* p = nullptr;
    uint64_t cap = 0;
    bool pinned;
    explicit HostBuf(bool pin) : pinned(pin) {}
    void release() {
        if (!p) return;
        if (pinned) bgr_host_free(p); else free(p);
        p = nullptr; cap = 0;
    }
    bool ensure(uint64_t bytes) {
        if (bytes <= cap) return true;
        release();
        uint64_t want = bytes + bytes / 4 + 4096;
        if (pinned) { if (bgr_host_alloc(want, &p) != BGR_OK) return false; }
        else { p = malloc(want); if (!p) return false; }
        cap = want;
        return true;
    }
    ~HostBuf() { release(); }
};

struct Pinned {  // the page-locked buffers of one batch in flight: sources / targets of the async copies
    // reads travel as 2-bit planes (read_pack.h): fw3 words, N bitmap, and the N-mask words of the few reads with an N
    HostBuf fw3{true}, hasn{true}, nm_idx{true}, nm_val{true}, offs{true}, paths{true}, poffs{true}, status{true};
    // text route (bgr_align_fasta_text): the piece of the file as it is, and the two record streams as they come back
    HostBuf text{true}, ptext{true}, ntext{true};
    HostBuf info{true};  // text route under -b with progress blocks: one word per record of the piece (bgr_text_batch.record_info_out)
    std::vector<bgr_text_stage*> stages;  // per device of the run: where this set's piece waits in HBM (bgr_text_stage_upload), made on first use
    uint64_t nm_count = 0;
    uint32_t max_len = 0;
    ~Pinned() { for (bgr_text_stage* st : stages) bgr_text_stage_destroy(st); }
};

// Page-locked buffers cost ~0.2 s per GB to allocate: a finished run leaves its sets here for the next bgr_align_all of the
// process (bgr_host_cache_release frees them).
// (heap objects that are never destroyed: a static destructor would release page-locked memory, streams and HBM buffers after the HIP
// runtime's own teardown at exit; a process that wants them gone calls bgr_host_cache_release -- the CLI does at the end of main)
std::mutex& g_pin_cache_m = *new std::mutex;
std::vector<std::unique_ptr<Pinned>>& g_pin_cache = *new std::vector<std::unique_ptr<Pinned>>;

// Something the reference prints to stdout between two reads of the input order: a file name (aligner.cpp:559,576) or, in
// exhaustive mode, the block its worker prints after every tenth getReads() call (alignerExhaustive.cpp:306-316).  A
// mark sits in front of read `pos` of its batch; the ordered writer prints it when it gets there.
struct Mark {
    uint64_t pos;
    int kind;          // 0 = progress block, 1 = line of text
    std::string text;
};

struct Batch {
    uint64_t index = 0;
    std::vector<Mark> marks;                              // ascending pos
    std::shared_ptr<MappedFile> file;                     // keeps header/sequence slices valid
    std::vector<std::unique_ptr<ParsedChunk>> chunks;
    std::shared_ptr<std::vector<std::unique_ptr<ParsedChunk>>> shared_chunks;  // a chunk group cut into several batches (base cap): its storage, shared
    std::vector<std::pair<const ParsedChunk*, std::pair<uint32_t, uint32_t>>> spans;  // chunk, [first, last) records
    uint64_t n = 0, bases = 0, path_cap = 0;
    std::unique_ptr<Pinned> pin;                          // attached by the gatherer, handed back by the formatter
    std::vector<RecSlice> recs;                           // flattened view of the records of this batch
    // text route: the batch is bytes [t_begin, t_end) of `file` (a piece that starts at a header line); once mapped as text,
    // dev_text is set and the record streams sit in pin->ptext / pin->ntext
    bool text_piece = false, dev_text = false, fastq_piece = false;
    uint64_t t_begin = 0, t_end = 0, p_bytes = 0, n_bytes = 0;
    // a FASTQ piece crosses PCIe without its '+' and quality lines: fq_parts = where the gatherer may cut it into independent parts
    // ({first byte, number of the line it lies in}, FastqPlan::piece_parts), t_gathered = bytes of header + read lines staged
    std::vector<std::pair<uint64_t, uint64_t>> fq_parts;
    std::vector<std::shared_ptr<const std::vector<uint32_t>>> fq_nl;  // per part: the newline offsets of the plan's chunk the part lies in (from that chunk's first byte), or null
    uint64_t fq_chunk_bytes = 0;
    uint64_t t_gathered = 0;
    // text route under -b with progress blocks: the batch is a piece [t_begin, t_end) of its file whatever route mapped it in the end; the
    // writer counts getReads() iterations (record attempts) itself.  n_attempts = iterations of the piece; the device's per-record words sit in
    // pin->info, a piece that fell back to the host parser lists the iteration of each of its reads in att_of_read
    bool text_origin = false, first_of_file = false, last_of_file = false;
    uint64_t n_attempts = 0;
    std::vector<uint64_t> att_of_read;
    unsigned dev = 0;                                     // which device of the run maps the batch (round robin in input order)
    int rc = BGR_OK;
    std::string err;
};

template <typename T>
class Channel {  // bounded FIFO
public:
    explicit Channel(size_t cap) : cap_(cap) {}
    bool push(T v) {
        std::unique_lock<std::mutex> l(m_);
        cv_space_.wait(l, [&] { return q_.size() < cap_ || closed_; });
        if (closed_) return false;
        q_.push_back(std::move(v));
        cv_item_.notify_one();
        return true;
    }
    bool try_pop(T& v) {
        std::lock_guard<std::mutex> l(m_);
        if (q_.empty()) return false;
        v = std::move(q_.front());
        q_.pop_front();
        cv_space_.notify_one();
        return true;
    }
    bool pop(T& v) {
        std::unique_lock<std::mutex> l(m_);
        cv_item_.wait(l, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        v = std::move(q_.front());
        q_.pop_front();
        cv_space_.notify_one();
        return true;
    }
    void close() {
        std::lock_guard<std::mutex> l(m_);
        closed_ = true;
        cv_item_.notify_all();
        cv_space_.notify_all();
    }
private:
    std::mutex m_;
    std::condition_variable cv_item_, cv_space_;
    std::deque<T> q_;
    size_t cap_;
    bool closed_ = false;
};

const uint64_t kRefBatch = 10000;  // records per getReads() call (alignerExhaustive.cpp:270)

// Where the reference's exhaustive worker prints its periodic block at -t 1: `iter` starts at 1 (aligner.h:103) and
// `iter++ % 10 == 0` is tested after every getReads() call, over all input files, so the block follows calls 10, 20, ...
// A call is 10000 loop iterations of getReads (record attempts, accepted or not); the parser counts them per chunk.
struct ProgressMarker {
    bool on = false;
    uint64_t calls_before = 0;  // calls made for earlier files
    uint64_t next_fc = 10;      // the next call of the current file (1-based) that is followed by a block
    void begin_file() { next_fc = 10 - calls_before % 10; }
    // Chunk `ch` holds iterations [base_iter, base_iter + ch.iters) of the file: indices (chunk-relative, ascending) of the
    // records in front of which a block is due; ch.recs.size() = behind its last record.
    void chunk(const ParsedChunk& ch, uint64_t base_iter, std::vector<uint64_t>& idx_out) {
        idx_out.clear();
        while (on && next_fc * kRefBatch < base_iter + ch.iters) {
            const uint64_t rel = next_fc * kRefBatch - base_iter;  // first iteration of the call after the block
            idx_out.push_back((uint64_t)(std::lower_bound(ch.rec_iter.begin(), ch.rec_iter.end(), rel) - ch.rec_iter.begin()));
            next_fc += 10;
        }
    }
    // End of a file of `total_iters` iterations (>= 1: even an empty file costs one call): blocks still due.
    unsigned end_file(uint64_t total_iters) {
        const uint64_t calls = (total_iters + kRefBatch - 1) / kRefBatch;
        unsigned due = 0;
        while (on && next_fc <= calls) { ++due; next_fc += 10; }
        calls_before += calls;
        return due;
    }
};

// to_string(v) + '.'  (aligner.cpp:600-609).  Two digits per step from a table, written back to front into their final place
// (the per-digit divide loop this replaces was the largest single cost of the host pipeline: ~50 ns per read).
static const char kDigitPairs[] =
    "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263"
    "646566676869707172737475767778798081828384858687888990919293949596979899";
inline char* put_int(char* o, int32_t v) {
    uint32_t u = (uint32_t)v;
    if (v < 0) { *o++ = '-'; u = 0u - u; }
    const unsigned len = u < 10 ? 1 : u < 100 ? 2 : u < 1000 ? 3 : u < 10000 ? 4 : u < 100000 ? 5 : u < 1000000 ? 6 : u < 10000000 ? 7 :
                         u < 100000000 ? 8 : u < 1000000000 ? 9 : 10;
    char* e = o + len;
    char* w = e;
    while (u >= 100) {
        const uint32_t q = u / 100, r = u - q * 100;
        w -= 2;
        memcpy(w, kDigitPairs + 2 * r, 2);
        u = q;
    }
    if (u >= 10) memcpy(w - 2, kDigitPairs + 2 * u, 2);
    else w[-1] = (char)('0' + u);
    *e++ = '.';
    return e;
}

// ---- correction mode (-c): recoverPath / getUnitig / compactionEnd (aligner.cpp:270-302, utils.cpp:171-179) ------
struct Unitigs {
    const char* seqs = nullptr;
    const uint64_t* offs = nullptr;
    uint64_t n = 0;
    uint32_t k = 0;
};
inline char rc_char(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A'; }  // utils.cpp:52-59
inline void oriented_unitig(const Unitigs& u, int32_t id, std::string& out) {  // getUnitig: forward, or reverse complement for id < 0
    out.clear();
    const uint64_t i = (uint64_t)(id < 0 ? -(int64_t)id : id);
    if (i == 0 || i > u.n) return;  // unitigs[0] is "" (aligner.cpp:408)
    const char* s = u.seqs + u.offs[i - 1];
    const uint64_t len = u.offs[i] - u.offs[i - 1];
    if (id > 0) { out.assign(s, len); return; }
    out.resize(len);
    for (uint64_t j = 0; j < len; ++j) out[j] = rc_char(s[len - 1 - j]);
}
// The read as spelled by its path: unitigs glued on their k-1 overlaps, then substr(offset, read length).
// false = the reference's "bug compaction" exit (no orientation of the next unitig continues the walk).
bool recover_path(const Unitigs& u, const int32_t* path, uint64_t n, uint32_t read_len, std::string& walk, std::string& tmp, std::string& rc) {
    const uint32_t K1 = u.k - 1;
    oriented_unitig(u, path[1], walk);
    for (uint64_t i = 2; i < n; ++i) {
        oriented_unitig(u, path[i], tmp);
        if (walk.empty() || tmp.empty() || walk.size() < K1 || tmp.size() < K1) return false;
        if (walk.compare(walk.size() - K1, K1, tmp, 0, K1) == 0) { walk.append(tmp, K1, std::string::npos); continue; }
        rc.resize(tmp.size());
        for (size_t j = 0; j < tmp.size(); ++j) rc[j] = rc_char(tmp[tmp.size() - 1 - j]);
        if (walk.compare(walk.size() - K1, K1, rc, 0, K1) == 0) { walk.append(rc, K1, std::string::npos); continue; }
        return false;
    }
    const int32_t off = path[0];
    if (off < 0 || (uint64_t)off > walk.size()) return false;  // std::out_of_range in the reference
    walk.erase(0, (size_t)off);
    if (walk.size() > read_len) walk.resize(read_len);
    return true;
}

// Records lo..hi of a batch as the reference writes them: mapped -> "header\n" + "int." * n + "\n" into pbuf
// (alignerGreedy.cpp:406-411), the others -> "header\nread\n" into nbuf (alignerGreedy.cpp:421-427).
void format_range(const Batch& b, uint64_t lo, uint64_t hi, std::string& pbuf, std::string& nbuf) {
    const int32_t* paths = static_cast<const int32_t*>(b.pin->paths.p);
    const uint64_t* poffs = static_cast<const uint64_t*>(b.pin->poffs.p);
    uint64_t pmax = 0, nmax = 0;  // exact upper bounds, so the loop below writes through raw pointers
    for (uint64_t i = lo; i < hi; ++i) {
        const RecSlice& r = b.recs[i];
        const uint64_t np = poffs[i + 1] - poffs[i];
        if (np) pmax += r.hl + 2 + 12 * np; else nmax += (uint64_t)r.hl + r.sl + 2;
    }
    pbuf.resize(pmax);
    nbuf.resize(nmax);
    char* po = pmax ? &pbuf[0] : nullptr;
    char* no = nmax ? &nbuf[0] : nullptr;
    char* const p0 = po;
    for (uint64_t i = lo; i < hi; ++i) {
        const RecSlice& r = b.recs[i];
        if (poffs[i + 1] > poffs[i]) {
            memcpy(po, r.h, r.hl); po += r.hl;
            *po++ = '\n';
            for (uint64_t j = poffs[i]; j < poffs[i + 1]; ++j) po = put_int(po, paths[j]);
            *po++ = '\n';
        } else {
            memcpy(no, r.h, r.hl); no += r.hl;
            *no++ = '\n';
            memcpy(no, r.s, r.sl); no += r.sl;
            *no++ = '\n';
        }
    }
    pbuf.resize(pmax ? (size_t)(po - p0) : 0);
}

// The same with the two opt-in output variants: correction mode (mapped reads are written as header + corrected read,
// alignerGreedy.cpp:394-404) and the no-overlap split (reads without any anchor go to a third buffer).
// Returns false on the reference's "bug compaction" condition (aligner.cpp:280-283: it prints "bug compaction", the walk
// so far and the unitig that does not continue it, and exits); the buffers then hold the records BEFORE that read and
// `bug` the two strings the reference prints.
bool format_range_ext(const Batch& b, uint64_t lo, uint64_t hi, const Unitigs* correct, bool split_no_overlap, std::string& pbuf,
                      std::string& nbuf, std::string& obuf, std::string& bug) {
    const int32_t* paths = static_cast<const int32_t*>(b.pin->paths.p);
    const uint64_t* poffs = static_cast<const uint64_t*>(b.pin->poffs.p);
    const uint8_t* status = static_cast<const uint8_t*>(b.pin->status.p);
    pbuf.clear(); nbuf.clear(); obuf.clear();
    std::string walk, tmp, rc;
    for (uint64_t i = lo; i < hi; ++i) {
        const RecSlice& r = b.recs[i];
        const uint64_t np = poffs[i + 1] - poffs[i];
        if (np) {
            if (correct && (np < 2 || !recover_path(*correct, paths + poffs[i], np, r.sl, walk, tmp, rc))) {
                bug = walk + " " + tmp;
                return false;
            }
            pbuf.append(r.h, r.hl);
            pbuf.push_back('\n');
            if (correct) {
                if (status[i] & BGR_ST_RC) {  // the path was found on the reverse complement: turn the spelled read back
                    rc.resize(walk.size());
                    for (size_t j = 0; j < walk.size(); ++j) rc[j] = rc_char(walk[walk.size() - 1 - j]);
                    pbuf.append(rc);
                } else {
                    pbuf.append(walk);
                }
            } else {
                char num[16];
                for (uint64_t j = poffs[i]; j < poffs[i + 1]; ++j) { char* e = put_int(num, paths[j]); pbuf.append(num, (size_t)(e - num)); }
            }
            pbuf.push_back('\n');
        } else {
            std::string& o = (split_no_overlap && (status[i] & BGR_ST_MASK) == BGR_ST_NOANCHOR) ? obuf : nbuf;
            o.append(r.h, r.hl);
            o.push_back('\n');
            o.append(r.s, r.sl);
            o.push_back('\n');
        }
    }
    return true;
}

}  // namespace

extern "C" void bgr_host_cache_release(void) {
    std::lock_guard<std::mutex> l(g_pin_cache_m);
    g_pin_cache.clear();
}

namespace {

// Bytes [begin, end) of one input file (end beyond the file: to its end).  `mf`: the file's image when the caller has opened it already
// (the lanes of a split run share the images; a FIFO can be opened only once), else the producer opens it when it gets there.
struct InputSpan {
    std::string file;
    std::shared_ptr<MappedFile> mf;
    uint64_t begin = 0, end = ~0ull;
};

// One pipeline: the devices first_device .. first_device + n_gpus - 1 of `opt` map `inputs` in order into ONE pair of output files.
// `cancel` (optional): shared with the other lanes of a split run -- a lane that fails stops them all.
int align_all_impl(bgr_graph* graph, const bgr_params* prm, const bgr_run_options* opt, const std::vector<InputSpan>& inputs,
                   const char* paths_file, const char* notaligned_file, uint64_t counters_out[5], double* mapping_seconds, std::atomic<bool>* cancel) {
    const unsigned n_gpus = std::max<uint32_t>(1, opt->n_gpus);
    unsigned threads = std::max<uint32_t>(1, opt->threads);
    // The threads of this run (and the page-locked memory they allocate) on the NUMA node of the devices they feed: a copy engine
    // reading staging buffers across the socket link runs at about half its rate.  Only when all devices of the run share a node;
    // the calling thread's affinity is restored at the end, and the pool of host threads is clamped to the CPUs of the restricted
    // set (opt->numa = 1, or the option numa = 0, leaves the affinity alone).
    cpu_set_t old_aff, want_aff;
    bool aff_changed = false;
    if (!opt->numa && bgr::opt("numa") != 0) {
        CPU_ZERO(&want_aff);
        bool same = true;
        std::string first_list;
        for (unsigned g = 0; g < n_gpus && same; ++g) {
            char list[512];
            if (bgr_device_local_cpus((int)(opt->first_device + g), list, sizeof(list)) != BGR_OK) { same = false; break; }
            if (g == 0) first_list = list; else if (first_list != list) same = false;
        }
        if (same && !first_list.empty() && sched_getaffinity(0, sizeof(old_aff), &old_aff) == 0) {
            int n_set = 0;
            const char* c = first_list.c_str();
            while (*c) {  // "a-b,c,d-e"
                char* e = nullptr;
                long a = strtol(c, &e, 10), b = a;
                if (e == c) break;
                if (*e == '-') { c = e + 1; b = strtol(c, &e, 10); }
                for (long i = a; i <= b && i < CPU_SETSIZE; ++i) if (CPU_ISSET(i, &old_aff)) { CPU_SET(i, &want_aff); ++n_set; }
                c = *e == ',' ? e + 1 : e;
                if (*e != ',' && *e != 0) break;
            }
            if (n_set > 0 && sched_setaffinity(0, sizeof(want_aff), &want_aff) == 0) {
                aff_changed = true;
                threads = std::min<unsigned>(threads, (unsigned)n_set);  // (a pool larger than the CPU set it may run on only oversubscribes it)
            }
        }
    }
    struct RestoreAffinity { bool on; cpu_set_t* old; ~RestoreAffinity() { if (on) sched_setaffinity(0, sizeof(cpu_set_t), old); } } restore_aff{aff_changed, &old_aff};
    // Defaults: 128k reads per batch keeps the page-locked staging small (it costs ~0.2 s per GB to allocate) and the
    // pipeline fine-grained; the parser chunk is a thread's share of a batch.
    // (one launch addresses its path arena with 32 bits: a batch stays below 4 M reads and ~1 G bases)
    const bool writes = prm->mode != BGR_MODE_EXHAUSTIVE || opt->write_exhaustive;
    const bool correction = opt->correction && prm->mode != BGR_MODE_EXHAUSTIVE;  // alignPartExhaustive ignores -c
    const bool progress_blocks = opt->echo_files && prm->mode == BGR_MODE_EXHAUSTIVE;
    // Text route: the device parses, packs, maps and formats (bgr_align_fasta_text); the host only moves bytes.  FASTA, the
    // reference's two output files and no -b progress blocks (those count getReads() calls, which only the host parser tracks).
    // Correction mode too (the device spells the reads from its 2-bit unitig store) unless the graph has non-ACGT unitig characters.
    bgr_graph_info_t gi_route;
    if (bgr_graph_info(graph, &gi_route) != BGR_OK) return BGR_E_ARG;
    // (round 4: -b WITH its progress blocks too, for FASTA -- the device says what became of every record, bgr_text_batch.record_info_out, and
    // the ordered writer counts getReads() calls from that; FASTQ with progress blocks stays on the host route)
    const bool text_route = opt->route == 0 && !(correction && gi_route.has_exceptions) && !opt->no_overlap_file && !(progress_blocks && opt->fastq);
    const bool text_progress = text_route && progress_blocks;
    const uint64_t batch_reads = std::min<uint64_t>(opt->batch_reads ? opt->batch_reads : (text_route ? 1ull << 18 : 1ull << 17), 4ull << 20);
    // text route: bytes of a batch.  At most kTextPieceMax: the device addresses the two formatted streams with 32 bits, and a piece of B
    // bytes gives at most 12 B + (its own bytes) of records (--batch beyond ~1.9 M reads of 150 bp is cut to that)
    const uint64_t kTextPieceMax = 320ull << 20;
    const uint64_t piece_bytes = std::min<uint64_t>(std::max<uint64_t>(batch_reads * 170, 4096), kTextPieceMax - (16ull << 20));
    uint64_t batch_bases_cap = 1ull << 30;
    if (const int64_t cap = bgr::opt("test.bases_cap")) batch_bases_cap = (uint64_t)cap;  // (test hook: walk the cut with small inputs)
    const uint64_t chunk_bytes = opt->chunk_bytes ? opt->chunk_bytes
                                                  : std::min<uint64_t>(8ull << 20, std::max<uint64_t>(256ull << 10, batch_reads * 170 / threads));
    Unitigs unitigs;
    if (correction) {
        if (bgr_graph_unitigs(graph, &unitigs.seqs, &unitigs.offs, &unitigs.n) != BGR_OK) return BGR_E_ARG;
        bgr_graph_info_t gi0;
        if (bgr_graph_info(graph, &gi0) != BGR_OK) return BGR_E_ARG;
        unitigs.k = gi0.k;
    }
    FILE* ovlF = nullptr;
    if (opt->no_overlap_file) {
        ovlF = fopen(opt->no_overlap_file, "wb");
        if (!ovlF) return bgr::set_error(BGR_E_IO, "bgr_align_all: cannot open the no-overlap file");
    }
    const bool extended = correction || ovlF != nullptr;
    bgr_graph_info_t gi;
    if (bgr_graph_info(graph, &gi) != BGR_OK) return BGR_E_ARG;

    FILE* pathF = fopen(paths_file, "wb");        // aligner.h:85
    FILE* notF = fopen(notaligned_file, "wb");    // aligner.h:86
    if (!pathF || !notF) {
        if (pathF) fclose(pathF);
        if (notF) fclose(notF);
        return bgr::set_error(BGR_E_IO, "bgr_align_all: cannot open the output files");
    }

    // two aligners (streams) per device so that consecutive batches overlap copy and compute (text route: the piece is already on its
    // way to the device when a worker takes the batch; measured 185 / 163 / 149 Mreads/s end to end with 2 / 3 / 4 workers per device)
    unsigned per_dev = 2;
    if (const int64_t w = bgr::opt("workers_per_device")) per_dev = (unsigned)w;  // (tuning / experiments)
    std::vector<bgr_aligner*> aligners;
    for (unsigned g = 0; g < n_gpus; ++g) {
        for (unsigned j = 0; j < per_dev; ++j) {
            bgr_aligner* a = nullptr;
            int rc = bgr_aligner_create(graph, (int)(opt->first_device + g), &a);
            if (rc != BGR_OK) {
                for (auto* x : aligners) bgr_aligner_destroy(x);
                fclose(pathF); fclose(notF);
                return rc;
            }
            // (nobody reads bgr_aligner_kernel_times here: no events around the kernels -- they cost a 262 144-read piece's mapping launch a tenth of its time)
            (void)bgr_aligner_set_knob(a, BGR_KNOB_KERNEL_EVENTS, 0);
            aligners.push_back(a);
        }
    }

    auto t_start = std::chrono::steady_clock::now();
    WorkerPool pool(threads + 2);  // + 2: the stage threads mostly wait inside run()
    // A fixed pool of batch objects circulates producer -> GPU workers -> writer -> producer, so the pinned
    // buffers are allocated once and the number of batches in flight is bounded.
    size_t extra_sets = text_route ? 3 : 0;  // text route: a set is held from the read of its piece until its streams are written
    if (bgr::opt("extra_sets") >= 0) extra_sets = (size_t)bgr::opt("extra_sets");
    const size_t max_batches = aligners.size() * 2 + 2 + extra_sets;
    Channel<std::unique_ptr<Batch>> to_gather(max_batches), to_out(max_batches), free_batches(max_batches);
    std::vector<std::unique_ptr<Channel<std::unique_ptr<Batch>>>> to_gpu;  // one queue per device: its workers and its share of the batches
    for (unsigned g = 0; g < n_gpus; ++g) to_gpu.push_back(std::make_unique<Channel<std::unique_ptr<Batch>>>(max_batches));
    std::atomic<size_t> created{0};
    auto take_batch = [&](std::unique_ptr<Batch>& b) {  // reuse a finished batch; create one only while below the cap
        if (free_batches.try_pop(b)) return true;
        if (created.fetch_add(1) < max_batches) { b = std::make_unique<Batch>(); return true; }
        created.fetch_sub(1);
        return free_batches.pop(b);
    };
    const bool timing = bgr::opt("timing") != 0;
    pool.timing = timing;
    // text route: room for a piece's two record streams.  A mapped 150 bp read leaves ~27 bytes of its 165 in `paths`, an unmapped one all
    // of them in the other file: a quarter of the piece each to start with (page-locked memory costs ~0.2 s per GB: less of it, and a fresh
    // process reaches its rate sooner); a stream that does not fit comes back as BGR_E_CAPACITY and the set's buffer grows once (-c, whose
    // paths stream is header + read, starts at the full size)
    const uint64_t out_div = correction ? 1 : 4;
    // FASTQ on the text route: only the header and read lines of a piece go to the device (option fastq_gather = 0: the four-line records as they are)
    const bool fastq_gather = bgr::opt("fastq_gather") != 0;
    std::atomic<uint64_t> us_parse{0}, us_gather{0}, us_gpu{0}, us_format{0}, us_write{0}, us_alloc{0};
    auto now_us = []() { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    std::atomic<bool> failed_here{false};
    std::atomic<bool>& failed = cancel ? *cancel : failed_here;  // (a split run: one flag for all lanes)
    bool failed_first = false;  // this pipeline recorded the failure (first_rc / first_err are its own)
    std::mutex err_m;
    std::string first_err;
    int first_rc = BGR_OK;
    auto fail = [&](int rc, const std::string& msg) {
        std::lock_guard<std::mutex> l(err_m);
        if (!failed.exchange(true)) { first_rc = rc; first_err = msg; failed_first = true; }
    };
    // The page-locked buffers cost ~0.2 s per GB to allocate, so they are few (one set per batch between gather and
    // format), sized from the input up front, allocated by their own thread while the parsers already run, and reused.
    const size_t n_pins = aligners.size() * 2 + 1 + extra_sets;
    Channel<std::unique_ptr<Pinned>> free_pins(n_pins + 1);
    std::thread pin_allocator([&]() {
        const uint64_t ta0 = now_us();
        uint64_t total_in = 0, max_file = 0;
        for (const InputSpan& in : inputs) {
            uint64_t sz = 0;
            struct stat st;
            if (in.mf) sz = in.mf->size;
            else if (stat(in.file.c_str(), &st) == 0) sz = (uint64_t)st.st_size;
            sz = std::min(sz, in.end) - std::min(sz, in.begin);
            total_in += sz;
            max_file = std::max(max_file, sz);
        }
        const uint64_t group0 = std::max<uint64_t>(threads, (batch_reads * 170) / chunk_bytes);
        const uint64_t est_bytes = std::min<uint64_t>(max_file, opt->fastq ? batch_reads * 160 : group0 * chunk_bytes);
        const uint64_t est_n = std::min<uint64_t>(batch_reads + batch_reads / 4, est_bytes / 16 + 1);
        const size_t need = est_bytes ? (size_t)std::min<uint64_t>(n_pins, (total_in + est_bytes - 1) / est_bytes + 1) : 1;
        const uint64_t est_piece = std::min<uint64_t>(max_file, opt->fastq ? batch_reads * 330 : piece_bytes + piece_bytes / 16);
        const size_t need_text = est_piece ? (size_t)std::min<uint64_t>(n_pins, (total_in + est_piece - 1) / est_piece + 1) : 1;
        for (size_t i = 0; i < n_pins; ++i) {
            std::unique_ptr<Pinned> pn;
            {
                std::lock_guard<std::mutex> l(g_pin_cache_m);  // a set left by an earlier run of this process
                if (!g_pin_cache.empty()) { pn = std::move(g_pin_cache.back()); g_pin_cache.pop_back(); }
            }
            if (!pn) pn = std::make_unique<Pinned>();
            if (text_route) {  // best effort: the stages grow what turns out too small
                if (i < need_text) (void)(pn->text.ensure(est_piece + 64) && pn->ptext.ensure(est_piece / out_div + 4096) && pn->ntext.ensure(est_piece / out_div + 4096));
            } else if (i < need) {
                (void)(pn->fw3.ensure(bgr::packed_plane_words(est_n, est_bytes) * 8) && pn->hasn.ensure((est_n / 32 + 2) * 4) && pn->offs.ensure((est_n + 1) * 8) &&
                       pn->paths.ensure((8 * est_n + 4096) * 4) && pn->poffs.ensure((est_n + 1) * 8) && pn->status.ensure(est_n + 1));
            }
            if (!free_pins.push(std::move(pn))) break;
        }
        us_alloc += now_us() - ta0;
    });

    // ---- stage 1: parse + gather -----------------------------------------------------------------------
    std::thread producer([&]() {
        uint64_t next_index = 0;
        ProgressMarker marker;
        marker.on = progress_blocks;
        std::vector<Mark> pending;      // marks waiting for the next batch (they go in front of its first read)
        std::vector<uint64_t> mark_idx;
        auto open_batch = [&](std::unique_ptr<Batch>& b, const std::shared_ptr<MappedFile>& mf) {
            if (!take_batch(b)) return false;
            b->file = mf;
            b->bases = 0;
            b->text_piece = b->dev_text = b->fastq_piece = false;
            b->fq_parts.clear();
            b->fq_nl.clear();
            b->t_gathered = 0;
            b->text_origin = b->first_of_file = b->last_of_file = false;
            b->n_attempts = 0;
            b->att_of_read.clear();
            b->marks.clear();
            for (auto& m : pending) { m.pos = 0; b->marks.push_back(std::move(m)); }
            pending.clear();
            return true;
        };
        auto emit = [&](std::unique_ptr<Batch> b) {  // number the batch in input order and hand it to the gatherer
            b->index = next_index++;
            b->n = b->recs.size();
            return to_gather.push(std::move(b));
        };
        auto drop = [&](std::unique_ptr<Batch> b) {  // an opened batch that got neither reads nor marks
            b->chunks.clear(); b->shared_chunks.reset(); b->file.reset(); b->recs.clear();
            free_batches.push(std::move(b));
        };
        bool ok = true;
        for (size_t fi = 0; fi < inputs.size() && !failed && ok; ++fi) {  // aligner.cpp:552-586: the files of the comma-separated list, in order
            const InputSpan& in = inputs[fi];
            const std::string& file = in.file;
            if (opt->echo_files) pending.push_back({0, 1, file});  // aligner.cpp:559,576  cout<<file<<endl
            std::shared_ptr<MappedFile> mf = in.mf;
            if (!mf) {
                mf = std::make_shared<MappedFile>();
                std::string err;
                if (!mf->open(file, err)) { fail(BGR_E_IO, "read file: " + err); break; }
            }
            // the span of the file this pipeline maps: all of it, or -- a lane of a split run -- from one record start to another
            const uint64_t s_begin = std::min(in.begin, mf->size), s_end = std::max(s_begin, std::min(in.end, mf->size));
            const char* const sdata = mf->data + s_begin;
            const uint64_t ssize = s_end - s_begin;
            if (opt->fastq && (s_begin != 0 || s_end != mf->size)) { fail(BGR_E_ARG, "bgr_align_all: a FASTQ file is mapped whole (its records are counted from its first line)"); break; }
            marker.begin_file();
            uint64_t file_iters = 0;  // getReads() iterations of this file handed on so far
            if (opt->fastq) {
                // FASTQ: newline counts first (they fix which line of a record every chunk starts in), then the chunks
                // group by group, so that the later stages already work on the first batches while the rest is parsed
                bgr::FastqPlan plan(mf->data, mf->size, chunk_bytes);
                if (text_route && fastq_gather) plan.keep_newlines();  // (the count pass keeps the newline positions: the gather does not scan again)
                const size_t nc = plan.chunks();
                uint64_t tp0 = now_us();
                std::unique_ptr<Batch> b;
                if (text_route) {
                    // text route: every record in front of the file's last getReads() call boundary is a plain four-line record -- pieces of
                    // whole records go to the device as they are, cut while the newline counts of the chunks behind them are still being
                    // taken (a boundary is final once the records counted so far reach beyond it); the rest of the file (the phantom record
                    // at its end, truncated tails: aligner.cpp:51-68) goes through the sequential state machine on the host, behind them
                    const size_t wave = std::max<size_t>((size_t)threads * 8, 64);
                    uint64_t r0 = 0, o0 = 0;
                    for (size_t c_done = 0; ok && !failed;) {
                        const size_t c_end = std::min(nc, c_done + wave);
                        tp0 = now_us();
                        if (c_end > c_done) pool.run(c_end - c_done, [&](size_t c) { plan.count_chunk(c_done + c); }, 1);
                        plan.extend_counts(c_end);
                        c_done = c_end;
                        const bool last = c_done >= nc;
                        if (last) plan.finish_counts();
                        us_parse += now_us() - tp0;
                        const uint64_t counted = plan.records_counted();
                        const uint64_t limit = last ? plan.par_records() : (counted ? ((counted - 1) / 10000) * 10000 : 0);
                        while (ok && !failed && r0 < limit && (last || r0 + batch_reads <= limit)) {
                            uint64_t r1 = std::min<uint64_t>(limit, r0 + batch_reads), o1 = plan.record_offset(r1);
                            while (o1 - o0 > (1ull << 30) && r1 > r0 + 1) {  // long reads: a piece stays under the 2 GiB of one call
                                r1 = r0 + (r1 - r0) / 2;
                                o1 = plan.record_offset(r1);
                            }
                            std::unique_ptr<Batch> tb;
                            if (!open_batch(tb, mf)) { ok = false; break; }
                            tb->text_piece = o1 - o0 < (1ull << 31);  // (one record of 2 GiB: the host parser takes it)
                            tb->fastq_piece = true;
                            tb->t_begin = o0; tb->t_end = o1;
                            if (fastq_gather) {
                                plan.piece_parts(o0, o1, r0, tb->fq_parts);
                                tb->fq_chunk_bytes = plan.chunk_bytes();
                                for (const auto& pt : tb->fq_parts) tb->fq_nl.push_back(plan.newlines_of((size_t)(pt.first / plan.chunk_bytes())));
                                plan.release_newlines((size_t)(o1 / plan.chunk_bytes()));  // (chunks wholly in front of the next piece)
                            }
                            if (!tb->text_piece) {
                                tb->chunks.clear();
                                tb->chunks.push_back(std::make_unique<ParsedChunk>());
                                bgr::parse_fastq_records(mf->data, o0, o1, *tb->chunks[0]);
                                tb->recs = tb->chunks[0]->recs;
                            }
                            ok = emit(std::move(tb));
                            r0 = r1; o0 = o1;
                        }
                        if (last) break;
                    }
                    if (ok && !failed) {
                        ParsedChunk tl;
                        tp0 = now_us();
                        bgr::parse_fastq_from(mf->data, o0, mf->size, tl);
                        us_parse += now_us() - tp0;
                        const std::vector<RecSlice>& rs = tl.recs;
                        size_t lo = 0;
                        while (lo < rs.size() && ok) {
                            std::unique_ptr<Batch> hb;
                            if (!open_batch(hb, mf)) { ok = false; break; }
                            const size_t take = std::min<size_t>(rs.size() - lo, (size_t)batch_reads);
                            hb->recs.assign(rs.begin() + lo, rs.begin() + lo + take);
                            lo += take;
                            ok = emit(std::move(hb));
                        }
                    }
                    continue;
                }
                // FASTQ: newline counts first (they fix which line of a record every chunk starts in), then the chunks
                // group by group, so that the later stages already work on the first batches while the rest is parsed
                pool.run(nc, [&](size_t c) { plan.count_chunk(c); }, 1);
                plan.finish_counts();
                us_parse += now_us() - tp0;
                auto feed = [&](const ParsedChunk& ch) {  // slices point into the file image only (no joined storage)
                    const std::vector<RecSlice>& rs = ch.recs;
                    marker.chunk(ch, file_iters, mark_idx);
                    file_iters += ch.iters;
                    size_t lo = 0, mi = 0;
                    while (lo < rs.size() && ok) {
                        if (!b && !open_batch(b, mf)) { ok = false; break; }
                        size_t take = std::min<size_t>(rs.size() - lo, (size_t)batch_reads - b->recs.size());
                        uint64_t acc = b->bases;
                        for (size_t j = 0; j < take; ++j) {  // long reads: close the batch at ~1 G bases
                            acc += rs[lo + j].sl;
                            if (acc >= batch_bases_cap) { take = j + 1; break; }
                        }
                        b->bases = acc;
                        for (; mi < mark_idx.size() && mark_idx[mi] < lo + take; ++mi) b->marks.push_back({b->recs.size() + (mark_idx[mi] - lo), 0, ""});
                        b->recs.insert(b->recs.end(), rs.begin() + lo, rs.begin() + lo + take);
                        lo += take;
                        if (b->recs.size() >= batch_reads || b->bases >= batch_bases_cap) ok = emit(std::move(b));
                    }
                    for (; mi < mark_idx.size(); ++mi) {  // due behind the chunk's last record
                        if (b) b->marks.push_back({b->recs.size(), 0, ""}); else pending.push_back({0, 0, ""});
                    }
                };
                const size_t group = std::max<size_t>(threads, (size_t)((batch_reads * 330) / chunk_bytes));
                bool tail = plan.sequential_only();
                for (size_t c = 0; c < nc && !tail && ok && !failed; c += group) {
                    const size_t c_end = std::min(nc, c + group);
                    std::vector<ParsedChunk> fq(c_end - c);
                    for (auto& ch : fq) ch.track_iters = marker.on;
                    std::vector<char> done(c_end - c, 1);
                    tp0 = now_us();
                    pool.run(c_end - c, [&](size_t j) { done[j] = plan.parse_chunk(c + j, fq[j]) ? 1 : 0; }, 1);
                    us_parse += now_us() - tp0;
                    for (size_t j = 0; j < fq.size() && ok; ++j) {
                        feed(fq[j]);
                        if (!done[j]) { tail = true; break; }  // this chunk ran into the sequential tail: nothing after it is parsed here
                    }
                }
                if (ok && !failed) {
                    ParsedChunk tl;
                    tl.track_iters = marker.on;
                    tp0 = now_us();
                    plan.parse_tail(tl);
                    us_parse += now_us() - tp0;
                    feed(tl);
                }
                for (unsigned d = marker.end_file(std::max<uint64_t>(1, file_iters)); d; --d) {
                    if (b) b->marks.push_back({b->recs.size(), 0, ""}); else pending.push_back({0, 0, ""});
                }
                if (b) {
                    if (ok && (!b->recs.empty() || !b->marks.empty())) ok = emit(std::move(b));
                    else drop(std::move(b));
                }
                continue;
            }
            if (text_route) {  // pieces of the file as they are: the device finds the records (the worker falls back per piece)
                std::vector<uint64_t> cuts = bgr::split_fasta(sdata, ssize, piece_bytes);
                for (size_t c = 0; c < cuts.size() && !failed && ok; ++c) {
                    std::unique_ptr<Batch> b;
                    if (!open_batch(b, mf)) { ok = false; break; }
                    b->text_piece = true;
                    b->dev_text = false;
                    b->t_begin = s_begin + cuts[c];
                    b->t_end = s_begin + (c + 1 < cuts.size() ? cuts[c + 1] : ssize);
                    b->text_origin = true;
                    b->first_of_file = c == 0;
                    b->last_of_file = c + 1 == cuts.size();
                    ok = emit(std::move(b));
                }
                continue;
            }
            std::vector<uint64_t> starts = bgr::split_fasta(sdata, ssize, chunk_bytes);
            size_t c = 0;
            while (c < starts.size() && !failed && ok) {
                // as many chunks as it takes to reach ~batch_reads (estimated from bytes), at least `threads`
                size_t group = std::max<size_t>(threads, (size_t)((batch_reads * 170) / chunk_bytes));
                size_t c_end = std::min(starts.size(), c + group);
                std::unique_ptr<Batch> b;
                if (!open_batch(b, mf)) { ok = false; break; }
                b->chunks.resize(c_end - c);
                for (auto& ch : b->chunks) { ch = std::make_unique<ParsedChunk>(); ch->track_iters = marker.on; }
                Batch* bp = b.get();
                const uint64_t tp0 = now_us();
                pool.run(c_end - c, [&](size_t j) {
                    uint64_t e = (c + j + 1 < starts.size()) ? starts[c + j + 1] : ssize;
                    bgr::parse_fasta_chunk(sdata, starts[c + j], e, gi.k, *bp->chunks[j]);
                }, 1);
                size_t total = 0;
                for (auto& ch : b->chunks) total += ch->recs.size();
                b->recs.reserve(total);
                // (one launch addresses its path arena with 32 bits: a group of chunks with very long records -- a large --chunk-bytes,
                // many threads -- is handed on in pieces of at most ~1 G bases, like the FASTQ feeder does)
                // every batch cut from the group keeps the group's storage alive (the records point into it)
                auto keep = std::make_shared<std::vector<std::unique_ptr<ParsedChunk>>>(std::move(b->chunks));
                b->chunks.clear();
                b->shared_chunks = keep;
                for (auto& ch : *keep) {
                    marker.chunk(*ch, file_iters, mark_idx);
                    file_iters += ch->iters;
                    size_t mi = 0;
                    for (size_t ri = 0; ri <= ch->recs.size() && ok; ++ri) {
                        for (; mi < mark_idx.size() && mark_idx[mi] == ri; ++mi) b->marks.push_back({b->recs.size(), 0, ""});
                        if (ri == ch->recs.size()) break;
                        b->recs.push_back(ch->recs[ri]);
                        b->bases += ch->recs[ri].sl;
                        if (b->bases >= batch_bases_cap) {  // hand this piece on, go on in a fresh batch
                            ok = emit(std::move(b));
                            if (!ok || !open_batch(b, mf)) { ok = false; break; }
                            b->shared_chunks = keep;
                        }
                    }
                    if (!ok) break;
                }
                us_parse += now_us() - tp0;
                c = c_end;
                if (!ok) break;
                if (c >= starts.size())
                    for (unsigned d = marker.end_file(std::max<uint64_t>(1, file_iters)); d; --d) b->marks.push_back({b->recs.size(), 0, ""});
                if (b->recs.empty() && b->marks.empty()) { drop(std::move(b)); continue; }
                ok = ok && emit(std::move(b));
            }
        }
        if (ok && !failed && !pending.empty()) {  // what is printed after the last read: an empty batch carries it to the writer
            std::unique_ptr<Batch> b;
            if (open_batch(b, nullptr)) emit(std::move(b));
        }
        to_gather.close();
    });

    // ---- stage 1b: gather the sequences of a batch into (pooled) pinned memory ---------------------------
    // false = the batch could not be staged (the error is recorded); it still travels on, so that the writer sees every
    // index and recycles every batch (a dropped batch would leave the producer waiting for a free one for ever)
    auto gather = [&](Batch& b) {
        uint64_t bases = 0;
        if (!b.pin && !free_pins.pop(b.pin)) { fail(BGR_E_INTERNAL, "pinned buffer pool closed"); return false; }
        const uint64_t tg0 = now_us();
        if (!b.pin->offs.ensure((b.n + 1) * 8)) { fail(BGR_E_HIP, bgr_last_error()); return false; }
        uint64_t* offs = static_cast<uint64_t*>(b.pin->offs.p);
        for (uint64_t i = 0; i < b.n; ++i) { offs[i] = bases; bases += b.recs[i].sl; }
        offs[b.n] = bases;
        b.bases = bases;
        // typical paths are a handful of ints; the worker fetches again with the full bound if not.  A batch large enough
        // for bgr_align_batch to map it in pieces gets the full bound at once (there is no single result to fetch again).
        b.path_cap = 2 * (bases + 8 * b.n) >= (1ull << 31) ? bases + 8 * b.n + 8 : 8 * b.n + 4096;
        if (!b.pin->fw3.ensure(bgr::packed_plane_words(b.n, bases) * 8) || !b.pin->hasn.ensure((b.n / 32 + 2) * 4) || !b.pin->paths.ensure(b.path_cap * 4) ||
            !b.pin->poffs.ensure((b.n + 1) * 8) || !b.pin->status.ensure(b.n + 1)) { fail(BGR_E_HIP, bgr_last_error()); return false; }
        uint64_t* fw3 = static_cast<uint64_t*>(b.pin->fw3.p);
        uint32_t* hasn = static_cast<uint32_t*>(b.pin->hasn.p);
        us_alloc += now_us() - tg0;
        const uint64_t tg1 = now_us();
        // pack (instead of copy) the sequences into the page-locked plane: every thread a range of whole bitmap words
        const uint64_t per = (((b.n + threads - 1) / threads) + 31) & ~31ull;
        Batch* bp = &b;
        struct Part { std::vector<uint32_t> idx; std::vector<uint64_t> val; uint32_t max_len = 0; };
        std::vector<Part> parts(threads);
        pool.run(threads, [&](size_t t) {
            const uint64_t lo = t * per, hi = std::min<uint64_t>(bp->n, lo + per);
            if (lo >= hi) return;
            Part& pt = parts[t];
            std::vector<uint64_t> nm;
            memset(hasn + lo / 32, 0, ((hi - lo + 31) / 32) * 4);
            for (uint64_t i = lo; i < hi; ++i) {
                const uint32_t len = bp->recs[i].sl, words = (len + 31) >> 5;
                if (nm.size() < words) nm.resize(words);
                const uint64_t w0 = bgr::packed_word_offset(offs[i], i);
                pt.max_len = std::max(pt.max_len, len);
                if (bgr::pack_read(bp->recs[i].s, len, fw3 + w0, nm.data())) {
                    hasn[i >> 5] |= 1u << (i & 31);
                    for (uint32_t j = 0; j < words; ++j) { pt.idx.push_back((uint32_t)(w0 + j)); pt.val.push_back(nm[j]); }
                }
            }
        }, 2);
        uint64_t nmc = 0;
        b.pin->max_len = 0;
        for (const Part& pt : parts) { nmc += pt.idx.size(); b.pin->max_len = std::max(b.pin->max_len, pt.max_len); }
        b.pin->nm_count = nmc;
        if (nmc) {
            if (!b.pin->nm_idx.ensure(nmc * 4) || !b.pin->nm_val.ensure(nmc * 8)) { fail(BGR_E_HIP, bgr_last_error()); return false; }
            uint64_t at = 0;
            for (const Part& pt : parts) {
                if (pt.idx.empty()) continue;
                memcpy(static_cast<uint32_t*>(b.pin->nm_idx.p) + at, pt.idx.data(), pt.idx.size() * 4);
                memcpy(static_cast<uint64_t*>(b.pin->nm_val.p) + at, pt.val.data(), pt.val.size() * 8);
                at += pt.idx.size();
            }
        }
        us_gather += now_us() - tg1;
        return true;
    };
    // text route: the piece of the file into page-locked memory as it is (pread / memcpy in parallel), output buffers sized
    auto stage_text = [&](Batch& b) {
        if (!b.pin && !free_pins.pop(b.pin)) { fail(BGR_E_INTERNAL, "pinned buffer pool closed"); return false; }
        const uint64_t tg0 = now_us();
        const uint64_t bytes = b.t_end - b.t_begin;
        if (!b.pin->text.ensure(bytes + 64) || !b.pin->ptext.ensure(bytes / out_div + 4096) || !b.pin->ntext.ensure(bytes / out_div + 4096)) { fail(BGR_E_HIP, bgr_last_error()); return false; }
        us_alloc += now_us() - tg0;
        const uint64_t tg1 = now_us();
        Batch* bp = &b;
        if (b.pin->stages.size() < n_gpus) b.pin->stages.resize(n_gpus, nullptr);
        bgr_text_stage*& st = b.pin->stages[b.dev];
        // (a set cached by an earlier run of the process may carry the stage of another device in this place: the index is relative to the run)
        if (st && bgr_text_stage_device(st) != (int)(opt->first_device + b.dev)) { bgr_text_stage_destroy(st); st = nullptr; }
        if (!st && bgr_text_stage_create((int)(opt->first_device + b.dev), &st) != BGR_OK) { fail(BGR_E_HIP, bgr_last_error()); return false; }
        if (b.fastq_piece && !b.fq_parts.empty()) {
            // FASTQ: the header and read lines of every part, gathered where the part's bytes would have gone (never more than they are);
            // the parts travel one after the other and lie back to back on the device
            const size_t np = b.fq_parts.size();
            std::vector<const char*> ptrs(np);
            std::vector<uint64_t> lens(np);
            pool.run(np, [&](size_t j) {
                const uint64_t lo = bp->fq_parts[j].first, hi = j + 1 < np ? bp->fq_parts[j + 1].first : bp->t_end;
                char* dst = static_cast<char*>(bp->pin->text.p) + (lo - bp->t_begin);
                ptrs[j] = dst;
                const std::vector<uint32_t>* nl = j < bp->fq_nl.size() ? bp->fq_nl[j].get() : nullptr;
                lens[j] = nl ? bgr::fastq_gather_lines_at(bp->file->data, lo, hi, bp->fq_parts[j].second, dst, nl->data(), nl->size(), (lo / bp->fq_chunk_bytes) * bp->fq_chunk_bytes)
                             : bgr::fastq_gather_lines(bp->file->data, lo, hi, bp->fq_parts[j].second, dst);
            }, 2);
            us_gather += now_us() - tg1;
            b.t_gathered = 0;
            for (uint64_t l : lens) b.t_gathered += l;
            if (bgr_text_stage_upload_parts(st, (uint32_t)np, ptrs.data(), lens.data()) != BGR_OK) { fail(BGR_E_HIP, bgr_last_error()); return false; }
            return true;
        }
        const uint64_t part = 4ull << 20;
        const size_t parts = (size_t)((bytes + part - 1) / part);
        std::atomic<bool> okc{true};
        pool.run(parts, [&](size_t j) {
            const uint64_t lo = j * part, len = std::min<uint64_t>(part, bytes - lo);
            if (!bp->file->copy_out(static_cast<char*>(bp->pin->text.p) + lo, bp->t_begin + lo, len)) okc = false;
        }, 2);
        us_gather += now_us() - tg1;
        if (!okc) { fail(BGR_E_IO, "read error on the read file"); return false; }
        // ... and on towards its device at once, on the stage's own copy stream: the worker's call finds it there (or waits on the device)
        if (bgr_text_stage_upload(st, static_cast<const char*>(b.pin->text.p), bytes) != BGR_OK) { fail(BGR_E_HIP, bgr_last_error()); return false; }
        return true;
    };
    // a text piece through the host parser after all (the device found it irregular, or the file has shown to be): the exact
    // getReads state machine, chunk-parallel, then the batch goes the host route (gather, bgr_align_batch_packed, host formatter)
    auto host_parse_piece = [&](Batch& b) {
        const uint64_t tp0 = now_us();
        const char* base = b.file->data + b.t_begin;
        const uint64_t bytes = b.t_end - b.t_begin;
        if (b.fastq_piece) {  // whole four-line records
            b.chunks.clear();
            b.chunks.push_back(std::make_unique<ParsedChunk>());
            bgr::parse_fastq_records(base, 0, bytes, *b.chunks[0]);
            b.recs = b.chunks[0]->recs;
            b.n = b.recs.size();
            b.dev_text = false;
            us_parse += now_us() - tp0;
            return;
        }
        std::vector<uint64_t> starts = bgr::split_fasta(base, bytes, chunk_bytes);
        b.chunks.clear();
        b.chunks.resize(starts.size());
        for (auto& ch : b.chunks) { ch = std::make_unique<ParsedChunk>(); ch->track_iters = text_progress; }
        Batch* bp = &b;
        pool.run(starts.size(), [&](size_t j) {
            const uint64_t e = j + 1 < starts.size() ? starts[j + 1] : bytes;
            bgr::parse_fasta_chunk(base, starts[j], e, gi.k, *bp->chunks[j]);
        }, 1);
        b.recs.clear();
        b.att_of_read.clear();
        b.n_attempts = 0;
        for (auto& ch : b.chunks) {
            b.recs.insert(b.recs.end(), ch->recs.begin(), ch->recs.end());
            if (text_progress) for (uint32_t it : ch->rec_iter) b.att_of_read.push_back(b.n_attempts + it);  // (iterations of the piece: chunk after chunk)
            b.n_attempts += ch->iters;
        }
        b.n = b.recs.size();
        b.dev_text = false;
        us_parse += now_us() - tp0;
    };
    std::thread gatherer([&]() {
        std::unique_ptr<Batch> b;
        while (to_gather.pop(b)) {
            b->dev = (unsigned)(b->index % n_gpus);
            // this file is not of the device's shape: host route from here on.  So is a piece beyond what one text call takes: a piece is
            // cut at record starts, so a very long record can carry it past the 2 GiB of a call, or past kTextPieceMax bytes, the bound that
            // keeps the u32 offsets of the formatted streams exact (a path int is at most 12 characters and consumes at least one base)
            if (b->text_piece && !b->fastq_piece && (b->file->irregular_pieces.load() >= 2 || b->t_end - b->t_begin > kTextPieceMax)) {
                host_parse_piece(*b);
                b->text_piece = false;
            }
            if (failed || !(b->text_piece ? stage_text(*b) : gather(*b))) { if (!to_out.push(std::move(b))) break; continue; }
            const unsigned dv = b->dev;
            if (!to_gpu[dv]->push(std::move(b))) break;
        }
        to_gather.close();  // (after a failure: unblock the producer)
        for (auto& q : to_gpu) q->close();
    });

    // ---- stage 2: GPU workers --------------------------------------------------------------------------
    std::vector<std::thread> workers;
    std::atomic<unsigned> live_workers{(unsigned)aligners.size()};
    for (size_t w = 0; w < aligners.size(); ++w) {
        workers.emplace_back([&, w]() {
            std::unique_ptr<Batch> b;
            Channel<std::unique_ptr<Batch>>& my_q = *to_gpu[w / per_dev];  // (aligners are created device by device, per_dev each)
            while (my_q.pop(b)) {
                if (!failed && b->text_piece) {  // FASTA bytes in, record bytes out (bgr_align_fasta_text)
                    const uint64_t tq0 = now_us();
                    bgr_text_batch tb;
                    memset(&tb, 0, sizeof(tb));
                    tb.struct_size = sizeof(tb);
                    tb.text = static_cast<const char*>(b->pin->text.p);
                    tb.text_bytes = b->t_end - b->t_begin;
                    tb.stage = b->pin->stages.size() > b->dev ? b->pin->stages[b->dev] : nullptr;
                    tb.fastq = b->fastq_piece ? 1u : 0u;
                    if (b->fastq_piece && !b->fq_parts.empty()) {  // gathered: header and read lines only, held by the stage (parts with gaps on the host)
                        tb.text = nullptr;
                        tb.text_bytes = b->t_gathered;
                        tb.fastq = 2u;
                    }
                    tb.want_output = writes ? (correction ? 2u : 1u) : 0u;
                    tb.paths_out = static_cast<char*>(b->pin->ptext.p);
                    tb.paths_cap = b->pin->ptext.cap;
                    tb.notaligned_out = static_cast<char*>(b->pin->ntext.p);
                    tb.notaligned_cap = b->pin->ntext.cap;
                    int rc = BGR_OK;
                    if (text_progress) {  // what became of every record: the writer's progress blocks
                        const uint64_t words = tb.text_bytes / 24 + 1024;
                        if (!b->pin->info.ensure(words * 4)) rc = BGR_E_HIP;
                        tb.record_info_out = static_cast<uint32_t*>(b->pin->info.p);
                        tb.record_info_cap = words;
                    }
                    if (rc == BGR_OK) rc = bgr_align_fasta_text(aligners[w], prm, &tb);
                    if (rc == BGR_E_CAPACITY) {  // unusually long records: the same device results into larger buffers
                        if (!b->pin->ptext.ensure(tb.paths_bytes + 64) || !b->pin->ntext.ensure(tb.notaligned_bytes + 64)) rc = BGR_E_HIP;
                        else {
                            tb.paths_out = static_cast<char*>(b->pin->ptext.p); tb.paths_cap = b->pin->ptext.cap;
                            tb.notaligned_out = static_cast<char*>(b->pin->ntext.p); tb.notaligned_cap = b->pin->ntext.cap;
                            rc = bgr_aligner_fetch_text(aligners[w], &tb);
                        }
                    }
                    if (rc != BGR_OK) fail(rc, bgr_last_error());
                    else if (tb.irregular) {  // (2 = correction mode met a path that spells no walk: the host formatter reproduces the reference's exit)
                        if (tb.irregular == 1) b->file->irregular_pieces.fetch_add(1);
                        host_parse_piece(*b);
                        b->text_piece = false;
                        if (!gather(*b)) { if (!to_out.push(std::move(b))) break; continue; }
                    } else {
                        b->dev_text = true;
                        b->n = tb.n_accepted;
                        b->n_attempts = tb.n_records;
                        b->p_bytes = tb.paths_bytes;
                        b->n_bytes = tb.notaligned_bytes;
                    }
                    us_gpu += now_us() - tq0;
                }
                if (!failed && !b->text_piece) {
                    const uint64_t tq0 = now_us();
                    bgr_packed_reads pk;
                    pk.read_offsets = static_cast<const uint64_t*>(b->pin->offs.p);
                    pk.fw3 = static_cast<const uint64_t*>(b->pin->fw3.p);
                    pk.hasn = static_cast<const uint32_t*>(b->pin->hasn.p);
                    pk.nm_index = static_cast<const uint32_t*>(b->pin->nm_idx.p);
                    pk.nm_value = static_cast<const uint64_t*>(b->pin->nm_val.p);
                    pk.nm_count = b->pin->nm_count;
                    pk.max_read_len = b->pin->max_len;
                    int rc = bgr_align_batch_packed(aligners[w], prm, &pk, b->n, static_cast<int32_t*>(b->pin->paths.p), b->path_cap,
                                                    static_cast<uint64_t*>(b->pin->poffs.p), static_cast<uint8_t*>(b->pin->status.p));
                    if (rc == BGR_E_CAPACITY) {  // unusually long paths: fetch the same device results again into a full-size buffer
                        b->path_cap = b->bases + 8 * b->n + 8;
                        if (!b->pin->paths.ensure(b->path_cap * 4)) rc = BGR_E_HIP;
                        else rc = bgr_aligner_fetch(aligners[w], b->n, static_cast<int32_t*>(b->pin->paths.p), b->path_cap, static_cast<uint64_t*>(b->pin->poffs.p),
                                                    static_cast<uint8_t*>(b->pin->status.p));
                    }
                    if (rc != BGR_OK) fail(rc, bgr_last_error());
                    us_gpu += now_us() - tq0;
                }
                if (!to_out.push(std::move(b))) break;
            }
            if (--live_workers == 0) to_out.close();
        });
    }

    // ---- stage 3: format (range-parallel) in batch order, stage 4: one thread writes the formatted buffers ----------
    struct OutBufs { std::vector<std::string> pb, nb, ob; };
    // what the writer hands to the I/O thread: formatted buffers (host route) or the batch itself, whose page-locked buffers hold the
    // two record streams as the device wrote them (text route; the batch is recycled once they are on their way to the files)
    struct IoItem { std::unique_ptr<OutBufs> bufs; std::unique_ptr<Batch> batch; };
    Channel<IoItem> to_io(2);
    Channel<std::unique_ptr<OutBufs>> free_bufs(4);
    for (int i = 0; i < 3; ++i) {
        auto ob = std::make_unique<OutBufs>();
        ob->pb.resize(threads); ob->nb.resize(threads); ob->ob.resize(threads);
        free_bufs.push(std::move(ob));
    }
    auto recycle_batch = [&](std::unique_ptr<Batch>& b) {  // hand the batch (and its pinned buffers) back to the producer
        if (!b) return;
        if (b->pin) free_pins.push(std::move(b->pin));
        b->recs.clear(); b->marks.clear(); b->chunks.clear(); b->shared_chunks.reset(); b->file.reset(); b->fq_nl.clear();
        free_batches.push(std::move(b));
    };
    std::thread io_thread([&]() {
        IoItem it;
        while (to_io.pop(it)) {
            const uint64_t tw0 = now_us();
            if (it.batch) {
                Batch* tbp = it.batch.get();
                pool.run(2, [&](size_t w) {  // the files are independent streams: one writer each
                    if (w == 0) {
                        if (tbp->p_bytes && fwrite(tbp->pin->ptext.p, 1, tbp->p_bytes, pathF) != tbp->p_bytes) fail(BGR_E_IO, "write to the paths file failed");
                    } else {
                        if (tbp->n_bytes && fwrite(tbp->pin->ntext.p, 1, tbp->n_bytes, notF) != tbp->n_bytes) fail(BGR_E_IO, "write to the notAligned file failed");
                    }
                }, 4);
                us_write += now_us() - tw0;
                recycle_batch(it.batch);
                continue;
            }
            std::unique_ptr<OutBufs>& o = it.bufs;
            pool.run(2, [&](size_t w) {  // the files are independent streams: one writer each
                for (unsigned t = 0; t < threads; ++t) {
                    if (w == 0) {
                        if (!o->pb[t].empty() && fwrite(o->pb[t].data(), 1, o->pb[t].size(), pathF) != o->pb[t].size()) fail(BGR_E_IO, "write to the paths file failed");
                    } else {
                        if (!o->nb[t].empty() && fwrite(o->nb[t].data(), 1, o->nb[t].size(), notF) != o->nb[t].size()) fail(BGR_E_IO, "write to the notAligned file failed");
                        if (ovlF && !o->ob[t].empty() && fwrite(o->ob[t].data(), 1, o->ob[t].size(), ovlF) != o->ob[t].size()) fail(BGR_E_IO, "write to the no-overlap file failed");
                    }
                }
            }, 4);
            us_write += now_us() - tw0;
            free_bufs.push(std::move(o));
        }
    });
    std::thread writer([&]() {
        std::map<uint64_t, std::unique_ptr<Batch>> pending;
        uint64_t want = 0;
        bool stop_writing_after_this = false, wrote_last = false;
        std::unique_ptr<Batch> b;
        std::vector<char> okv(threads, 1);
        std::vector<std::string> bugv(threads);
        // aligner.h:68 counters as the reference's single worker has them when it prints (atomic<uint>: 32-bit wrap-around)
        uint32_t c_reads = 0, c_aligned = 0, c_failed = 0, c_noov = 0, c_overlaps = 0;
        const uint32_t K1 = gi.k - 1;
        auto count_reads = [&](const Batch& bt, uint64_t lo, uint64_t hi) {  // exhaustive mode: alignerExhaustive.cpp:35-58
            const uint8_t* st = static_cast<const uint8_t*>(bt.pin->status.p);
            for (uint64_t i = lo; i < hi; ++i) {
                ++c_reads;
                if ((st[i] & BGR_ST_MASK) == BGR_ST_ALIGNED) ++c_aligned; else ++c_failed;
                const uint32_t L = bt.recs[i].sl;
                c_overlaps += L >= K1 ? L - K1 + 1 : 1;  // overlaps += listOverlap.size(): every position (aligner.cpp:318-342)
            }
        };
        auto print_block = [&]() {  // alignerExhaustive.cpp:306-316
            const uint32_t got = c_aligned + c_failed;
            std::cout << "Read : " << c_reads << std::endl;
            std::cout << "No Overlap : " << c_noov << " Percent : " << (100 * float(c_noov)) / c_reads << std::endl;
            std::cout << "Got Overlap : " << got << " Percent : " << (100 * float(got)) / c_reads << std::endl;
            std::cout << "Overlap and Aligned : " << c_aligned << " Percent : " << (100 * float(c_aligned)) / got << std::endl;
            std::cout << "Overlap but no aligne: " << c_failed << " Percent : " << (100 * float(c_failed)) / got << std::endl;
            const auto secs = std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - t_start).count();
            std::cout << "Reads/seconds : " << c_reads / (uint64_t)(secs + 1) << std::endl;
            // (the reference divides by zero here -- and dies -- when no read has been counted yet)
            std::cout << "Overlap per reads : " << (got ? c_overlaps / got : 0u) << std::endl;
            std::cout << std::endl;
        };
        // text route under -b with progress blocks: the writer counts getReads() iterations itself.  A batch is a piece of its file; the
        // device reports what became of each of its records (pin->info: kept << 31 | mapped << 30 | length, record = iteration), a piece that
        // went through the host parser lists the iteration of each of its reads.  Same rule as ProgressMarker::chunk / end_file on the host route.
        ProgressMarker wmark;
        wmark.on = text_progress;
        uint64_t w_file_iters = 0;
        auto count_attempts = [&](const Batch& bt, uint64_t lo, uint64_t hi, uint64_t& rd) {  // iterations [lo, hi) of the piece; rd: cursor into its reads (fallback form)
            if (bt.dev_text) {
                const uint32_t* info = static_cast<const uint32_t*>(bt.pin->info.p);
                for (uint64_t i = lo; i < hi; ++i) {
                    const uint32_t v = info[i];
                    if (!(v >> 31)) continue;
                    ++c_reads;
                    if (v & 0x40000000u) ++c_aligned; else ++c_failed;
                    const uint32_t L = v & 0x3FFFFFFFu;
                    c_overlaps += L >= K1 ? L - K1 + 1 : 1;
                }
            } else {
                uint64_t r1 = rd;
                while (r1 < bt.att_of_read.size() && bt.att_of_read[r1] < hi) ++r1;
                if (r1 > rd) count_reads(bt, rd, r1);
                rd = r1;
            }
        };
        auto text_progress_batch = [&](const Batch& bt) {
            if (bt.first_of_file) { wmark.begin_file(); w_file_iters = 0; }
            uint64_t at = 0, rd = 0;
            while (wmark.next_fc * kRefBatch < w_file_iters + bt.n_attempts) {  // a block is due in front of iteration `rel` of this piece
                const uint64_t rel = wmark.next_fc * kRefBatch > w_file_iters ? wmark.next_fc * kRefBatch - w_file_iters : 0;
                count_attempts(bt, at, rel, rd);
                at = rel;
                print_block();
                wmark.next_fc += 10;
            }
            count_attempts(bt, at, bt.n_attempts, rd);
            w_file_iters += bt.n_attempts;
            if (bt.last_of_file)
                for (unsigned d = wmark.end_file(std::max<uint64_t>(1, w_file_iters)); d; --d) print_block();
        };
        while (to_out.pop(b)) {
            pending[b->index] = std::move(b);
            while (!pending.empty() && pending.begin()->first == want) {
                std::unique_ptr<Batch> cur = std::move(pending.begin()->second);
                pending.erase(pending.begin());
                ++want;
                struct Recycle {  // hand the batch (and its pinned buffers) back to the producer, unless the I/O thread got it
                    decltype(recycle_batch)& fn; std::unique_ptr<Batch>& b;
                    ~Recycle() { fn(b); }
                } recycle{recycle_batch, cur};
                if (!failed) {  // what the reference prints between reads, in input order
                    uint64_t at = 0;
                    const bool own_count = text_progress && cur->text_origin;  // (its marks are file names only; the iterations are counted below)
                    for (const Mark& mk : cur->marks) {
                        if (progress_blocks && !own_count) { count_reads(*cur, at, mk.pos); at = mk.pos; }
                        if (mk.kind == 1) std::cout << mk.text << std::endl; else print_block();
                    }
                    if (progress_blocks && !own_count) count_reads(*cur, at, cur->n);
                    if (own_count) text_progress_batch(*cur);
                }
                if ((failed && !stop_writing_after_this) || !writes || wrote_last || cur->n == 0) continue;
                if (cur->dev_text) {  // the record streams are ready as they are
                    IoItem item;
                    item.batch = std::move(cur);
                    to_io.push(std::move(item));
                    continue;
                }
                std::unique_ptr<OutBufs> o;
                if (!free_bufs.pop(o)) continue;
                std::vector<std::string>&pb = o->pb, &nb = o->nb, &ob = o->ob;
                const uint64_t per = (cur->n + threads - 1) / threads;
                Batch* cp = cur.get();
                const uint64_t tf0 = now_us();
                pool.run(threads, [&](size_t t) {
                    pb[t].clear(); nb[t].clear(); ob[t].clear();
                    uint64_t lo = t * per, hi = std::min<uint64_t>(cp->n, lo + per);
                    if (lo >= hi) return;
                    if (!extended) format_range(*cp, lo, hi, pb[t], nb[t]);
                    else okv[t] = format_range_ext(*cp, lo, hi, correction ? &unitigs : nullptr, ovlF != nullptr, pb[t], nb[t], ob[t], bugv[t]) ? 1 : 0;
                }, 3);
                // "bug compaction": like the reference, everything before the offending read is written, nothing after it
                unsigned t_stop = threads;
                for (unsigned t = 0; t < threads; ++t)
                    if (!okv[t]) { t_stop = t; break; }
                if (t_stop < threads) {
                    fail(BGR_E_COMPACTION, "bug compaction\n" + bugv[t_stop]);
                    for (unsigned t = t_stop + 1; t < threads; ++t) { pb[t].clear(); nb[t].clear(); ob[t].clear(); }
                    for (auto& ok1 : okv) ok1 = 1;
                    stop_writing_after_this = true;
                }
                us_format += now_us() - tf0;
                IoItem item;
                item.bufs = std::move(o);
                to_io.push(std::move(item));  // the formatted records never point into the batch: it can be recycled now
                if (stop_writing_after_this) wrote_last = true;
            }
        }
        to_io.close();
    });

    producer.join();
    gatherer.join();
    for (auto& t : workers) t.join();
    writer.join();
    io_thread.join();
    pin_allocator.join();
    {   // the page-locked sets wait for the next run of this process (bgr_host_cache_release)
        std::unique_ptr<Pinned> pn;
        std::lock_guard<std::mutex> l(g_pin_cache_m);
        while (free_pins.try_pop(pn)) if (g_pin_cache.size() < 96) g_pin_cache.push_back(std::move(pn));  // (the lanes of a split run on eight devices hold 64 sets)
    }
    free_pins.close();
    free_batches.close();
    free_bufs.close();
    fclose(pathF);
    fclose(notF);
    if (ovlF) fclose(ovlF);
    uint64_t tot[5] = {0, 0, 0, 0, 0};
    for (auto* a : aligners) {
        uint64_t c5[5];
        if (!failed && bgr_aligner_counters(a, c5) == BGR_OK) for (int j = 0; j < 5; ++j) tot[j] += c5[j];
        bgr_aligner_destroy(a);
    }
    if (counters_out) memcpy(counters_out, tot, sizeof(tot));
    if (mapping_seconds) *mapping_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    if (timing)
        fprintf(stderr, "bgreat: stage busy time (s): parse %.3f  alloc %.3f  gather %.3f  gpu(sum over %zu workers) %.3f  format %.3f  write %.3f\n",
                us_parse / 1e6, us_alloc / 1e6, us_gather / 1e6, aligners.size(), us_gpu / 1e6, us_format / 1e6, us_write / 1e6);
    if (timing)  // thread CPU seconds spent inside the pool's tasks, by stage
        fprintf(stderr, "bgreat: pool CPU time (s): parse %.3f  gather/pack %.3f  format %.3f  write %.3f  other %.3f\n", pool.cpu_us[1] / 1e6,
                pool.cpu_us[2] / 1e6, pool.cpu_us[3] / 1e6, pool.cpu_us[4] / 1e6, pool.cpu_us[0] / 1e6);
    if (failed_first) return bgr::set_error(first_rc, first_err);
    if (failed) return bgr::set_error(BGR_E_INTERNAL, "stopped: another lane of the run failed");
    return BGR_OK;
}

}  // namespace



// ---- split run: one pipeline per device (bgr_run_options.split_output) ---------------------------------------------------------
// The reference's N workers share ONE reader and ONE writer under a mutex each (alignerGreedy.cpp:372-377,407-411).  Here every
// device gets a lane of its own -- producer, gatherer, stream workers, ordered writer, output pair -- over a contiguous share of the
// input, so nothing is shared between devices but the graph and the page cache; `cat` of the pairs in device order is the -t 1 stream.
static int align_all_lanes(bgr_graph* graph, const bgr_params* prm, const bgr_run_options* opt, const std::vector<std::string>& files,
                           const char* paths_file, const char* notaligned_file, uint64_t counters_out[5], double* mapping_seconds) {
    const unsigned n = std::max<uint32_t>(1, opt->n_gpus);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::shared_ptr<MappedFile>> imgs;
    uint64_t total = 0;
    for (const std::string& f : files) {
        auto mf = std::make_shared<MappedFile>();
        std::string err;
        if (!mf->open(f, err)) return bgr::set_error(BGR_E_IO, "read file: " + err);
        if (opt->echo_files) std::cout << f << std::endl;  // aligner.cpp:559,576 (nothing else is printed while such a run maps)
        total += mf->size;
        imgs.push_back(std::move(mf));
    }
    // lane d: the bytes between cut d and cut d + 1 of the files taken together; a cut is the first record start at or behind d / n of
    // the bytes (fasta_cut_at: where an independent reader is in the sequential reader's state), so every lane parses its share exactly
    // as the one reader would
    struct Cut { size_t file; uint64_t off; };
    std::vector<Cut> cuts(n + 1);
    cuts[0] = {0, 0};
    cuts[n] = {files.size(), 0};
    for (unsigned d = 1; d < n; ++d) {
        const uint64_t target = (uint64_t)((unsigned __int128)total * d / n);
        uint64_t base = 0;
        Cut c = {files.size(), 0};
        for (size_t f = 0; f < imgs.size(); ++f) {
            const uint64_t sz = imgs[f]->size;
            if (target < base + sz) {
                const uint64_t at = target > base ? bgr::fasta_cut_at(imgs[f]->data, sz, target - base) : 0;
                c = at < sz ? Cut{f, at} : Cut{f + 1, 0};
                break;
            }
            base += sz;
        }
        if (c.file < cuts[d - 1].file || (c.file == cuts[d - 1].file && c.off < cuts[d - 1].off)) c = cuts[d - 1];
        cuts[d] = c;
    }
    // the graph on every device of the run before the lanes start: bgr_aligner_create would bring it there too, but the lanes run side by side
    // and the graph's table of resident copies is not made for concurrent writers (a run behind bgr_devices_init finds every copy in place)
    const bool one_device = bgr::opt("test.lanes_on_one_device") != 0;
    for (unsigned d = 0; d < (one_device ? 1u : n); ++d) {
        const int rc = bgr_graph_upload(graph, (int)(opt->first_device + d));
        if (rc != BGR_OK) return rc;
    }
    std::atomic<bool> cancel{false};
    std::vector<int> rcs(n, BGR_OK);
    std::vector<std::string> errs(n);
    std::vector<std::array<uint64_t, 5>> cnt(n);
    std::vector<double> lane_end(n, 0.0), lane_secs(n, 0.0);
    std::vector<std::thread> lanes;
    for (unsigned d = 0; d < n; ++d) {
        lanes.emplace_back([&, d]() {
            std::vector<InputSpan> in;
            for (size_t f = cuts[d].file; f < files.size() && (f < cuts[d + 1].file || (f == cuts[d + 1].file && cuts[d + 1].off > 0)); ++f) {
                InputSpan sp;
                sp.file = files[f];
                sp.mf = imgs[f];
                sp.begin = f == cuts[d].file ? cuts[d].off : 0;
                sp.end = f == cuts[d + 1].file ? cuts[d + 1].off : ~0ull;
                if (std::min(sp.end, imgs[f]->size) > sp.begin || imgs[f]->size == 0) in.push_back(std::move(sp));
            }
            bgr_run_options o = *opt;
            o.n_gpus = 1;
            // (BGREAT_TEST_LANES_ON_ONE_DEVICE=1, a test hook: every lane on the first device -- the split run's code path on a one-GPU box)
            o.first_device = one_device ? opt->first_device : opt->first_device + d;
            o.threads = std::max<uint32_t>(1, opt->threads / n);
            o.echo_files = 0;
            o.split_output = 0;
            const std::string pf = std::string(paths_file) + "." + std::to_string(d), nf = std::string(notaligned_file) + "." + std::to_string(d);
            cnt[d].fill(0);
            double secs = 0;
            rcs[d] = align_all_impl(graph, prm, &o, in, pf.c_str(), nf.c_str(), cnt[d].data(), &secs, &cancel);
            lane_end[d] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            lane_secs[d] = secs;
            if (bgr::opt("timing")) fprintf(stderr, "bgreat: lane %u (device %u): %llu reads in %.3f s, done %.3f s after the start of the run\n", d, o.first_device, (unsigned long long)cnt[d][0], secs, lane_end[d]);
            if (rcs[d] != BGR_OK) errs[d] = bgr_last_error();  // (the message is the lane thread's own)
        });
    }
    for (auto& t : lanes) t.join();
    if (mapping_seconds) {  // as for one pipeline: from the (first) pipeline's start -- output files open, aligners made -- to the (last) one's end
        double first = 1e300, last = 0;
        for (unsigned d = 0; d < n; ++d) { first = std::min(first, lane_end[d] - lane_secs[d]); last = std::max(last, lane_end[d]); }
        *mapping_seconds = std::max(0.0, last - first);
    }
    int rc = BGR_OK;
    std::string err;
    for (unsigned d = 0; d < n; ++d)   // the lane that failed, not the ones it stopped
        if (rcs[d] != BGR_OK && (rc == BGR_OK || (rc == BGR_E_INTERNAL && err.rfind("stopped:", 0) == 0))) { rc = rcs[d]; err = errs[d]; }
    if (rc != BGR_OK) return bgr::set_error(rc, err);
    if (counters_out) for (int j = 0; j < 5; ++j) { counters_out[j] = 0; for (unsigned d = 0; d < n; ++d) counters_out[j] += cnt[d][j]; }
    return BGR_OK;
}

extern "C" int bgr_align_all(bgr_graph* graph, const bgr_params* prm, const bgr_run_options* opt, const char* reads_csv,
                             const char* paths_file, const char* notaligned_file, uint64_t counters_out[5], double* mapping_seconds) {
    if (!graph || !prm || !opt || !reads_csv || !paths_file || !notaligned_file) return bgr::set_error(BGR_E_ARG, "bgr_align_all: null argument");
    if (opt->struct_size != sizeof(bgr_run_options))
        return bgr::set_error(BGR_E_ARG, "bgr_align_all: bgr_run_options.struct_size is not this library's sizeof(bgr_run_options): the caller was built against another header (zero the struct, set struct_size)");
    std::vector<std::string> files;  // aligner.cpp:552-586: comma-separated list
    {
        const std::string list(reads_csv);
        size_t last = 0;
        for (size_t i = 0; i <= list.size(); ++i) {
            if (i != list.size() && list[i] != ',') continue;
            files.push_back(list.substr(last, i - last));
            last = i + 1;
        }
    }
    const bool progress_blocks = opt->echo_files && prm->mode == BGR_MODE_EXHAUSTIVE;
    const bool correction = opt->correction && prm->mode != BGR_MODE_EXHAUSTIVE;
    if (opt->split_output && opt->n_gpus > 1 && !opt->fastq && !progress_blocks && !correction && !opt->no_overlap_file)
        return align_all_lanes(graph, prm, opt, files, paths_file, notaligned_file, counters_out, mapping_seconds);
    std::vector<InputSpan> inputs(files.size());
    for (size_t i = 0; i < files.size(); ++i) inputs[i].file = files[i];
    return align_all_impl(graph, prm, opt, inputs, paths_file, notaligned_file, counters_out, mapping_seconds, nullptr);
}
