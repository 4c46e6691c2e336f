// capi.hip -- implementation of the C-ABI declared in include/bgreat_gpu.h.  Thin: argument checks, HIP memory
// and stream management, launch geometry; the algorithm lives in graph_build.cpp (index) and
// the *_kernels.hip files (mapping; launch interface align_kernels.h).  There is no CPU mapping path in this library.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bgreat_gpu.h"
#include "align_kernels.h"
#include "fanout.h"
#include "fastx.h"
#include "anchor_index.h"
#include "graph_build.h"
#include "launch_plan.h"
#include "read_pack.h"
#include "text_kernels.h"
#include "options.h"

namespace {

thread_local std::string tl_err;
int fail(int code, const std::string& msg) { tl_err = msg; return code; }
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(BGR_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

const uint32_t kRetryCtr = 6, kRetrySubsetCtr = 7;  // words of `cursor`: reads the last exhaustive pass hands back / reads of the list it maps when it runs again
const int kTimerRing = 64;   // launches between two drains of the timers
const int kTimerSlots = 8;   // kernels of one launch timed separately (pre-pass, passes)

}  // namespace

namespace bgr {
int set_error(int code, const std::string& msg) { return fail(code, msg); }  // for pipeline.cpp
}

struct bgr_graph {
    bgr::HostGraph host;  // empty when adopted from a device blob
    std::vector<char> ascii;            // the unitig characters as given (only graphs built from sequences have them):
    std::vector<uint64_t> ascii_offs;   // correction mode spells reads from these, like the reference's vector<string>
    BgrBlobHeader header;
    struct Dev { void* ptr; bool owned; };
    std::map<int, Dev> dev;
    uint32_t fanout_method = 0;  // how bgr_devices_init moved the blob between devices last time
};

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        // diagnostic (bgr_set_option("poison_device_buffers", 1); tools/fuzz_*.py, the GPU suite): fresh device memory usually reads as zeroes, recycled memory of
        // a long-lived process does not -- fill every new buffer with a pattern so that a kernel that reads what nothing has written shows in ANY run
        const bool poison = bgr::opt("poison_device_buffers") != 0;
        if (e == hipSuccess && poison) { e = hipMemset(p, 0xA5, want); if (e == hipSuccess) e = hipDeviceSynchronize(); }  // (the fill runs on the null stream: the aligner's streams do not wait for it)
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct bgr_text_stage {  // one piece of text on its way to / resident in a device: buffer, copy stream, "it has arrived" event
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev = nullptr, ev0 = nullptr;  // ev0: BGREAT_TIMING only, start of the copy
    DevBuf buf;
    uint64_t bytes = 0;
    bool timing = false, pending = false;
    double copy_ms = 0, copy_bytes = 0;
    void settle() {  // BGREAT_TIMING: duration of the last copy (it has completed)
        if (!timing || !pending) return;
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev0, ev) == hipSuccess) { copy_ms += ms; copy_bytes += (double)bytes; }
        pending = false;
    }
};

struct bgr_aligner {
    bgr_graph* graph = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    BgrDeviceGraph dg;
    // the text route (bgr_align_fasta_text): the piece, its records, the formatted streams
    DevBuf tx_in, tx_sums, tx_state, tx_rec, tx_idx, tx_accrec, tx_accsrc, tx_offs, tx_psz, tx_nsz, tx_poff, tx_noff, tx_pout, tx_nout, tx_info;
    uint64_t tx_n_acc = 0, tx_pbytes = 0, tx_nbytes = 0;
    bool blocking_sync = bgr::opt("blocking_sync") != 0;
    hipEvent_t ev_wait = nullptr;
    const uint8_t* tx_text = nullptr;  // where the last call's piece lies in HBM (tx_in, or the caller's stage)
    uint32_t tx_flip = 0;              // which of the two info blocks the current piece uses
    uint32_t tx_epoch = 0, tx_ticket[2] = {0, 0};   // the one-launch kernels' chains: epoch of the last launch; tickets earlier launches took (parse, format)
    bool tx_written = false;           // the streams lie in tx_pout / tx_nout (the format launch wrote them: every stretch ended below the capacities)
    uint32_t tx_want = 0;              // its want_output (2 = correction mode: mapped reads as spelled by their paths)
    double tx_phase_s[5] = {0, 0, 0, 0, 0};  // BGREAT_TIMING: host wall seconds to the call's four waits (mark, records, mapping + sizes, streams) + calls

    DevBuf in_reads, in_offs, pk_fw3, pk_nm, pk_hasn, results, arena, ovf, ovf2, lst, deepbuf, retry, retry2, small, csr_sums, csr_poffs, csr_status, csr_paths;  // small: cursor[16] u32 @0, counters[5] u64 @64
    struct DeepRun {  // the last pass of the exhaustive launch in flight, as enqueued: settle_launch runs it again for reads whose table filled up
        bool open = false;
        bgr::BatchIO io;
        bgr::KernelParams kp;
        BgrDeviceGraph dg;
        bgr::LaunchCfg cfg;
        uint32_t per_wave_lds = 0, path_cap = 0, frames = 0, memo_cap = 0, runs = 0;
    } deep;
    bgr::PlanDevice plan_dev;     // CUs, LDS, resident waves per kernel: asked once
    bool plan_dev_known = false;
    uint64_t last_n = 0;
    DevBuf wave_times;            // diagnostic builds only (-DBGR_PHASE_TIMING)
    uint64_t wave_times_n = 0;
    uint64_t ticket_serial = 0;       // bgr_align_batch_begin: tickets handed out; the batch of the last one is in flight until its wait
    bool ticket_open = false;
    std::vector<uint64_t> ticket_offs;  // that batch's offsets made relative (kept alive for the asynchronous copy)
    uint32_t last_launch[4] = {0, 0, 0, 0};
    uint32_t cfg_waves = 0, cfg_blocks_per_cu = 0, cfg_lds_mphf = 0;
    bool exh_filter = bgr::opt("exh_filter") != 0;  // exhaustive mode through the minimizer filter too (option exh_filter = 0: without)
    // bgr_aligner_set_knob (test / diagnostic hooks, read here instead of from the environment on every launch)
    uint32_t knob_frame_cap = 0, knob_search = 0, knob_debug_stop = 0, knob_greedy_fast = 0, knob_exh_fast = 0, knob_anc_fast = 0, knob_memo_cap = 0, knob_prepass = 0, knob_no_events = 0;
    uint64_t knob_split_limit = 0;
    uint32_t knob_overlap = 0;      // BGR_KNOB_BATCH_OVERLAP
    bgr_aligner* twin = nullptr;    // second stream + buffers for the overlapped form of bgr_align_batch (created on first use)
    bool is_twin = false;
    int num_cus = 0;
    size_t lds_per_cu = 0;
    hipEvent_t ev[kTimerRing][kTimerSlots + 1];  // ev[i][0] = start of launch i, ev[i][j] = behind its j-th kernel
    int ev_marks[kTimerRing];                    // kernels timed in launch i
    int ev_used = 0;
    uint64_t t_launches = 0;
    double t_ms = 0, t_slot_ms[kTimerSlots] = {0, 0, 0, 0, 0, 0, 0, 0};
    const char* t_slot_name[kTimerSlots] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

namespace {

// Wait for the aligner's stream.  By default hipStreamSynchronize (the runtime spins: lowest latency, one busy CPU per waiting thread);
// with BGREAT_BLOCKING_SYNC=1 an event made with hipEventBlockingSync is recorded and waited for instead: the thread sleeps until the
// interrupt, so more stream workers per device than CPUs to spare can overlap their calls (bgr_align_all's text route).
hipError_t wait_stream(bgr_aligner* a) {
    if (!a->blocking_sync) return hipStreamSynchronize(a->stream);
    hipError_t e = hipEventRecord(a->ev_wait, a->stream);
    return e == hipSuccess ? hipEventSynchronize(a->ev_wait) : e;
}

int drain_timers(bgr_aligner* a) {
    if (a->ev_used == 0) return BGR_OK;
    HIP_TRY(hipStreamSynchronize(a->stream));
    for (int i = 0; i < a->ev_used; ++i) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, a->ev[i][0], a->ev[i][a->ev_marks[i]]));
        a->t_ms += ms;
        for (int j = 0; j < a->ev_marks[i]; ++j) {
            HIP_TRY(hipEventElapsedTime(&ms, a->ev[i][j], a->ev[i][j + 1]));
            a->t_slot_ms[j] += ms;
        }
        ++a->t_launches;
    }
    a->ev_used = 0;
    return BGR_OK;
}

}  // namespace

extern "C" {

const char* bgr_last_error(void) { return tl_err.c_str(); }

int bgr_set_option(const char* name, int64_t value) {
    bgr::Option* o = name ? bgr::find_option(name) : nullptr;
    if (!o) return fail(BGR_E_ARG, std::string("bgr_set_option: unknown option '") + (name ? name : "") + "'");
    if (value < o->lo || value > o->hi) return fail(BGR_E_ARG, std::string("bgr_set_option: value out of range for '") + name + "'");
    o->value.store(value, std::memory_order_relaxed);
    return BGR_OK;
}
int bgr_get_option(const char* name, int64_t* value) {
    bgr::Option* o = name ? bgr::find_option(name) : nullptr;
    if (!o || !value) return fail(BGR_E_ARG, std::string("bgr_get_option: unknown option '") + (name ? name : "") + "'");
    *value = o->value.load(std::memory_order_relaxed);
    return BGR_OK;
}
const char* bgr_option_name(uint32_t index, const char** what) {
    size_t n;
    bgr::Option* t = bgr::option_table(&n);
    if (index >= n) return nullptr;
    if (what) *what = t[index].what;
    return t[index].name;
}
void bgr_set_build_threads(uint32_t threads) { bgr::set_build_threads(threads); }

int bgr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- index ------------------------------------------------------------------------------------------
int bgr_graph_build(uint32_t k, uint64_t n_unitigs, const char* seqs, const uint64_t* offsets, double gamma, bgr_graph** out) {
    return bgr_graph_build_ex(k, n_unitigs, seqs, offsets, gamma, 0, out);
}

int bgr_graph_build_ex(uint32_t k, uint64_t n_unitigs, const char* seqs, const uint64_t* offsets, double gamma, uint32_t flags, bgr_graph** out) {
    if (!out || (n_unitigs && (!seqs || !offsets))) return fail(BGR_E_ARG, "bgr_graph_build: null argument");
    if (flags & ~(uint32_t)(BGR_BUILD_ANCHORS | BGR_BUILD_NO_EVICTIONS)) return fail(BGR_E_ARG, "bgr_graph_build: unknown flag");
    bgr_graph* g = new bgr_graph();
    std::string err;
    uint64_t zero[1] = {0};
    if (!bgr::build_graph(k, n_unitigs, seqs, n_unitigs ? offsets : zero, gamma > 0 ? gamma : 0.0, flags, g->host, err)) {
        delete g;
        return fail(BGR_E_ARG, err);
    }
    g->header = *g->host.header();
    const uint64_t nu = g->header.n_unitigs;  // loading stopped at the first short sequence
    g->ascii.assign(seqs + (nu ? offsets[0] : 0), seqs + (nu ? offsets[nu] : 0));
    g->ascii_offs.resize(nu + 1);
    for (uint64_t i = 0; i <= nu; ++i) g->ascii_offs[i] = nu ? offsets[i] - offsets[0] : 0;
    *out = g;
    return BGR_OK;
}

int bgr_graph_unitigs(const bgr_graph* g, const char** seqs, const uint64_t** offsets, uint64_t* n) {
    if (!g || !seqs || !offsets || !n) return fail(BGR_E_ARG, "bgr_graph_unitigs: null argument");
    if (g->ascii_offs.empty()) return fail(BGR_E_ARG, "bgr_graph_unitigs: this graph was created from a blob and carries no unitig characters");
    *seqs = g->ascii.data();
    *offsets = g->ascii_offs.data();
    *n = g->ascii_offs.size() - 1;
    return BGR_OK;
}

int bgr_graph_build_from_fasta(const char* path, uint32_t k, double gamma, bgr_graph** out) {
    return bgr_graph_build_from_fasta_ex(path, k, gamma, 0, out);
}

int bgr_graph_build_from_fasta_ex(const char* path, uint32_t k, double gamma, uint32_t flags, bgr_graph** out) {
    if (!path || !out) return fail(BGR_E_ARG, "bgr_graph_build_from_fasta: null argument");
    std::vector<char> seqs;
    std::vector<uint64_t> offs;
    std::string err;
    if (!bgr::read_unitig_fasta(path, k, seqs, offs, err)) return fail(BGR_E_IO, err);
    return bgr_graph_build_ex(k, offs.size() - 1, seqs.data(), offs.data(), gamma, flags, out);
}

int bgr_graph_anchor_lookup(const bgr_graph* g, uint64_t kmer, uint64_t* index_out, uint64_t* position_out) {
    if (!g || !index_out) return fail(BGR_E_ARG, "bgr_graph_anchor_lookup: null argument");
    if (g->host.blob.empty()) return fail(BGR_E_ARG, "bgr_graph_anchor_lookup: graph has no host blob");
    if (!g->header.anc_n) return fail(BGR_E_ARG, "bgr_graph_anchor_lookup: the graph was built without BGR_BUILD_ANCHORS");
    const uint64_t idx = bgr::anchor_lookup(g->host.header(), g->host.base(), kmer);
    *index_out = idx;
    if (position_out) *position_out = idx == ~0ULL ? 0 : reinterpret_cast<const uint64_t*>(g->host.base() + g->header.off_anc_pos)[idx];
    return BGR_OK;
}

int bgr_graph_key_lookup(const bgr_graph* g, uint64_t key, uint32_t* slot_out) {
    if (!g || !slot_out) return fail(BGR_E_ARG, "bgr_graph_key_lookup: null argument");
    if (g->host.blob.empty()) return fail(BGR_E_ARG, "bgr_graph_key_lookup: graph has no host blob");
    *slot_out = bgr::host_lookup(g->host.header(), g->host.base(), key);
    return BGR_OK;
}

const void* bgr_graph_blob(const bgr_graph* g, uint64_t* bytes) {
    if (!g || g->host.blob.empty()) { if (bytes) *bytes = 0; return nullptr; }
    if (bytes) *bytes = g->header.blob_bytes;
    return g->host.blob.data();
}

int bgr_graph_from_blob(const void* blob, uint64_t bytes, bgr_graph** out) {
    if (!blob || !out) return fail(BGR_E_ARG, "bgr_graph_from_blob: null argument");
    std::string err;
    if (!bgr::validate_blob(blob, bytes, err)) return fail(BGR_E_ARG, err);
    bgr_graph* g = new bgr_graph();
    if (!g->host.blob.reset((bytes + 7) / 8)) { delete g; return fail(BGR_E_ARG, "bgr_graph_from_blob: out of memory"); }
    memcpy(g->host.blob.data(), blob, bytes);
    g->header = *g->host.header();
    *out = g;
    return BGR_OK;
}

int bgr_graph_info(const bgr_graph* g, bgr_graph_info_t* o) {
    if (!g || !o) return fail(BGR_E_ARG, "bgr_graph_info: null argument");
    const BgrBlobHeader& h = g->header;
    memset(o, 0, sizeof(*o));
    o->k = h.k; o->n_levels = 2; o->n_unitigs = h.n_unitigs; o->n_keys = h.n_placed + h.n_fallback;
    o->n_left_keys = h.n_left_keys; o->n_right_keys = h.n_right_keys; o->n_fallback = h.n_fallback;
    o->total_bases = h.total_bases; o->blob_bytes = h.blob_bytes; o->mphf_bytes = h.n_buckets * 4;
    o->max_unitig_len = h.max_unitig_len; o->has_exceptions = h.has_exc; o->has_anchors = h.anc_n ? 1 : 0; o->gamma = h.gamma;
    return BGR_OK;
}

void bgr_graph_destroy(bgr_graph* g) {
    if (!g) return;
    for (auto& kv : g->dev) {
        if (kv.second.owned && kv.second.ptr) {
            if (hipSetDevice(kv.first) == hipSuccess) (void)hipFree(kv.second.ptr);
        }
    }
    delete g;
}

int bgr_graph_upload(bgr_graph* g, int device) {
    if (!g) return fail(BGR_E_ARG, "bgr_graph_upload: null graph");
    if (g->dev.count(device)) return BGR_OK;
    if (g->host.blob.empty()) return fail(BGR_E_ARG, "bgr_graph_upload: graph has no host blob (adopted graphs live on one device)");
    HIP_TRY(hipSetDevice(device));
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, g->header.blob_bytes));
    hipError_t e = hipMemcpy(p, g->host.blob.data(), g->header.blob_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);  // (from pageable memory on the null stream; the aligners' streams are non-blocking: the blob is there before any of them starts)
    if (e != hipSuccess) { (void)hipFree(p); return fail(BGR_E_HIP, std::string("blob H2D: ") + hipGetErrorString(e)); }
    g->dev[device] = {p, true};
    return BGR_OK;
}

// ---- one-off distribution of the graph over the GPUs of one process -------------------------------------------------------
// The blob goes from the host to the first device once and from there device to device over xGMI: as ONE RCCL broadcast
// (a communicator per device from ncclCommInitAll; librccl is looked up at run time, so the library has no link-time
// dependency on it), or, when RCCL is not available or refuses, as peer copies in a doubling schedule (1 -> 2 -> 4 -> 8
// holders: every round uses disjoint point-to-point links).
namespace {

struct Rccl {
    typedef int (*CommInitAll_t)(void** comms, int n, const int* devs);
    typedef int (*Group_t)(void);
    typedef int (*Broadcast_t)(const void* send, void* recv, size_t count, int dtype, int root, void* comm, hipStream_t stream);
    typedef int (*CommDestroy_t)(void* comm);
    void* lib = nullptr;
    CommInitAll_t CommInitAll = nullptr;
    Group_t GroupStart = nullptr, GroupEnd = nullptr;
    Broadcast_t Broadcast = nullptr;
    CommDestroy_t CommDestroy = nullptr;
    bool load() {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = reinterpret_cast<CommInitAll_t>(dlsym(lib, "ncclCommInitAll"));
        GroupStart = reinterpret_cast<Group_t>(dlsym(lib, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<Group_t>(dlsym(lib, "ncclGroupEnd"));
        Broadcast = reinterpret_cast<Broadcast_t>(dlsym(lib, "ncclBroadcast"));
        CommDestroy = reinterpret_cast<CommDestroy_t>(dlsym(lib, "ncclCommDestroy"));
        return CommInitAll && GroupStart && GroupEnd && Broadcast && CommDestroy;
    }
};

// ptr[i] on device devs[i]; ptr[0] holds the blob.  false = RCCL not usable here (nothing has been copied, or partly:
// the caller then overwrites with peer copies).
bool fanout_rccl(const std::vector<int>& devs, const std::vector<void*>& ptr, uint64_t bytes, std::string& why) {
    Rccl r;
    if (!r.load()) { why = "librccl not loadable"; return false; }
    const int n = (int)devs.size();
    std::vector<void*> comms(n, nullptr);
    if (r.CommInitAll(comms.data(), n, devs.data()) != 0) { why = "ncclCommInitAll failed"; return false; }
    std::vector<hipStream_t> streams(n, nullptr);
    bool ok = true;
    for (int i = 0; i < n && ok; ++i) ok = hipSetDevice(devs[i]) == hipSuccess && hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking) == hipSuccess;
    if (ok) {
        ok = r.GroupStart() == 0;
        for (int i = 0; i < n && ok; ++i) {
            ok = hipSetDevice(devs[i]) == hipSuccess &&
                 r.Broadcast(ptr[i], ptr[i], (size_t)bytes, /*ncclChar*/ 0, /*root rank*/ 0, comms[i], streams[i]) == 0;
        }
        ok = r.GroupEnd() == 0 && ok;
    }
    for (int i = 0; i < n; ++i) {
        if (streams[i]) { if (hipSetDevice(devs[i]) == hipSuccess) { ok = hipStreamSynchronize(streams[i]) == hipSuccess && ok; (void)hipStreamDestroy(streams[i]); } }
    }
    for (void* c : comms) if (c) (void)r.CommDestroy(c);
    if (!ok) why = "RCCL broadcast failed";
    return ok;
}

// one round of the doubling schedule: concurrent peer copies over disjoint links, complete on return
bool peer_round(const std::vector<bgr::FanoutCopy>& round, uint64_t bytes, std::string& why) {
    std::vector<hipStream_t> streams;
    std::vector<int> sdev;
    bool ok = true;
    for (const bgr::FanoutCopy& c : round) {
        if (hipSetDevice(c.src_dev) != hipSuccess) { ok = false; why = "hipSetDevice failed"; break; }
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, c.src_dev, c.dst_dev) == hipSuccess && can) {
            hipError_t pe = hipDeviceEnablePeerAccess(c.dst_dev, 0);  // direct xGMI path; without it the runtime stages through the host
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
        }
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { ok = false; why = "hipStreamCreate failed"; break; }
        streams.push_back(st);
        sdev.push_back(c.src_dev);
        hipError_t e = hipMemcpyPeerAsync(c.dst, c.dst_dev, c.src, c.src_dev, bytes, st);
        if (e != hipSuccess) { ok = false; why = std::string("hipMemcpyPeerAsync: ") + hipGetErrorString(e); break; }
    }
    for (size_t j = 0; j < streams.size(); ++j) {  // (also after a failure: nothing may still be writing when the buffers are released)
        if (hipSetDevice(sdev[j]) == hipSuccess) {
            hipError_t e = hipStreamSynchronize(streams[j]);
            if (e != hipSuccess && ok) { ok = false; why = std::string("peer copy: ") + hipGetErrorString(e); }
            (void)hipStreamDestroy(streams[j]);
        }
    }
    return ok;
}

}  // namespace

int bgr_devices_init(bgr_graph* g, int first_device, uint32_t n_devices, uint32_t how) {
    if (!g) return fail(BGR_E_ARG, "bgr_devices_init: null graph");
    if (n_devices == 0) n_devices = 1;
    if (how > BGR_FANOUT_PEER) return fail(BGR_E_ARG, "bgr_devices_init: unknown distribution method");
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd == 0) return fail(BGR_E_HIP, "no HIP device available");
    if (first_device < 0 || (uint64_t)first_device + n_devices > (uint64_t)nd) return fail(BGR_E_ARG, "bgr_devices_init: device range outside the visible devices");
    int rc = bgr_graph_upload(g, first_device);  // host -> first device (idempotent)
    if (rc != BGR_OK) return rc;
    const uint64_t bytes = g->header.blob_bytes;
    bgr::FanoutOps ops;
    ops.alloc = [bytes](int dev, void** out) { return hipSetDevice(dev) == hipSuccess && hipMalloc(out, bytes) == hipSuccess; };
    ops.release = [](int dev, void* p) { if (p && hipSetDevice(dev) == hipSuccess) (void)hipFree(p); };
    ops.broadcast = [bytes](const std::vector<int>& devs, const std::vector<void*>& ptr, std::string& why) { return fanout_rccl(devs, ptr, bytes, why); };
    ops.peer_round = [bytes](const std::vector<bgr::FanoutCopy>& round, std::string& why) { return peer_round(round, bytes, why); };
    // (bookkeeping in fanout.h: new buffers are registered only once they hold the blob; after a failure they are released and
    // g->dev is unchanged, so bgr_graph_upload / bgr_aligner_create can still bring the blob to a device on their own)
    std::map<int, void*> resident;
    for (auto& kv : g->dev) resident[kv.first] = kv.second.ptr;
    g->fanout_method = 0;
    const bgr::FanoutResult r = bgr::fanout_blob(first_device, n_devices, how, resident, ops);
    if (!r.error.empty()) return fail(r.hip_error ? BGR_E_HIP : BGR_E_ARG, "bgr_devices_init: " + r.error);
    for (auto& kv : resident) if (!g->dev.count(kv.first)) g->dev[kv.first] = {kv.second, true};
    g->fanout_method = (uint32_t)r.method;
    return BGR_OK;
}

uint32_t bgr_devices_method(const bgr_graph* g) { return g ? g->fanout_method : 0; }

const void* bgr_graph_device_blob(const bgr_graph* g, int device) {
    if (!g) return nullptr;
    auto it = g->dev.find(device);
    return it == g->dev.end() ? nullptr : it->second.ptr;
}

int bgr_graph_adopt_device_blob(int device, const void* dev_blob, uint64_t bytes, bgr_graph** out) {
    if (!dev_blob || !out || bytes < sizeof(BgrBlobHeader)) return fail(BGR_E_ARG, "bgr_graph_adopt_device_blob: bad argument");
    HIP_TRY(hipSetDevice(device));
    bgr_graph* g = new bgr_graph();
    hipError_t e = hipMemcpy(&g->header, dev_blob, sizeof(BgrBlobHeader), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { delete g; return fail(BGR_E_HIP, std::string("header D2H: ") + hipGetErrorString(e)); }
    std::string verr;  // the same checks as for a host blob: the kernels index HBM with these numbers
    if (!bgr::validate_blob_header(&g->header, bytes, verr)) {
        delete g;
        return fail(BGR_E_ARG, "bgr_graph_adopt_device_blob: " + verr);
    }
    g->dev[device] = {const_cast<void*>(dev_blob), false};
    *out = g;
    return BGR_OK;
}

// ---- mapping ----------------------------------------------------------------------------------------
int bgr_aligner_create(bgr_graph* g, int device, bgr_aligner** out) {
    if (!g || !out) return fail(BGR_E_ARG, "bgr_aligner_create: null argument");
    int nd = 0;
    hipError_t e0 = hipGetDeviceCount(&nd);
    if (e0 != hipSuccess || nd == 0) return fail(BGR_E_HIP, "no HIP device available (this library has no CPU mapping path)");
    if (device < 0 || device >= nd) return fail(BGR_E_ARG, "bgr_aligner_create: device index out of range");
    int rc = bgr_graph_upload(g, device);
    if (rc != BGR_OK) return rc;
    HIP_TRY(hipSetDevice(device));
    bgr_aligner* a = new bgr_aligner();
    a->graph = g;
    a->device = device;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { delete a; return fail(BGR_E_HIP, hipGetErrorString(e)); }
    a->num_cus = prop.multiProcessorCount;
    a->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : 65536;
    e = hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete a; return fail(BGR_E_HIP, hipGetErrorString(e)); }
    for (int i = 0; i < kTimerRing; ++i) {
        for (int j = 0; j <= kTimerSlots; ++j) {
            if (hipEventCreate(&a->ev[i][j]) != hipSuccess) {
                delete a;
                return fail(BGR_E_HIP, "hipEventCreate failed");
            }
        }
    }
    if (a->blocking_sync && hipEventCreateWithFlags(&a->ev_wait, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) a->blocking_sync = false;
    bgr::resolve_device_graph(&g->header, g->dev[device].ptr, a->dg);
    e = a->small.ensure(256);
    if (e == hipSuccess) e = hipMemset(a->small.p, 0, a->small.cap);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);  // (the fill runs on the null stream, which the aligner's non-blocking stream does not wait for)
    if (e != hipSuccess) { delete a; return fail(BGR_E_HIP, hipGetErrorString(e)); }
    *out = a;
    return BGR_OK;
}

void bgr_aligner_destroy(bgr_aligner* a) {
    if (!a) return;
    if (a->tx_phase_s[4] > 0 && bgr::opt("timing"))
        fprintf(stderr, "bgreat: text calls %.0f on device %d: to record count %.3f s, records %.3f s, mapping + sizes %.3f s, streams out %.3f s\n", a->tx_phase_s[4], a->device,
                a->tx_phase_s[0], a->tx_phase_s[1], a->tx_phase_s[2], a->tx_phase_s[3]);
    if (a->twin) { bgr_aligner_destroy(a->twin); a->twin = nullptr; }
    if (hipSetDevice(a->device) == hipSuccess) {
        if (a->stream) (void)hipStreamSynchronize(a->stream);
        a->in_reads.release(); a->in_offs.release(); a->pk_fw3.release(); a->pk_nm.release(); a->pk_hasn.release(); a->results.release(); a->arena.release(); a->ovf.release(); a->ovf2.release(); a->lst.release(); a->deepbuf.release(); a->retry.release(); a->retry2.release(); a->small.release();
        a->csr_sums.release(); a->csr_poffs.release(); a->csr_status.release(); a->csr_paths.release(); a->wave_times.release();
        for (DevBuf* b : {&a->tx_in, &a->tx_sums, &a->tx_state, &a->tx_rec, &a->tx_idx, &a->tx_accrec, &a->tx_accsrc, &a->tx_offs,
                          &a->tx_psz, &a->tx_nsz, &a->tx_poff, &a->tx_noff, &a->tx_pout, &a->tx_nout, &a->tx_info}) b->release();
        for (int i = 0; i < kTimerRing; ++i) for (int j = 0; j <= kTimerSlots; ++j) (void)hipEventDestroy(a->ev[i][j]);
        if (a->ev_wait) (void)hipEventDestroy(a->ev_wait);
        if (a->stream) (void)hipStreamDestroy(a->stream);
    }
    delete a;
}

int bgr_aligner_configure(bgr_aligner* a, uint32_t waves_per_block, uint32_t blocks_per_cu, uint32_t lds_mphf) {
    if (!a || waves_per_block > 16 || lds_mphf > 2) return fail(BGR_E_ARG, "bgr_aligner_configure: bad argument");
    a->cfg_waves = waves_per_block;
    a->cfg_blocks_per_cu = blocks_per_cu;
    a->cfg_lds_mphf = lds_mphf;
    return BGR_OK;
}

int bgr_aligner_set_knob(bgr_aligner* a, uint32_t knob, uint64_t value) {
    if (!a) return fail(BGR_E_ARG, "bgr_aligner_set_knob: null aligner");
    switch (knob) {
        case BGR_KNOB_EXH_FRAME_CAP: a->knob_frame_cap = (uint32_t)std::min<uint64_t>(value, 1u << 20); return BGR_OK;
        case BGR_KNOB_EXH_SEARCH: if (value > 2) break; a->knob_search = (uint32_t)value; return BGR_OK;
        case BGR_KNOB_BATCH_SPLIT_LIMIT: a->knob_split_limit = value; return BGR_OK;
        case BGR_KNOB_DEBUG_STOP: a->knob_debug_stop = (uint32_t)value; return BGR_OK;
        case BGR_KNOB_BATCH_OVERLAP: a->knob_overlap = (uint32_t)value; return BGR_OK;
        case BGR_KNOB_GREEDY_FAST: if (value > 1) break; a->knob_greedy_fast = (uint32_t)value; return BGR_OK;
        case BGR_KNOB_EXH_FAST: if (value > 1) break; a->knob_exh_fast = (uint32_t)value; return BGR_OK;
        case BGR_KNOB_ANCHORS_FAST: if (value > 1) break; a->knob_anc_fast = (uint32_t)value; return BGR_OK;
        case BGR_KNOB_GREEDY_PREPASS: if (value > 1) break; a->knob_prepass = (uint32_t)value; return BGR_OK;
        case BGR_KNOB_KERNEL_EVENTS: if (value > 1) break; a->knob_no_events = value ? 0u : 1u; return BGR_OK;
        case BGR_KNOB_EXH_MEMO_CAP: a->knob_memo_cap = (uint32_t)std::min<uint64_t>(value, 1u << 24); return BGR_OK;
        default: break;
    }
    return fail(BGR_E_ARG, "bgr_aligner_set_knob: unknown knob or value out of range");
}

// What the planner needs of the graph / device / caller's tuning (launch_plan.h)
static bgr::PlanGraph plan_graph_of(const BgrBlobHeader& h) {
    bgr::PlanGraph g;
    g.k = h.k; g.slot_fill_x100 = h.slot_fill_x100; g.table_bytes = (uint32_t)std::min<uint64_t>((uint64_t)h.n_buckets * 4, 0xFFFFFFFFull);
    g.total_bases = h.total_bases; g.n_unitigs = h.n_unitigs; g.n_buckets = h.n_buckets; g.max_unitig_len = h.max_unitig_len;
    g.anc_n = h.anc_n; g.anc_active_levels = h.anc_active_levels; g.has_exc = h.has_exc != 0;
    return g;
}
static bgr::PlanTuning plan_tuning_of(const bgr_aligner* a) {
    bgr::PlanTuning t;
    t.cfg_waves = a->cfg_waves; t.cfg_blocks_per_cu = a->cfg_blocks_per_cu; t.cfg_lds_mphf = a->cfg_lds_mphf;
    t.frame_cap = a->knob_frame_cap; t.search = a->knob_search; t.memo_cap = a->knob_memo_cap;
    t.no_greedy_fast = a->knob_greedy_fast != 0; t.no_exh_fast = a->knob_exh_fast != 0; t.no_anc_fast = a->knob_anc_fast != 0;
    return t;
}

// The planner by itself, on numbers (no device, no graph object): what tests/test_launch_plan.py sweeps on the CPU.
extern "C" int bgr_plan_launch(const bgr_plan_input* in, bgr_plan_output* out) {
    if (!in || !out) return fail(BGR_E_ARG, "bgr_plan_launch: null argument");
    bgr::PlanGraph g;
    g.k = in->k; g.slot_fill_x100 = in->slot_fill_x100; g.table_bytes = in->table_bytes; g.total_bases = in->graph_bases; g.n_unitigs = in->n_unitigs;
    g.n_buckets = in->table_bytes / 4; g.max_unitig_len = in->max_unitig_len; g.anc_n = in->anchors ? 1 : 0; g.anc_active_levels = in->anchor_levels; g.has_exc = in->has_exceptions != 0;
    bgr::PlanDevice d;
    if (in->num_cus) d.num_cus = in->num_cus;
    if (in->lds_per_cu) d.lds_per_cu = in->lds_per_cu;
    for (int i = 0; i < 7; ++i) if (in->resident_waves[i]) d.resident[i] = in->resident_waves[i];
    bgr::PlanTuning t;
    t.cfg_waves = in->cfg_waves; t.cfg_blocks_per_cu = in->cfg_blocks_per_cu; t.cfg_lds_mphf = in->cfg_lds_mphf;
    bgr::PlanBatch b;
    b.mode = in->mode; b.max_mismatch = in->max_mismatch; b.partial = in->partial; b.max_read_len = in->max_read_len; b.n_reads = in->n_reads; b.total_bases = in->total_bases;
    const bgr::LaunchPlan P = bgr::plan_launch(g, d, t, b);
    memset(out, 0, sizeof(*out));
    if (P.error) return fail(BGR_E_ARG, P.error);
    const bgr::LaunchCfg* cf[6] = {&P.cfg, &P.cfg_fast, &P.cfg_x4, &P.cfg_a4, &P.cfg_mid, &P.cfg_deep};
    const bool used[6] = {true, P.fast_pass, P.x4_pass, P.a4_pass, P.mid_pass, P.two_pass};
    for (int i = 0; i < 6; ++i) {
        if (!used[i]) continue;
        out->pass[i].used = 1; out->pass[i].blocks = cf[i]->blocks; out->pass[i].waves_per_block = cf[i]->waves_per_block;
        out->pass[i].lds_bytes = cf[i]->lds_bytes; out->pass[i].table_staged = cf[i]->stage_mphf;
    }
    out->level_search = P.level_search; out->deep_only = P.deep_only; out->x4_levels = P.x4_levels; out->memo_cap = P.memo_cap;
    out->deep_scratch_bytes = P.two_pass ? (uint64_t)P.cfg_deep.blocks * P.cfg_deep.waves_per_block * P.deep_stride * 4 : 0;
    out->arena_ints = P.arena_cap;
    return BGR_OK;
}

// The mapping launch of one batch: the geometry comes from plan_launch (launch_plan.h, a pure function of numbers), this function sizes the
// buffers and enqueues.  planes_ready: the aligner's 2-bit planes (pk_fw3 / pk_nm / pk_hasn) already hold the batch
// (bgr_align_batch_packed copied them in); else they are made from the ASCII reads at d_reads by the pre-pass.
// d_src_off (may be null): where each read's characters start in d_reads when they lie scattered in a text (text route); reads_bytes: bytes of d_reads.
static int align_device_impl(bgr_aligner* a, const bgr_params* p, const void* d_reads, const void* d_read_offsets, uint64_t n_reads,
                             uint64_t total_bases, uint32_t max_read_len, bool planes_ready, const void* d_src_off = nullptr, uint64_t reads_bytes = 0,
                             bool cursor_is_zero = false) {
    if (!a || !p) return fail(BGR_E_ARG, "bgr_align_device: null argument");
    if (p->mode > BGR_MODE_ANCHORS) return fail(BGR_E_ARG, "bgr_align_device: unknown mode");
    if (p->mode == BGR_MODE_ANCHORS && !a->graph->header.anc_n)
        return fail(BGR_E_ARG, "bgr_align_device: BGR_MODE_ANCHORS needs a graph built with BGR_BUILD_ANCHORS");
    a->last_n = n_reads;
    a->deep.open = false;
    a->deep.runs = 0; a->deep.memo_cap = 0;
    if (n_reads == 0) return BGR_OK;
    if ((!d_reads && !planes_ready) || !d_read_offsets) return fail(BGR_E_ARG, "bgr_align_device: null device buffer");
    if (n_reads >= 0xFFFFFFFFull) return fail(BGR_E_ARG, "bgr_align_device: more than 2^32-2 reads in one batch");
    HIP_TRY(hipSetDevice(a->device));
    if (a->ev_used == kTimerRing) { int rc = drain_timers(a); if (rc != BGR_OK) return rc; }
    HIP_TRY(a->results.ensure(n_reads * 8));

    // ---- launch geometry (launch_plan.h) ------------------------------------------------------------
    if (!a->plan_dev_known) {
        a->plan_dev.num_cus = (uint32_t)a->num_cus;
        a->plan_dev.lds_per_cu = a->lds_per_cu;
        for (uint32_t m = 0; m < 7; ++m) a->plan_dev.resident[m] = bgr::resident_waves_per_cu(m);
        a->plan_dev_known = true;
    }
    bgr::PlanBatch pb;
    pb.mode = p->mode; pb.max_mismatch = p->max_mismatch; pb.partial = p->partial; pb.max_read_len = max_read_len;
    pb.n_reads = n_reads; pb.total_bases = total_bases;
    const bgr::LaunchPlan P = bgr::plan_launch(plan_graph_of(a->graph->header), a->plan_dev, plan_tuning_of(a), pb);
    if (P.error) return fail(BGR_E_ARG, P.error);
    const bgr::LaunchCfg &cfg = P.cfg, &cfg_deep = P.cfg_deep, &cfg_mid = P.cfg_mid, &cfg_fast = P.cfg_fast, &cfg_x4 = P.cfg_x4, &cfg_a4 = P.cfg_a4;
    const bool level_search = P.level_search, two_pass = P.two_pass, deep_only = P.deep_only, mid_pass = P.mid_pass, fast_pass = P.fast_pass, x4_pass = P.x4_pass, a4_pass = P.a4_pass;
    const uint32_t waves = cfg.waves_per_block, wfast = P.wfast;
    if (two_pass) HIP_TRY(a->deepbuf.ensure((uint64_t)cfg_deep.blocks * cfg_deep.waves_per_block * P.deep_stride * 4));
    HIP_TRY(a->arena.ensure(P.arena_cap * 4));
    a->last_launch[0] = cfg.blocks; a->last_launch[1] = waves * 64; a->last_launch[2] = cfg.lds_bytes; a->last_launch[3] = cfg.stage_mphf | (level_search && !deep_only ? 2u : 0u) | (fast_pass ? 4u : 0u);
    if (a4_pass) { a->last_launch[0] = cfg_a4.blocks; a->last_launch[1] = cfg_a4.waves_per_block * 64; a->last_launch[2] = cfg_a4.lds_bytes; a->last_launch[3] = 4u; }
    if (x4_pass) { a->last_launch[0] = cfg_x4.blocks; a->last_launch[1] = cfg_x4.waves_per_block * 64; a->last_launch[2] = cfg_x4.lds_bytes; a->last_launch[3] = cfg_x4.stage_mphf | (level_search ? 2u : 0u) | 4u; }
    if (fast_pass) { a->last_launch[0] = cfg_fast.blocks; a->last_launch[1] = cfg_fast.waves_per_block * 64; a->last_launch[2] = cfg_fast.lds_bytes; a->last_launch[3] = cfg_fast.stage_mphf | 4u; }

    // A launch handed ASCII reads maps them straight from the characters: the mapping kernels stage each read's 2-bit words themselves (round 5:
    // the pre-pass was 4-11 % of a launch and existed only to write 40 B per read that the next kernel read back).  BGR_KNOB_GREEDY_PREPASS 1 = the
    // pre-pass over the ASCII bytes and 2-bit planes as in rounds 2-4 (what bgr_align_batch_packed's host-packed planes still use)
    // (greedy and exhaustive mode: their several-reads-per-wave kernels stage their groups, the one-read-per-wave kernels go through load_packed.  Anchors
    // mode keeps the pre-pass: its four-reads-per-wave kernel -- BooPHF arithmetic at 96 VGPRs -- lost more inside than the pre-pass cost: 934 vs 965-1 000 Mreads/s)
    // ... and so do launches without a several-reads-per-wave pass (-i, budgets beyond 254, exception planes, the knobs): the one-read-per-wave kernels alone are
    // faster from planes (depth-first 11.8 vs 12.8 ms per 5 M reads, level search 14.0 vs 14.3 per 2 M)
    const bool inline_pack = (fast_pass || x4_pass) && !planes_ready && d_reads && !a->knob_prepass;
    if (!reads_bytes) reads_bytes = total_bases;  // (reads end to end: the buffer holds exactly their bases)
    if (!inline_pack) {
        HIP_TRY(a->pk_fw3.ensure(P.plane_words * 8));
        HIP_TRY(a->pk_nm.ensure(P.plane_words * 8));
        HIP_TRY(a->pk_hasn.ensure((n_reads + 31) / 32 * 4 + 4));
    }
    bgr::BatchIO io;
    io.ascii = inline_pack ? static_cast<const uint8_t*>(d_reads) : nullptr;
    io.ascii_src = inline_pack ? static_cast<const uint32_t*>(d_src_off) : nullptr;
    io.ascii_bytes = reads_bytes;
    io.fw3 = static_cast<const uint64_t*>(a->pk_fw3.p);
    io.nmw = static_cast<const uint64_t*>(a->pk_nm.p);
    io.hasn = static_cast<const uint32_t*>(a->pk_hasn.p);
    io.read_offs = static_cast<const uint64_t*>(d_read_offsets);
    io.n_reads = (uint32_t)n_reads;
    io.words_per_read = P.words;
    io.path_cap = P.path_cap;
    io.arena_cap = (uint32_t)P.arena_cap;
    io.arena_chunk = P.arena_chunk;
    io.frames_per_wave = P.frames;
    io.ovf_list = nullptr;
    io.subset = nullptr;
    io.deep_scratch = nullptr;
    io.deep_stride = (uint32_t)P.deep_stride;
    io.level_search = level_search ? 1u : 0u;
    io.search_iters = P.search_iters;
    io.deep_memo_cap = P.memo_cap;
    io.wide_scan = P.wide_scan ? 1u : 0u;
    io.greedy_multi = 0;
    io.queue = nullptr;
    io.q_cap = 0;
    io.gen_list = nullptr;
    io.gen_ctr = 8;
    io.exh4 = 0;
    io.anc4 = 0;
    io.subset_ctr = 2;
    io.ovf_ctr = 2;
    io.wave_times = nullptr;
    io.task_ctr = 10;  // (cursor[0..15] are zeroed in front of every launch; 10 is used by nothing else)
    if (fast_pass) {
        const uint64_t grid_waves = (uint64_t)cfg_fast.blocks * cfg_fast.waves_per_block;
        HIP_TRY(a->ovf.ensure(grid_waves * P.q_cap * 8));
        HIP_TRY(a->ovf2.ensure(n_reads * 4));
    }
    if (two_pass) {
        HIP_TRY(a->retry.ensure(n_reads * 4));  // what the last pass hands back for another run (a table that filled up)
        if (!deep_only) {
            HIP_TRY(a->ovf.ensure(n_reads * 4));
            io.ovf_list = static_cast<uint32_t*>(a->ovf.p);
            if (mid_pass) HIP_TRY(a->ovf2.ensure(n_reads * 4));
        }
    }
    if (deep_only) {
        io.level_search = 0;
        io.frames_per_wave = P.frames_deep;
        io.deep_scratch = static_cast<uint32_t*>(a->deepbuf.p);
        io.ovf_list = static_cast<uint32_t*>(a->retry.p);
        io.ovf_ctr = kRetryCtr;
    }
    io.results = static_cast<uint2*>(a->results.p);
    io.arena = static_cast<int32_t*>(a->arena.p);
    io.cursor = static_cast<uint32_t*>(a->small.p);
    bgr::KernelParams kp = {p->max_mismatch, p->effort, p->partial, p->mode, a->knob_debug_stop};
    // the filter in front of a large key table pays where many read positions are probed per anchor (greedy scans: chr1-scale graph
    // 1 020 -> 1 184 Mreads/s, L2 requests per read 135 -> 28).  The exhaustive scan meets its first hit within a few positions; with
    // fingerprints behind it the filter cost it more than it saved (4-allele graph: 697 without, 664 with), with the bucket's keys compared
    // directly behind it, it pays there too (4-allele graph 702 -> 720, chr1-scale graph 1 111 -> 1 231): on by default for the minimizer
    // kind (option exh_filter = 0: without); the one-hash kind of short k stays off in exhaustive mode
    BgrDeviceGraph dgl = a->dg;
    if (p->mode == BGR_MODE_EXHAUSTIVE && !(dgl.filter_kind == BGR_FILTER_MINIMIZER && a->exh_filter)) dgl.bloom = nullptr;

    if (!cursor_is_zero) HIP_TRY(hipMemsetAsync(a->small.p, 0, 64, a->stream));  // cursor[0..15]: arena cursor, overflow flag, list counters (the text form: the parse launch cleared them)
    {   // the waves of a several-reads-per-wave kernel own the first grid x chunk ints of the arena by their number: what the cursor hands out lies behind
        const uint64_t own = fast_pass ? P.fast_rows
                           : x4_pass   ? (uint64_t)cfg_x4.blocks * cfg_x4.waves_per_block * P.arena_chunk
                           : a4_pass   ? n_reads * bgr::kA4PathInts : 0;
        io.arena_own = (uint32_t)own;   // (< 2^32: plan_launch refuses a launch whose arena is larger)
    }
    // HIP events on the aligner's stream: one in front of the launch, one behind every kernel of it (bgr_aligner_kernel_times)
    int marks = 0;
    // (timing a kernel is not free: the events make the runtime dispatch with completion stamps and keep the kernels of a launch apart -- three events around two
    // kernels cost a 262 144-read launch 16 of its 167 us, 1 730 -> 1 560 Mreads/s, and stamps taken by the dispatch packets themselves (hipExtLaunchKernelGGL)
    // cost the same, profiles/r05_mode_by_batch_size.txt; BGR_KNOB_KERNEL_EVENTS 0: none -- bgr_align_all without its timing option)
    const bool timed = !a->knob_no_events;
    auto mark = [&](const char* name) -> hipError_t {
        if (!timed || marks >= kTimerSlots) return hipSuccess;
        a->t_slot_name[marks] = name;
        return hipEventRecord(a->ev[a->ev_used][++marks], a->stream);
    };
    if (timed) HIP_TRY(hipEventRecord(a->ev[a->ev_used][0], a->stream));
    hipError_t e = hipSuccess;
    if (!planes_ready && !inline_pack) {
        HIP_TRY(hipMemsetAsync(a->pk_hasn.p, 0, (n_reads + 31) / 32 * 4, a->stream));
        e = d_src_off ? bgr::launch_pack_reads_at(static_cast<const uint8_t*>(d_reads), static_cast<const uint32_t*>(d_src_off), io.read_offs, io.n_reads, reads_bytes, total_bases,
                                                  static_cast<uint64_t*>(a->pk_fw3.p), static_cast<uint64_t*>(a->pk_nm.p), static_cast<uint32_t*>(a->pk_hasn.p), a->stream)
                      : bgr::launch_pack_reads(static_cast<const uint8_t*>(d_reads), io.read_offs, io.n_reads, total_bases, static_cast<uint64_t*>(a->pk_fw3.p),
                                               static_cast<uint64_t*>(a->pk_nm.p), static_cast<uint32_t*>(a->pk_hasn.p), a->stream);
        if (e != hipSuccess) return fail(BGR_E_HIP, std::string("pre-pass launch: ") + hipGetErrorString(e));
        HIP_TRY(mark("bgr_pack_reads_kernel"));
    }
    if (fast_pass) {
        // ONE launch of the sixteen-reads-per-wave kernel: every wave maps its share of the batch and then works off its own queue of
        // follow-up items (the next anchors of a read whose first ones failed, then its reverse complement: alignerGreedy.cpp:41-56).
        // Reads the kernel does not take (N, very long paths) are mapped from scratch by the general kernel right behind: with an
        // empty list its workgroups exit at once.
        bgr::BatchIO iof = io;
        iof.greedy_multi = 1;
        iof.words_per_read = wfast;
        iof.queue = static_cast<uint2*>(a->ovf.p);
        iof.q_cap = P.q_cap;
        iof.gen_list = static_cast<uint32_t*>(a->ovf2.p);
        iof.gen_ctr = 8;
#ifdef BGR_PHASE_TIMING
        HIP_TRY(a->wave_times.ensure((uint64_t)cfg_fast.blocks * cfg_fast.waves_per_block * 32));
        iof.wave_times = static_cast<unsigned long long*>(a->wave_times.p);
        a->wave_times_n = (uint64_t)cfg_fast.blocks * cfg_fast.waves_per_block;
#endif
        e = bgr::launch_align(dgl, iof, kp, cfg_fast, a->stream);
        if (e != hipSuccess) return fail(BGR_E_HIP, std::string("kernel launch (sixteen-reads-per-wave kernel): ") + hipGetErrorString(e));
        HIP_TRY(mark("bgr_align_greedy_multi_kernel (all reads, retries in the launch)"));
        io.subset = static_cast<uint32_t*>(a->ovf2.p);
        io.subset_ctr = 8;
    }
    if (a4_pass) {
        HIP_TRY(a->lst.ensure(n_reads * 4));
        bgr::BatchIO ioa = io;
        ioa.anc4 = P.a4_lanes;
        ioa.words_per_read = wfast;
        ioa.subset = nullptr;
        ioa.ovf_list = static_cast<uint32_t*>(a->lst.p);
        ioa.ovf_ctr = 5;
        e = bgr::launch_align(dgl, ioa, kp, cfg_a4, a->stream);
        if (e != hipSuccess) return fail(BGR_E_HIP, std::string("kernel launch (anchors, four reads per wave): ") + hipGetErrorString(e));
        HIP_TRY(mark("bgr_align_anchors4_kernel (all reads)"));
        io.subset = static_cast<uint32_t*>(a->lst.p);
        io.subset_ctr = 5;
    }
    if (x4_pass) {
        HIP_TRY(a->lst.ensure(n_reads * 4));
        bgr::BatchIO iox = io;
        iox.exh4 = P.x4_levels;
        iox.words_per_read = wfast;
        iox.level_search = 0;
        iox.subset = nullptr;
        iox.ovf_list = static_cast<uint32_t*>(a->lst.p);
        iox.ovf_ctr = 5;
#ifdef BGR_PHASE_TIMING
        HIP_TRY(a->wave_times.ensure((uint64_t)cfg_x4.blocks * cfg_x4.waves_per_block * 32));
        iox.wave_times = static_cast<unsigned long long*>(a->wave_times.p);
        a->wave_times_n = (uint64_t)cfg_x4.blocks * cfg_x4.waves_per_block;
#endif
        e = bgr::launch_align(dgl, iox, kp, cfg_x4, a->stream);
        if (e != hipSuccess) return fail(BGR_E_HIP, std::string("kernel launch (exhaustive, four reads per wave): ") + hipGetErrorString(e));
        HIP_TRY(mark("bgr_align_exhaustive4_kernel (all reads)"));
        io.subset = static_cast<uint32_t*>(a->lst.p);
        io.subset_ctr = 5;
    }
    e = bgr::launch_align(dgl, io, kp, cfg, a->stream);
    if (e != hipSuccess) return fail(BGR_E_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    HIP_TRY(mark(p->mode == BGR_MODE_GREEDY ? (fast_pass ? "bgr_align_greedy_kernel (listed reads)" : "bgr_align_greedy_kernel")
                 : p->mode == BGR_MODE_ANCHORS ? (a4_pass ? "bgr_align_anchors_kernel (listed reads)" : "bgr_align_anchors_kernel")
                 : deep_only ? "bgr_align_exhaustive_kernel (HBM state, remembered calls)" : level_search ? "bgr_align_exhaustive_dp_kernel" : "bgr_align_exhaustive_kernel"));
    bgr::BatchIO io2 = io;  // the last pass as enqueued (deep_only: the launch above)
    if (two_pass && !deep_only) {  // always enqueued: with an empty list its waves exit at once (no host round trip in between)
        const uint32_t* pending = io.ovf_list;
        uint32_t pending_ctr = 2;
        if (mid_pass) {  // depth-first search, LDS stack, over what the level search listed; its own overflow goes to list 2
            bgr::BatchIO iom = io;
            iom.level_search = 0;
            iom.frames_per_wave = P.frames_mid;
            iom.subset = io.ovf_list;
            iom.subset_ctr = 2;
            iom.ovf_list = static_cast<uint32_t*>(a->ovf2.p);
            iom.ovf_ctr = 3;
            e = bgr::launch_align(dgl, iom, kp, cfg_mid, a->stream);
            if (e != hipSuccess) return fail(BGR_E_HIP, std::string("kernel launch (depth-first pass): ") + hipGetErrorString(e));
            HIP_TRY(mark("bgr_align_exhaustive_kernel (listed reads)"));
            pending = iom.ovf_list;
            pending_ctr = 3;
        }
        io2.frames_per_wave = P.frames_deep;
        io2.subset = pending;
        io2.subset_ctr = pending_ctr;
        io2.ovf_list = static_cast<uint32_t*>(a->retry.p);
        io2.ovf_ctr = kRetryCtr;
        io2.deep_scratch = static_cast<uint32_t*>(a->deepbuf.p);
        io2.level_search = 0;
        e = bgr::launch_align(dgl, io2, kp, cfg_deep, a->stream);
        if (e != hipSuccess) return fail(BGR_E_HIP, std::string("kernel launch (last pass): ") + hipGetErrorString(e));
        HIP_TRY(mark("bgr_align_exhaustive_kernel (HBM state, remembered calls; listed reads)"));
    }
    if (two_pass) {  // what settle_launch needs to run the last pass again for the reads it handed back
        a->deep.open = true;
        a->deep.io = io2; a->deep.kp = kp; a->deep.dg = dgl; a->deep.cfg = cfg_deep;
        a->deep.per_wave_lds = bgr::deep_lds_bytes_per_wave(max_read_len);
        a->deep.path_cap = P.path_cap; a->deep.frames = P.frames_deep; a->deep.memo_cap = P.memo_cap;
        a->deep.runs = 1;
    }
    if (timed) {
        a->ev_marks[a->ev_used] = marks;
        ++a->ev_used;
    }
    return BGR_OK;
}

// Behind a mapping launch, with its stream waited for and cursor[0 .. 15] on the host (`cur`): the arena must not have overflowed, and in
// exhaustive mode the last pass may have handed reads back whose table of remembered calls filled up (exh_memo, exhaustive_kernels.hip):
// those run again -- the last pass only, over that list -- with a table sixteen times as large and as few waves as the list has reads, until
// none is left.  A search visits at most positions x halves x 2 nodes, so this ends; what can end it early is the device's memory.
static int settle_launch(bgr_aligner* a, uint32_t* cur) {
    if (cur[1]) return fail(BGR_E_INTERNAL, "path arena overflow (internal sizing error)");
    if (!a->deep.open) return BGR_OK;
    while (cur[kRetryCtr]) {
        const uint32_t n_retry = cur[kRetryCtr];
        auto& D = a->deep;
        if (D.runs >= bgr::kDeepRuns || D.memo_cap >= (1u << 28))
            return fail(BGR_E_NOMEM, "exhaustive search: a read's table of remembered calls outgrew 2^28 entries per wave (8 GiB)");
        D.memo_cap *= 16;
        const uint64_t stride = bgr::deep_scratch_words(D.path_cap, D.frames, D.memo_cap);
        if (stride > 0xFFFFFFFFull) return fail(BGR_E_NOMEM, "exhaustive search: a read's table of remembered calls outgrew the 16 GiB a wave can address");
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        const uint64_t room = (uint64_t)free_b + a->deepbuf.cap;
        uint64_t waves = std::min<uint64_t>(n_retry, (uint64_t)D.cfg.blocks * D.cfg.waves_per_block);
        waves = std::min<uint64_t>(waves, std::max<uint64_t>(1, room / 2 / (stride * 4)));
        if (stride * 4 > room - room / 8) return fail(BGR_E_NOMEM, "exhaustive search: not enough device memory for a read's table of remembered calls");
        HIP_TRY(a->deepbuf.ensure(waves * stride * 4));
        // the list handed back becomes the list to map; the kernel appends to the other one
        DevBuf& other = a->retry2;
        HIP_TRY(other.ensure((uint64_t)n_retry * 4));
        bgr::BatchIO io = D.io;
        io.subset = io.ovf_list;
        io.subset_ctr = kRetrySubsetCtr;
        io.ovf_list = static_cast<uint32_t*>(other.p);
        io.ovf_ctr = kRetryCtr;
        io.deep_scratch = static_cast<uint32_t*>(a->deepbuf.p);
        io.deep_stride = (uint32_t)stride;
        io.deep_memo_cap = D.memo_cap;
        uint32_t ctr[2] = {0, n_retry};  // cursor[kRetryCtr] = 0, cursor[kRetrySubsetCtr] = n
        static_assert(kRetrySubsetCtr == kRetryCtr + 1, "the two counters are written with one copy");
        HIP_TRY(hipMemcpyAsync(static_cast<uint32_t*>(a->small.p) + kRetryCtr, ctr, 8, hipMemcpyHostToDevice, a->stream));
        bgr::LaunchCfg cfg = D.cfg;
        cfg.waves_per_block = (uint32_t)std::min<uint64_t>(cfg.waves_per_block, waves);
        cfg.blocks = (uint32_t)std::max<uint64_t>(1, waves / cfg.waves_per_block);
        cfg.lds_bytes = bgr::kLdsFixed + cfg.waves_per_block * D.per_wave_lds;
        hipError_t e = bgr::launch_align(D.dg, io, D.kp, cfg, a->stream);
        if (e != hipSuccess) return fail(BGR_E_HIP, std::string("kernel launch (last pass, run again): ") + hipGetErrorString(e));
        HIP_TRY(hipMemcpyAsync(cur, a->small.p, 64, hipMemcpyDeviceToHost, a->stream));
        HIP_TRY(wait_stream(a));
        if (cur[1]) return fail(BGR_E_INTERNAL, "path arena overflow (internal sizing error)");
        std::swap(a->retry, a->retry2);  // (D.io.ovf_list names the list the next round maps)
        D.io.ovf_list = io.ovf_list;
        ++D.runs;
    }
    a->deep.open = false;
    return BGR_OK;
}
// ... for callers that have not looked at the cursor yet (exhaustive launches only: nothing to settle otherwise but the arena flag, which every fetch checks)
static int settle_launch_sync(bgr_aligner* a) {
    if (!a->deep.open) return BGR_OK;
    uint32_t cur[16];
    HIP_TRY(hipMemcpyAsync(cur, a->small.p, sizeof(cur), hipMemcpyDeviceToHost, a->stream));
    HIP_TRY(wait_stream(a));
    return settle_launch(a, cur);
}

int bgr_align_device(bgr_aligner* a, const bgr_params* p, const void* d_reads, const void* d_read_offsets, uint64_t n_reads,
                     uint64_t total_bases, uint32_t max_read_len) {
    return align_device_impl(a, p, d_reads, d_read_offsets, n_reads, total_bases, max_read_len, false);
}

uint64_t bgr_packed_plane_words(uint64_t n_reads, uint64_t total_bases) { return bgr::packed_plane_words(n_reads, total_bases); }

int bgr_pack_reads(const char* reads, const uint64_t* read_offsets, uint64_t n, uint64_t* fw3, uint32_t* hasn, uint32_t* nm_index, uint64_t* nm_value,
                   uint64_t nm_cap, uint64_t* nm_count, uint32_t* max_read_len) {
    if (!read_offsets || !fw3 || !hasn || !nm_count || (n && !reads)) return fail(BGR_E_ARG, "bgr_pack_reads: null argument");
    const uint64_t base = read_offsets[0];
    memset(hasn, 0, ((n + 31) / 32) * 4);
    uint64_t cnt = 0;
    uint32_t longest = 0;
    std::vector<uint64_t> nm;
    for (uint64_t r = 0; r < n; ++r) {
        const uint64_t off = read_offsets[r] - base, len64 = read_offsets[r + 1] - read_offsets[r];
        if (len64 > 0x7FFFFFFFull) return fail(BGR_E_ARG, "bgr_pack_reads: read longer than 2^31 bases");
        const uint32_t len = (uint32_t)len64, words = (len + 31) >> 5;
        longest = std::max(longest, len);
        if (nm.size() < words) nm.resize(words);
        const uint64_t w0 = bgr::packed_word_offset(off, r);
        if (bgr::pack_read(reads + read_offsets[r], len, fw3 + w0, nm.data())) {
            hasn[r >> 5] |= 1u << (r & 31);
            if (cnt + words > nm_cap || !nm_index || !nm_value) return fail(BGR_E_CAPACITY, "bgr_pack_reads: N-mask list too small");
            for (uint32_t j = 0; j < words; ++j) { nm_index[cnt] = (uint32_t)(w0 + j); nm_value[cnt] = nm[j]; ++cnt; }
        }
    }
    *nm_count = cnt;
    if (max_read_len) *max_read_len = longest;
    return BGR_OK;
}

int bgr_align_batch_packed(bgr_aligner* a, const bgr_params* p, const bgr_packed_reads* pk, uint64_t n, int32_t* paths_out, uint64_t paths_cap,
                           uint64_t* path_offsets, uint8_t* status) {
    if (!a || !p || !pk || !path_offsets || (n && (!pk->read_offsets || !pk->fw3 || !pk->hasn || !status))) return fail(BGR_E_ARG, "bgr_align_batch_packed: null argument");
    if (pk->nm_count && (!pk->nm_index || !pk->nm_value)) return fail(BGR_E_ARG, "bgr_align_batch_packed: null N-mask list");
    path_offsets[0] = 0;
    a->last_n = 0;
    if (n == 0) return BGR_OK;
    if (pk->read_offsets[0] != 0) return fail(BGR_E_ARG, "bgr_align_batch_packed: read_offsets must start at 0 (they address the planes)");
    const uint64_t total = pk->read_offsets[n];
    if (n >= 0x7FFFFFFFull || 2 * (total + 16 * n) >= 0xFFFFFFFFull - (256ull << 20))
        return fail(BGR_E_ARG, "bgr_align_batch_packed: batch too large for one launch (2*(bases + 16*reads) must stay below 2^32 - 2^28); pack it in pieces");
    HIP_TRY(hipSetDevice(a->device));
    const uint64_t plane_words = bgr::packed_plane_words(n, total);
    HIP_TRY(a->in_offs.ensure((n + 1) * 8));
    HIP_TRY(a->pk_fw3.ensure(plane_words * 8));
    HIP_TRY(a->pk_nm.ensure(plane_words * 8));
    HIP_TRY(a->pk_hasn.ensure((n + 31) / 32 * 4 + 4));
    HIP_TRY(hipMemcpyAsync(a->in_offs.p, pk->read_offsets, (n + 1) * 8, hipMemcpyHostToDevice, a->stream));
    HIP_TRY(hipMemcpyAsync(a->pk_fw3.p, pk->fw3, ((total >> 5) + n + 1) * 8, hipMemcpyHostToDevice, a->stream));
    HIP_TRY(hipMemcpyAsync(a->pk_hasn.p, pk->hasn, (n + 31) / 32 * 4, hipMemcpyHostToDevice, a->stream));
    if (pk->nm_count) {  // the N-mask words of the few reads that hold an N: a sparse list, scattered into the plane on the device
        const uint64_t voff = (pk->nm_count * 4 + 7) & ~7ull;  // indices, then the values 8-byte aligned, in the ASCII staging buffer (unused here)
        HIP_TRY(a->in_reads.ensure(voff + pk->nm_count * 8));
        char* d = static_cast<char*>(a->in_reads.p);
        HIP_TRY(hipMemcpyAsync(d, pk->nm_index, pk->nm_count * 4, hipMemcpyHostToDevice, a->stream));
        HIP_TRY(hipMemcpyAsync(d + voff, pk->nm_value, pk->nm_count * 8, hipMemcpyHostToDevice, a->stream));
        hipError_t e = bgr::launch_scatter_words(reinterpret_cast<const uint32_t*>(d), reinterpret_cast<const uint64_t*>(d + voff), pk->nm_count,
                                                 static_cast<uint64_t*>(a->pk_nm.p), plane_words, a->stream);
        if (e != hipSuccess) return fail(BGR_E_HIP, std::string("N-mask scatter launch: ") + hipGetErrorString(e));
    }
    uint32_t max_len = pk->max_read_len;
    if (!max_len) {
        for (uint64_t i = 0; i < n; ++i) max_len = std::max<uint32_t>(max_len, (uint32_t)std::min<uint64_t>(pk->read_offsets[i + 1] - pk->read_offsets[i], 0x7FFFFFFFull));
    }
    // (the host buffers may be reused once the copies are done: bgr_aligner_fetch below synchronises the stream)
    int rc = align_device_impl(a, p, nullptr, a->in_offs.p, n, total, max_len, true);
    if (rc != BGR_OK) return rc;
    return bgr_aligner_fetch(a, n, paths_out, paths_cap, path_offsets, status);
}

// ---- text form: FASTA bytes in, record bytes out (text_kernels.hip) -------------------------------------------------------------
static int fetch_text_impl(bgr_aligner* a, bgr_text_batch* b) {
    b->paths_bytes = a->tx_pbytes;
    b->notaligned_bytes = a->tx_nbytes;
    if (!b->want_output || a->tx_n_acc == 0) return BGR_OK;
    if (a->tx_pbytes > b->paths_cap || a->tx_nbytes > b->notaligned_cap || (a->tx_pbytes && !b->paths_out) || (a->tx_nbytes && !b->notaligned_out))
        return fail(BGR_E_CAPACITY, "bgr_align_fasta_text: output buffer too small (paths_bytes / notaligned_bytes say what is needed; bgr_aligner_fetch_text)");
    if (!a->tx_written) {   // (correction mode; streams larger than the buffers the format launch wrote into; a second fetch)
    HIP_TRY(a->tx_pout.ensure(a->tx_pbytes + 64));
    HIP_TRY(a->tx_nout.ensure(a->tx_nbytes + 64));
    hipError_t e = a->tx_want == 2
        ? bgr::launch_text_correct_write(a->dg, a->tx_text, static_cast<const uint2*>(a->results.p), static_cast<const int32_t*>(a->arena.p), static_cast<const uint4*>(a->tx_rec.p),
                                         static_cast<const uint32_t*>(a->tx_accrec.p), (uint32_t)a->tx_n_acc, static_cast<const uint32_t*>(a->tx_poff.p),
                                         static_cast<const uint32_t*>(a->tx_noff.p), static_cast<const uint32_t*>(a->tx_idx.p), static_cast<uint8_t*>(a->tx_pout.p),
                                         static_cast<uint8_t*>(a->tx_nout.p), a->stream)
        : bgr::launch_text_write(a->tx_text, static_cast<const uint2*>(a->results.p), static_cast<const int32_t*>(a->arena.p),
                                 static_cast<const uint4*>(a->tx_rec.p), static_cast<const uint32_t*>(a->tx_accrec.p), (uint32_t)a->tx_n_acc,
                                 static_cast<const uint32_t*>(a->tx_poff.p), static_cast<const uint32_t*>(a->tx_noff.p), static_cast<uint8_t*>(a->tx_pout.p),
                                 static_cast<uint8_t*>(a->tx_nout.p), a->stream);
    if (e != hipSuccess) return fail(BGR_E_HIP, std::string("text write launch: ") + hipGetErrorString(e));
    a->tx_written = true;
    }
    if (a->tx_pbytes) HIP_TRY(hipMemcpyAsync(b->paths_out, a->tx_pout.p, a->tx_pbytes, hipMemcpyDeviceToHost, a->stream));
    if (a->tx_nbytes) HIP_TRY(hipMemcpyAsync(b->notaligned_out, a->tx_nout.p, a->tx_nbytes, hipMemcpyDeviceToHost, a->stream));
    HIP_TRY(wait_stream(a));
    return BGR_OK;
}

int bgr_aligner_fetch_text(bgr_aligner* a, bgr_text_batch* b) {
    if (!a || !b) return fail(BGR_E_ARG, "bgr_aligner_fetch_text: null argument");
    if (b->struct_size != sizeof(bgr_text_batch)) return fail(BGR_E_ARG, "bgr_aligner_fetch_text: bgr_text_batch.struct_size is not this library's sizeof(bgr_text_batch)");
    HIP_TRY(hipSetDevice(a->device));
    return fetch_text_impl(a, b);
}

int bgr_text_stage_create(int device, bgr_text_stage** out) {
    if (!out) return fail(BGR_E_ARG, "bgr_text_stage_create: null argument");
    HIP_TRY(hipSetDevice(device));
    bgr_text_stage* s = new bgr_text_stage();
    s->device = device;
    hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    s->timing = bgr::opt("timing") != 0;
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev, s->timing ? hipEventDefault : hipEventDisableTiming);
    if (e == hipSuccess && s->timing) e = hipEventCreate(&s->ev0);
    if (e != hipSuccess) { bgr_text_stage_destroy(s); return fail(BGR_E_HIP, std::string("bgr_text_stage_create: ") + hipGetErrorString(e)); }
    *out = s;
    return BGR_OK;
}

void bgr_text_stage_destroy(bgr_text_stage* s) {
    if (!s) return;
    if (hipSetDevice(s->device) == hipSuccess) {
        if (s->stream) { (void)hipStreamSynchronize(s->stream); s->settle(); (void)hipStreamDestroy(s->stream); }
        if (s->timing && s->copy_ms > 0) fprintf(stderr, "bgreat: stage on device %d: %.0f MB up in %.3f s of copies = %.1f GB/s\n", s->device, s->copy_bytes / 1e6, s->copy_ms / 1e3, s->copy_bytes / s->copy_ms / 1e6);
        if (s->ev) (void)hipEventDestroy(s->ev);
        if (s->ev0) (void)hipEventDestroy(s->ev0);
        s->buf.release();
    }
    delete s;
}

int bgr_text_stage_device(const bgr_text_stage* s) { return s ? s->device : -1; }

int bgr_text_stage_upload(bgr_text_stage* s, const char* text, uint64_t bytes) {
    if (!s || (bytes && !text)) return fail(BGR_E_ARG, "bgr_text_stage_upload: null argument");
    if (bytes >= (1ull << 31)) return fail(BGR_E_ARG, "bgr_text_stage_upload: piece of 2 GiB or more; cut it");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));  // (a piece still on its way: the buffer may grow, i.e. move)
    s->settle();
    HIP_TRY(s->buf.ensure(bytes + 128));
    if (s->timing) { HIP_TRY(hipEventRecord(s->ev0, s->stream)); s->pending = true; }
    HIP_TRY(hipMemsetAsync(static_cast<char*>(s->buf.p) + (bytes & ~63ull), 0, 128, s->stream));   // (from an aligned address: one fill; the copies below lay the text over its start)
    if (bytes) HIP_TRY(hipMemcpyAsync(s->buf.p, text, bytes, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipEventRecord(s->ev, s->stream));
    s->bytes = bytes;
    return BGR_OK;
}

int bgr_text_stage_upload_parts(bgr_text_stage* s, uint32_t n_parts, const char* const* parts, const uint64_t* part_bytes) {
    if (!s || (n_parts && (!parts || !part_bytes))) return fail(BGR_E_ARG, "bgr_text_stage_upload_parts: null argument");
    uint64_t bytes = 0;
    for (uint32_t i = 0; i < n_parts; ++i) { if (part_bytes[i] && !parts[i]) return fail(BGR_E_ARG, "bgr_text_stage_upload_parts: null part"); bytes += part_bytes[i]; }
    if (bytes >= (1ull << 31)) return fail(BGR_E_ARG, "bgr_text_stage_upload_parts: piece of 2 GiB or more; cut it");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));  // (a piece still on its way: the buffer may grow, i.e. move)
    s->settle();
    HIP_TRY(s->buf.ensure(bytes + 128));
    if (s->timing) { HIP_TRY(hipEventRecord(s->ev0, s->stream)); s->pending = true; }
    HIP_TRY(hipMemsetAsync(static_cast<char*>(s->buf.p) + (bytes & ~63ull), 0, 128, s->stream));   // (from an aligned address: one fill; the copies below lay the text over its start)
    uint64_t at = 0;
    for (uint32_t i = 0; i < n_parts; ++i) {
        if (part_bytes[i]) HIP_TRY(hipMemcpyAsync(static_cast<char*>(s->buf.p) + at, parts[i], part_bytes[i], hipMemcpyHostToDevice, s->stream));
        at += part_bytes[i];
    }
    HIP_TRY(hipEventRecord(s->ev, s->stream));
    s->bytes = bytes;
    return BGR_OK;
}

static double wall_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int bgr_align_fasta_text(bgr_aligner* a, const bgr_params* p, bgr_text_batch* b) {
    if (!a || !p || !b) return fail(BGR_E_ARG, "bgr_align_fasta_text: null argument");
    if (b->struct_size != sizeof(bgr_text_batch)) return fail(BGR_E_ARG, "bgr_align_fasta_text: bgr_text_batch.struct_size is not this library's sizeof(bgr_text_batch): the caller was built against another header (zero the struct, set struct_size)");
    if (b->text_bytes && !b->text && !b->stage) return fail(BGR_E_ARG, "bgr_align_fasta_text: null argument");
    if (b->record_info_out && b->text_bytes >= (1ull << 30)) return fail(BGR_E_ARG, "bgr_align_fasta_text: record_info_out holds a read's length in 30 bits: pieces below 2^30 bytes");
    double tw = wall_now();
    auto lap = [&](int i) { const double t = wall_now(); a->tx_phase_s[i] += t - tw; tw = t; };
    if (b->text_bytes >= (1ull << 31)) return fail(BGR_E_ARG, "bgr_align_fasta_text: piece of 2 GiB or more; cut it");
    if (b->want_output > 2) return fail(BGR_E_ARG, "bgr_align_fasta_text: want_output must be 0, 1 or 2");
    if (b->fastq > 2) return fail(BGR_E_ARG, "bgr_align_fasta_text: fastq must be 0, 1 (four-line records) or 2 (header and read lines only)");
    const uint32_t rec_lines = b->fastq == 2 ? 2u : b->fastq ? 4u : 0u;  // lines per record of a FASTQ piece (0: FASTA, records start at '>' lines)
    if (b->fastq && b->text_bytes && b->text && b->text[b->text_bytes - 1] != '\n' && !b->stage)
        return fail(BGR_E_ARG, "bgr_align_fasta_text: a FASTQ piece holds whole four-line records and ends with a newline");
    if (b->want_output == 2 && (a->graph->header.has_exc || p->mode == BGR_MODE_EXHAUSTIVE))
        return fail(BGR_E_ARG, "bgr_align_fasta_text: correction on the device needs a graph of ACGT-only unitigs and greedy mode (format such a run on the host)");
    if (b->record_info_out && (b->want_output == 2 || b->record_info_cap < b->text_bytes / 24 + 1024))
        return fail(BGR_E_ARG, "bgr_align_fasta_text: record_info_out needs room for text_bytes / 24 + 1024 words and is not available in correction mode");
    a->tx_want = b->want_output;
    b->irregular = 0;
    b->n_records = b->n_accepted = b->paths_bytes = b->notaligned_bytes = 0;
    a->last_n = 0;
    a->tx_n_acc = a->tx_pbytes = a->tx_nbytes = 0;
    if (b->text_bytes == 0) return BGR_OK;
    HIP_TRY(hipSetDevice(a->device));
    const uint32_t nbytes = (uint32_t)b->text_bytes;
    const uint8_t* text = nullptr;
    // 1. the piece in HBM (zero padded): uploaded ahead of this call by a stage (bgr_text_stage_upload, a copy stream of its own), or copied now
    if (b->stage) {
        if (b->stage->device != a->device || b->stage->bytes != b->text_bytes) return fail(BGR_E_ARG, "bgr_align_fasta_text: the stage holds another piece / lives on another device");
        HIP_TRY(hipStreamWaitEvent(a->stream, b->stage->ev, 0));
        text = static_cast<const uint8_t*>(b->stage->buf.p);
    } else {
        if (!b->text) return fail(BGR_E_ARG, "bgr_align_fasta_text: null text");
        HIP_TRY(a->tx_in.ensure((uint64_t)nbytes + 128));
        text = static_cast<const uint8_t*>(a->tx_in.p);
        HIP_TRY(hipMemsetAsync(static_cast<char*>(a->tx_in.p) + (nbytes & ~63ull), 0, 128, a->stream));   // (64 zero bytes behind the piece at least; from an aligned address: ONE fill, and the copy below lays the text over its start)
        HIP_TRY(hipMemcpyAsync(a->tx_in.p, b->text, nbytes, hipMemcpyHostToDevice, a->stream));
    }
    a->tx_text = text;
    // 2. records: starts, extents, shape, accept test; accepted records compacted in input order.  Room for one record per 24 bytes of
    // text (a sequencing read with its header is several times that): a piece with more record starts goes to the host parser.
    const uint32_t R_cap = nbytes / 24 + 1024;
    // (tickets and chains of the two one-launch kernels, text_kernels.hip: never cleared between launches -- every launch gets a fresh epoch)
    {
        const uint64_t chain_words = std::max<uint64_t>(3ull * bgr::text_tiles(nbytes), 2ull * bgr::format_tiles(R_cap));
        const size_t had = a->tx_state.cap;
        HIP_TRY(a->tx_state.ensure(64 + chain_words * 8));
        if (a->tx_state.cap != had || a->tx_epoch + 2 > BGR_TEXT_EPOCH_MAX) {
            const bool fresh = a->tx_state.cap != had;
            HIP_TRY(hipMemsetAsync(a->tx_state.p, 0, a->tx_state.cap, a->stream));
            a->tx_epoch = fresh ? (uint32_t)bgr::opt("test.text_epoch") : 0u;   // (0 but for the test hook)
            a->tx_ticket[0] = a->tx_ticket[1] = 0;
        }
    }
    uint32_t* tickets = static_cast<uint32_t*>(a->tx_state.p);
    uint64_t* chains = reinterpret_cast<uint64_t*>(static_cast<char*>(a->tx_state.p) + 64);
    HIP_TRY(a->tx_sums.ensure((2ull * bgr::scan_tiles(R_cap) + 16) * 4));   // (correction mode's scans)
    for (DevBuf* d : {&a->tx_idx, &a->tx_accrec, &a->tx_accsrc, &a->tx_psz, &a->tx_nsz, &a->tx_poff, &a->tx_noff})
        HIP_TRY(d->ensure(((uint64_t)R_cap + 4) * 4));
    HIP_TRY(a->tx_rec.ensure(((uint64_t)R_cap + 1) * 16));
    HIP_TRY(a->tx_offs.ensure(((uint64_t)R_cap + 2) * 8));
    uint32_t* sums2 = static_cast<uint32_t*>(a->tx_sums.p);
    // TXT_INFO_WORDS u32 behind cursor / counters / CSR total -- two such blocks, taken in turn: a piece's parse launch clears the block of the NEXT piece (and the
    // mapping launch's cursor), so that no fill stands in front of either (the blocks start zero: bgr_aligner_create)
    a->tx_flip ^= 1u;   // (here, behind everything that can fail before the launch: a block is cleared by the launch in front of the one that uses it)
    uint32_t* info = reinterpret_cast<uint32_t*>(static_cast<char*>(a->small.p) + 192 + 32 * a->tx_flip);
    uint32_t* info_next = reinterpret_cast<uint32_t*>(static_cast<char*>(a->small.p) + 192 + 32 * (a->tx_flip ^ 1u));
    static_assert(TXT_INFO_WORDS * 4 == 32, "two info blocks of 32 bytes at small + 192");
    hipError_t e = bgr::launch_text_parse(text, nbytes, rec_lines, a->dg.k, tickets, a->tx_ticket[0], ++a->tx_epoch, chains, static_cast<uint4*>(a->tx_rec.p),
                                          static_cast<uint32_t*>(a->tx_idx.p), static_cast<uint32_t*>(a->tx_accrec.p), static_cast<uint32_t*>(a->tx_accsrc.p),
                                          static_cast<uint64_t*>(a->tx_offs.p), info, R_cap, static_cast<uint32_t*>(a->small.p), 16, info_next, TXT_INFO_WORDS, a->stream);
    if (e == hipSuccess) a->tx_ticket[0] += bgr::text_tiles(nbytes);   // (as many tickets as workgroups will take)
    if (e != hipSuccess) return fail(BGR_E_HIP, std::string("text record launches: ") + hipGetErrorString(e));
    uint32_t h[TXT_INFO_WORDS];
    HIP_TRY(hipMemcpyAsync(h, info, sizeof(h), hipMemcpyDeviceToHost, a->stream));
    HIP_TRY(wait_stream(a));
    lap(0);
    const uint32_t R = h[TXT_INFO_N_REC];
    b->n_records = R;
    if (R == 0 || R > R_cap) { b->irregular = 1; return BGR_OK; }  // no record start in the bytes, or far more than a read file has: the host parser decides
    lap(1);
    if (h[TXT_INFO_IRREGULAR]) { b->irregular = 1; return BGR_OK; }
    const uint32_t n_acc = h[TXT_INFO_N_ACC], bases = h[TXT_INFO_BASES], max_len = h[TXT_INFO_MAX_LEN];
    b->n_accepted = n_acc;
    if (n_acc == 0) {
        if (b->record_info_out) memset(b->record_info_out, 0, (size_t)R * 4);  // (nothing kept: every record a dropped one)
        return BGR_OK;
    }
    // 3. the mapping launch, its reads where they lie in the text (planes by the pre-pass, or -- greedy mode -- staged by the mapping kernels themselves)
    int rc = align_device_impl(a, p, text, a->tx_offs.p, n_acc, bases, max_len, false, a->tx_accsrc.p, nbytes, true);
    if (rc != BGR_OK) return rc;
    a->last_n = 0;  // (bgr_aligner_fetch has no host read_offsets to pair its rows with: the text form hands out text)
    a->tx_n_acc = n_acc;
    // Everything behind the mapping launch reads its results.  In exhaustive mode those are final only once the launch is settled (reads the last pass
    // handed back for a larger table are mapped again: settle_launch) -- which hardly ever happens: so the kernels below are enqueued at once, the
    // cursor comes back with the wait this call makes anyway, and only a launch that did hand reads back is settled and the kernels enqueued AGAIN
    // (a wait between the mapping launch and them cost -b a fifth of its end-to-end rate)
    uint64_t pcap_dev = 0, ncap_dev = 0;
    a->tx_written = false;
    for (int again = 0; again < 2; ++again) {
    if (b->record_info_out) {  // what became of every record (the -b progress blocks of the caller), on its way to the host behind the mapping launch
        HIP_TRY(a->tx_info.ensure((uint64_t)R * 4));
        e = bgr::launch_text_record_info(static_cast<const uint4*>(a->tx_rec.p), static_cast<const uint32_t*>(a->tx_idx.p), static_cast<const uint2*>(a->results.p), R,
                                         static_cast<uint32_t*>(a->tx_info.p), a->stream);
        if (e != hipSuccess) return fail(BGR_E_HIP, std::string("record info launch: ") + hipGetErrorString(e));
        HIP_TRY(hipMemcpyAsync(b->record_info_out, a->tx_info.p, (size_t)R * 4, hipMemcpyDeviceToHost, a->stream));
    }
    if (!b->want_output) {  // counters only: the launch still has to be settled (arena flag; exhaustive mode: reads the last pass handed back)
        uint32_t cur[16];
        HIP_TRY(hipMemcpyAsync(cur, a->small.p, sizeof(cur), hipMemcpyDeviceToHost, a->stream));
        HIP_TRY(wait_stream(a));
        const bool handed_back = a->deep.open && cur[kRetryCtr] != 0;
        rc = settle_launch(a, cur);
        if (rc != BGR_OK || !handed_back || !b->record_info_out) return rc;
        continue;  // (the record info was made from results that were not final)
    }
    // 4. sizes of the records, stream offsets, the bytes
    if (b->want_output == 2) {  // (tx_idx is free again behind the compaction: it takes the corrected reads' lengths)
        HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(info + TXT_INFO_BUG), -1, 1, a->stream));
        e = bgr::launch_text_correct_sizes(a->dg, static_cast<const uint2*>(a->results.p), static_cast<const int32_t*>(a->arena.p), static_cast<const uint4*>(a->tx_rec.p),
                                           static_cast<const uint32_t*>(a->tx_accrec.p), n_acc, static_cast<uint32_t*>(a->tx_psz.p), static_cast<uint32_t*>(a->tx_nsz.p),
                                           static_cast<uint32_t*>(a->tx_idx.p), info + TXT_INFO_BUG, a->stream);
    } else {
        // sizes, offsets and bytes in one launch, into device buffers of the caller's capacities (at most twice the piece + 1 MB each: a piece whose
        // streams need more -- paths of very many unitigs -- is written by bgr_text_write_kernel from the offsets once the totals are known)
        pcap_dev = b->paths_out ? std::min<uint64_t>(b->paths_cap, 2ull * nbytes + (1ull << 20)) : 0;
        ncap_dev = b->notaligned_out ? std::min<uint64_t>(b->notaligned_cap, 2ull * nbytes + (1ull << 20)) : 0;
        HIP_TRY(a->tx_pout.ensure(pcap_dev + 64));
        HIP_TRY(a->tx_nout.ensure(ncap_dev + 64));
        e = bgr::launch_text_format(text, static_cast<const uint2*>(a->results.p), static_cast<const int32_t*>(a->arena.p), static_cast<const uint4*>(a->tx_rec.p),
                                    static_cast<const uint32_t*>(a->tx_accrec.p), n_acc, tickets + 1, a->tx_ticket[1], ++a->tx_epoch, chains, static_cast<uint32_t*>(a->tx_poff.p),
                                    static_cast<uint32_t*>(a->tx_noff.p), static_cast<uint8_t*>(a->tx_pout.p), static_cast<uint8_t*>(a->tx_nout.p), pcap_dev, ncap_dev, info, a->stream);
        if (n_acc && e == hipSuccess) a->tx_ticket[1] += bgr::format_tiles(n_acc);   // (as many tickets as workgroups ran: launch_text_format launches nothing for zero reads)
    }
    if (e == hipSuccess && b->want_output == 2) e = bgr::launch_scan2_u32(static_cast<const uint32_t*>(a->tx_psz.p), static_cast<const uint32_t*>(a->tx_nsz.p), static_cast<uint32_t*>(a->tx_poff.p),
                                                   static_cast<uint32_t*>(a->tx_noff.p), n_acc, nullptr, sums2, info + TXT_INFO_PBYTES, info + TXT_INFO_NBYTES, a->stream);
    if (e != hipSuccess) return fail(BGR_E_HIP, std::string("text size launches: ") + hipGetErrorString(e));
    uint32_t all[64], h2[16];   // the cursor words (cursor[1]: arena overflow flag) and the info block lie in the same 256 bytes: one copy
    HIP_TRY(hipMemcpyAsync(all, a->small.p, sizeof(all), hipMemcpyDeviceToHost, a->stream));
    HIP_TRY(wait_stream(a));
    memcpy(h2, all, sizeof(h2));
    memcpy(h, all + (info - static_cast<const uint32_t*>(a->small.p)), sizeof(h));
    lap(2);
    if (h2[1]) return fail(BGR_E_INTERNAL, "path arena overflow (internal sizing error)");
    if (a->deep.open) {
        const bool handed_back = h2[kRetryCtr] != 0;
        rc = settle_launch(a, h2);
        if (rc != BGR_OK) return rc;
        if (handed_back) continue;  // sizes and offsets once more, from the final results
    }
    break;
    }
    if (b->want_output == 2 && h[TXT_INFO_BUG] != 0xFFFFFFFFu) {  // a path that does not spell a walk: the reference prints "bug compaction" and exits
        b->irregular = 2;                                          // (aligner.cpp:280-283); the caller reproduces that on the host
        a->tx_n_acc = 0;
        return BGR_OK;
    }
    a->tx_pbytes = h[TXT_INFO_PBYTES];
    a->tx_nbytes = h[TXT_INFO_NBYTES];
    a->tx_written = b->want_output == 1 && a->tx_pbytes <= pcap_dev && a->tx_nbytes <= ncap_dev;   // (every workgroup's stretch ended below the capacities)
    const int frc = fetch_text_impl(a, b);
    lap(3);
    a->tx_phase_s[4] += 1;
    return frc;
}

int bgr_aligner_sync(bgr_aligner* a) {
    if (!a) return fail(BGR_E_ARG, "bgr_aligner_sync: null aligner");
    HIP_TRY(hipSetDevice(a->device));
    HIP_TRY(hipStreamSynchronize(a->stream));
    return settle_launch_sync(a);  // (exhaustive mode: reads the last pass handed back are mapped before the results count as final)
}

int bgr_aligner_device_results(bgr_aligner* a, void** d_results, void** d_arena, void** d_cursor) {
    if (!a) return fail(BGR_E_ARG, "bgr_aligner_device_results: null aligner");
    if (d_results) *d_results = a->results.p;
    if (d_arena) *d_arena = a->arena.p;
    if (d_cursor) *d_cursor = a->small.p;
    return BGR_OK;
}

// results -> CSR on the device in two steps: the number of path ints of the launch (fetch_total), then the dense arrays and
// their copies into the caller's memory (fetch_copy: relative path_offsets[0..n], status[0..n), total ints at paths_out)
static int fetch_total(bgr_aligner* a, uint64_t n, uint64_t* total_out) {
    HIP_TRY(hipSetDevice(a->device));
    if (a->deep.open) { const int src = settle_launch_sync(a); if (src != BGR_OK) return src; }
    else HIP_TRY(wait_stream(a));
    const uint64_t nb = (n + 4095) / 4096;
    HIP_TRY(a->csr_sums.ensure(nb * 4 + 64));
    HIP_TRY(a->csr_poffs.ensure((n + 1) * 8));
    HIP_TRY(a->csr_status.ensure(n));
    unsigned long long* d_total = reinterpret_cast<unsigned long long*>(static_cast<char*>(a->small.p) + 128);
    hipError_t e = bgr::launch_csr(static_cast<const uint2*>(a->results.p), static_cast<const int32_t*>(a->arena.p), (uint32_t)n,
                                   static_cast<uint32_t*>(a->csr_sums.p), d_total, nullptr, nullptr, nullptr, 0, 0, a->stream);
    if (e != hipSuccess) return fail(BGR_E_HIP, std::string("csr launch: ") + hipGetErrorString(e));
    uint64_t hs[17];  // cursor[0..1] @0, path-int total @128
    HIP_TRY(hipMemcpyAsync(hs, a->small.p, sizeof(hs), hipMemcpyDeviceToHost, a->stream));
    HIP_TRY(wait_stream(a));
    const uint32_t* cur = reinterpret_cast<const uint32_t*>(hs);
    if (cur[1]) return fail(BGR_E_INTERNAL, "path arena overflow (internal sizing error)");
    *total_out = hs[16];
    return BGR_OK;
}
static int fetch_copy(bgr_aligner* a, uint64_t n, uint64_t total, int32_t* paths_out, uint64_t* path_offsets, uint8_t* status, bool with_end = true) {
    unsigned long long* d_total = reinterpret_cast<unsigned long long*>(static_cast<char*>(a->small.p) + 128);
    HIP_TRY(a->csr_paths.ensure(total * 4 + 16));
    hipError_t e = bgr::launch_csr(static_cast<const uint2*>(a->results.p), static_cast<const int32_t*>(a->arena.p), (uint32_t)n,
                                   static_cast<uint32_t*>(a->csr_sums.p), d_total, static_cast<unsigned long long*>(a->csr_poffs.p),
                                   static_cast<int32_t*>(a->csr_paths.p), static_cast<uint8_t*>(a->csr_status.p), (uint32_t)std::min<uint64_t>(total, 0xFFFFFFFFull), 1, a->stream);
    if (e != hipSuccess) return fail(BGR_E_HIP, std::string("csr launch: ") + hipGetErrorString(e));
    // (with_end = false: entry n, the end of the last read, is left to the caller -- it is the first entry of the next piece)
    HIP_TRY(hipMemcpyAsync(path_offsets, a->csr_poffs.p, (n + (with_end ? 1 : 0)) * 8, hipMemcpyDeviceToHost, a->stream));
    HIP_TRY(hipMemcpyAsync(status, a->csr_status.p, n, hipMemcpyDeviceToHost, a->stream));
    if (total) HIP_TRY(hipMemcpyAsync(paths_out, a->csr_paths.p, total * 4, hipMemcpyDeviceToHost, a->stream));
    HIP_TRY(wait_stream(a));
    return BGR_OK;
}

int bgr_aligner_fetch(bgr_aligner* a, uint64_t n, int32_t* paths_out, uint64_t paths_cap, uint64_t* path_offsets, uint8_t* status) {
    if (!a || !path_offsets || !status || (paths_cap && !paths_out)) return fail(BGR_E_ARG, "bgr_aligner_fetch: null argument");
    if (n != a->last_n) return fail(BGR_E_ARG, "bgr_aligner_fetch: n_reads differs from the last bgr_align_device call");
    path_offsets[0] = 0;
    if (n == 0) return BGR_OK;
    uint64_t total = 0;
    int rc = fetch_total(a, n, &total);
    if (rc != BGR_OK) return rc;
    if (total > paths_cap) return fail(BGR_E_CAPACITY, "bgr_aligner_fetch: paths_out too small");
    return fetch_copy(a, n, total, paths_out, path_offsets, status);
}

// One piece of a batch up to (not including) the copies out: H2D of the characters and offsets, the mapping launch.
static int batch_piece_launch(bgr_aligner* a, const bgr_params* p, const char* reads, const uint64_t* read_offsets, uint64_t n) {
    HIP_TRY(hipSetDevice(a->device));
    const uint64_t base = read_offsets[0], total = read_offsets[n] - base;
    uint32_t max_len = 0;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t l = read_offsets[i + 1] - read_offsets[i];
        if (l > 0x7FFFFFFFull) return fail(BGR_E_ARG, "bgr_align_batch: read longer than 2^31 bases");
        max_len = std::max<uint32_t>(max_len, (uint32_t)l);
    }
    HIP_TRY(a->in_reads.ensure(total + 16));
    HIP_TRY(a->in_offs.ensure((n + 1) * 8));
    HIP_TRY(hipMemcpyAsync(a->in_reads.p, reads + base, total, hipMemcpyHostToDevice, a->stream));
    if (base == 0) {
        HIP_TRY(hipMemcpyAsync(a->in_offs.p, read_offsets, (n + 1) * 8, hipMemcpyHostToDevice, a->stream));
        HIP_TRY(hipStreamSynchronize(a->stream));
    } else {
        std::vector<uint64_t> rel(n + 1);
        for (uint64_t i = 0; i <= n; ++i) rel[i] = read_offsets[i] - base;
        HIP_TRY(hipMemcpyAsync(a->in_offs.p, rel.data(), (n + 1) * 8, hipMemcpyHostToDevice, a->stream));
        HIP_TRY(hipStreamSynchronize(a->stream));
    }
    return bgr_align_device(a, p, a->in_reads.p, a->in_offs.p, n, total, max_len);
}

// A large batch in pieces on several streams (this aligner's and its twins', one host thread each): the copies of one piece run under the
// kernels of the others, and -- with more streams than two -- the link never waits for a stream that is busy fetching its results (two
// streams kept it 2/3 busy: 243 Mreads/s per 5 M-read call; kOverlapStreams = 4: see DESIGN 9).  A piece's place in paths_out is known
// once the pieces in front of it have counted their path ints (fetch_total); results are those of one launch over the whole batch.
static const uint64_t kOverlapMinReads = 512 * 1024;
static const unsigned kOverlapMaxStreams = 4, kOverlapMaxPieces = 2 * kOverlapMaxStreams;
static int align_batch_overlapped(bgr_aligner* a, const bgr_params* p, const char* reads, const uint64_t* read_offsets, uint64_t n,
                                  int32_t* paths_out, uint64_t paths_cap, uint64_t* path_offsets, uint8_t* status) {
    unsigned n_streams = kOverlapMaxStreams;
    n_streams = (unsigned)std::min<int64_t>((int64_t)kOverlapMaxStreams, std::max<int64_t>(2, bgr::opt("overlap_streams")));  // (option overlap_streams: A/B measurements)
    const unsigned n_pieces = 2 * n_streams;
    bgr_aligner* al[kOverlapMaxStreams] = {a, nullptr, nullptr, nullptr};
    for (unsigned t = 1; t < n_streams; ++t) {  // the twins: a chain a -> twin -> twin's twin ..., created on first use
        bgr_aligner* prev = al[t - 1];
        if (!prev->twin) {
            int rc = bgr_aligner_create(a->graph, a->device, &prev->twin);
            if (rc != BGR_OK) return rc;
            prev->twin->is_twin = true;
        }
        bgr_aligner* tw = prev->twin;
        tw->cfg_waves = a->cfg_waves; tw->cfg_blocks_per_cu = a->cfg_blocks_per_cu; tw->cfg_lds_mphf = a->cfg_lds_mphf;
        tw->knob_frame_cap = a->knob_frame_cap; tw->knob_search = a->knob_search; tw->knob_debug_stop = a->knob_debug_stop;
        tw->knob_greedy_fast = a->knob_greedy_fast; tw->knob_exh_fast = a->knob_exh_fast; tw->knob_anc_fast = a->knob_anc_fast; tw->knob_memo_cap = a->knob_memo_cap; tw->knob_prepass = a->knob_prepass; tw->knob_no_events = a->knob_no_events;
        al[t] = tw;
    }
    uint64_t cut[kOverlapMaxPieces + 1];
    for (unsigned k = 0; k <= n_pieces; ++k) cut[k] = n * k / n_pieces;
    std::mutex mu;
    std::condition_variable cv;
    uint64_t totals[kOverlapMaxPieces];
    bool known[kOverlapMaxPieces];
    for (unsigned k = 0; k < n_pieces; ++k) { totals[k] = 0; known[k] = false; }
    int first_rc = BGR_OK;
    std::string first_err;
    auto give_up = [&](int rc, const char* msg) {
        std::lock_guard<std::mutex> l(mu);
        if (first_rc == BGR_OK) { first_rc = rc; first_err = msg ? msg : ""; }
        cv.notify_all();
    };
    auto work = [&](unsigned t) {
        for (unsigned k = t; k < n_pieces; k += n_streams) {
            { std::lock_guard<std::mutex> l(mu); if (first_rc != BGR_OK) return; }
            const uint64_t i0 = cut[k], cnt = cut[k + 1] - cut[k];
            uint64_t tot = 0;
            int rc = cnt ? batch_piece_launch(al[t], p, reads, read_offsets + i0, cnt) : BGR_OK;
            if (rc == BGR_OK && cnt) rc = fetch_total(al[t], cnt, &tot);
            if (rc != BGR_OK) { give_up(rc, bgr_last_error()); return; }
            uint64_t before = 0;
            {
                std::unique_lock<std::mutex> l(mu);
                totals[k] = tot;
                known[k] = true;
                cv.notify_all();
                cv.wait(l, [&] { if (first_rc != BGR_OK) return true; for (unsigned j = 0; j < k; ++j) if (!known[j]) return false; return true; });
                if (first_rc != BGR_OK) return;
                for (unsigned j = 0; j < k; ++j) before += totals[j];
            }
            if (before + tot > paths_cap) { give_up(BGR_E_CAPACITY, "bgr_align_batch: paths_out too small"); return; }
            if (cnt) {
                rc = fetch_copy(al[t], cnt, tot, paths_out ? paths_out + before : nullptr, path_offsets + i0, status + i0, false);
                if (rc != BGR_OK) { give_up(rc, bgr_last_error()); return; }
                if (before) for (uint64_t i = i0; i < i0 + cnt; ++i) path_offsets[i] += before;  // (entry i0 + cnt belongs to the next piece / the end)
            }
        }
    };
    std::vector<std::thread> helpers;
    try {
        for (unsigned t = 1; t < n_streams; ++t) helpers.emplace_back(work, t);
        work(0u);
    } catch (...) {  // (bad_alloc in a piece, or no thread to be had: the helpers that run must still be joined)
        give_up(BGR_E_INTERNAL, "bgr_align_batch: out of memory");
    }
    for (auto& h : helpers) h.join();
    if (first_rc != BGR_OK) return fail(first_rc, first_err);
    uint64_t all = 0;
    for (unsigned k = 0; k < n_pieces; ++k) all += totals[k];
    path_offsets[n] = all;
    a->last_n = 0;  // several launches: bgr_aligner_fetch has nothing to re-read
    return BGR_OK;
}

int bgr_align_batch(bgr_aligner* a, const bgr_params* p, const char* reads, const uint64_t* read_offsets, uint64_t n,
                    int32_t* paths_out, uint64_t paths_cap, uint64_t* path_offsets, uint8_t* status) {
    if (!a || !p || !path_offsets || (n && (!reads || !read_offsets || !status))) return fail(BGR_E_ARG, "bgr_align_batch: null argument");
    path_offsets[0] = 0;
    a->last_n = 0;
    if (n == 0) return BGR_OK;
    HIP_TRY(hipSetDevice(a->device));
    const uint64_t base = read_offsets[0], total = read_offsets[n] - base;
    // One launch addresses its path arena with 32 bits: a batch beyond that (~13 M reads of 150 bp) is mapped in pieces.
    // (BGR_KNOB_BATCH_SPLIT_LIMIT: tests lower the limit to walk this path with small inputs)
    const uint64_t lim = a->knob_split_limit ? std::max<uint64_t>(4096, a->knob_split_limit) : 0xFFFFFFFFull - (256ull << 20);
    if (n > 1 && (2 * (total + 16 * n) >= lim || n >= 0x7FFFFFFFull)) {
        uint64_t w = 0;
        for (uint64_t i0 = 0; i0 < n;) {
            uint64_t i1 = i0 + 1;  // longest piece below half the limit (at least one read)
            while (i1 < n && 2 * ((read_offsets[i1 + 1] - read_offsets[i0]) + 16 * (i1 + 1 - i0)) < lim / 2 && i1 + 1 - i0 < (1ull << 30)) ++i1;
            int rc = bgr_align_batch(a, p, reads, read_offsets + i0, i1 - i0, paths_out ? paths_out + w : nullptr, paths_cap - w, path_offsets + i0, status + i0);
            if (rc != BGR_OK) return rc;
            const uint64_t got = path_offsets[i1];
            for (uint64_t i = i0; i <= i1; ++i) path_offsets[i] += w;
            w += got;
            i0 = i1;
        }
        a->last_n = 0;  // several launches: bgr_aligner_fetch has nothing to re-read
        return BGR_OK;
    }
    // a large batch: in pieces on two streams, copies under kernels (BGR_KNOB_BATCH_OVERLAP = 1 turns it off)
    if (!a->is_twin && !a->knob_overlap && n >= kOverlapMinReads) return align_batch_overlapped(a, p, reads, read_offsets, n, paths_out, paths_cap, path_offsets, status);
    int rc = batch_piece_launch(a, p, reads, read_offsets, n);
    if (rc != BGR_OK) return rc;
    return bgr_aligner_fetch(a, n, paths_out, paths_cap, path_offsets, status);
}

// ---- asynchronous form over host buffers ---------------------------------------------------------------------------------------
int bgr_align_batch_begin(bgr_aligner* a, const bgr_params* p, const char* reads, const uint64_t* read_offsets, uint64_t n, bgr_ticket* ticket) {
    if (!a || !p || !ticket || (n && (!reads || !read_offsets))) return fail(BGR_E_ARG, "bgr_align_batch_begin: null argument");
    if (a->ticket_open) return fail(BGR_E_ARG, "bgr_align_batch_begin: this aligner still has a batch in flight (wait for its ticket first)");
    ticket->aligner = a;
    ticket->n_reads = n;
    ticket->serial = ++a->ticket_serial;
    a->last_n = 0;
    if (n == 0) { a->ticket_open = true; return BGR_OK; }
    HIP_TRY(hipSetDevice(a->device));
    const uint64_t base = read_offsets[0], total = read_offsets[n] - base;
    if (n >= 0x7FFFFFFFull || 2 * (total + 16 * n) >= 0xFFFFFFFFull - (256ull << 20))
        return fail(BGR_E_ARG, "bgr_align_batch_begin: batch too large for one launch (2 * (bases + 16 * reads) must stay below 2^32 - 2^28); bgr_align_batch cuts such a batch");
    uint32_t max_len = 0;
    a->ticket_offs.resize(n + 1);
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t l = read_offsets[i + 1] - read_offsets[i];
        if (l > 0x7FFFFFFFull) return fail(BGR_E_ARG, "bgr_align_batch_begin: read longer than 2^31 bases");
        max_len = std::max<uint32_t>(max_len, (uint32_t)l);
        a->ticket_offs[i] = read_offsets[i] - base;
    }
    a->ticket_offs[n] = total;
    HIP_TRY(a->in_reads.ensure(total + 16));
    HIP_TRY(a->in_offs.ensure((n + 1) * 8));
    HIP_TRY(hipMemcpyAsync(a->in_reads.p, reads + base, total, hipMemcpyHostToDevice, a->stream));
    HIP_TRY(hipMemcpyAsync(a->in_offs.p, a->ticket_offs.data(), (n + 1) * 8, hipMemcpyHostToDevice, a->stream));
    const int rc = bgr_align_device(a, p, a->in_reads.p, a->in_offs.p, n, total, max_len);
    if (rc != BGR_OK) return rc;
    a->ticket_open = true;
    return BGR_OK;
}

int bgr_align_batch_test(const bgr_ticket* t) {
    if (!t || !t->aligner) return fail(BGR_E_ARG, "bgr_align_batch_test: null ticket");
    bgr_aligner* a = t->aligner;
    if (!a->ticket_open || t->serial != a->ticket_serial) return fail(BGR_E_ARG, "bgr_align_batch_test: stale ticket");
    if (hipSetDevice(a->device) != hipSuccess) return fail(BGR_E_HIP, "hipSetDevice failed");
    const hipError_t e = hipStreamQuery(a->stream);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) return 0;
    return fail(BGR_E_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(e));
}

int bgr_align_batch_wait(const bgr_ticket* t, int32_t* paths_out, uint64_t paths_cap, uint64_t* path_offsets, uint8_t* status) {
    if (!t || !t->aligner || !path_offsets) return fail(BGR_E_ARG, "bgr_align_batch_wait: null argument");
    bgr_aligner* a = t->aligner;
    if (!a->ticket_open || t->serial != a->ticket_serial) return fail(BGR_E_ARG, "bgr_align_batch_wait: stale ticket (every begin is waited for once, in order)");
    path_offsets[0] = 0;
    if (t->n_reads == 0) { a->ticket_open = false; return BGR_OK; }
    const int rc = bgr_aligner_fetch(a, t->n_reads, paths_out, paths_cap, path_offsets, status);
    if (rc != BGR_E_CAPACITY) a->ticket_open = false;  // (too small a paths buffer: the results stay, wait again with a larger one)
    return rc;
}

int bgr_aligner_counters(bgr_aligner* a, uint64_t out[5]) {
    if (!a || !out) return fail(BGR_E_ARG, "bgr_aligner_counters: null argument");
    HIP_TRY(hipSetDevice(a->device));
    HIP_TRY(hipStreamSynchronize(a->stream));
    { const int src = settle_launch_sync(a); if (src != BGR_OK) return src; }  // (exhaustive mode: reads the last pass handed back are mapped -- and counted -- first)
    HIP_TRY(hipMemcpy(out, static_cast<char*>(a->small.p) + 64, 40, hipMemcpyDeviceToHost));
    for (bgr_aligner* tw = a->twin; tw; tw = tw->twin) {  // the pieces of overlapped batches its other streams mapped
        uint64_t t[5];
        HIP_TRY(hipStreamSynchronize(tw->stream));
        { const int src = settle_launch_sync(tw); if (src != BGR_OK) return src; }
        HIP_TRY(hipMemcpy(t, static_cast<char*>(tw->small.p) + 64, 40, hipMemcpyDeviceToHost));
        for (int i = 0; i < 5; ++i) out[i] += t[i];
    }
    return BGR_OK;
}

int bgr_aligner_reset_counters(bgr_aligner* a) {
    if (!a) return fail(BGR_E_ARG, "bgr_aligner_reset_counters: null aligner");
    HIP_TRY(hipSetDevice(a->device));
    // (on the aligner's own stream: a fill on the null stream is not ordered with a non-blocking stream's kernels)
    HIP_TRY(hipMemsetAsync(static_cast<char*>(a->small.p) + 64, 0, 40, a->stream));
    HIP_TRY(hipStreamSynchronize(a->stream));
    for (bgr_aligner* tw = a->twin; tw; tw = tw->twin) {
        HIP_TRY(hipMemsetAsync(static_cast<char*>(tw->small.p) + 64, 0, 40, tw->stream));
        HIP_TRY(hipStreamSynchronize(tw->stream));
    }
    return BGR_OK;
}

int bgr_aligner_kernel_time(bgr_aligner* a, uint64_t* launches, double* total_ms) {
    if (!a) return fail(BGR_E_ARG, "bgr_aligner_kernel_time: null aligner");
    HIP_TRY(hipSetDevice(a->device));
    int rc = drain_timers(a);
    if (rc != BGR_OK) return rc;
    if (launches) *launches = a->t_launches;
    if (total_ms) *total_ms = a->t_ms;
    for (bgr_aligner* tw = a->twin; tw; tw = tw->twin) {  // the pieces of overlapped batches its other streams mapped
        rc = drain_timers(tw);
        if (rc != BGR_OK) return rc;
        if (launches) *launches += tw->t_launches;
        if (total_ms) *total_ms += tw->t_ms;
    }
    return BGR_OK;
}

int bgr_aligner_kernel_times(bgr_aligner* a, uint64_t* launches, double slot_ms[8], const char* slot_names[8]) {
    if (!a || !slot_ms) return fail(BGR_E_ARG, "bgr_aligner_kernel_times: null argument");
    HIP_TRY(hipSetDevice(a->device));
    int rc = drain_timers(a);
    if (rc != BGR_OK) return rc;
    if (launches) *launches = a->t_launches;
    for (bgr_aligner* tw = a->twin; tw; tw = tw->twin) {
        rc = drain_timers(tw);
        if (rc != BGR_OK) return rc;
        if (launches) *launches += tw->t_launches;
    }
    for (int j = 0; j < kTimerSlots; ++j) {
        slot_ms[j] = a->t_slot_ms[j];
        const char* name = a->t_slot_name[j];
        for (bgr_aligner* tw = a->twin; tw; tw = tw->twin) {  // (all streams run the same launch sequence: slot j is the same kernel)
            if (tw->t_slot_ms[j] <= 0) continue;
            slot_ms[j] += tw->t_slot_ms[j];
            if (!name) name = tw->t_slot_name[j];
        }
        if (slot_names) slot_names[j] = slot_ms[j] > 0 ? name : nullptr;
    }
    return BGR_OK;
}

int bgr_aligner_reset_kernel_time(bgr_aligner* a) {
    if (!a) return fail(BGR_E_ARG, "bgr_aligner_reset_kernel_time: null aligner");
    HIP_TRY(hipSetDevice(a->device));
    int rc = drain_timers(a);
    if (rc != BGR_OK) return rc;
    a->t_launches = 0;
    a->t_ms = 0;
    for (double& x : a->t_slot_ms) x = 0;
    if (a->twin) return bgr_aligner_reset_kernel_time(a->twin);
    return BGR_OK;
}

int bgr_aligner_launch_info(bgr_aligner* a, uint32_t out[4]) {
    if (!a || !out) return fail(BGR_E_ARG, "bgr_aligner_launch_info: null argument");
    memcpy(out, a->last_launch, sizeof(a->last_launch));
    return BGR_OK;
}

#ifdef BGR_PHASE_TIMING
// diagnostic builds only (not part of include/bgreat_gpu.h): the last launch's per-wave time stamps, four u64 per wave (100 MHz ticks)
int bgr_debug_wave_times(bgr_aligner* a, uint64_t* out, uint64_t cap_waves, uint64_t* n_waves) {
    if (!a || !n_waves) return BGR_E_ARG;
    HIP_TRY(hipSetDevice(a->device));
    HIP_TRY(hipStreamSynchronize(a->stream));
    *n_waves = a->wave_times_n;
    const uint64_t n = std::min(cap_waves, a->wave_times_n);
    if (n && out) HIP_TRY(hipMemcpy(out, a->wave_times.p, n * 32, hipMemcpyDeviceToHost));
    return BGR_OK;
}
#endif

int bgr_aligner_last_pass_runs(const bgr_aligner* a, uint32_t* runs, uint32_t* memo_cap) {
    if (!a) return fail(BGR_E_ARG, "bgr_aligner_last_pass_runs: null aligner");
    if (runs) *runs = a->deep.runs;
    if (memo_cap) *memo_cap = a->deep.memo_cap;
    return BGR_OK;
}

int bgr_aligner_pass_counts(bgr_aligner* a, uint32_t out[4]) {
    if (!a || !out) return fail(BGR_E_ARG, "bgr_aligner_pass_counts: null argument");
    HIP_TRY(hipSetDevice(a->device));
    HIP_TRY(hipStreamSynchronize(a->stream));
    uint32_t cur[16];
    HIP_TRY(hipMemcpy(cur, a->small.p, sizeof(cur), hipMemcpyDeviceToHost));
    out[0] = cur[2]; out[1] = cur[3]; out[2] = cur[5]; out[3] = cur[8];
    return BGR_OK;
}

int bgr_device_local_cpus(int device, char* cpulist_out, uint64_t cap) {
    if (!cpulist_out || cap < 2) return fail(BGR_E_ARG, "bgr_device_local_cpus: null argument");
    char bus[64];
    HIP_TRY(hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device));
    for (char* c = bus; *c; ++c) *c = (char)tolower(*c);
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/local_cpulist";
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return fail(BGR_E_IO, "bgr_device_local_cpus: cannot read " + path);
    const size_t got = fread(cpulist_out, 1, (size_t)cap - 1, f);
    fclose(f);
    cpulist_out[got] = 0;
    for (size_t i = 0; i < got; ++i) if (cpulist_out[i] == '\n') cpulist_out[i] = 0;
    if (!cpulist_out[0]) return fail(BGR_E_IO, "bgr_device_local_cpus: empty cpulist");
    return BGR_OK;
}

int bgr_device_alloc(int device, uint64_t bytes, void** out) {
    if (!out) return fail(BGR_E_ARG, "bgr_device_alloc: null argument");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMalloc(out, bytes ? bytes : 1));
    return BGR_OK;
}
int bgr_device_free(int device, void* p) {
    if (!p) return BGR_OK;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(p));
    return BGR_OK;
}
int bgr_device_upload(int device, void* dst_device, const void* src_host, uint64_t bytes) {
    if (bytes && (!dst_device || !src_host)) return fail(BGR_E_ARG, "bgr_device_upload: null argument");
    HIP_TRY(hipSetDevice(device));
    if (bytes) { HIP_TRY(hipMemcpy(dst_device, src_host, bytes, hipMemcpyHostToDevice)); HIP_TRY(hipStreamSynchronize(nullptr)); }  // (the callers' kernels run on non-blocking streams)
    return BGR_OK;
}
int bgr_device_download(int device, void* dst_host, const void* src_device, uint64_t bytes) {
    if (bytes && (!dst_host || !src_device)) return fail(BGR_E_ARG, "bgr_device_download: null argument");
    HIP_TRY(hipSetDevice(device));
    if (bytes) HIP_TRY(hipMemcpy(dst_host, src_device, bytes, hipMemcpyDeviceToHost));
    return BGR_OK;
}

// Page-locked host memory.  hipHostMalloc pins 4 KB pages at ~4.4 GB/s on this platform, from any number of threads; an anonymous
// mapping with transparent huge pages asked for, touched once per 2 MB and then registered pins at ~26 GB/s (tools/ubench/pin_alloc.hip) --
// bgr_align_all takes ~0.7 GB of staging sets while its pipeline ramps up, 0.17 s of a 0.6 s run of 100 M reads.  Buffers of 2 MB and
// more go that way (hipHostMalloc when anything in it fails); the registry tells bgr_host_free how a pointer was obtained.
// (never destroyed: the pipeline's cache of staging sets frees its buffers from a static destructor of its own, in whatever order)
static std::mutex& g_host_m = *new std::mutex;
static std::map<void*, uint64_t>& g_host_mapped = *new std::map<void*, uint64_t>;  // registered anonymous mappings: pointer -> mapped bytes

int bgr_host_alloc(uint64_t bytes, void** out) {
    if (!out) return fail(BGR_E_ARG, "bgr_host_alloc: null argument");
    void* p = nullptr;
    if (bytes >= (2ull << 20) && bgr::opt("huge_pinned")) {
        const uint64_t len = (bytes + (2ull << 20) - 1) & ~((2ull << 20) - 1);
        void* m = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m != MAP_FAILED) {
            (void)madvise(m, len, MADV_HUGEPAGE);
            for (uint64_t o = 0; o < len; o += 2ull << 20) static_cast<volatile char*>(m)[o] = 0;  // fault the (huge) pages in before they are pinned
            if (hipHostRegister(m, len, hipHostRegisterDefault) == hipSuccess) {
                std::lock_guard<std::mutex> l(g_host_m);
                g_host_mapped[m] = len;
                *out = m;
                return BGR_OK;
            }
            (void)hipGetLastError();
            munmap(m, len);
        }
    }
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) return fail(BGR_E_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    *out = p;
    return BGR_OK;
}

int bgr_host_free(void* p) {
    if (!p) return BGR_OK;
    uint64_t len = 0;
    {
        std::lock_guard<std::mutex> l(g_host_m);
        auto it = g_host_mapped.find(p);
        if (it != g_host_mapped.end()) { len = it->second; g_host_mapped.erase(it); }
    }
    if (len) {
        hipError_t e = hipHostUnregister(p);
        munmap(p, len);
        if (e != hipSuccess) return fail(BGR_E_HIP, std::string("hipHostUnregister: ") + hipGetErrorString(e));
        return BGR_OK;
    }
    hipError_t e = hipHostFree(p);
    if (e != hipSuccess) return fail(BGR_E_HIP, std::string("hipHostFree: ") + hipGetErrorString(e));
    return BGR_OK;
}

// ---- read files / output records (host) ---------------------------------------------------------------
struct bgr_readset { bgr::ReadSet rs; };

int bgr_readset_load(const char* path, int fastq, uint32_t k, bgr_readset** out) {
    return bgr_readset_load_parallel(path, fastq, k, 1, 0, out);
}

int bgr_readset_load_parallel(const char* path, int fastq, uint32_t k, uint32_t threads, uint64_t chunk_bytes, bgr_readset** out) {
    if (!path || !out) return fail(BGR_E_ARG, "bgr_readset_load: null argument");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(BGR_E_IO, std::string("cannot open read file ") + path);
    std::vector<char> buf;
    char tmp[1 << 16];
    size_t got;
    while ((got = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + got);
    fclose(f);
    bgr_readset* r = new bgr_readset();
    r->rs.clear();
    bgr::parse_reads_parallel(buf.data(), buf.size(), fastq != 0, k, threads ? threads : 1, chunk_bytes ? chunk_bytes : (8u << 20), r->rs);
    *out = r;
    return BGR_OK;
}
uint64_t bgr_readset_count(const bgr_readset* rs) { return rs ? rs->rs.count() : 0; }
int bgr_readset_view(const bgr_readset* rs, const char** reads, const uint64_t** read_offsets, const char** headers, const uint64_t** header_offsets) {
    if (!rs) return fail(BGR_E_ARG, "bgr_readset_view: null readset");
    if (reads) *reads = rs->rs.reads.data();
    if (read_offsets) *read_offsets = rs->rs.read_offs.data();
    if (headers) *headers = rs->rs.headers.data();
    if (header_offsets) *header_offsets = rs->rs.header_offs.data();
    return BGR_OK;
}
void bgr_readset_destroy(bgr_readset* rs) { delete rs; }

int bgr_write_records(void* paths_file, void* notaligned_file, uint64_t n, const char* headers, const uint64_t* hoffs,
                      const char* reads, const uint64_t* roffs, const int32_t* paths, const uint64_t* poffs) {
    if (!paths_file || !notaligned_file || (n && (!headers || !hoffs || !reads || !roffs || !poffs))) return fail(BGR_E_ARG, "bgr_write_records: null argument");
    FILE* pf = static_cast<FILE*>(paths_file);
    FILE* nf = static_cast<FILE*>(notaligned_file);
    std::string pbuf, nbuf;
    pbuf.reserve(1 << 20);
    nbuf.reserve(1 << 20);
    char num[16];
    for (uint64_t i = 0; i < n; ++i) {
        const char* h = headers + hoffs[i];
        size_t hn = hoffs[i + 1] - hoffs[i];
        if (poffs[i + 1] > poffs[i]) {  // alignerGreedy.cpp:406-411 + printPath aligner.cpp:600-609
            pbuf.append(h, hn);
            pbuf.push_back('\n');
            for (uint64_t j = poffs[i]; j < poffs[i + 1]; ++j) {
                int32_t v = paths[j];
                uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
                int len = 0;
                do { num[len++] = (char)('0' + u % 10); u /= 10; } while (u);
                if (v < 0) pbuf.push_back('-');
                while (len) pbuf.push_back(num[--len]);
                pbuf.push_back('.');
            }
            pbuf.push_back('\n');
            if (pbuf.size() > (1 << 20) - 4096) { if (fwrite(pbuf.data(), 1, pbuf.size(), pf) != pbuf.size()) return fail(BGR_E_IO, "write to paths file failed"); pbuf.clear(); }
        } else {  // alignerGreedy.cpp:421-427
            nbuf.append(h, hn);
            nbuf.push_back('\n');
            nbuf.append(reads + roffs[i], roffs[i + 1] - roffs[i]);
            nbuf.push_back('\n');
            if (nbuf.size() > (1 << 20) - 4096) { if (fwrite(nbuf.data(), 1, nbuf.size(), nf) != nbuf.size()) return fail(BGR_E_IO, "write to notAligned file failed"); nbuf.clear(); }
        }
    }
    if (!pbuf.empty() && fwrite(pbuf.data(), 1, pbuf.size(), pf) != pbuf.size()) return fail(BGR_E_IO, "write to paths file failed");
    if (!nbuf.empty() && fwrite(nbuf.data(), 1, nbuf.size(), nf) != nbuf.size()) return fail(BGR_E_IO, "write to notAligned file failed");
    return BGR_OK;
}

}  // extern "C"
