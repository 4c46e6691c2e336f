// options.h -- process-wide tuning and diagnostic options of the library.  Nothing in the library reads the environment: a host sets an
// option through bgr_set_option() (the CLI: --set name=value), before the objects that look at it are created.  Table: INTEGRATION.md 5.
#ifndef BGREAT_AMD_OPTIONS_H
#define BGREAT_AMD_OPTIONS_H

#include <stdint.h>
#include <string.h>

#include <atomic>

namespace bgr {

struct Option {
    const char* name;
    std::atomic<int64_t> value;
    int64_t lo, hi;
    const char* what;
};

// (a function-local table: one instance per process, whichever translation unit asks first)
inline Option* option_table(size_t* n) {
    static Option t[] = {
        {"timing", {0}, 0, 1, "per-stage timing lines on stderr (pipeline, text calls, index build)"},
        {"blocking_sync", {0}, 0, 1, "aligners wait for their stream with a blocking event (the thread sleeps) instead of spinning"},
        {"numa", {1}, 0, 1, "bgr_align_all pins its threads to the NUMA node of the devices they feed when all share one"},
        {"workers_per_device", {0}, 0, 16, "bgr_align_all: stream workers per device (0 = default: 2)"},
        {"extra_sets", {-1}, -1, 32, "bgr_align_all: staging sets beyond one per worker (-1 = default)"},
        {"fastq_gather", {1}, 0, 1, "FASTQ pieces cross PCIe without their '+' and quality lines"},
        {"overlap_streams", {4}, 2, 4, "bgr_align_batch of >= 512 k reads: internal streams"},
        {"huge_pinned", {1}, 0, 1, "bgr_host_alloc: buffers of 2 MB and more from huge-page mappings registered with the runtime"},
        {"exh_filter", {1}, 0, 1, "exhaustive mode probes large key tables through the minimizer filter"},
        {"build_filter", {-1}, -1, 2, "index build: filter in front of the key table: -1 by table size, 0 none, 1 one hash, 2 minimizer-blocked (forced onto small graphs by tests)"},
        {"poison_device_buffers", {0}, 0, 1, "diagnostic: every new device buffer is filled with a pattern (a kernel that reads what nothing wrote shows in any run)"},
        {"test.bases_cap", {0}, 0, INT64_MAX, "test hook: bases per batch of bgr_align_all (walks the cut of large batches with small inputs)"},
        {"test.lanes_on_one_device", {0}, 0, 1, "test hook: every lane of a split run / every device of --gpus N is device 0 (a one-GPU box walks the N-device code)"},
        {"test.text_epoch", {0}, 0, 0x3FFFFF, "test hook: the epoch the text form's chains start from when their state is (re)allocated (walks the 22-bit wrap with a few pieces)"},
    };
    *n = sizeof(t) / sizeof(t[0]);
    return t;
}
inline Option* find_option(const char* name) {
    size_t n;
    Option* t = option_table(&n);
    for (size_t i = 0; i < n; ++i)
        if (strcmp(t[i].name, name) == 0) return &t[i];
    return nullptr;
}
inline int64_t opt(const char* name) {
    Option* o = find_option(name);
    return o ? o->value.load(std::memory_order_relaxed) : 0;
}

}  // namespace bgr

#endif
