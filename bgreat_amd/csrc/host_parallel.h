// host_parallel.h -- small helpers of the host-side index build: contiguous-range thread fan-out, a parallel
// sort, and the BGREAT_TIMING phase timer.
#ifndef BGREAT_AMD_HOST_PARALLEL_H
#define BGREAT_AMD_HOST_PARALLEL_H

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include "options.h"

namespace bgr {

// fn(begin, end, tid) over [0, n) cut into `T` contiguous ranges, one thread each (inline when T == 1).
template <class Fn>
inline void parallel_ranges(unsigned T, uint64_t n, Fn fn) {
    if (T <= 1 || n < 2) { fn((uint64_t)0, n, 0u); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t) {
        uint64_t b = n * t / T, e = n * (t + 1) / T;
        th.emplace_back([=] { fn(b, e, t); });
    }
    for (auto& x : th) x.join();
}

// Sorts v with T threads: T sorted runs, then rounds of pairwise merges (each merge on its own thread).
inline void parallel_sort(std::vector<uint64_t>& v, unsigned T) {
    if (T <= 1 || v.size() < (1u << 16)) { std::sort(v.begin(), v.end()); return; }
    std::vector<uint64_t> cut(T + 1);
    for (unsigned t = 0; t <= T; ++t) cut[t] = v.size() * t / T;
    parallel_ranges(T, T, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t t = b; t < e; ++t) std::sort(v.begin() + cut[t], v.begin() + cut[t + 1]);
    });
    while (cut.size() > 2) {
        size_t pairs = (cut.size() - 1) / 2;
        std::vector<std::thread> th;
        for (size_t p = 0; p < pairs; ++p)
            th.emplace_back([&, p] { std::inplace_merge(v.begin() + cut[2 * p], v.begin() + cut[2 * p + 1], v.begin() + cut[2 * p + 2]); });
        for (auto& x : th) x.join();
        std::vector<uint64_t> nc;
        for (size_t i = 0; i < cut.size(); i += 2) nc.push_back(cut[i]);
        if ((cut.size() - 1) % 2) nc.push_back(cut.back());
        cut.swap(nc);
    }
}

struct PhaseTimer {  // BGREAT_TIMING=1: per-phase wall time of the index build on stderr
    bool on = opt("timing") != 0;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[build] %-10s %.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    }
};

}  // namespace bgr
#endif
