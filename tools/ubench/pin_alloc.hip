// tools/ubench/pin_alloc.hip -- page-locked allocations (hipHostMalloc) of the size of a staging set, one after the other and from several threads at once:
// does the pinning parallelise?  (bgr_align_all allocates its staging sets while the pipeline ramps up.)
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipFree(nullptr);
    const size_t bytes = 90ull << 20;
    const int n = 8;
    for (int T : {1, 2, 4, 8}) {
        std::vector<void*> p(n, nullptr);
        const double t0 = now();
        std::vector<std::thread> ts;
        for (int t = 0; t < T; ++t) ts.emplace_back([&, t]() { for (int i = t; i < n; i += T) if (hipHostMalloc(&p[i], bytes, hipHostMallocDefault) != hipSuccess) abort(); });
        for (auto& t : ts) t.join();
        const double t1 = now();
        for (void* q : p) hipHostFree(q);
        printf("%d thread(s): %d x %zu MB page-locked in %.3f s (%.2f GB/s), freed in %.3f s\n", T, n, bytes >> 20, t1 - t0, n * bytes / (t1 - t0) / 1e9, now() - t1);
    }
    // the same memory from malloc + hipHostRegister
    {
        std::vector<void*> p(n);
        const double t0 = now();
        for (int i = 0; i < n; ++i) { p[i] = aligned_alloc(4096, bytes); if (hipHostRegister(p[i], bytes, hipHostRegisterDefault) != hipSuccess) { printf("hipHostRegister failed\n"); return 0; } }
        const double t1 = now();
        printf("malloc + hipHostRegister: %.3f s (%.2f GB/s)\n", t1 - t0, n * bytes / (t1 - t0) / 1e9);
        for (void* q : p) { hipHostUnregister(q); free(q); }
    }
    // anonymous mapping with transparent huge pages asked for, then registered
    for (int huge : {0, 1}) {
        std::vector<void*> p(n);
        const double t0 = now();
        bool ok = true;
        for (int i = 0; i < n && ok; ++i) {
            p[i] = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | (huge ? 0 : MAP_POPULATE), -1, 0);
            if (p[i] == MAP_FAILED) { ok = false; break; }
            if (huge) { madvise(p[i], bytes, MADV_HUGEPAGE); for (size_t o = 0; o < bytes; o += 2u << 20) static_cast<volatile char*>(p[i])[o] = 0; }
            if (hipHostRegister(p[i], bytes, hipHostRegisterDefault) != hipSuccess) ok = false;
        }
        const double t1 = now();
        printf("mmap%s + hipHostRegister: %s %.3f s (%.2f GB/s)\n", huge ? " + MADV_HUGEPAGE + touch per 2 MB" : " + MAP_POPULATE", ok ? "" : "FAILED", t1 - t0, n * bytes / (t1 - t0) / 1e9);
        if (ok) for (void* q : p) { hipHostUnregister(q); munmap(q, bytes); }
    }
    FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    if (f) { char b[128] = {0}; if (fgets(b, 127, f)) printf("THP: %s", b); fclose(f); }
    return 0;
}
