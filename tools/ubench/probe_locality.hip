// tools/ubench/probe_locality.hip -- what a wave pays for 64 dword probes into a table that does not fit L2, as a function of how many
// ADJACENT lanes share a 64-byte line (1 = every lane its own random line; 10 = a run of ten lanes in one line, which is what a
// minimizer-blocked filter gives the scan of consecutive read positions).  Prints probes per second per table size.
//   hipcc --offload-arch=gfx950 -O3 probe_locality.hip -o probe_locality && ./probe_locality
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int G>
__global__ void __launch_bounds__(256) probe(const uint32_t* tab, uint32_t line_mask, uint32_t rounds, uint32_t* out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t grp = (t / G) * 0x9E3779B1u + r * 0x85EBCA6Bu;   // one random line per run of G adjacent lanes
        const uint32_t line = mix(grp) & line_mask;
        const uint32_t word = mix(t * 31u + r) & 15u;
        acc ^= tab[(size_t)line * 16 + word];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    uint32_t* out; hipMalloc(&out, 64);
    for (size_t mb : {2, 16, 64, 256, 2048}) {
        const size_t bytes = mb << 20;
        uint32_t* tab; if (hipMalloc(&tab, bytes) != hipSuccess) return 1;
        hipMemset(tab, 1, bytes);
        const uint32_t line_mask = (uint32_t)(bytes / 64 - 1);
        const uint32_t blocks = 256 * 32, rounds = 256;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto run = [&](auto kern, const char* name) {
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, tab, line_mask, 8u, out);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, tab, line_mask, rounds, out);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("table %5zu MB  lanes per line %-3s  %.1f G probes/s  (%.3f ms)\n", mb, name, (double)blocks * 256 * rounds / ms / 1e6, ms);
        };
        run(probe<1>, "1"); run(probe<4>, "4"); run(probe<10>, "10"); run(probe<16>, "16"); run(probe<64>, "64");
        hipFree(tab);
    }
    return 0;
}
