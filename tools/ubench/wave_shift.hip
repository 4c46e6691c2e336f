// tools/ubench/wave_shift.hip -- DPP whole-wave shifts on gfx950 (wave_shl:1 / wave_shr:1): what they do at the ends of the wave and what
// they cost next to a row-local DPP move and a ds_bpermute.  The sliding-window minimum of the minimizer filter is built from them.
//   hipcc --offload-arch=gfx950 -O3 wave_shift.hip -o wave_shift && ./wave_shift
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 2048
#define R4(x) x x x x
#define BODY(ins, mod) \
    asm volatile(R4(ins " %0, %0 " mod "\n" ins " %1, %1 " mod "\n" ins " %2, %2 " mod "\n" ins " %3, %3 " mod "\n" \
                    ins " %4, %4 " mod "\n" ins " %5, %5 " mod "\n" ins " %6, %6 " mod "\n" ins " %7, %7 " mod "\n") \
                 : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]))
#define BODY3(ins, mod) \
    asm volatile(R4(ins " %0, %0, %0 " mod "\n" ins " %1, %1, %1 " mod "\n" ins " %2, %2, %2 " mod "\n" ins " %3, %3, %3 " mod "\n" \
                    ins " %4, %4, %4 " mod "\n" ins " %5, %5, %5 " mod "\n" ins " %6, %6, %6 " mod "\n" ins " %7, %7, %7 " mod "\n") \
                 : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]))

template <int KIND>
__global__ void __launch_bounds__(256) rate(uint32_t* out) {
    uint32_t b[8];
    for (int i = 0; i < 8; ++i) b[i] = threadIdx.x * 7 + i;
    const uint32_t addr = ((threadIdx.x + 1) & 63) * 4;
    for (int r = 0; r < REP; ++r) {
        if (KIND == 0) BODY("v_mov_b32_dpp", "wave_shl:1 row_mask:0xf bank_mask:0xf");
        if (KIND == 1) BODY3("v_min_u32_dpp", "wave_shl:1 row_mask:0xf bank_mask:0xf");
        if (KIND == 2) BODY("v_mov_b32_dpp", "row_shl:1 row_mask:0xf bank_mask:0xf");
        if (KIND == 3) BODY3("v_min_u32_dpp", "row_shl:1 row_mask:0xf bank_mask:0xf");
        if (KIND == 4) for (int q = 0; q < 4; ++q) for (int i = 0; i < 8; ++i) b[i] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)addr, (int)b[i]);
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= b[i];
    if (s == 0x12345) out[0] = s;
}
__global__ void semantics(uint32_t* p) {
    const uint32_t x = 100 + threadIdx.x;
    p[threadIdx.x] = (uint32_t)__builtin_amdgcn_update_dpp((int)7777, (int)x, 0x130, 0xF, 0xF, false);        // wave_shl:1, old = 7777
    p[64 + threadIdx.x] = (uint32_t)__builtin_amdgcn_update_dpp((int)7777, (int)x, 0x138, 0xF, 0xF, false);   // wave_shr:1
    p[128 + threadIdx.x] = (uint32_t)__builtin_amdgcn_update_dpp((int)7777, (int)x, 0x130, 0xF, 0xF, true);   // wave_shl:1, bound_ctrl
}
int main() {
    uint32_t* d; if (hipMalloc(&d, 4096) != hipSuccess) return 1;
    hipLaunchKernelGGL(semantics, dim3(1), dim3(64), 0, 0, d);
    uint32_t h[192];
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    const char* names[3] = {"wave_shl:1 (old=7777)", "wave_shr:1 (old=7777)", "wave_shl:1 bound_ctrl"};
    for (int k = 0; k < 3; ++k) {
        printf("%s: lane0=%u lane1=%u lane15=%u lane16=%u lane31=%u lane32=%u lane62=%u lane63=%u\n", names[k], h[64 * k], h[64 * k + 1], h[64 * k + 15], h[64 * k + 16],
               h[64 * k + 31], h[64 * k + 32], h[64 * k + 62], h[64 * k + 63]);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* kn[5] = {"v_mov_b32_dpp wave_shl:1", "v_min_u32_dpp wave_shl:1", "v_mov_b32_dpp row_shl:1", "v_min_u32_dpp row_shl:1", "ds_bpermute_b32"};
    for (int waves_per_simd : {1, 2, 4, 8}) {
        for (int kind = 0; kind < 5; ++kind) {
            const uint32_t blocks = 256 * waves_per_simd;   // 4 waves per block = one per SIMD of a CU
            auto launch = [&]() {
                switch (kind) {
                    case 0: hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(256), 0, 0, d); break;
                    case 1: hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(256), 0, 0, d); break;
                    case 2: hipLaunchKernelGGL(rate<2>, dim3(blocks), dim3(256), 0, 0, d); break;
                    case 3: hipLaunchKernelGGL(rate<3>, dim3(blocks), dim3(256), 0, 0, d); break;
                    default: hipLaunchKernelGGL(rate<4>, dim3(blocks), dim3(256), 0, 0, d); break;
                }
            };
            launch();
            hipEventRecord(e0, 0);
            launch();
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double inst_per_simd = (double)REP * 32 * waves_per_simd;
            printf("%d waves/SIMD  %-26s %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", waves_per_simd, kn[kind], ms * 1e-3 * 2.4e9 / inst_per_simd);
        }
    }
    return 0;
}
