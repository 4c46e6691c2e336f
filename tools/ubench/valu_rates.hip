// tools/ubench/valu_rates.hip -- issue cost of the integer VALU ops the mapping kernels lean on (gfx950).
// One wave per SIMD would under-fill the pipe, so 8 waves per SIMD run the same dependent-free streams;
// prints cycles per wave-instruction per SIMD.   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 4096
template <int OP>
__global__ void k(uint64_t* out, uint64_t seed) {
    uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    uint32_t b0 = (uint32_t)a0, b1 = (uint32_t)a1, b2 = (uint32_t)a2, b3 = (uint32_t)a3;
    uint32_t s = (uint32_t)(seed & 31) + 1;
    for (int i = 0; i < REP; ++i) {
        if (OP == 0) { b0 = b0 + s; b1 = b1 + s; b2 = b2 + s; b3 = b3 + s; }                          // v_add_u32
        if (OP == 1) { a0 = a0 << s; a1 = a1 << s; a2 = a2 << s; a3 = a3 << s; a0 |= 1; a1 |= 1; a2 |= 1; a3 |= 1; }  // v_lshlrev_b64 + or
        if (OP == 2) { b0 = b0 << s | 1; b1 = b1 << s | 1; b2 = b2 << s | 1; b3 = b3 << s | 1; }      // v_lshl_or_b32
        if (OP == 3) { b0 = b0 * 0x9E3779B1u + 1; b1 = b1 * 0x9E3779B1u + 1; b2 = b2 * 0x9E3779B1u + 1; b3 = b3 * 0x9E3779B1u + 1; }  // v_mul_lo_u32
        if (OP == 4) { b0 = __umulhi(b0, 0x9E3779B1u) + s; b1 = __umulhi(b1, 0x9E3779B1u) + s; b2 = __umulhi(b2, 0x9E3779B1u) + s; b3 = __umulhi(b3, 0x9E3779B1u) + s; }
        if (OP == 5) { b0 = __popc(b0) + s * b0; b1 = __popc(b1) + b1; b2 = __popc(b2) + b2; b3 = __popc(b3) + b3; }  // v_bcnt
        if (OP == 6) { a0 = a0 * 0xBF58476D1CE4E5B9ULL + 1; a1 = a1 * 0xBF58476D1CE4E5B9ULL + 1; a2 = a2 * 0xBF58476D1CE4E5B9ULL + 1; a3 = a3 * 0xBF58476D1CE4E5B9ULL + 1; }  // 64x64 mul
        if (OP == 7) { a0 = a0 + a1; a1 = a1 + a2; a2 = a2 + a3; a3 = a3 + a0; }                      // 64-bit add
        if (OP == 8) { b0 = __builtin_amdgcn_alignbit(b0, b1, s); b1 = __builtin_amdgcn_alignbit(b1, b2, s); b2 = __builtin_amdgcn_alignbit(b2, b3, s); b3 = __builtin_amdgcn_alignbit(b3, b0, s); }
        if (OP == 9) { a0 = a0 >> s | 1ULL << 63; a1 = a1 >> s | 1ULL << 63; a2 = a2 >> s | 1ULL << 63; a3 = a3 >> s | 1ULL << 63; }  // v_lshrrev_b64
        if (OP == 10) { b0 = (b0 & 0xFFFF) * 48 + b0; b1 = (b1 & 0xFFFF) * 48 + b1; b2 = (b2 & 0xFFFF) * 48 + b2; b3 = (b3 & 0xFFFF) * 48 + b3; }  // v_mad_u32_u24
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ b0 ^ b1 ^ b2 ^ b3;
}
template <int OP>
void run(const char* name, int ops_per_iter, uint64_t* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8, threads = 256;   // 8 waves per SIMD on every CU
    k<OP><<<blocks, threads>>>(d, 12345);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<OP><<<blocks, threads>>>(d, 12345 + r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = 5.0 * blocks * (threads / 64) * (double)REP * ops_per_iter;   // total wave-instructions
    double per_simd = wave_instr / 1024.0;                                            // 1024 SIMDs
    double ns = ms * 1e6 / per_simd;
    printf("%-28s %.3f ns per wave-instruction per SIMD  (= %.2f cycles at 2.1 GHz, %.2f at 2.4)\n", name, ns, ns * 2.1, ns * 2.4);
}
int main() {
    uint64_t* d; hipMalloc(&d, 256 * 8 * 256 * 8);
    run<0>("v_add_u32", 4, d);
    run<2>("v_lshl_or_b32", 4, d);
    run<8>("v_alignbit_b32", 4, d);
    run<10>("v_and + v_mad_u32_u24", 8, d);
    run<5>("v_bcnt_u32 (+mul/add)", 9, d);
    run<3>("v_mul_lo_u32 + add", 8, d);
    run<4>("v_mul_hi_u32 + add", 8, d);
    run<7>("64-bit add (2 ops)", 8, d);
    run<1>("v_lshlrev_b64 + v_or", 8, d);
    run<9>("v_lshrrev_b64 + v_or", 8, d);
    run<6>("64x64->64 mul + add", 4, d);
    return 0;
}
