// tools/ubench/valu_rates.hip -- issue cost of the VALU instructions the mapping kernels lean on (gfx950), as a function
// of how many waves share a SIMD.  Inline assembly, so the instruction stream is exactly what is named: every wave runs
// REP iterations of 32 instructions of ONE kind over 8 independent register chains (no dependent back-to-back pair closer
// than 8 instructions).  Prints cycles per wave-instruction per SIMD = (event time x 2.4 GHz) / (instructions per SIMD).
//   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 2048
#define R4(x) x x x x
// 8 chains on 32-bit registers: ins dst, src0, src1 with dst = chain register
#define BODY32(ins) \
    asm volatile(R4(ins " %0, %8, %0\n" ins " %1, %8, %1\n" ins " %2, %8, %2\n" ins " %3, %8, %3\n" \
                    ins " %4, %8, %4\n" ins " %5, %8, %5\n" ins " %6, %8, %6\n" ins " %7, %8, %7\n") \
                 : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]) : "v"(s))
#define BODY32_3(ins) /* three-operand: dst = f(chain, s, chain) */ \
    asm volatile(R4(ins " %0, %0, %8, %0\n" ins " %1, %1, %8, %1\n" ins " %2, %2, %8, %2\n" ins " %3, %3, %8, %3\n" \
                    ins " %4, %4, %8, %4\n" ins " %5, %5, %8, %5\n" ins " %6, %6, %8, %6\n" ins " %7, %7, %8, %7\n") \
                 : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]) : "v"(s))
#define BODY32_1(ins) /* one source */ \
    asm volatile(R4(ins " %0, %0\n" ins " %1, %1\n" ins " %2, %2\n" ins " %3, %3\n" ins " %4, %4\n" ins " %5, %5\n" ins " %6, %6\n" ins " %7, %7\n") \
                 : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]))
#define BODY64(ins) /* 64-bit shift: dst pair, shift amount, src pair */ \
    asm volatile(R4(ins " %0, %8, %0\n" ins " %1, %8, %1\n" ins " %2, %8, %2\n" ins " %3, %8, %3\n" \
                    ins " %4, %8, %4\n" ins " %5, %8, %5\n" ins " %6, %8, %6\n" ins " %7, %8, %7\n") \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(s))
#define BODYDPP(ins) \
    asm volatile(R4(ins " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" ins " %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" \
                    ins " %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n" ins " %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n" \
                    ins " %4, %4, %4 row_mirror row_mask:0xf bank_mask:0xf\n" ins " %5, %5, %5 row_mirror row_mask:0xf bank_mask:0xf\n" \
                    ins " %6, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xf\n" ins " %7, %7, %7 row_half_mirror row_mask:0xf bank_mask:0xf\n") \
                 : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]))
#define BODYRL /* v_readlane_b32 into 8 SGPRs (then nothing: the scalar results are dead but asm volatile keeps them) */ \
    asm volatile(R4("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9\n" \
                    "v_readlane_b32 s24, %4, 11\n v_readlane_b32 s25, %5, 13\n v_readlane_b32 s26, %6, 15\n v_readlane_b32 s27, %7, 17\n") \
                 : : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27")
#define BODYSALU /* scalar ALU for comparison */ \
    asm volatile(R4("s_add_u32 s20, s20, s28\n s_add_u32 s21, s21, s28\n s_lshl_b32 s22, s22, 1\n s_and_b32 s23, s23, s28\n" \
                    "s_add_u32 s24, s24, s28\n s_xor_b32 s25, s25, s28\n s_lshr_b32 s26, s26, 1\n s_or_b32 s27, s27, s28\n") \
                 : : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "scc")
template <int OP>
__global__ void __launch_bounds__(256) k(uint64_t* out, uint32_t seed) {
    uint32_t b[8];
    uint64_t a[8];
    for (int i = 0; i < 8; ++i) { b[i] = seed * (2 * i + 1) + threadIdx.x; a[i] = (uint64_t)b[i] * 0x9E3779B97F4A7C15ULL; }
    uint32_t s = (seed & 15) + 1;
    for (int i = 0; i < REP; ++i) {
        if (OP == 0) BODY32("v_add_u32");
        if (OP == 1) BODY32("v_xor_b32");
        if (OP == 2) BODY32("v_lshlrev_b32");
        if (OP == 3) BODY32_3("v_lshl_or_b32");
        if (OP == 4) BODY32_3("v_alignbit_b32");
        if (OP == 5) BODY32_1("v_bfrev_b32");
        if (OP == 6) BODY32("v_bcnt_u32_b32");
        if (OP == 7) BODY32("v_mul_lo_u32");
        if (OP == 8) BODY32("v_mul_hi_u32");
        if (OP == 9) BODY32_3("v_mad_u32_u24");
        if (OP == 10) BODY64("v_lshlrev_b64");
        if (OP == 11) BODY64("v_lshrrev_b64");
        if (OP == 12) BODYDPP("v_add_u32_dpp");
        if (OP == 13) BODYRL;
        if (OP == 14) BODY32("v_min_u32");
        if (OP == 15) BODY32_3("v_and_or_b32");
        if (OP == 16) BODY32_3("v_xad_u32");
        if (OP == 17) BODYSALU;
        if (OP == 18) BODY32_3("v_bfe_u32");
        if (OP == 19) BODY32_3("v_add3_u32");
        if (OP == 20) BODY32("v_and_b32");
        if (OP == 21) BODY32("v_or_b32");
        if (OP == 22) BODY32("v_sub_u32");
        if (OP == 23) BODY32_1("v_mov_b32");
        if (OP == 24) BODY32_1("v_not_b32");
        if (OP == 25) BODY32("v_lshrrev_b32");
        if (OP == 26) BODY32("v_cndmask_b32");   // (e32: vcc implicit)
        if (OP == 30) asm volatile(R4("v_cndmask_b32_e64 %0, %8, %0, s[20:21]\n v_cndmask_b32_e64 %1, %8, %1, s[20:21]\n v_cndmask_b32_e64 %2, %8, %2, s[20:21]\n v_cndmask_b32_e64 %3, %8, %3, s[20:21]\n"
                                     "v_cndmask_b32_e64 %4, %8, %4, s[20:21]\n v_cndmask_b32_e64 %5, %8, %5, s[20:21]\n v_cndmask_b32_e64 %6, %8, %6, s[20:21]\n v_cndmask_b32_e64 %7, %8, %7, s[20:21]\n")
                                  : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]) : "v"(s) : "s20", "s21");
        if (OP == 31) asm volatile(R4("v_cmp_lt_u32 vcc, %8, %0\n v_cmp_lt_u32 vcc, %8, %1\n v_cmp_lt_u32 vcc, %8, %2\n v_cmp_lt_u32 vcc, %8, %3\n"
                                     "v_cmp_lt_u32 vcc, %8, %4\n v_cmp_lt_u32 vcc, %8, %5\n v_cmp_lt_u32 vcc, %8, %6\n v_cmp_lt_u32 vcc, %8, %7\n")
                                  : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]) : "v"(s) : "vcc");
        if (OP == 32) asm volatile(R4("v_cmp_lt_u32 vcc, %8, %0\n v_cndmask_b32 %0, %8, %0, vcc\n v_cmp_lt_u32 vcc, %8, %1\n v_cndmask_b32 %1, %8, %1, vcc\n"
                                     "v_cmp_lt_u32 vcc, %8, %2\n v_cndmask_b32 %2, %8, %2, vcc\n v_cmp_lt_u32 vcc, %8, %3\n v_cndmask_b32 %3, %8, %3, vcc\n")
                                  : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "v"(s), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]) : "vcc");
        if (OP == 33) BODY32("v_lshlrev_b32");
        if (OP == 27) BODY32("v_max_u32");
        if (OP == 28) BODY32("v_subrev_u32");
        if (OP == 29) BODY32("v_ashrrev_i32");
        // round 3: the rest of the opcodes the mapping kernels' dynamic mix contains (tools/bbcount.py prices with these)
        if (OP == 40) asm volatile(R4("v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %1, s[20:21], %8, %9, %1\n v_mad_u64_u32 %2, s[20:21], %8, %9, %2\n v_mad_u64_u32 %3, s[20:21], %8, %9, %3\n"
                                     "v_mad_u64_u32 %4, s[20:21], %8, %9, %4\n v_mad_u64_u32 %5, s[20:21], %8, %9, %5\n v_mad_u64_u32 %6, s[20:21], %8, %9, %6\n v_mad_u64_u32 %7, s[20:21], %8, %9, %7\n")
                                  : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(s), "v"(b[0]) : "s20", "s21");
        if (OP == 41) asm volatile(R4("v_bitop3_b32 %0, %0, %8, %0 bitop3:0x48\n v_bitop3_b32 %1, %1, %8, %1 bitop3:0x48\n v_bitop3_b32 %2, %2, %8, %2 bitop3:0x48\n v_bitop3_b32 %3, %3, %8, %3 bitop3:0x48\n"
                                     "v_bitop3_b32 %4, %4, %8, %4 bitop3:0x48\n v_bitop3_b32 %5, %5, %8, %5 bitop3:0x48\n v_bitop3_b32 %6, %6, %8, %6 bitop3:0x48\n v_bitop3_b32 %7, %7, %8, %7 bitop3:0x48\n")
                                  : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]) : "v"(s));
        if (OP == 42) BODY32_3("v_lshl_add_u32");
        if (OP == 43) BODY32_3("v_perm_b32");
        if (OP == 44) BODY32_1("v_ffbl_b32");
        if (OP == 45) asm volatile(R4("v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %1, %2\n v_cmp_lt_u64 vcc, %2, %3\n v_cmp_lt_u64 vcc, %3, %4\n"
                                     "v_cmp_lt_u64 vcc, %4, %5\n v_cmp_lt_u64 vcc, %5, %6\n v_cmp_lt_u64 vcc, %6, %7\n v_cmp_lt_u64 vcc, %7, %0\n")
                                  : : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "vcc");
        if (OP == 46) asm volatile(R4("v_lshl_add_u64 %0, %0, 3, %1\n v_lshl_add_u64 %1, %1, 3, %2\n v_lshl_add_u64 %2, %2, 3, %3\n v_lshl_add_u64 %3, %3, 3, %4\n"
                                     "v_lshl_add_u64 %4, %4, 3, %5\n v_lshl_add_u64 %5, %5, 3, %6\n v_lshl_add_u64 %6, %6, 3, %7\n v_lshl_add_u64 %7, %7, 3, %0\n")
                                  : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
        if (OP == 47) asm volatile(R4("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %4\n v_mov_b64 %4, %5\n v_mov_b64 %5, %6\n v_mov_b64 %6, %7\n v_mov_b64 %7, %0\n")
                                  : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
        if (OP == 48) asm volatile(R4("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                                     "v_mov_b32_dpp %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n"
                                     "v_mov_b32_dpp %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                                     "v_mov_b32_dpp %6, %6 row_half_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 row_mirror row_mask:0xf bank_mask:0xf\n")
                                  : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
        if (OP == 49) asm volatile(R4("v_cndmask_b32 %0, %8, %0, vcc\n v_cndmask_b32 %1, %8, %1, vcc\n v_cndmask_b32 %2, %8, %2, vcc\n v_cndmask_b32 %3, %8, %3, vcc\n"
                                     "v_cndmask_b32 %4, %8, %4, vcc\n v_cndmask_b32 %5, %8, %5, vcc\n v_cndmask_b32 %6, %8, %6, vcc\n v_cndmask_b32 %7, %8, %7, vcc\n")
                                  : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]) : "v"(s) : "vcc");
        if (OP == 50) asm volatile(R4("v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s22, %2\n v_readfirstlane_b32 s23, %3\n"
                                     "v_readfirstlane_b32 s24, %4\n v_readfirstlane_b32 s25, %5\n v_readfirstlane_b32 s26, %6\n v_readfirstlane_b32 s27, %7\n")
                                  : : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        if (OP == 51) asm volatile(R4("v_writelane_b32 %0, s28, 3\n v_writelane_b32 %1, s28, 5\n v_writelane_b32 %2, s28, 7\n v_writelane_b32 %3, s28, 9\n"
                                     "v_writelane_b32 %4, s28, 11\n v_writelane_b32 %5, s28, 13\n v_writelane_b32 %6, s28, 15\n v_writelane_b32 %7, s28, 17\n")
                                  : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]) : : "s28");
        if (OP == 52) BODY32("v_add_co_u32_e32");   // (vcc implicit)
        if (OP == 53) asm volatile(R4("v_max_u32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_max_u32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n"
                                     "v_max_u32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_max_u32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n"
                                     "v_max_u32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_max_u32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n"
                                     "v_max_u32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_max_u32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n")
                                  : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]) : "v"(s));
        if (OP == 54) BODY32("v_and_b32_e64");      // a plain op in its 64-bit (VOP3) encoding
        if (OP == 55) BODY32_3("v_bfi_b32");
        if (OP == 56) BODY32("v_sub_u32_e64");
    }
    uint64_t x = 0;
    for (int i = 0; i < 8; ++i) x ^= a[i] ^ b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
template <int OP>
void run(const char* name, uint64_t* d) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%-18s", name);
    for (int wps : {1, 2, 4, 6, 8}) {              // waves per SIMD: 256 CUs x wps workgroups of 4 waves
        const int blocks = 256 * wps, threads = 256;
        k<OP><<<blocks, threads>>>(d, 12345);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) k<OP><<<blocks, threads>>>(d, 12345 + r);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double wave_instr = 5.0 * blocks * (threads / 64) * (double)REP * 32;
        const double ns = ms * 1e6 / (wave_instr / 1024.0);
        printf("  w%d: %5.2f cyc", wps, ns * 2.4);
    }
    printf("\n");
}
int main() {
    uint64_t* d; (void)hipMalloc(&d, 256 * 8 * 256 * 8);
    int clk_khz = 0; (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("device clock attribute %d kHz; cycles below = event time x 2.4 GHz / wave-instructions per SIMD (1024 SIMDs)\n", clk_khz);
    printf("columns: waves resident per SIMD (w1 = one wave alone)\n");
    run<0>("v_add_u32", d);
    run<1>("v_xor_b32", d);
    run<20>("v_and_b32", d);
    run<21>("v_or_b32", d);
    run<22>("v_sub_u32", d);
    run<28>("v_subrev_u32", d);
    run<23>("v_mov_b32", d);
    run<24>("v_not_b32", d);
    run<30>("v_cndmask e64 sgpr", d);
    run<31>("v_cmp_lt_u32", d);
    run<32>("cmp+cndmask pairs", d);
    run<2>("v_lshlrev_b32", d);
    run<25>("v_lshrrev_b32", d);
    run<29>("v_ashrrev_i32", d);
    run<27>("v_max_u32", d);
    run<14>("v_min_u32", d);
    run<3>("v_lshl_or_b32", d);
    run<15>("v_and_or_b32", d);
    run<16>("v_xad_u32", d);
    run<19>("v_add3_u32", d);
    run<18>("v_bfe_u32", d);
    run<4>("v_alignbit_b32", d);
    run<5>("v_bfrev_b32", d);
    run<6>("v_bcnt_u32_b32", d);
    run<9>("v_mad_u32_u24", d);
    run<7>("v_mul_lo_u32", d);
    run<8>("v_mul_hi_u32", d);
    run<10>("v_lshlrev_b64", d);
    run<11>("v_lshrrev_b64", d);
    run<12>("v_add_u32_dpp", d);
    run<13>("v_readlane_b32", d);
    run<17>("SALU mix", d);
    run<40>("v_mad_u64_u32", d);
    run<41>("v_bitop3_b32", d);
    run<42>("v_lshl_add_u32", d);
    run<43>("v_perm_b32", d);
    run<44>("v_ffbl_b32", d);
    run<45>("v_cmp_lt_u64", d);
    run<46>("v_lshl_add_u64", d);
    run<47>("v_mov_b64", d);
    run<48>("v_mov_b32_dpp", d);
    run<49>("v_cndmask e32 vcc", d);
    run<50>("v_readfirstlane", d);
    run<51>("v_writelane_b32", d);
    run<52>("v_add_co_u32", d);
    run<53>("v_max_u32_sdwa", d);
    run<54>("v_and_b32_e64", d);
    run<55>("v_bfi_b32", d);
    run<56>("v_sub_u32_e64", d);
    return 0;
}
