// valu_more.hip -- SIMD cycles per wave64 instruction on gfx950 for the instructions that could replace the single-pipe ones of
// the node step (same method as valu_pipes.hip: the last of w waves per SIMD to finish, cycles / (w x instructions)).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define REP8(x) x x x x x x x x
#define OPS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20", "s21"
#define I8(ins) ins(0) ins(1) ins(2) ins(3) ins(4) ins(5) ins(6) ins(7)
template <int KIND>
__global__ __launch_bounds__(64) void k_rate(uint32_t iters, float seed, unsigned long long *cycles, float *sink) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    float b = seed * 0.5f, c = seed * 0.25f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t i = 0; i < iters; ++i) {
#define A(n) "v_fma_mix_f32 %" #n ", %8, %9, %" #n " op_sel_hi:[1,0,0]\n"
        if (KIND == 0) { REP8(asm volatile(I8(A) : OPS);) }
#undef A
#define A(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
        if (KIND == 1) { REP8(asm volatile(I8(A) : OPS);) }
#undef A
#define A(n) "v_or_b32_sdwa %" #n ", %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
        if (KIND == 2) { REP8(asm volatile(I8(A) : OPS);) }
#undef A
#define A(n) "v_cvt_f32_ubyte0_sdwa %" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n"
        if (KIND == 3) { REP8(asm volatile(I8(A) : OPS);) }
#undef A
#define A(n) "v_cmp_class_f32 vcc, %" #n ", %8\n"
        if (KIND == 4) { REP8(asm volatile(I8(A) : OPS);) }
#undef A
#define A(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
        if (KIND == 5) { REP8(asm volatile(I8(A) : OPS);) }
#undef A
#define A(n) "v_lshl_add_u32 %" #n ", %" #n ", 3, %9\n"
        if (KIND == 6) { REP8(asm volatile(I8(A) : OPS);) }
#undef A
#define A(n) "v_or3_b32 %" #n ", %" #n ", %8, %9\n"
        if (KIND == 7) { REP8(asm volatile(I8(A) : OPS);) }
#undef A
#define A(n) "v_add_f32_sdwa %" #n ", %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n"
        if (KIND == 8) { REP8(asm volatile(I8(A) : OPS);) }
#undef A
#define A(n) "v_cmp_lt_u32 vcc, %" #n ", %8\n"
        if (KIND == 9) { REP8(asm volatile(I8(A) : OPS);) }
#undef A
#define A(n) "v_mul_legacy_f32 %" #n ", %" #n ", %8\n"
        if (KIND == 10) { REP8(asm volatile(I8(A) : OPS);) }
#undef A
#define A(n) "v_subrev_u32 %" #n ", %8, %" #n "\n"
        if (KIND == 11) { REP8(asm volatile(I8(A) : OPS);) }
#undef A
#define A(n) "v_ashrrev_i32 %" #n ", 3, %" #n "\n"
        if (KIND == 12) { REP8(asm volatile(I8(A) : OPS);) }
#undef A
#define A(n) "v_bcnt_u32_b32 %" #n ", %8, %" #n "\n"
        if (KIND == 13) { REP8(asm volatile(I8(A) : OPS);) }
#undef A
#define A(n) "v_fma_f32 %" #n ", |%8|, -%9, %" #n "\n"
        if (KIND == 14) { REP8(asm volatile(I8(A) : OPS);) }
#undef A
#define A(n) "v_add_f32 %" #n ", 0x3f4ccccd, %" #n "\n"
        if (KIND == 15) { REP8(asm volatile(I8(A) : OPS);) }
#undef A
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (r == 12345.678f) sink[threadIdx.x] = r;
}
template <int KIND>
static void run(const char *name, int n_cu, unsigned long long *d_cyc, float *d_sink) {
    const uint32_t iters = 2000;
    printf("%-34s", name);
    for (int w : {1, 2, 4}) {
        const int grid = n_cu * 4 * w;
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_rate<KIND>), dim3(grid), dim3(64), 0, 0, iters, 1.0f, d_cyc, d_sink);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(grid);
        (void)hipMemcpy(h.data(), d_cyc, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("  w%d: %5.2f", w, (double)h[grid - 1] / ((double)iters * 64.0 * w));
    }
    printf("\n");
}
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    unsigned long long *d_cyc; float *d_sink;
    (void)hipMalloc(&d_cyc, sizeof(unsigned long long) * p.multiProcessorCount * 64); (void)hipMalloc(&d_sink, 256);
    const int n = p.multiProcessorCount;
    run<0>("v_fma_mix_f32 (f16 x f32 + f32)", n, d_cyc, d_sink);
    run<1>("v_perm_b32", n, d_cyc, d_sink);
    run<2>("v_or_b32_sdwa (byte select)", n, d_cyc, d_sink);
    run<3>("v_cvt_f32_ubyte0_sdwa", n, d_cyc, d_sink);
    run<4>("v_cmp_class_f32", n, d_cyc, d_sink);
    run<5>("v_add3_u32", n, d_cyc, d_sink);
    run<6>("v_lshl_add_u32", n, d_cyc, d_sink);
    run<7>("v_or3_b32", n, d_cyc, d_sink);
    run<8>("v_add_f32_sdwa", n, d_cyc, d_sink);
    run<9>("v_cmp_lt_u32", n, d_cyc, d_sink);
    run<10>("v_mul_legacy_f32", n, d_cyc, d_sink);
    run<11>("v_subrev_u32", n, d_cyc, d_sink);
    run<12>("v_ashrrev_i32", n, d_cyc, d_sink);
    run<13>("v_bcnt_u32_b32", n, d_cyc, d_sink);
    run<14>("v_fma_f32 with |a|, -b modifiers", n, d_cyc, d_sink);
    run<15>("v_add_f32 with a literal", n, d_cyc, d_sink);
    return 0;
}
