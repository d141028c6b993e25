// valu_issue_patterns.hip -- what keeps two waves of a SIMD from issuing VALU instructions side by side?  Cycles per VALU
// instruction at 1 / 2 / 4 / 5 / 8 waves per SIMD for instruction streams with dependencies, exec-mask changes, VCC traffic, LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define REP8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(64) void k_pat(uint32_t iters, float seed, unsigned long long *cycles, float *sink) {
    __shared__ float lds[64];
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    float b = seed * 0.5f, c = seed * 0.25f;
    lds[threadIdx.x] = seed;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t i = 0; i < iters; ++i) {
        if (KIND == 0) {        // 8 independent simple chains (reference)
            REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 1) { // ONE dependent chain of simple ops
            REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 2) { // two dependent chains
            REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 3) { // independent simple ops with an exec save / restore pair around every 4
            REP8(asm volatile("s_mov_b64 s[20:21], exec\n s_and_b64 exec, exec, s[20:21]\n v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n s_mov_b64 exec, s[20:21]\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s20", "s21");)
        } else if (KIND == 4) { // v_cmp to vcc + s_and_saveexec + 3 VALU + restore (a divergent if)
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n s_or_b64 exec, exec, s[20:21]\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s20", "s21", "vcc");)
        } else if (KIND == 5) { // 8 VALU with 8 SALU interleaved
            REP8(asm volatile("v_add_f32 %0, %0, %8\n s_add_u32 s20, s20, 1\n v_add_f32 %1, %1, %8\n s_add_u32 s21, s21, 1\n v_add_f32 %2, %2, %8\n s_add_u32 s20, s20, 1\n v_add_f32 %3, %3, %8\n s_add_u32 s21, s21, 1\n v_add_f32 %4, %4, %8\n s_add_u32 s20, s20, 1\n v_add_f32 %5, %5, %8\n s_add_u32 s21, s21, 1\n v_add_f32 %6, %6, %8\n s_add_u32 s20, s20, 1\n v_add_f32 %7, %7, %8\n s_add_u32 s21, s21, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s20", "s21", "scc");)
        } else if (KIND == 6) { // v_cndmask with vcc from a v_cmp every other instruction (VCC ping-pong)
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
        } else if (KIND == 7) { // dependent chain alternating simple and complex: fma -> max -> fma -> max
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_max_f32 %0, %0, %8\n v_fma_f32 %0, %0, %8, %9\n v_max_f32 %0, %0, %8\n v_fma_f32 %0, %0, %8, %9\n v_max_f32 %0, %0, %8\n v_fma_f32 %0, %0, %8, %9\n v_max_f32 %0, %0, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 8) { // the slab test of one child as the compiler writes it: 6 cvt, 6 fma, max3, max, min3, min, cmp, cndmask (dependencies as in the kernel)
            REP8(asm volatile("v_cvt_f32_ubyte0 %0, %8\n v_cvt_f32_ubyte1 %1, %8\n v_cvt_f32_ubyte2 %2, %8\n v_cvt_f32_ubyte0 %3, %9\n v_cvt_f32_ubyte1 %4, %9\n v_cvt_f32_ubyte2 %5, %9\n"
                              "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
                              "v_max3_f32 %0, %0, %1, %2\n v_max_f32 %0, %0, %8\n v_min3_f32 %3, %3, %4, %5\n v_min_f32 %3, %3, %9\n v_cmp_le_f32 vcc, %0, %3\n v_cndmask_b32 %6, %6, %8, vcc"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
        } else if (KIND == 9) { // 8 independent VALU + one ds_read / ds_write pair and a wait
            REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n ds_write_b32 %10, %2\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n ds_read_b32 %4, %10\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n s_waitcnt lgkmcnt(0)\n v_add_f32 %7, %7, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "v"((uint32_t)(threadIdx.x * 4)) : "memory");)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + lds[(threadIdx.x + 1) & 63];
    if (r == 12345.678f) sink[threadIdx.x] = r;
}
template <int KIND>
static void run(const char *name, double valu_per_rep, int n_cu, unsigned long long *d_cyc, float *d_sink) {
    const uint32_t iters = 2000;
    printf("%-58s", name);
    for (int w : {1, 2, 4, 5, 8}) {
        const int grid = n_cu * 4 * w;
        hipLaunchKernelGGL((k_pat<KIND>), dim3(grid), dim3(64), 0, 0, iters, 1.0f, d_cyc, d_sink);
        hipLaunchKernelGGL((k_pat<KIND>), dim3(grid), dim3(64), 0, 0, iters, 1.0f, d_cyc, d_sink);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(grid);
        (void)hipMemcpy(h.data(), d_cyc, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("  w%d: %5.2f", w, (double)h[grid - 1] / ((double)iters * 8.0 * valu_per_rep * w));
    }
    printf("\n");
}
int main() {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    printf("%s: SIMD cycles per VALU instruction at w waves per SIMD\n", prop.gcnArchName);
    unsigned long long *d_cyc; float *d_sink;
    (void)hipMalloc((void **)&d_cyc, sizeof(unsigned long long) * n_cu * 4 * 8);
    (void)hipMalloc((void **)&d_sink, 256);
    run<0>("8 independent v_add chains", 8, n_cu, d_cyc, d_sink);
    run<1>("ONE dependent v_add chain", 8, n_cu, d_cyc, d_sink);
    run<2>("two dependent v_add chains", 8, n_cu, d_cyc, d_sink);
    run<3>("independent v_add, exec saved / masked / restored per 8", 8, n_cu, d_cyc, d_sink);
    run<4>("v_cmp + s_and_saveexec + 3 v_add + s_or exec + 4 v_add", 8, n_cu, d_cyc, d_sink);
    run<5>("v_add with an s_add between every two", 8, n_cu, d_cyc, d_sink);
    run<6>("v_cmp -> v_cndmask pairs through vcc", 8, n_cu, d_cyc, d_sink);
    run<7>("dependent chain fma -> max -> fma -> max", 8, n_cu, d_cyc, d_sink);
    run<8>("one child's slab test (6 cvt, 6 fma, max3, max, min3, min, cmp, cndmask)", 18, n_cu, d_cyc, d_sink);
    run<9>("8 v_add + ds_write + ds_read + lgkmcnt wait", 8, n_cu, d_cyc, d_sink);
    return 0;
}
