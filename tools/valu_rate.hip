// valu_rate.hip -- what a gfx950 SIMD issues per cycle, by instruction kind and waves per SIMD.
// Settles the unit of "VALU issue slots" used in DESIGN.md section 4 (is a wave64 VALU instruction 2 or 4 cycles of a SIMD?).
// Build: hipcc --offload-arch=gfx950 -O2 tools/valu_rate.hip -o nvidia-optix-ray-tracer_amd/lib/valu_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(64) void k_rate(uint32_t iters, float seed, unsigned long long *cycles, float *sink) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    float b = seed * 0.5f, c = seed * 0.25f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
    uint32_t u0 = __float_as_uint(a0), u1 = __float_as_uint(a1), u2 = __float_as_uint(a2), u3 = __float_as_uint(a3);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t i = 0; i < iters; ++i) {
        if (KIND == 0) {          // v_fma_f32, 8 independent chains
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 1) {   // v_pk_fma_f32, 4 independent chains (x2 to make 8 instructions)
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                              "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
        } else if (KIND == 2) {   // v_cvt_f32_ubyte0..3
            REP8(asm volatile("v_cvt_f32_ubyte0 %0, %8\n v_cvt_f32_ubyte1 %1, %8\n v_cvt_f32_ubyte2 %2, %8\n v_cvt_f32_ubyte3 %3, %8\n"
                              "v_cvt_f32_ubyte0 %4, %9\n v_cvt_f32_ubyte1 %5, %9\n v_cvt_f32_ubyte2 %6, %9\n v_cvt_f32_ubyte3 %7, %9"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(u0), "v"(u1));)
        } else if (KIND == 3) {   // v_max3_f32
            REP8(asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                              "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 4) {   // v_pk_fma_f16
            REP8(asm volatile("v_pk_fma_f16 %0, %0, %4, %5\n v_pk_fma_f16 %1, %1, %4, %5\n v_pk_fma_f16 %2, %2, %4, %5\n v_pk_fma_f16 %3, %3, %4, %5\n"
                              "v_pk_fma_f16 %0, %0, %4, %5\n v_pk_fma_f16 %1, %1, %4, %5\n v_pk_fma_f16 %2, %2, %4, %5\n v_pk_fma_f16 %3, %3, %4, %5"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(b), "v"(c));)
        } else if (KIND == 5) {   // v_perm_b32
            REP8(asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5\n"
                              "v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(b), "v"(c));)
        } else if (KIND == 6) {   // v_rcp_f32 (transcendental)
            REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                              "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 7) {   // v_pk_max_f16 / v_pk_min_f16
            REP8(asm volatile("v_pk_max_f16 %0, %0, %4\n v_pk_min_f16 %1, %1, %4\n v_pk_max_f16 %2, %2, %5\n v_pk_min_f16 %3, %3, %5\n"
                              "v_pk_max_f16 %0, %0, %4\n v_pk_min_f16 %1, %1, %4\n v_pk_max_f16 %2, %2, %5\n v_pk_min_f16 %3, %3, %5"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(b), "v"(c));)
        } else if (KIND == 8) {   // v_cndmask_b32 with vcc + v_cmp
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                              "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + __uint_as_float(u0 ^ u1 ^ u2 ^ u3);
    if (r == 12345.678f) sink[threadIdx.x] = r;
}

template <int KIND>
static void run(const char *name, int n_cu, unsigned long long *d_cyc, float *d_sink) {
    const uint32_t iters = 4000;               // x 64 instructions
    for (int w : {1, 2, 3, 4, 5, 8}) {
        const int grid = n_cu * 4 * w;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k_rate<KIND>), dim3(grid), dim3(64), 0, 0, iters, 1.0f, d_cyc, d_sink);      // warm-up
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_rate<KIND>), dim3(grid), dim3(64), 0, 0, iters, 1.0f, d_cyc, d_sink);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(grid);
        hipMemcpy(h.data(), d_cyc, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[grid / 2], mx = (double)h[grid - 1];
        const double instr = (double)iters * 64.0;
        // per SIMD: w waves x instr instructions in `med` cycles (all waves of a SIMD run concurrently)
        printf("%-16s waves/SIMD %d: %8.3f ms  wave cycles median %9.0f max %9.0f -> %.3f cycles per wave-instruction per SIMD (%.2f per wave)\n",
               name, w, ms, med, mx, med / (instr * w), med / instr);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz\n", prop.gcnArchName, n_cu, prop.clockRate);
    unsigned long long *d_cyc; float *d_sink;
    hipMalloc((void **)&d_cyc, sizeof(unsigned long long) * n_cu * 4 * 8);
    hipMalloc((void **)&d_sink, 256);
    run<0>("v_fma_f32", n_cu, d_cyc, d_sink);
    run<1>("v_pk_fma_f32", n_cu, d_cyc, d_sink);
    run<2>("v_cvt_f32_ubyte", n_cu, d_cyc, d_sink);
    run<3>("v_max3_f32", n_cu, d_cyc, d_sink);
    run<4>("v_pk_fma_f16", n_cu, d_cyc, d_sink);
    run<5>("v_perm_b32", n_cu, d_cyc, d_sink);
    run<6>("v_rcp_f32", n_cu, d_cyc, d_sink);
    run<7>("v_pk_minmax_f16", n_cu, d_cyc, d_sink);
    run<8>("v_cmp+v_cndmask", n_cu, d_cyc, d_sink);
    return 0;
}
