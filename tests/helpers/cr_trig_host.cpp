// Host compile of the PRODUCT's csrc/cr_trig.h (the pose kernel's sinf / cosf / acosf / asinf / atan2f: the platform's
// double function + a double-double slow path), so that the CPU suite can sweep it against the oracle without a GPU.
// On the host the fast path calls libm's double functions where the device calls ocml's: by construction that does
// not change a result.  Test infrastructure.
#include "../../nvidia-optix-ray-tracer_amd/csrc/cr_trig.h"
#include <stdint.h>
#include <string.h>

static inline float one(int which, float a, float b, bool slow) {
    switch (which) {
        case 0: return hrt::sinf_cr(a, slow);
        case 1: return hrt::cosf_cr(a, slow);
        case 2: return hrt::acosf_cr(a, slow);
        case 3: return hrt::asinf_cr(a, slow);
        default: return hrt::atan2f_cr(a, b, slow);
    }
}
extern "C" void host_trig(int which, const float *a, const float *b, uint64_t n, int force_slow, float *out) {
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n; ++i) out[i] = one(which, a[i], b ? b[i] : 0.0f, force_slow != 0);
}
extern "C" void host_trig_bits(int which, uint32_t first, uint32_t stride, uint64_t count, int force_slow, float *out) {
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < count; ++i) {
        const uint32_t bits = first + (uint32_t)(i * stride);
        float x; memcpy(&x, &bits, 4);
        out[i] = one(which, x, 0.0f, force_slow != 0);
    }
}
