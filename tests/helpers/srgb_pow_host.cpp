// Host compile of the PRODUCT's csrc/srgb_pow.h (the arithmetic k_finalize / k_to_rgba8 / k_color_to_float4 run on the
// device: doubles with +,-,*,/ and fma only, no libm), so that the CPU suite can sweep it against the oracle over every
// float in [0, 1] without a GPU.  Test infrastructure.
#include "../../nvidia-optix-ray-tracer_amd/csrc/srgb_pow.h"
#include <stdint.h>
#include <string.h>

extern "C" void host_pow_inv_gamma_bits(uint32_t first, uint64_t count, float *out) {
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < count; ++i) {
        const uint32_t b = first + (uint32_t)i;
        float x; memcpy(&x, &b, 4);
        out[i] = hrt::pow_inv_gamma(x);
    }
}
