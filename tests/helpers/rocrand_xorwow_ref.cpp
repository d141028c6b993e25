// Test helper (CPU tests only): runs rocRAND's host-callable xorwow engine so that the oracle's
// sub-sequence skip-ahead (T^(2^67 k) over GF(2)) can be pinned against an independent,
// third-party implementation of the same recurrence.  rocRAND scrambles the seed with other
// constants than cuRAND, so the oracle is called with rocRAND's constants for this comparison.
//   build: hipcc -O1 -shared -fPIC -x c++ -D__HIP_PLATFORM_AMD__ rocrand_xorwow_ref.cpp -I/opt/rocm/include
#include <rocrand/rocrand_xorwow.h>
#include <cstdint>

extern "C" void rocrand_xorwow_draw(unsigned long long seed, unsigned long long subsequence, unsigned long long offset,
                                    uint32_t *out, int n) {
    rocrand_device::xorwow_engine e(seed, subsequence, offset);
    for (int i = 0; i < n; ++i) out[i] = e();
}
