// bvh8_host_api.cpp -- the host-only entry points of include/hrt.h: the BVH8 builder over host triangles and the blob helpers.
// No HIP in here: the file is part of libhrt.so and, alone with bvh8_build.cpp, of the sanitizer build (make asan:
// HRT_HOST_ONLY), which runs the builder under AddressSanitizer / UBSan on the CPU.
#include "../../include/hrt.h"
#include "bvh8.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace hrt {
#ifdef HRT_HOST_ONLY
namespace { thread_local std::string g_host_error; }
static int host_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_host_error = buf;
    return code;
}
#else
int fail(HrtContext *ctx, int code, const char *fmt, ...);       // hrt_api.cpp
#define host_fail(code, ...) fail(nullptr, code, __VA_ARGS__)
#endif
}  // namespace hrt

using namespace hrt;

extern "C" {

#ifdef HRT_HOST_ONLY
const char *hrt_last_error(const HrtContext *) { return g_host_error.c_str(); }
#endif

int alloc_bvh_blob(size_t n_nodes, size_t n_prims, const float *lo, const float *hi, HrtBvhBlob *out) {
    std::memset(out, 0, sizeof *out);
    out->n_nodes = n_nodes; out->n_triangles = n_prims;
    out->nodes = std::malloc(std::max<size_t>(1, sizeof(Bvh8Node) * n_nodes));
    out->triangles = std::malloc(std::max<size_t>(1, sizeof(PrimRecord) * n_prims));
    if (!out->nodes || !out->triangles) { std::free(out->nodes); std::free(out->triangles); std::memset(out, 0, sizeof *out); return HRT_ERR_OOM; }
    for (int a = 0; a < 3; ++a) { out->bounds[a] = lo[a]; out->bounds[3 + a] = hi[a]; }
    return HRT_OK;
}

int hrt_host_build_bvh8(const float *h_triangles, uint32_t n_triangles, HrtBvhBlob *out) {
    if (!out || (n_triangles && !h_triangles)) return HRT_ERR_INVALID;
    std::vector<BuildPrim> prims(n_triangles);
    for (uint32_t p = 0; p < n_triangles; ++p) {
        BuildPrim &bp = prims[p]; std::memset(&bp, 0, sizeof bp);
        const float *v = h_triangles + 9 * (size_t)p;
        for (int a = 0; a < 3; ++a) {
            bp.rec.a[a] = v[a]; bp.rec.b[a] = v[3 + a] - v[a]; bp.rec.c[a] = v[6 + a] - v[a];
            bp.lo[a] = std::fmin(v[a], std::fmin(v[3 + a], v[6 + a]));
            bp.hi[a] = std::fmax(v[a], std::fmax(v[3 + a], v[6 + a]));
        }
        bp.rec.prim = p; bp.rec.inst = 0; bp.rec.kind = kPrimKindTriangle;
    }
    Bvh8 b;
    build_bvh8(prims, b, 0);
    const char *err = validate_bvh8(b);
    if (err[0]) return host_fail(HRT_ERR_STATE, "bvh8 validation: %s", err);
    const int rc = alloc_bvh_blob(b.nodes.size(), b.prims.size(), b.lo, b.hi, out);
    if (rc != HRT_OK) return rc;
    if (!b.nodes.empty()) std::memcpy(out->nodes, b.nodes.data(), sizeof(Bvh8Node) * b.nodes.size());
    if (!b.prims.empty()) std::memcpy(out->triangles, b.prims.data(), sizeof(PrimRecord) * b.prims.size());      // (memcpy from NULL is undefined even for 0 bytes: UBSan)
    return HRT_OK;
}

void hrt_host_free(HrtBvhBlob *blob) {
    if (!blob) return;
    std::free(blob->nodes); std::free(blob->triangles);
    std::memset(blob, 0, sizeof *blob);
}

}  // extern "C"
