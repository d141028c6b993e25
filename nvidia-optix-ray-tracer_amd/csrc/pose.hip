// pose.hip -- the per-frame particle pose update of the reference's Time mode, on the device (gfx950).
//
// Replaces the host loop at src/Global/RendererTime.cu:436-472 and the H2D copy of the instance array that
// follows it (:476-478): per particle
//     factor  = frame / (frame_count - 1)                                    (:449-452)
//     shift   = position + (velocity * duration / frame_count) * frame       (:455-458)
//     quat    = slerp(cur.quat, next.quat, factor)                           (:296-340)
//     rotate  = quatToEuler(quat)              [degrees]                     (:343-370)
//     T       = Shift(offset + shift) * (Rx * Ry * Rz) * Scale(scale)        (include/Global/DeviceFunctions.cuh:43-148)
// and T's first three rows are written into OptixInstance.transform.
//
// The float operation order is the reference's (4x4 products accumulate left to right starting from 0).
// Reference behaviour kept on purpose:
//   * the aggregate returns of slerp put the w-expression into .x, x into .y, y into .z, z into .w (:313-318,
//     :334-339), and quatToEuler reads the result by name -- the components arrive rotated by one place;
//   * the Euler angles are those of a Z*Y*X rotation but are composed as Rx*Ry*Rz (:127-130).
// sin / cos / acos / asin / atan2: the reference calls its platform's float functions -- third-party arithmetic that differs
// by an ULP between platforms -- pinned here and in the oracle as the CORRECTLY ROUNDED float of the exact value
// (csrc/cr_trig.h), so that every transform entry, and with it every frame rendered from posed instances, is bit-exact
// against the oracle.  One thread per particle; 48 + 48 B read, 48 B written.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cr_trig.h"
#include "device_types.h"

#pragma clang fp contract(off)

namespace hrt {

namespace {

constexpr float kPi = 3.1415926f;                 // PI, include/Global/DeviceFunctions.cuh:19

__device__ __forceinline__ float sin_f(float x) { return sinf_cr(x); }
__device__ __forceinline__ float cos_f(float x) { return cosf_cr(x); }
__device__ __forceinline__ float acos_f(float x) { return acosf_cr(x); }
__device__ __forceinline__ float asin_f(float x) { return asinf_cr(x); }
__device__ __forceinline__ float atan2_f(float y, float x) { return atan2f_cr(y, x); }

struct Quat { float x, y, z, w; };
struct Mat4 { float m[4][4]; };

__device__ __forceinline__ Mat4 mul(const Mat4 &a, const Mat4 &b) {
    Mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float sum = 0.0f;
            for (int n = 0; n < 4; ++n) sum += a.m[i][n] * b.m[n][j];
            r.m[i][j] = sum;
        }
    return r;
}

__device__ __forceinline__ Mat4 identity() {
    Mat4 r;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = i == j ? 1.0f : 0.0f;
    return r;
}

__device__ __forceinline__ Mat4 rotation(float degree, int axis) {
    const float theta = degree * kPi / 180.0f;
    const float c = cos_f(theta), s = sin_f(theta);
    Mat4 r = identity();
    if (axis == 0) { r.m[1][1] = c; r.m[1][2] = -s; r.m[2][1] = s; r.m[2][2] = c; }
    else if (axis == 1) { r.m[0][0] = c; r.m[0][2] = s; r.m[2][0] = -s; r.m[2][2] = c; }
    else { r.m[0][0] = c; r.m[0][1] = -s; r.m[1][0] = s; r.m[1][1] = c; }
    return r;
}

// slerp with the reference's component placement: returns {w-expr, x-expr, y-expr, z-expr} as {x, y, z, w}
__device__ __forceinline__ Quat slerp(Quat q1, Quat q2, float t) {
    float dot = q1.w * q2.w + q1.x * q2.x + q1.y * q2.y + q1.z * q2.z;
    if (dot < 0.0f) { q2.w = -q2.w; q2.x = -q2.x; q2.y = -q2.y; q2.z = -q2.z; dot = -dot; }
    if (dot > 0.9995f) {
        Quat r = {q1.w + t * (q2.w - q1.w), q1.x + t * (q2.x - q1.x), q1.y + t * (q2.y - q1.y), q1.z + t * (q2.z - q1.z)};
        const float mag = sqrtf(r.w * r.w + r.x * r.x + r.y * r.y + r.z * r.z);
        if (mag > 0.0f) { r.w /= mag; r.x /= mag; r.y /= mag; r.z /= mag; }
        return r;
    }
    const float theta_0 = acos_f(dot);
    const float theta = theta_0 * t;
    const float sin_theta = sin_f(theta);
    const float sin_theta_0 = sin_f(theta_0);
    const float s0 = cos_f(theta) - dot * sin_theta / sin_theta_0;
    const float s1 = sin_theta / sin_theta_0;
    return {(s0 * q1.w) + (s1 * q2.w), (s0 * q1.x) + (s1 * q2.x), (s0 * q1.y) + (s1 * q2.y), (s0 * q1.z) + (s1 * q2.z)};
}

__device__ __forceinline__ void quat_to_euler_degrees(const Quat &q, float *deg) {
    const float sinr_cosp = 2.0f * (q.w * q.x + q.y * q.z);
    const float cosr_cosp = 1.0f - 2.0f * (q.x * q.x + q.y * q.y);
    const float roll = atan2_f(sinr_cosp, cosr_cosp);
    const float sinp = 2.0f * (q.w * q.y - q.z * q.x);
    const float pitch = fabsf(sinp) >= 1.0f ? copysignf(kPi / 2.0f, sinp) : asin_f(sinp);
    const float siny_cosp = 2.0f * (q.w * q.z + q.x * q.y);
    const float cosy_cosp = 1.0f - 2.0f * (q.y * q.y + q.z * q.z);
    const float yaw = atan2_f(siny_cosp, cosy_cosp);
    deg[0] = roll * 180.0f / kPi; deg[1] = pitch * 180.0f / kPi; deg[2] = yaw * 180.0f / kPi;
}

}  // namespace

__global__ __launch_bounds__(256) void k_pose_instances(PoseArgs a) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.n) return;
    const float4 qc = a.current[3 * (size_t)i], pc = a.current[3 * (size_t)i + 1], vc = a.current[3 * (size_t)i + 2];
    const float4 qn = a.next[3 * (size_t)i];
    // HrtParticleState: {quat.xyzw}{position.xyz, velocity.x}{velocity.yz, pad, pad}
    const float pos[3] = {pc.x, pc.y, pc.z}, vel[3] = {pc.w, vc.x, vc.y};
    const float fcount = (float)a.frame_count, fframe = (float)a.frame;
    const float factor = a.frame_count > 1u ? fframe / (float)(a.frame_count - 1u) : 1.0f;
    float shift[3];
    for (int k = 0; k < 3; ++k) {
        const float total = vel[k] * a.duration;
        const float per_frame = total / fcount;
        shift[k] = a.mesh_mode ? a.offset[k] + per_frame * fframe : a.offset[k] + (pos[k] + per_frame * fframe);
    }
    float4 *out = reinterpret_cast<float4 *>(reinterpret_cast<unsigned char *>(a.instances) + (size_t)(a.first_instance + i) * 80u);
    if (a.mesh_mode) {
        // Mesh mode (RendererMesh.cu:379-391): rotation (0,0,0) -- the rotation matrices are exact identities (cos 0 = 1,
        // sin 0 = 0), so the product Shift * R * Scale is the scale on the diagonal and the shift in the last column
        out[0] = make_float4(a.scale[0], 0.0f, 0.0f, shift[0]);
        out[1] = make_float4(0.0f, a.scale[1], 0.0f, shift[1]);
        out[2] = make_float4(0.0f, 0.0f, a.scale[2], shift[2]);
        return;
    }
    const Quat q = slerp(Quat{qc.x, qc.y, qc.z, qc.w}, Quat{qn.x, qn.y, qn.z, qn.w}, factor);
    float deg[3];
    quat_to_euler_degrees(q, deg);
    Mat4 s = identity(), sc = identity();
    s.m[0][3] = shift[0]; s.m[1][3] = shift[1]; s.m[2][3] = shift[2];
    sc.m[0][0] = a.scale[0]; sc.m[1][1] = a.scale[1]; sc.m[2][2] = a.scale[2];
    const Mat4 r = mul(mul(rotation(deg[0], 0), rotation(deg[1], 1)), rotation(deg[2], 2));
    const Mat4 t = mul(mul(s, r), sc);
    out[0] = make_float4(t.m[0][0], t.m[0][1], t.m[0][2], t.m[0][3]);
    out[1] = make_float4(t.m[1][0], t.m[1][1], t.m[1][2], t.m[1][3]);
    out[2] = make_float4(t.m[2][0], t.m[2][1], t.m[2][2], t.m[2][3]);
}

// the five pinned functions over arrays (hrt_debug_trig: the parity tests sweep them against the oracle)
__global__ __launch_bounds__(256) void k_debug_trig(int which, const float *a, const float *b, uint32_t first, uint32_t stride, uint64_t n, int force_slow, float *out) {
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256ull) {
        const float x = a ? a[i] : __uint_as_float(first + (uint32_t)(i * stride));
        const float y = b ? b[i] : 0.0f;
        float r;
        switch (which) {
            case 0: r = sinf_cr(x, force_slow != 0); break;
            case 1: r = cosf_cr(x, force_slow != 0); break;
            case 2: r = acosf_cr(x, force_slow != 0); break;
            case 3: r = asinf_cr(x, force_slow != 0); break;
            default: r = atan2f_cr(x, y, force_slow != 0); break;
        }
        out[i] = r;
    }
}
void launch_debug_trig(int which, const float *a, const float *b, uint32_t first, uint32_t stride, uint64_t n, int force_slow, float *out, hipStream_t s) {
    if (n == 0) return;
    const uint64_t blocks = (n + 255u) / 256u;
    hipLaunchKernelGGL(k_debug_trig, dim3((uint32_t)(blocks < 65536u ? blocks : 65536u)), dim3(256), 0, s, which, a, b, first, stride, n, force_slow, out);
}

void launch_pose_instances(const PoseArgs &a, hipStream_t s) {
    if (a.n == 0) return;
    hipLaunchKernelGGL(k_pose_instances, dim3((a.n + 255u) / 256u), dim3(256), 0, s, a);
}

}  // namespace hrt
