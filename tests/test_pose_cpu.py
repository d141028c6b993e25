"""Time-mode pose pipeline -- the oracle's restatement of slerp / quatToEuler /
constructTransformMatrix (src/Global/RendererTime.cu:296-370, include/Global/DeviceFunctions.cuh:43-148)
against closed forms evaluated independently in float64, its known answers, and the committed fixture.

The reference cannot be built here (CUDA + OptiX headers), so these pin the restatement against the
mathematics the reference's code states, including its two quirks: the positional aggregate returns of
slerp (the {w,x,y,z} expressions land in .x .y .z .w) and Euler angles of a Z*Y*X rotation composed as Rx*Ry*Rz.
"""
import json
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden" / "pose_vectors.json"
PI_F = float(np.float32(3.1415926))


def _rot(deg, axis):
    t = np.deg2rad(deg) * (PI_F / np.pi)          # the reference's truncated PI
    c, s = np.cos(t), np.sin(t)
    m = np.eye(4)
    if axis == 0:
        m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s, s, c
    elif axis == 1:
        m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
    else:
        m[0, 0], m[0, 1], m[1, 0], m[1, 1] = c, -s, s, c
    return m


def _transform64(shift, rot, scale):
    s = np.eye(4)
    s[:3, 3] = shift
    sc = np.diag([scale[0], scale[1], scale[2], 1.0])
    return (s @ _rot(rot[0], 0) @ _rot(rot[1], 1) @ _rot(rot[2], 2) @ sc)[:3].reshape(12)


def _slerp64(q1, q2, t):
    """float64 slerp on fields named x,y,z,w = q[0..3]; returns the reference's placement [w', x', y', z']."""
    q1, q2 = np.asarray(q1, np.float64), np.asarray(q2, np.float64)
    dot = float(q1 @ q2)
    if dot < 0:
        q2, dot = -q2, -dot
    if dot > 0.9995:
        r = q1 + t * (q2 - q1)
        r = r / np.linalg.norm(r)
    else:
        th0 = np.arccos(dot)
        th = th0 * t
        r = (np.cos(th) - dot * np.sin(th) / np.sin(th0)) * q1 + (np.sin(th) / np.sin(th0)) * q2
    x, y, z, w = r
    return np.array([w, x, y, z])


def _euler64(q):
    x, y, z, w = (float(v) for v in q)
    roll = np.arctan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y))
    sinp = 2 * (w * y - z * x)
    pitch = np.copysign(PI_F / 2, sinp) if abs(sinp) >= 1 else np.arcsin(sinp)
    yaw = np.arctan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z))
    return np.array([roll, pitch, yaw]) * 180.0 / PI_F


def test_construct_transform_known_answers(oracle):
    ident = oracle.construct_transform((0, 0, 0), (0, 0, 0), (1, 1, 1))
    assert np.array_equal(ident, np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32))
    # the shipped ground sphere: shift only (files/config.json:26)
    g = oracle.construct_transform((0, 0, -1000.5), (0, 0, 0), (1, 1, 1))
    assert np.array_equal(g, np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, -1000.5], np.float32))
    # 90 degrees about z, uniform scale 2, shift (1,2,3): x -> 2y, y -> -2x
    m = oracle.construct_transform((1, 2, 3), (0, 0, 90), (2, 2, 2)).reshape(3, 4)
    want = np.array([[0, -2, 0, 1], [2, 0, 0, 2], [0, 0, 2, 3]], np.float64)
    assert np.abs(m - want).max() < 2e-7       # cos(90 deg) with PI = 3.1415926f is 2.7e-8, not 0
    assert m[0, 0] != 0.0                      # ... and the restatement keeps that


def test_construct_transform_matches_float64_composition(oracle):
    rng = np.random.default_rng(3)
    worst = 0.0
    for _ in range(300):
        shift = rng.uniform(-3, 3, 3)
        rot = rng.uniform(-180, 180, 3)
        scale = rng.uniform(0.2, 2.5, 3)
        got = oracle.construct_transform(shift, rot, scale)
        want = _transform64(np.float32(shift), np.float32(rot), np.float32(scale))
        worst = max(worst, np.abs(got - want).max())
    assert worst < 5e-6
    # rotation order is Rx * Ry * Rz (DeviceFunctions.cuh:127-130), not Rz * Ry * Rx
    m = oracle.construct_transform((0, 0, 0), (90, 0, 90), (1, 1, 1)).reshape(3, 4)[:, :3]
    assert np.abs(m - (_rot(90, 0) @ _rot(90, 2))[:3, :3]).max() < 1e-6
    assert np.abs(m - (_rot(90, 2) @ _rot(90, 0))[:3, :3]).max() > 0.5


def test_slerp_branches_and_placement(oracle):
    rng = np.random.default_rng(4)
    for _ in range(300):
        q1 = rng.normal(size=4)
        q1 /= np.linalg.norm(q1)
        q2 = rng.normal(size=4)
        q2 /= np.linalg.norm(q2)
        t = rng.uniform(0, 1)
        got = oracle.slerp(q1, q2, t)
        want = _slerp64(np.float32(q1), np.float32(q2), np.float32(t))
        assert np.abs(got - want).max() < 3e-6
        assert abs(np.linalg.norm(got.astype(np.float64)) - 1.0) < 1e-5
    q1 = np.float32([0.1, 0.2, 0.3, 0.9273618])
    q2 = np.float32([0.5, -0.5, 0.5, 0.5])
    # endpoints, with the positional placement: out = [w, x, y, z] of the input
    assert np.abs(oracle.slerp(q1, q2, 0.0) - q1[[3, 0, 1, 2]]).max() < 1e-6
    assert np.abs(oracle.slerp(q1, q2, 1.0) - q2[[3, 0, 1, 2]]).max() < 1e-6
    # dot < 0: the second quaternion is negated (shortest arc)
    assert np.abs(oracle.slerp(q1, -q2, 1.0) - q2[[3, 0, 1, 2]]).max() < 1e-6
    # nearly equal quaternions take the normalised-lerp branch (dot > 0.9995)
    q3 = q1 + np.float32([1e-3, -1e-3, 0, 0])
    got = oracle.slerp(q1, q3, 0.5)
    mid = (q1.astype(np.float64) + q3) / 2
    mid /= np.linalg.norm(mid)
    assert np.abs(got - mid[[3, 0, 1, 2]]).max() < 1e-6


def test_quat_to_euler_against_float64_and_pitch_clamp(oracle):
    rng = np.random.default_rng(5)
    for _ in range(300):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        got = oracle.quat_to_euler(q)
        want = _euler64(np.float32(q))
        d = np.abs(got - want)
        d = np.minimum(d, np.abs(d - 360.0))
        assert d.max() < 2e-2 if abs(want[1]) > 89.0 else d.max() < 2e-3       # asin is ill-conditioned at the poles
    # identity (w = 1): all angles zero; 90 degrees about z: yaw = 90 * (pi / PI)
    assert np.array_equal(oracle.quat_to_euler([0, 0, 0, 1]), np.zeros(3, np.float32))
    s = np.sqrt(0.5)
    e = oracle.quat_to_euler([0, 0, s, s])
    assert abs(e[2] - 90.0) < 1e-4 and abs(e[0]) < 1e-6 and abs(e[1]) < 1e-6
    # |sinp| >= 1: pitch clamps to +-PI/2 -> 90 degrees exactly in the reference's own PI
    e = oracle.quat_to_euler(np.float32([0, s, 0, s]) * np.float32(1.0000002))
    assert abs(abs(e[1]) - 90.0) < 1e-5


def test_pose_transforms_frame_loop(oracle, hrt):
    """factor, shift and the chain slerp -> euler -> matrix, for the frame scalars of RendererTime.cu:425-470."""
    cur = hrt.scenes.particle_states(25, 0)
    nxt = hrt.scenes.particle_states(25, 1)
    dur, count = 0.5, 120
    first = oracle.pose_transforms(cur, nxt, dur, 0, count)
    last = oracle.pose_transforms(cur, nxt, dur, count - 1, count)
    for i in range(25):
        for f, out in ((0, first[i]), (count - 1, last[i])):
            q = oracle.slerp(cur[i, :4], nxt[i, :4], np.float32(f) / np.float32(count - 1))
            rot = oracle.quat_to_euler(q)
            shift = (cur[i, 4:7] + (cur[i, 7:10] * np.float32(dur) / np.float32(count)) * np.float32(f)).astype(np.float32)
            assert np.array_equal(out, oracle.construct_transform(shift, rot, (1, 1, 1)))
    # velocity (0, 0, -2) for 0.5 s: the last frame has moved by 119/120 of -1
    assert np.allclose(last[:, 11] - first[:, 11], -1.0 * 119 / 120, atol=1e-6)
    # offset and scale enter as in constructTransformMatrix(particleOffset + shift, rotate, particleScale)
    moved = oracle.pose_transforms(cur, nxt, dur, 7, count, offset=(1, 2, 3), scale=(2, 2, 2))
    plain = oracle.pose_transforms(cur, nxt, dur, 7, count)
    assert np.allclose(moved[:, [3, 7, 11]] - plain[:, [3, 7, 11]], [1, 2, 3], atol=1e-6)
    assert np.allclose(moved.reshape(-1, 3, 4)[:, :, :3], 2 * plain.reshape(-1, 3, 4)[:, :, :3], atol=1e-6)
    # a single frame: factor = 1 (RendererTime.cu:449-452)
    one = oracle.pose_transforms(cur, nxt, dur, 0, 1)
    q = oracle.slerp(cur[0, :4], nxt[0, :4], 1.0)
    assert np.array_equal(one[0], oracle.construct_transform(cur[0, 4:7], oracle.quat_to_euler(q), (1, 1, 1)))


def test_mesh_mode_drift_is_scale_and_shift_exactly(oracle):
    """Mesh mode (RendererMesh.cu:379-391) poses with rotation (0,0,0): the rotation matrices are exact identities, so
    constructTransformMatrix(shift, 0, scale) is the scale on the diagonal and the shift in the last column, with +0
    everywhere else -- which is what hrt_pose_instances writes in mesh_mode without running the matrix products."""
    rng = np.random.default_rng(9)
    for _ in range(200):
        shift = rng.uniform(-5, 5, 3).astype(np.float32)
        scale = rng.uniform(-2, 2, 3).astype(np.float32)
        m = oracle.construct_transform(shift, (0, 0, 0), scale)
        want = np.array([scale[0], 0, 0, shift[0], 0, scale[1], 0, shift[1], 0, 0, scale[2], shift[2]], np.float32)
        assert np.array_equal(m.view(np.uint32), want.view(np.uint32))          # bit patterns: the zeros are +0


def test_pose_golden_vectors(oracle):
    """Committed outputs of the oracle (tests/golden/make_golden.py): a change of the restatement shows up here."""
    g = json.loads(GOLDEN.read_text())
    cur = np.array(g["current"], np.float32)
    nxt = np.array(g["next"], np.float32)
    for case in g["cases"]:
        got = oracle.pose_transforms(cur, nxt, case["duration"], case["frame"], case["frame_count"], case["offset"], case["scale"])
        want = np.array(case["transforms_hex"], dtype=np.uint32).view(np.float32).reshape(-1, 12)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


# ---- the correctly rounded sinf / cosf / acosf / asinf / atan2f pin (csrc/cr_trig.h on the product side) ----
def _host_trig():
    """Host compile of the product's csrc/cr_trig.h (tests/helpers/cr_trig_host.cpp)."""
    import ctypes as C
    import subprocess
    src = Path(__file__).resolve().parent / "helpers" / "cr_trig_host.cpp"
    so = src.with_name("libcr_trig_host.so")
    hdrs = [src.parents[2] / "nvidia-optix-ray-tracer_amd" / "csrc" / h for h in ("cr_trig.h", "srgb_pow.h")]
    if not so.exists() or so.stat().st_mtime < max(f.stat().st_mtime for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", str(src), "-o", str(so)])
    H = C.CDLL(str(so))
    H.host_trig.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
    H.host_trig_bits.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_void_p]
    return H


def _same_bits(a, b):
    return np.array_equal(a.view(np.uint32)[~(np.isnan(a) & np.isnan(b))], b.view(np.uint32)[~(np.isnan(a) & np.isnan(b))])


def atan2_test_pairs(n, seed):
    """(y, x) pairs for atan2: arbitrary bit patterns, comparable magnitudes, the axes, tiny and huge ratios, signed zeros."""
    rng = np.random.default_rng(seed)
    y = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32).copy()
    x = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32).copy()
    q = n // 4
    y[:q] = rng.normal(size=q).astype(np.float32); x[:q] = rng.normal(size=q).astype(np.float32)
    y[q:2 * q] = (rng.normal(size=q) * 10.0 ** rng.uniform(-30, 30, q)).astype(np.float32)
    x[q:2 * q] = (rng.normal(size=q) * 10.0 ** rng.uniform(-30, 30, q)).astype(np.float32)
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 3.4e38, -3.4e38, 1e-30, 0.5], np.float32)
    sy, sx = np.meshgrid(special, special)
    k = sy.size
    y[2 * q:2 * q + k] = sy.ravel(); x[2 * q:2 * q + k] = sx.ravel()
    return y, x


def test_trig_pin_host_compile_of_the_product_header_matches_the_oracle(oracle):
    """sinf / cosf / acosf / asinf / atan2f of the pose pipeline are pinned on both sides as the correctly rounded float.  The
    oracle gets there with libm doubles + __float128, the product (csrc/cr_trig.h) with the platform's double function + a
    double-double slow path.  Sweeps over bit patterns of ALL floats (strided), dense in [-1, 1] for the inverse functions,
    for the normal path AND with every value forced through the slow path (which in production one call in ~500 000 takes)."""
    H = _host_trig()
    for which in (oracle.TRIG_SIN, oracle.TRIG_COS):
        for slow, stride in ((0, 509), (1, 8191)):
            count = (1 << 32) // stride
            got = np.empty(count, np.float32)
            H.host_trig_bits(which, 0, stride, count, slow, got.ctypes.data)
            assert _same_bits(got, oracle.trig_bits(which, 0, stride, count)), (which, slow)
        # the pose pipeline's own range, densely: |x| <= 4
        for sign in (0, 0x80000000):
            for slow, stride in ((0, 127), (1, 2039)):
                count = 0x40800000 // stride
                got = np.empty(count, np.float32)
                H.host_trig_bits(which, sign, stride, count, slow, got.ctypes.data)
                assert _same_bits(got, oracle.trig_bits(which, sign, stride, count)), (which, sign, slow)
    for which in (oracle.TRIG_ACOS, oracle.TRIG_ASIN):
        for sign in (0, 0x80000000):
            for slow, stride in ((0, 127), (1, 2039)):
                count = 0x3F800000 // stride + 1
                got = np.empty(count, np.float32)
                H.host_trig_bits(which, sign, stride, count, slow, got.ctypes.data)
                assert _same_bits(got, oracle.trig_bits(which, sign, stride, count)), (which, sign, slow)
        ends = np.array([1.0, -1.0, 0.99999994, -0.99999994, 1.0000001, 0.0, -0.0, 1e-45, 2.0, np.nan], np.float32)
        for slow in (0, 1):
            got = np.empty_like(ends)
            H.host_trig(which, ends.ctypes.data, None, ends.size, slow, got.ctypes.data)
            assert _same_bits(got, oracle.trig(which, ends)), (which, slow)
    for slow, n in ((0, 1_000_000), (1, 200_000)):
        y, x = atan2_test_pairs(n, 5)
        got = np.empty(n, np.float32)
        H.host_trig(oracle.TRIG_ATAN2, y.ctypes.data, x.ctypes.data, n, slow, got.ctypes.data)
        assert _same_bits(got, oracle.trig(oracle.TRIG_ATAN2, y, x)), slow


def test_trig_pin_is_the_nearest_float_and_libm_is_within_an_ulp(oracle):
    """(1) The oracle's shortcut (libm double, __float128 only near a midpoint) gives the value decided in __float128
    everywhere sampled, (2) known answers, (3) the platform's float libm -- what a build of the reference would call -- is
    within 1 ULP of the pin on the pose pipeline's ranges (tolerance cross-check; it is NOT what parity is defined on)."""
    stride = 4099
    count = (1 << 32) // stride
    for which in (oracle.TRIG_SIN, oracle.TRIG_COS, oracle.TRIG_ACOS, oracle.TRIG_ASIN):
        assert _same_bits(oracle.trig_bits(which, 0, stride, count), oracle.trig_bits(which, 0, stride, count, force_exact=True)), which
    y, x = atan2_test_pairs(200_000, 9)
    assert _same_bits(oracle.trig(oracle.TRIG_ATAN2, y, x), oracle.trig(oracle.TRIG_ATAN2, y, x, force_exact=True))
    f = np.float32
    pi = f(np.pi)
    assert oracle.trig(oracle.TRIG_SIN, [0.0])[0] == 0 and oracle.trig(oracle.TRIG_COS, [0.0])[0] == 1
    assert oracle.trig(oracle.TRIG_ACOS, [1.0])[0] == 0 and oracle.trig(oracle.TRIG_ACOS, [-1.0])[0] == pi
    assert oracle.trig(oracle.TRIG_ASIN, [1.0])[0] == f(np.pi / 2) and oracle.trig(oracle.TRIG_ACOS, [0.0])[0] == f(np.pi / 2)
    assert oracle.trig(oracle.TRIG_ATAN2, [0.0], [-1.0])[0] == pi and oracle.trig(oracle.TRIG_ATAN2, [1.0], [1.0])[0] == f(np.pi / 4)
    assert oracle.trig(oracle.TRIG_ATAN2, [-1.0], [0.0])[0] == f(-np.pi / 2)
    rng = np.random.default_rng(2)
    ang = rng.uniform(-4, 4, 200_000).astype(np.float32)
    unit = rng.uniform(-1, 1, 200_000).astype(np.float32)

    def ulps(a, b):
        return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64)).max()
    for which, arg in ((oracle.TRIG_SIN, ang), (oracle.TRIG_COS, ang), (oracle.TRIG_ACOS, unit), (oracle.TRIG_ASIN, unit)):
        assert ulps(oracle.trig(which, arg), oracle.trig(which, arg, libm=True)) <= 1, which
    yy, xx = rng.normal(size=200_000).astype(np.float32), rng.normal(size=200_000).astype(np.float32)
    assert ulps(oracle.trig(oracle.TRIG_ATAN2, yy, xx), oracle.trig(oracle.TRIG_ATAN2, yy, xx, libm=True)) <= 1
