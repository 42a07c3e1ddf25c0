"""Checks that do NOT go through the oracle's own arithmetic (oracle and HIP path share
include/gie_math.h and one author — a wrong rounding or sign there is invisible to every parity test).

1. include/gie_math.h against double precision and hand-derived vectors: quaternion -> SE3 (se3.cuh:47-77),
   rigid inverse (:91-108), point transform (:123-149), pos2coord = floorf(p/w + 0.5f) (local_batch.h:250-258,
   pivots :129-166) with negative coordinates and half-voxel ties, the pinned atan2 polynomial against libm.
2. The projective OGM of the oracle (and through it of the HIP path, which equals it bit for bit) against an
   INDEPENDENT float64 numpy statement of the reference's kernels, written from the reference sources
   (vlp16_fast.cu:8-87 + vlp16_helper.h:35-65, realsense_fast.cu:9-94 + camera_helper.h:11-23): labels must agree
   on every voxel whose decision quantities are not within rounding distance of a threshold.
"""
import ctypes as C
import math
import os
import subprocess
from fractions import Fraction

import numpy as np
import pytest

import gie
from gie import scenes
from oracle_py import OracleMapper

HERE = os.path.dirname(os.path.abspath(__file__))
SHIM_SRC = os.path.join(HERE, "shims", "math_shim.c")
SHIM_SO = os.path.join(HERE, "shims", "libmath_shim.so")


def _load_shim(src, so, hdr):
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", so, "-lm"])
    lib = C.CDLL(so)
    fp = C.POINTER(C.c_float)
    lib.ms_from_quat.argtypes = [fp, fp, fp]
    lib.ms_inv.argtypes = [fp, fp]
    lib.ms_apply.argtypes = [fp, fp, fp]
    lib.ms_pos2coord.argtypes = [C.c_float, C.c_float]
    lib.ms_pos2coord.restype = C.c_int
    lib.ms_atan2f.argtypes = [C.c_float, C.c_float]
    lib.ms_atan2f.restype = C.c_float
    lib.ms_point_ok.argtypes = [C.c_float, C.c_float, C.c_float]
    lib.ms_point_ok.restype = C.c_int
    return lib


def _product_math():
    return _load_shim(SHIM_SRC, SHIM_SO, os.path.join(os.path.dirname(HERE), "include", "gie_math.h"))


def _oracle_math():
    return _load_shim(os.path.join(HERE, "shims", "oracle_math_shim.c"), os.path.join(HERE, "shims", "liboracle_math_shim.so"),
                      os.path.join(os.path.dirname(HERE), "oracle", "oracle_math.h"))


# every check below runs on BOTH statements of the geometry: the product's include/gie_math.h and the oracle's own oracle/oracle_math.h
@pytest.fixture(scope="module", params=["product", "oracle"])
def ms(request):
    return _product_math() if request.param == "product" else _oracle_math()


def test_the_two_statements_of_the_geometry_agree_bit_for_bit():
    """include/gie_math.h (kernels) and oracle/oracle_math.h (oracle) share no line since round 5; voxelisation floors their
    results, so they have to round identically everywhere: random and hand-picked inputs, every output compared as raw bits."""
    a, b = _product_math(), _oracle_math()
    rng = np.random.default_rng(2025)
    bits = lambda x: np.asarray(x, np.float32).view(np.uint32)
    quats = [(1, 0, 0, 0), (0, 0, 0, 1), (0, 1, 0, 0), (0.5, 0.5, 0.5, 0.5), (math.sqrt(0.5), 0, 0, -math.sqrt(0.5))]
    for _ in range(3000):
        v = rng.standard_normal(4)
        quats.append(tuple(v / np.linalg.norm(v)))
    for q in quats:
        t = rng.uniform(-2000, 2000, 3) * rng.choice([1.0, 1e-3, 1.0])
        ma, mb = _se3(a, q, t), _se3(b, q, t)
        assert np.array_equal(bits(ma), bits(mb)), q
        ia, ib = np.zeros(12, np.float32), np.zeros(12, np.float32)
        a.ms_inv(_f(ma.ravel())[1], ia.ctypes.data_as(C.POINTER(C.c_float)))
        b.ms_inv(_f(mb.ravel())[1], ib.ctypes.data_as(C.POINTER(C.c_float)))
        assert np.array_equal(bits(ia), bits(ib)), q
        for _k in range(4):
            pt = rng.uniform(-300, 300, 3)
            oa, ob = np.zeros(3, np.float32), np.zeros(3, np.float32)
            for lib, m, o in ((a, ia, oa), (b, ib, ob)):
                lib.ms_apply(_f(m)[1], _f(pt)[1], o.ctypes.data_as(C.POINTER(C.c_float)))
            assert np.array_equal(bits(oa), bits(ob)), (q, pt)
    ys = np.concatenate([rng.standard_normal(40000), [0.0, -0.0, 1.0, -1.0, 1e-30, -1e-30, 1e30, 0.41421357, 2.4142137, 0.41421354, 2.4142134]]).astype(np.float32)
    xs = np.concatenate([rng.standard_normal(40000), [0.0, 1.0, -1.0, 0.0, 1.0, -1.0, 1.0, 1.0, 1.0, 1.0, 1.0]]).astype(np.float32)
    ta = np.array([a.ms_atan2f(float(y), float(x)) for y, x in zip(ys, xs)], np.float32)
    tb = np.array([b.ms_atan2f(float(y), float(x)) for y, x in zip(ys, xs)], np.float32)
    assert np.array_equal(bits(ta), bits(tb))
    for w in (0.05, 0.1, 0.2, 0.25):
        ps = np.concatenate([rng.uniform(-60, 60, 20000), (np.arange(-400, 400) + 0.5) * w, np.arange(-400, 400) * w]).astype(np.float32)
        assert all(a.ms_pos2coord(float(p), w) == b.ms_pos2coord(float(p), w) for p in ps)
    for g in [(0, 0, 0), (1e6, -1e6, 5), (1.0000001e6, 0, 0), (0, float("nan"), 0), (float("inf"), 0, 0), (0, 0, -float("inf")), (3, 4, -2e6)]:
        assert a.ms_point_ok(*[float(v) for v in g]) == b.ms_point_ok(*[float(v) for v in g]), g


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _rot64(q):
    """Rotation matrix of a unit quaternion (w, x, y, z) in float64, the textbook form."""
    w, x, y, z = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _se3(ms, q, t):
    qa, qp = _f(q)
    ta, tp = _f(t)
    out, op = _f(np.zeros(12))
    ms.ms_from_quat(qp, tp, op)
    return out.reshape(3, 4)


def test_quaternion_to_se3_against_float64(ms):
    rng = np.random.default_rng(11)
    quats = [(1, 0, 0, 0), (0, 0, 0, 1), (0, 0, 0, -1), (math.cos(math.pi / 2), 0, 0, math.sin(math.pi / 2)),   # yaw +pi (two spellings)
             (math.cos(-math.pi / 2), 0, 0, math.sin(-math.pi / 2)), (math.sqrt(0.5), math.sqrt(0.5), 0, 0),
             (math.sqrt(0.5), 0, math.sqrt(0.5), 0), (math.sqrt(0.5), 0, 0, -math.sqrt(0.5)), (0.5, 0.5, 0.5, 0.5)]
    for _ in range(200):
        v = rng.standard_normal(4)
        quats.append(tuple(v / np.linalg.norm(v)))
    for q in quats:
        q32 = np.asarray(q, np.float32)
        t = rng.uniform(-500, 500, 3)
        m = _se3(ms, q32, t)
        r64 = _rot64(q32.astype(np.float64) )
        assert np.allclose(m[:, :3], r64, atol=4e-7, rtol=0), (q, m[:, :3] - r64)
        assert np.array_equal(m[:, 3], np.asarray(t, np.float32))
        # a yaw of +-pi maps x to -x and y to -y, z stays
    m = _se3(ms, (0, 0, 0, 1), (0, 0, 0))
    assert np.array_equal(np.sign(np.diag(m[:, :3])), [-1, -1, 1])
    # hand-derived: +90 deg about z takes (1,0,0) to (0,1,0)
    m = _se3(ms, (math.sqrt(0.5), 0, 0, math.sqrt(0.5)), (0, 0, 0))
    ma, mp = _f(m.reshape(-1))
    pa, pp = _f([1, 0, 0])
    o, op = _f(np.zeros(3))
    ms.ms_apply(mp, pp, op)
    assert np.allclose(o, [0, 1, 0], atol=2e-7)


def test_quaternion_to_se3_rounds_like_the_doubled_component_form(ms):
    """gie_se3_from_quat doubles the PRODUCTS of quaternion components; the reference doubles a component first and multiplies
    then (se3.cuh:53-75).  Doubling is exact in binary floating point, so the two must agree in every bit — which is what makes the
    header free to write the matrix its own way."""
    rng = np.random.default_rng(12)
    f = np.float32
    for _ in range(3000):
        q = rng.standard_normal(4)
        q = (q / np.linalg.norm(q)).astype(np.float32)
        if rng.random() < 0.1:
            q[rng.integers(0, 4)] = f(0.0)
        w, x, y, z = [f(v) for v in q]
        x2, y2, z2 = f(2) * x, f(2) * y, f(2) * z
        wx, wy, wz = x2 * w, y2 * w, z2 * w
        xx, xy, xz = x2 * x, y2 * x, z2 * x
        yy, yz, zz = y2 * y, z2 * y, z2 * z
        want = np.array([[f(1) - (yy + zz), xy - wz, xz + wy], [xy + wz, f(1) - (xx + zz), yz - wx], [xz - wy, yz + wx, f(1) - (xx + yy)]], np.float32)
        got = _se3(ms, q, (0.0, 0.0, 0.0))[:, :3]
        assert got.tobytes() == want.tobytes(), (q, got, want)


def test_rigid_inverse_and_transform_against_float64(ms):
    rng = np.random.default_rng(12)
    for _ in range(200):
        v = rng.standard_normal(4)
        q = (v / np.linalg.norm(v)).astype(np.float32)
        t = rng.uniform(-1000, 1000, 3).astype(np.float32)
        m = _se3(ms, q, t)
        ma, mp = _f(m.reshape(-1))
        inv, ip = _f(np.zeros(12))
        ms.ms_inv(mp, ip)
        inv = inv.reshape(3, 4)
        a = np.vstack([m.astype(np.float64), [0, 0, 0, 1]])
        b = np.vstack([inv.astype(np.float64), [0, 0, 0, 1]])
        assert np.allclose(b @ a, np.eye(4), atol=2e-4), (b @ a)
        assert np.allclose(b[:3, :3], a[:3, :3].T, atol=0)              # R^T exactly
        p = rng.uniform(-50, 50, 3).astype(np.float32)
        pa, pp = _f(p)
        o, op = _f(np.zeros(3))
        ms.ms_apply(mp, pp, op)
        want = a[:3, :3] @ p.astype(np.float64) + a[:3, 3]
        assert np.allclose(o, want, rtol=2e-6, atol=2e-4)


def test_pos2coord_hand_vectors_ties_and_negative_coordinates(ms):
    # floorf(p / w + 0.5f): round half UP (towards +inf), also for negative coordinates (local_batch.h:250-258)
    w = 0.5                                                # exactly representable: the ties below are exact
    for p, want in ((0.0, 0), (0.24, 0), (0.25, 1), (0.26, 1), (0.74, 1), (0.75, 2), (-0.24, 0), (-0.25, 0), (-0.26, -1),
                    (-0.74, -1), (-0.75, -1), (-0.76, -2), (1000.25, 2001), (-1000.25, -2000), (-1000.26, -2001)):
        assert ms.ms_pos2coord(p, w) == want, (p, want, ms.ms_pos2coord(p, w))
    # any voxel width: IEEE binary32 evaluation (numpy's float32 ops are correctly rounded) ...
    rng = np.random.default_rng(13)
    for w in (0.05, 0.1, 0.2, 0.0625):
        ps = np.concatenate([rng.uniform(-2000, 2000, 4000), (rng.integers(-40000, 40000, 2000) + 0.5) * w,
                             rng.integers(-40000, 40000, 2000) * w]).astype(np.float32)
        w32 = np.float32(w)
        for p in ps:
            got = ms.ms_pos2coord(float(p), w)
            want = int(np.floor(np.float32(np.float32(p / w32) + np.float32(0.5))))
            assert got == want, (p, w, got, want)
            # ... and never more than a rounding away from the exact value of floor(p/w + 1/2) on the stored inputs
            exact = math.floor(Fraction(float(p)) / Fraction(float(w32)) + Fraction(1, 2))
            assert abs(got - exact) <= 1
            qv = float(Fraction(float(p)) / Fraction(float(w32)) + Fraction(1, 2))
            tol = 4.0 * abs(qv) * 2.0 ** -23 + 1e-6        # two roundings of a value of this size
            if tol < qv % 1.0 < 1 - tol:
                assert got == exact, (p, w, got, exact)


def test_pivots_follow_the_reference_formulas():
    """_pvt = pos2coord(pos) - size/2, _msg_origin = _pvt * w (local_batch.h:129-142), through the oracle's C-ABI."""
    for size in ((32, 32, 16), (37, 29, 11)):
        cfg = gie.make_config(0.1, size, cutoff_dist=1.0)
        o = OracleMapper(cfg)
        try:
            for pos in ((0.0, 0.0, 0.0), (-1234.56, 789.01, -3.3), (0.05, -0.05, 0.149), (12.349, -7.651, 2.25)):
                o.set_pose(pos)
                want = tuple(int(math.floor(float(np.float32(np.float32(p) / np.float32(0.1)) + np.float32(0.5)))) - s // 2 for p, s in zip(pos, size))
                assert tuple(o.pivot()) == want, (pos, o.pivot(), want)
        finally:
            o.close()


def test_atan2_polynomial_against_libm(ms):
    rng = np.random.default_rng(14)
    ys = np.concatenate([rng.standard_normal(5000) * 10.0 ** rng.integers(-6, 6, 5000), [0.0, -0.0, 1.0, -1.0, 0.0, 0.0, 1e-30, -1e-30, 1e30]])
    xs = np.concatenate([rng.standard_normal(5000) * 10.0 ** rng.integers(-6, 6, 5000), [1.0, -1.0, 0.0, 0.0, 0.0, -0.0, -1e30, 1e30, -1e-30]])
    worst = 0.0
    for y, x in zip(ys.astype(np.float32), xs.astype(np.float32)):
        got = ms.ms_atan2f(float(y), float(x))
        if x == 0 and y == 0:
            assert got == 0.0
            continue
        want = math.atan2(float(y), float(x))
        err = abs(got - want)
        if abs(want) > 3.0:                               # +pi and -pi are the same direction
            err = min(err, abs(abs(got) - abs(want)))
        worst = max(worst, err)
        assert err < 6e-7, (y, x, got, want)              # 2 ulp of a value up to pi (the header's claim)
        assert (got >= 0) == (want >= 0) or abs(want) < 1e-6 or abs(abs(want) - math.pi) < 1e-6
    # quadrant conventions by hand
    assert abs(ms.ms_atan2f(1.0, 1.0) - math.pi / 4) < 3e-7 and abs(ms.ms_atan2f(1.0, -1.0) - 3 * math.pi / 4) < 3e-7
    assert abs(ms.ms_atan2f(-1.0, -1.0) + 3 * math.pi / 4) < 3e-7 and abs(ms.ms_atan2f(-1.0, 1.0) + math.pi / 4) < 3e-7
    assert ms.ms_atan2f(0.0, -1.0) == np.float32(math.pi) and ms.ms_atan2f(1.0, 0.0) == np.float32(math.pi / 2)


# ------------------------------------------------------------------ projective OGM, float64, from the reference sources

def _voxel_positions(pos, size, w):
    pvt = [int(math.floor(float(np.float32(np.float32(p) / np.float32(w)) + np.float32(0.5)))) - s // 2 for p, s in zip(pos, size)]
    z, y, x = np.meshgrid(np.arange(size[2]), np.arange(size[1]), np.arange(size[0]), indexing="ij")
    g = np.stack([x + pvt[0], y + pvt[1], z + pvt[2]], -1).astype(np.float64) * float(np.float32(w))     # coord2pos: crd * voxel_width
    return g


def _g2l(pos, q):
    r = _rot64(np.asarray(q, np.float64) / np.linalg.norm(q))
    return r.T, -r.T @ np.asarray(pos, np.float64)


def _multiscan_f64(pos, q, size, w, img, theta_inc, theta_min, phi_inc, phi_min, min_h, max_h, eps_m=1e-4):
    """VLP_FAST::setLocalOccupancy + VLP_HELPER::G2L in float64.  Returns (labels, sure) — sure = no decision quantity of
    the voxel is within rounding distance of its threshold."""
    g = _voxel_positions(pos, size, w)
    rt, t = _g2l(pos, q)
    l = g @ rt.T + t
    ring, scan = img.shape
    theta = np.arctan2(l[..., 1], l[..., 0])
    tt = (theta - theta_min) / theta_inc + 0.5
    ti = np.floor(tt).astype(np.int64) % scan
    hor = np.sqrt(l[..., 0] ** 2 + l[..., 1] ** 2)
    phi = np.arctan2(l[..., 2], hor)
    pp = (phi - phi_min) / phi_inc + 0.5
    pi_ = np.floor(pp).astype(np.int64)
    lab = np.zeros(g.shape[:-1], np.int8)
    inr = (pi_ >= 0) & (pi_ < ring)
    real = img.astype(np.float64)[np.clip(pi_, 0, ring - 1), ti]
    ok = inr & ~np.isnan(real) & (real > 0.3)
    ideal = hor
    free = ok & (ideal < real - 0.3)
    gap = ok & (ideal >= real - 0.3) & (ideal < real - 0.1)
    beyond = ok & (ideal > real + 0.1)
    band = ok & ~free & ~gap & ~beyond                     # the height gate decides
    occ = band & (g[..., 2] >= min_h) & (g[..., 2] <= max_h)
    lab[free] = 1
    lab[occ] = 2
    # a voxel is "sure" when none of the quantities its label hangs on is within rounding distance of its threshold
    # (bins: in units of a bin; ranges and heights: metres; fp32 evaluation of the reference is off by ~1e-6 of those)
    with np.errstate(invalid="ignore"):
        m_bin = np.minimum(np.minimum(tt % 1.0, 1 - tt % 1.0), np.minimum(pp % 1.0, 1 - pp % 1.0))
        m_rng = np.minimum(np.minimum(np.abs(ideal - (real - 0.3)), np.abs(ideal - (real - 0.1))), np.abs(ideal - (real + 0.1)))
        m_h = np.minimum(np.abs(g[..., 2] - min_h), np.abs(g[..., 2] - max_h))
    sure = (m_bin > 2e-3) & (hor > 1e-3) & (~ok | (m_rng > eps_m)) & (~band | (m_h > 1e-5)) & (np.isnan(real) | (np.abs(real - 0.3) > 1e-5))
    return lab, sure


def _depth_f64(pos, q, size, w, dep, cx, cy, fx, fy, valid_nan, min_h, max_h, eps_m=1e-4):
    """REALSENSE_FAST::setLocalOccupancy + CAM_HELPER::G2L in float64."""
    g = _voxel_positions(pos, size, w)
    rt, t = _g2l(pos, q)
    l = g @ rt.T + t
    rows, cols = dep.shape
    ideal = l[..., 0]
    with np.errstate(divide="ignore", invalid="ignore"):
        px = -l[..., 1] * fx / ideal + cx + 0.5
        py = -l[..., 2] * fy / ideal + cy + 0.5
    pxi, pyi = np.nan_to_num(np.floor(px), nan=-1.0, posinf=-1.0, neginf=-1.0), np.nan_to_num(np.floor(py), nan=-1.0, posinf=-1.0, neginf=-1.0)
    front = (ideal > 0.3) & (ideal <= 6.0)
    inimg = front & (pxi >= 0) & (pxi < cols) & (pyi >= 0) & (pyi < rows)
    real = dep.astype(np.float64)[np.clip(pyi, 0, rows - 1).astype(np.int64), np.clip(pxi, 0, cols - 1).astype(np.int64)]
    ok = inimg & ~(real <= 0.21)
    if valid_nan:
        real = np.where(np.isnan(real), 1000.0, real)
    else:
        ok &= ~np.isnan(real)
    wv = float(np.float32(w))
    free = ok & (ideal < real - wv)
    beyond = ok & (ideal > real + wv)
    occ = ok & ~free & ~beyond & (g[..., 2] >= min_h) & (g[..., 2] <= max_h)
    lab = np.zeros(g.shape[:-1], np.int8)
    lab[free] = 1
    lab[occ] = 2
    band = ok & ~free & ~beyond
    with np.errstate(invalid="ignore"):
        m_pix = np.minimum(np.minimum(px % 1.0, 1 - px % 1.0), np.minimum(py % 1.0, 1 - py % 1.0))
        m_front = np.minimum(np.abs(ideal - 0.3), np.abs(ideal - 6.0))
        m_rng = np.minimum(np.abs(ideal - (real - wv)), np.abs(ideal - (real + wv)))
        m_h = np.minimum(np.abs(g[..., 2] - min_h), np.abs(g[..., 2] - max_h))
        m_real = np.abs(real - 0.21)
    sure = (m_front > eps_m) & (~front | (m_pix > 2e-3)) & (~inimg | np.isnan(real) | (m_real > 1e-5)) & (~ok | (m_rng > eps_m)) & (~band | (m_h > 1e-5))
    return lab, sure


def _compare(lab_ref, lab_f64, sure, what):
    assert sure.mean() > 0.8, "%s: only %.3f of the voxels are clear of every threshold" % (what, sure.mean())
    bad = sure & (lab_ref != lab_f64)
    assert not bad.any(), "%s: %d voxels differ from the float64 statement of the reference (first at %s)" % (
        what, int(bad.sum()), np.argwhere(bad)[0])
    assert (lab_ref != lab_f64).mean() < 0.01             # and the rest are a handful of ties
    return int((lab_f64 == 1).sum()), int((lab_f64 == 2).sum())


def test_multiscan_ogm_against_float64_restatement_of_the_reference(oracle_lib):
    size, w = (48, 48, 16), 0.1
    world = scenes.BoxWorld(3, extent=(4.0, 4.0, 1.2), n_boxes=30, toggle_frac=0.0)
    cfg = gie.make_config(w, size, cutoff_dist=1.0, ogm_min_h=-0.6, ogm_max_h=0.7)
    o = OracleMapper(cfg)
    seen = [0, 0]
    try:
        for k in (0, 3, 7):
            pos, q = scenes.pose(k, w, delta_vox=5, yaw_deg=37.0)
            pts, _ = scenes.lidar_frame(world, k, pos, q, rings=16, az=1800, phi_min_deg=-15.0, phi_inc_deg=2.0, max_range=30.0)
            img = scenes.range_image(pts, scan_num=440, ring_num=16, phi_min_deg=-15.0, phi_inc_deg=2.0)
            kw = dict(theta_inc=2.0 * np.pi / 440, theta_min=-np.pi, phi_inc=np.radians(2.0), phi_min=np.radians(-15.0))
            o.set_pose(pos, q)
            o.ogm_multiscan(img, **kw)
            lab = o.read_ogm()["inst_type"]
            f32 = lambda v: float(np.float32(v))
            want, sure = _multiscan_f64(pos, q, size, w, img, f32(kw["theta_inc"]), f32(kw["theta_min"]), f32(kw["phi_inc"]), f32(kw["phi_min"]), f32(-0.6), f32(0.7))
            nf, no = _compare(lab, want, sure, "multiscan frame %d" % k)
            seen[0] += nf; seen[1] += no
            o.fuse(); o.batch_edt(); o.merge()
        assert seen[0] > 1000 and seen[1] > 50            # the scenes exercise both labels
    finally:
        o.close()


def test_depth_ogm_against_float64_restatement_of_the_reference(oracle_lib):
    size, w = (48, 40, 24), 0.05
    world = scenes.BoxWorld(4, extent=(2.0, 2.0, 1.0), n_boxes=25, toggle_frac=0.0)
    cfg = gie.make_config(w, size, cutoff_dist=1.0, ogm_min_h=-0.4, ogm_max_h=0.5)
    o = OracleMapper(cfg)
    seen = [0, 0]
    try:
        for k, valid_nan in ((1, False), (2, True), (5, False)):
            pos, q = scenes.pose(k, w, delta_vox=5, yaw_deg=47.0)
            dep = scenes.depth_frame(world, k, pos, q, rows=60, cols=80, fx=70.0, fy=70.0, cx=39.5, cy=29.5)
            if valid_nan:
                dep = dep.copy(); dep[10:20, 30:50] = np.nan
            o.set_pose(pos, q)
            o.ogm_depth(dep, cx=39.5, cy=29.5, fx=70.0, fy=70.0, valid_nan=valid_nan)
            lab = o.read_ogm()["inst_type"]
            want, sure = _depth_f64(pos, q, size, w, dep, 39.5, 29.5, 70.0, 70.0, valid_nan, float(np.float32(-0.4)), float(np.float32(0.5)))
            nf, no = _compare(lab, want, sure, "depth frame %d" % k)
            seen[0] += nf; seen[1] += no
            o.fuse(); o.batch_edt(); o.merge()
        assert seen[0] > 1000 and seen[1] > 50            # the scenes exercise both labels
    finally:
        o.close()


def _scan2d_f64(pos, q, size, w, rng, theta_inc, theta_min, min_h, max_h, eps_m=1e-4):
    """HOKUYO_FAST::setLocalOccupancy (hokuyo_fast.cu:9-81) + SCAN_HELPER::G2L (hokuyo_helper.h:17-33) in float64: the voxel centre
    in the sensor frame; theta = atan2(y, x) -> bin floor((theta - theta_min) / theta_inc + 0.5) modulo scan_num; the voxel is looked
    at only when |z| < voxel width (depth = horizontal range, else -1); range NaN or <= 0.3 -> nothing; ideal < real - 0.3 -> FREE;
    ideal > real + 0.3 -> nothing; else OCCUPIED inside the height gate."""
    g = _voxel_positions(pos, size, w)
    rt, t = _g2l(pos, q)
    l = g @ rt.T + t
    n = rng.shape[0]
    theta = np.arctan2(l[..., 1], l[..., 0])
    tt = (theta - theta_min) / theta_inc + 0.5
    ti = np.floor(tt).astype(np.int64) % n
    wv = float(np.float32(w))
    inplane = np.abs(l[..., 2]) < wv
    ideal = np.sqrt(l[..., 0] ** 2 + l[..., 1] ** 2)
    real = rng.astype(np.float64)[ti]
    ok = inplane & ~np.isnan(real) & (real > 0.3)
    free = ok & (ideal < real - 0.3)
    beyond = ok & (ideal > real + 0.3)
    band = ok & ~free & ~beyond
    occ = band & (g[..., 2] >= min_h) & (g[..., 2] <= max_h)
    lab = np.zeros(g.shape[:-1], np.int8)
    lab[free] = 1
    lab[occ] = 2
    with np.errstate(invalid="ignore"):
        m_bin = np.minimum(tt % 1.0, 1 - tt % 1.0)
        m_z = np.abs(np.abs(l[..., 2]) - wv)
        m_rng = np.minimum(np.abs(ideal - (real - 0.3)), np.abs(ideal - (real + 0.3)))
        m_h = np.minimum(np.abs(g[..., 2] - min_h), np.abs(g[..., 2] - max_h))
    sure = (m_z > 1e-5) & (~inplane | ((m_bin > 2e-3) & (ideal > 1e-3))) & (~ok | (m_rng > eps_m)) & (~band | (m_h > 1e-5)) \
        & (~inplane | np.isnan(real) | (np.abs(real - 0.3) > 1e-5))
    return lab, sure


def test_scan2d_ogm_against_float64_restatement_of_the_reference(oracle_lib):
    """The hokuyo path had no check outside the oracle's own arithmetic: a tilted sensor (roll and pitch, so that the
    |z| < voxel-width slab cuts through the volume obliquely), a NaN stretch and a too-short stretch in the scan."""
    size, w = (48, 48, 12), 0.1
    world = scenes.BoxWorld(6, extent=(4.0, 4.0, 1.0), n_boxes=30, toggle_frac=0.0)
    cfg = gie.make_config(w, size, cutoff_dist=1.0, ogm_min_h=-0.25, ogm_max_h=0.3)
    o = OracleMapper(cfg)
    seen = [0, 0]
    try:
        for k, tilt in ((0, 0.0), (3, 0.06), (6, -0.04)):
            pos, qy = scenes.pose(k, w, delta_vox=4, yaw_deg=31.0)
            # yaw * roll(tilt) * pitch(tilt / 2), Hamilton product, (w, x, y, z)
            def qmul(a, b):
                return (a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                        a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0])
            q = qmul(qmul(qy, (math.cos(tilt / 2), math.sin(tilt / 2), 0.0, 0.0)), (math.cos(tilt / 4), 0.0, math.sin(tilt / 4), 0.0))
            q = tuple(float(np.float32(v)) for v in q)
            _, rg = scenes.lidar_frame(world, k, pos, q, rings=1, az=360, phi_min_deg=0.0, max_range=30.0)
            r = np.where(np.isfinite(rg[0]), rg[0], np.nan).astype(np.float32)
            r[40:55] = np.nan
            r[200:210] = 0.25
            kw = dict(theta_inc=2.0 * np.pi / 360, theta_min=-np.pi + np.pi / 360)
            o.set_pose(pos, q)
            o.ogm_scan2d(r, **kw)
            lab = o.read_ogm()["inst_type"]
            f32 = lambda v: float(np.float32(v))
            want, sure = _scan2d_f64(pos, q, size, w, r, f32(kw["theta_inc"]), f32(kw["theta_min"]), f32(-0.25), f32(0.3))
            nf, no = _compare(lab, want, sure, "scan2d frame %d" % k)
            seen[0] += nf; seen[1] += no
            o.fuse(); o.batch_edt(); o.merge()
        assert seen[0] > 1000 and seen[1] > 30            # the scenes exercise both labels
    finally:
        o.close()


# ------------------------------------------------------------------ ray-casting OGM + fusion: a second statement
# Written from pntcld_raycast.cu:11-117, ray_cast.h:57-144, local_batch.h:114-126,250-258,303-350 and unify_helper.cuh:34-118 /
# voxmap_utils.cuh:182-200, not from the oracle: plain Python loops, every float operation an np.float32 operation in the order
# the reference writes them (no fused multiply-add: DESIGN.md deviation 4).  The pose has no rotation, so the sensor-to-map
# transform is a single float addition per coordinate and does not go through anybody's SE3 code.
_F = np.float32


def _pos2coord(p, w):
    return [int(np.floor(_F(_F(p[i]) / w) + _F(0.5))) for i in range(3)]          # floorf(p / w + 0.5f)


def _raycast_second_statement(origin, pts, pvt, size, w, min_h, max_h):
    X, Y, Z = size
    count = np.zeros((Z, Y, X), np.int32)
    occ = np.zeros((Z, Y, X), bool)
    inside = lambda c: 0 <= c[0] < X and 0 <= c[1] < Y and 0 <= c[2] < Z
    glb = [[_F(_F(p[i]) + _F(origin[i])) for i in range(3)] for p in pts]
    for g in glb:                                                                   # registerLocObs
        if g[2] >= min_h and g[2] <= max_h:
            c = [a - b for a, b in zip(_pos2coord(g, w), pvt)]
            if inside(c):
                occ[c[2], c[1], c[0]] = True
                count[c[2], c[1], c[0]] += 1

    def clear(cg):                                                                  # clearRayLoc on a global coordinate
        c = [a - b for a, b in zip(cg, pvt)]
        if inside(c):
            if occ[c[2], c[1], c[0]]:
                return False
            count[c[2], c[1], c[0]] -= 1
        return True                                                                 # (outside: type UNKNOWN, the add is dropped)

    max_length = _F(_F(_F(0.707) * _F(X)) * w)
    FLT_MAX = np.finfo(np.float32).max
    p0 = [_F(v) for v in origin]
    i0 = _pos2coord(p0, w)
    for p1 in glb:                                                                  # freeLocObs -> rayCastLoc
        i1 = _pos2coord(p1, w)
        clear(i0)
        if i0 == i1:
            continue
        d = [_F(p1[i] - p0[i]) for i in range(3)]
        ln = _F(np.sqrt(_F(_F(_F(d[0] * d[0]) + _F(d[1] * d[1])) + _F(d[2] * d[2]))))
        d = [_F(v / ln) for v in d]
        step, tmax, tdelta = [0] * 3, [FLT_MAX] * 3, [FLT_MAX] * 3
        cur = list(i0)
        for i in range(3):
            step[i] = 1 if d[i] > 0 else (-1 if d[i] < 0 else 0)
            if step[i]:
                border = _F(_F(_F(cur[i]) * w) + _F(_F(_F(step[i]) * w) * _F(0.5)))
                tmax[i] = _F(_F(border - p0[i]) / d[i])
                tdelta[i] = _F(w / _F(abs(d[i])))
        while True:
            if tmax[0] < tmax[1]:
                dim = 0 if tmax[0] < tmax[2] else 2
            else:
                dim = 1 if tmax[1] < tmax[2] else 2
            cur[dim] += step[dim]
            tmax[dim] = _F(tmax[dim] + tdelta[dim])
            if not clear(cur):
                break
            if cur == i1:
                break
            far = min(min(tmax[0], tmax[1]), tmax[2])
            if far > max_length or far > ln:
                break
    lab = np.where(count > 0, 2, np.where(count < 0, 1, 0)).astype(np.int8)        # getAllocKeys: OCCUPIED / FREE / untouched
    return count, lab


def _fuse_second_statement(occ_val, vox_type, count, thresh):
    """updateHashOGMWithPntCld on one voxel -> (occ_val, vox_type); types: 0 unknown, 1 free, 2 occupied."""
    if count == 0:
        return occ_val, vox_type
    val, a = (_F(250.0), _F(1.0)) if count > 0 else (_F(0.0), min(_F(1.0), _F(_F(-count) / _F(10.0))))
    prev = _F(occ_val) if vox_type != 0 else _F(0.0)
    v = _F(_F(a * val) + _F(_F(_F(1.0) - a) * prev))
    v = min(v, _F(254.0)); v = max(v, _F(1.0))
    o = int(v)
    return o, (2 if o > thresh else 1)


def _raycast_scene(k, rng):
    w = _F(0.1)
    origin = np.array([0.23 + 0.4 * k, -0.11 + 0.1 * k, 0.31], np.float32)
    n = 160
    az, el, r = rng.uniform(-np.pi, np.pi, n), rng.uniform(-0.5, 0.5, n), rng.uniform(0.15, 2.6, n)
    pts = np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)], -1).astype(np.float32)
    pts[:8] = pts[8:16]                                  # several points in one cell: counts above one
    pts[16] = [0.01, 0.0, 0.0]                           # a point in the sensor's own cell: the ray ends where it starts
    return w, origin, pts


def _check_raycast_and_fusion(make):
    size = (24, 20, 12)
    rng = np.random.default_rng(11)
    cfg = gie.make_config(0.1, size, cutoff_dist=1.0, ogm_min_h=-0.45, ogm_max_h=0.5)
    m = make(cfg)
    occ_val = np.zeros(size[::-1], np.int32)
    vtype = np.zeros(size[::-1], np.int8)
    known_block = {}
    try:
        for k in range(4):
            w, origin, pts = _raycast_scene(k, rng)
            m.set_pose(origin, (1.0, 0.0, 0.0, 0.0))
            pvt = list(m.pivot())
            assert pvt == [c - s // 2 for c, s in zip(_pos2coord(origin, w), size)]       # calculate_pivot_origin, local_batch.h:129-142
            m.ogm_pointcloud(pts)
            got = m.read_ogm()
            count, lab = _raycast_second_statement(origin, pts, pvt, size, w, _F(-0.45), _F(0.5))
            assert np.array_equal(got["ray_count"], count), "frame %d: %d cells differ in their ray count" % (k, int((got["ray_count"] != count).sum()))
            assert np.array_equal(got["inst_type"], lab), "frame %d: labels differ" % k
            assert (count > 1).any() and (count < -1).any() and (lab == 2).sum() > 50
            m.fuse()
            # the world the second statement keeps: voxels by GLOBAL coordinate (the volume moves 4 voxels per frame); a block exists
            # once a cell of it was touched (getAllocKeys keys every touched cell's block)
            zz, yy, xx = np.nonzero(count != 0)
            for x, y, z in zip(xx, yy, zz):
                g = (x + pvt[0], y + pvt[1], z + pvt[2])
                o, t = known_block.get(g, (0, 0))
                known_block[g] = _fuse_second_statement(o, t, int(count[z, y, x]), 180)
            g = np.array(sorted(known_block), np.int32)
            gv = m.query_global(g)
            want = np.array([known_block[tuple(v)] for v in g])
            assert np.array_equal(gv["occ_val"], want[:, 0]), "frame %d: fused occupancy values differ in %d voxels" % (k, int((gv["occ_val"] != want[:, 0]).sum()))
            got_t = np.where(gv["vox_type"] == 3, 1, gv["vox_type"])       # a frontier voxel (the merge of an earlier frame) is a free voxel
            assert np.array_equal(got_t, want[:, 1]), "frame %d: fused types differ" % k
            m.batch_edt(); m.merge()
        vals = np.array([v[0] for v in known_block.values()])
        assert len(np.unique(vals)) > 6                       # the low-pass filter was exercised: values between the extremes
    finally:
        m.close()


def test_raycast_ogm_and_fusion_against_a_second_statement(oracle_lib):
    _check_raycast_and_fusion(OracleMapper)


@pytest.mark.gpu
def test_raycast_ogm_and_fusion_against_a_second_statement_hip():
    _check_raycast_and_fusion(gie.Mapper)
