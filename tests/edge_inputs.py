"""Edge inputs of the sensor entry points, shared by the CPU (emulation) and GPU (HIP) suites:
empty scans, NaN / Inf / far-away / duplicate / zero-length points, poses at large negative
coordinates, range images and depth images full of invalid readings."""
import numpy as np

import gie
import parity
from gie import scenes


def _compare(a, b, tag):
    oa, ob = a.read_ogm(), b.read_ogm()
    assert np.array_equal(oa["ray_count"], ob["ray_count"]), tag
    assert np.array_equal(oa["inst_type"], ob["inst_type"]), tag
    for m in (a, b):
        m.fuse(); m.batch_edt(); m.merge()
    ra, rb = a.read_local(), b.read_local()
    for key in ("type", "dist_sq", "coc"):
        assert np.array_equal(ra[key], rb[key]), (tag, key)
    assert np.allclose(ra["edt"], rb["edt"], rtol=1e-6, atol=0), tag
    sa, sb = a.stats(), b.stats()
    for key in ("seeds_a", "seeds_b", "seeds_c", "visits_a", "visits_b", "visits_c", "blocks_total"):
        assert sa[key] == sb[key], (tag, key)
    return ra


def run(OracleMapper, Mapper):
    size = (40, 36, 20)
    cfg = gie.make_config(0.1, size, cutoff_dist=1.5, fast_mode=False)
    world = scenes.BoxWorld(7, extent=(4.0, 4.0, 1.5), n_boxes=30, toggle_frac=0.25)
    a, b = OracleMapper(cfg), Mapper(cfg)
    try:
        base = np.array([-1234.56, 789.01, -3.3], np.float32)          # far from the origin, negative coordinates
        rng = np.random.default_rng(3)
        known_before = 0
        for k in range(8):
            pos, q = scenes.pose(k, 0.1, delta_vox=4, yaw_deg=31.0)
            pts, _ = scenes.lidar_frame(world, k, pos, q, az=360, max_range=30.0)
            pos = tuple(np.float32(pos[i] + base[i]) for i in range(3))
            pts = pts.astype(np.float32)
            if k == 1:
                pts = np.zeros((0, 3), np.float32)                     # empty scan
            elif k == 2:
                bad = np.array([[np.nan, 0, 0], [0, np.inf, 0], [1, 1, -np.inf], [np.nan] * 3, [3e30, 0, 0], [0, -2e9, 1]], np.float32)
                pts = np.concatenate([pts[:50], bad, pts[50:], bad])   # invalid points in the middle and at the end
            elif k == 3:
                pts = np.concatenate([pts, pts[:200], np.zeros((5, 3), np.float32),          # duplicates, zero-length rays
                                      np.float32(1e-4) * rng.standard_normal((20, 3)).astype(np.float32),
                                      np.float32(5e4) * np.ones((3, 3), np.float32)])          # valid but 50 km away
            elif k == 4:
                pts = np.full((17, 3), np.nan, np.float32)             # nothing but invalid points
            for m in (a, b):
                m.set_pose(pos, q)
                m.ogm_pointcloud(pts)
            r = _compare(a, b, "cloud %d" % k)
            known = int((r["type"] != 0).sum())
            if k in (1, 4):
                assert known <= known_before                           # an empty scan observes nothing new
            known_before = known
        assert known_before > 0
        # range images / depth images that hold no valid reading, then mostly invalid ones
        for k, fill in enumerate((np.nan, np.inf, 0.0, -1.0, 0.29)):
            pos, q = scenes.pose(8 + k, 0.1, delta_vox=4, yaw_deg=31.0)
            pos = tuple(np.float32(pos[i] + base[i]) for i in range(3))
            img = np.full((16, 440), fill, np.float32)
            img[3, 100:140] = 2.5
            dep = np.full((60, 80), fill, np.float32)
            dep[20:30, 30:50] = 1.7
            for m in (a, b):
                m.set_pose(pos, q)
                if k % 2 == 0:
                    m.ogm_multiscan(img, theta_inc=2.0 * np.pi / 440, theta_min=-np.pi, phi_inc=np.radians(2.0), phi_min=np.radians(-15.0))
                else:
                    m.ogm_depth(dep, cx=39.5, cy=29.5, fx=70.0, fy=70.0, valid_nan=bool(k & 2))
            _compare(a, b, "image %d" % k)
    finally:
        a.close(); b.close()
