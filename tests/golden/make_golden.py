#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.

These are REGRESSION vectors produced by this repository's own CPU oracle (oracle/gie_oracle.c),
not by the reference: the reference ships no fixtures and cannot be built here.
They freeze the oracle's output so that an accidental change of the canonical semantics (or of
the scene generator) is caught, and give the GPU suite a check that does not depend on running
the oracle at test time.

    python tests/golden/make_golden.py
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "gie-mapping_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import parity  # noqa: E402
from oracle_py import OracleMapper  # noqa: E402

SCENARIOS = {
    "g_vlp16_48x48x16": parity.Scenario("g_vlp16", (48, 48, 16), sensor="multiscan", frames=12, delta_vox=5, yaw_deg=10.0),
    "g_mixed_48x40x24": parity.Scenario("g_mixed", (48, 40, 24), sensor="mixed", frames=12, delta_vox=5, yaw_deg=47.0),
    "g_raycast_32cube": parity.Scenario("g_ray", (32, 32, 32), voxel=0.05, sensor="lidar_points", frames=4, delta_vox=2,
                                        yaw_deg=15.0, extent=(0.7, 0.7, 0.7), n_boxes=12),
}


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff


def run(sc, make):
    """Per-frame CRCs of every externally visible array + the last frame's arrays in full."""
    m = make(sc.config())
    rows = []
    last = None
    for pos, q, kind, data, kw in sc.frames_iter():
        m.set_pose(pos, q)
        parity._feed(m, kind, data, kw)
        og = m.read_ogm()
        m.fuse()
        m.batch_edt()
        be = m.read_batch_edt()
        m.merge()
        r = m.read_local()
        st = m.stats()
        rows.append([crc(og["inst_type"]), crc(og["ray_count"]), crc(be["dist_sq"]), crc(be["coc"]), crc(r["type"]),
                     crc(r["dist_sq"]), crc(r["coc"]), st["seeds_a"], st["seeds_b"], st["seeds_c"], st["visits_a"],
                     st["visits_c"], st["levels_a"], st["levels_b"], st["levels_c"]])
        last = r
    m.close()
    return np.array(rows, dtype=np.int64), last


if __name__ == "__main__":
    for name, sc in SCENARIOS.items():
        rows, last = run(sc, OracleMapper)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), rows=rows, type=last["type"], dist_sq=last["dist_sq"],
                            coc=last["coc"].astype(np.int32), edt=last["edt"])
        print(name, rows.shape, "visits A/C", rows[:, 10].sum(), rows[:, 11].sum())
