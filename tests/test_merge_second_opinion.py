"""The merge stage against a second, independent statement of it (tests/merge_checker.py: written from the reference's kernels
and SURVEY App. E without looking at the oracle; dense numpy world, serial schedule).

Compared per map update, on multi-frame scenes of at most 48^3 voxels in which all three waves run:
  * the seeds obtainFrontiers hands to waves A / B / C (deterministic in the reference) — equal counts;
  * the distance of every known voxel inside the local volume after the waves — equal in all but a few voxels per million,
    although the two statements expand their frontiers in different orders (SURVEY §7 found it schedule-independent in its
    probes; here 34 voxel-frames of 1.4 million differ, by <= 0.07 voxel, the canonical schedule holding the smaller value:
    6-connected propagation of closest obstacles is not an exact EDT, and which of two equally near sources reaches a voxel
    first can matter one step on); where they differ both sides hold a witness and the two are less than half a voxel apart;
  * the stored global records (inside and outside the volume, on both sides): every valid (distance, closest obstacle) pair is a
    witness, |obstacle - voxel|^2 == distance, and where the two sides disagree outside the volume the difference is the
    schedule's (both are witnesses); the share of such voxels is reported and bounded."""
import numpy as np
import pytest

import gie
import parity
from merge_checker import MergeChecker
from oracle_py import OracleMapper

SCENES = [
    parity.Scenario("vlp16", (48, 48, 16), sensor="multiscan", frames=12, delta_vox=5, yaw_deg=10.0),
    parity.Scenario("depth", (48, 40, 24), sensor="depth", frames=10, delta_vox=5, yaw_deg=47.0),
    parity.Scenario("mixed_odd", (37, 29, 11), sensor="mixed", frames=9, delta_vox=3, yaw_deg=33.0),
    parity.Scenario("c5_hash_world", (40, 40, 24), voxel=0.05, sensor="labels", frames=8, delta_vox=8, yaw_deg=2.0, seed=5,
                    cutoff_dist=2.0, p_occ=0.01),
    parity.Scenario("c3_no_cutoff", (48, 48, 16), voxel=0.1, sensor="multiscan", frames=8, delta_vox=6, yaw_deg=12.0,
                    cutoff_dist=100.0, extent=(5.0, 5.0, 1.5)),
    parity.Scenario("fast_mode", (48, 40, 24), sensor="mixed", frames=9, delta_vox=5, yaw_deg=47.0, fast_mode=True),
]


def run_scene(sc, make_mapper):
    cfg = sc.config()
    m = make_mapper(cfg)
    frames = list(sc.frames_iter())
    # the world box: every volume of the drive + a margin for the waves outside it
    pv = []
    probe = make_mapper(cfg)
    for pos, q, *_ in frames:
        probe.set_pose(pos, q)
        pv.append(probe.pivot())
    probe.close()
    pv = np.array(pv)
    lo = pv.min(0) - 3
    hi = pv.max(0) + np.array(sc.size) + 3
    chk = MergeChecker(sc.size, cfg.cutoff_grids_sq, cfg.fast_mode, lo, hi)
    wr = chk.wr
    stats = {"voxels": 0, "inside_diff": 0, "outside_diff": 0, "outside_cmp": 0, "waves": [0, 0, 0]}
    try:
        for k, (pos, q, kind, data, kw) in enumerate(frames):
            m.set_pose(pos, q)
            parity._feed(m, kind, data, kw)
            m.fuse()
            T = m.read_local(edt=False, dist_sq=False, coc=False)["type"]
            m.batch_edt()
            e = m.read_batch_edt()
            m.merge()
            pvt = np.array(m.pivot())
            crd = pvt + np.array(sc.size) // 2                        # calculate_update_pivot (local_batch.h:159-166): round(pos / w) - wave_range / 2
            upvt = crd - wr // 2
            pd, seeds = chk.update(pvt, upvt, T, e["dist_sq"], e["coc"])
            st = m.stats()
            assert seeds == (st["seeds_a"], st["seeds_b"], st["seeds_c"]), "%s frame %d: seeds %s vs %s" % (
                sc.name, k, seeds, (st["seeds_a"], st["seeds_b"], st["seeds_c"]))
            for i, key in enumerate(("visits_a", "visits_b", "visits_c")):
                stats["waves"][i] += st[key]
            r = m.read_local()
            known = r["type"] != 0
            assert np.array_equal(r["type"], chk.T), "%s frame %d: types after the merge (FNT flips) differ" % (sc.name, k)
            bad = known & (r["dist_sq"] != pd)
            stats["voxels"] += int(known.sum()); stats["inside_diff"] += int(bad.sum())
            if bad.any():
                # the schedule showing inside the volume (measured: 34 voxel-frames of 1.4 million over these scenes, all in two
                # frames of one scene, the canonical schedule's distance the smaller one by <= 0.07 voxel): both sides must hold
                # a witness, and the two must be close
                assert bad.sum() <= 0.002 * known.sum(), "%s frame %d: distance inside the volume differs in %d of %d known voxels" % (
                    sc.name, k, int(bad.sum()), int(known.sum()))
                zz, yy, xx = np.nonzero(bad)
                v = np.stack([xx, yy, zz], -1)
                assert np.array_equal(((chk.pp[zz, yy, xx] + upvt - pvt - v) ** 2).sum(-1), pd[zz, yy, xx])
                assert np.array_equal(((r["coc"][zz, yy, xx].astype(np.int64) - pvt - v) ** 2).sum(-1), r["dist_sq"][zz, yy, xx])
                dsq = np.abs(np.sqrt(r["dist_sq"][bad].astype(np.float64)) - np.sqrt(pd[bad].astype(np.float64)))
                assert dsq.max() <= 0.5, "%s frame %d: the two schedules differ by %.2f voxels" % (sc.name, k, dsq.max())
            assert chk.witness_violations(pvt) == 0
            # stored records around the volume, both sides
            zz, yy, xx = np.nonzero(chk.g_type != 0)
            g = (np.stack([xx, yy, zz], -1) + chk.lo).astype(np.int32)
            gv = m.query_global(g)
            assert np.array_equal(gv["vox_type"], chk.g_type[zz, yy, xx]), "%s frame %d: stored types differ" % (sc.name, k)
            valid = (gv["dist_sq"] >= 0) & (gv["dist_sq"] < chk.invalid_dist) & (gv["coc"] <= 900000).all(-1)
            wit = ((gv["coc"].astype(np.int64) - g) ** 2).sum(-1)
            assert np.array_equal(wit[valid], gv["dist_sq"][valid]), "%s frame %d: a stored record of the mapper is not a witness" % (sc.name, k)
            loc = g - pvt
            outside = ~((loc >= 0) & (loc < np.array(sc.size))).all(-1)
            diff = gv["dist_sq"].astype(np.int64) != chk.g_dist[zz, yy, xx]
            assert (diff & ~outside).sum() <= bad.sum(), "%s frame %d: stored distances inside the volume differ where the pairs do not" % (sc.name, k)
            stats["outside_cmp"] += int(outside.sum()); stats["outside_diff"] += int((diff & outside).sum())
            od = diff & outside & valid
            if od.any():
                dd = np.sqrt(gv["dist_sq"][od].astype(np.float64)) - np.sqrt(chk.g_dist[zz, yy, xx][od].astype(np.float64))
                stats["outside_max"] = max(stats.get("outside_max", 0.0), float(np.abs(dd).max()))
                stats["outside_ours_smaller"] = stats.get("outside_ours_smaller", 0) + int((dd < 0).sum())
    finally:
        m.close()
    return stats


@pytest.mark.parametrize("sc", SCENES, ids=[s.name for s in SCENES])
def test_oracle_agrees_with_the_second_statement(oracle_lib, sc):
    st = run_scene(sc, OracleMapper)
    assert st["voxels"] > 0
    if sc.name in ("vlp16", "c5_hash_world", "c3_no_cutoff"):
        assert all(v > 0 for v in st["waves"]), "the scene does not exercise all three waves: %s" % (st["waves"],)
    assert st["inside_diff"] <= 5e-4 * st["voxels"], st
    # outside the volume the schedule shows (both sides are witnesses, checked above).  Measured with waves A and B in
    # block rounds: 0 / 0.1 % / 0.5 % of the stored records outside the volume (cut-off scenes / vlp16 / the no-cut-off scene), the
    # largest gap 2.4 voxels, either side the smaller one about equally often; with the level-synchronous waves of round 2 it was
    # 0 / 0.1 % / 0.13 % with the same largest gap.  Bounded, not required to be zero.
    assert st["outside_diff"] <= 0.006 * max(1, st["outside_cmp"]), st
    assert st.get("outside_max", 0.0) <= 3.0, st


@pytest.mark.gpu
@pytest.mark.parametrize("sc", SCENES[:4], ids=[s.name for s in SCENES[:4]])
def test_hip_agrees_with_the_second_statement(sc):
    st = run_scene(sc, gie.Mapper)
    assert st["voxels"] > 0
    assert st["outside_diff"] <= 0.006 * max(1, st["outside_cmp"]), st
