"""BASELINE configs 5 and 3 at their REAL sizes on one MI355X.  The scalar oracle cannot follow at these sizes,
so the results are pinned by what they have to be (size-independent properties), not by a second run:

* C5 — the 1024^3 volume at 0.05 m as 2x2x2 tiles of 512^3, all eight mappers in this process on one device (8 x ~11 GB),
  hash world under full observation, two map updates with the robot moving 8 voxels in between; the face layers travel
  device to device as in `bench.py --gpus 8` — update 0 with the fixed stream-ordered rounds the multi-GPU bench enqueues,
  update 1 until no tile changes.  Per tile: every voxel known; every distance witnessed by its closest obstacle; a closest
  obstacle inside the whole volume is a voxel the world holds an obstacle in at that frame and never closer than the exact
  EDT of the WHOLE 1024^3 world (oracle/edt_mt.c on all host cores) — equal to it in > 99.9 % of the voxels (BFS across a cut
  is not an exact EDT) —, one outside is a voxel one of the tiles remembers as occupied; and after the until-stable exchange no face
  voxel can be improved by the voxel across the cut (the tiles agree on their shared faces).
* C3 — 512^3 at 0.1 m, cutoff 100 m (= none: cfg/ugv_laser3D_params.yaml:28), fast_mode off, the 16-ring lidar through the
  projective OGM until waves A / B / C have flooded: no capacity error (GIE_ERRF_QUEUE / levels), witnesses everywhere,
  closest obstacles inside the volume exactly as far as the exact EDT of the volume's own obstacles, those outside at most.
"""
import numpy as np
import pytest

import gie
from gie import scenes, tiling

pytestmark = pytest.mark.gpu


def _labels_dev(torch, dev, pvt, size, frame):
    return scenes.hash_world_labels(
        pvt, size, frame, seed=5, p_occ=0.01, toggle_frac=0.25,
        arange=lambda n: torch.arange(n, dtype=torch.int64, device=dev),
        where=lambda c, a, b: torch.where(c, torch.tensor(a, dtype=torch.int8, device=dev), torch.tensor(b, dtype=torch.int8, device=dev)))


def _occupied_at(g, frame):
    """the hash world's label of global voxels g [n, 3] at `frame` (True = obstacle)"""
    h = scenes._mix32((g[:, 0] * 73856093) ^ (g[:, 1] * 19349669) ^ (g[:, 2] * 83492791) ^ (5 * 0x9e3779b1))
    base = h < int(0.01 * 4294967296.0)
    h2 = scenes._mix32(h ^ 0x5bd1e995)
    toggler = (h2 >> 8) < int(0.25 * 16777216.0)
    return base & (~toggler | (((h2 & 1) + int(frame)) % 2 == 0))


def test_c5_1024_cube_as_eight_tiles_of_512_on_one_gpu(oracle_lib):
    import torch
    from oracle_py import edt_mt
    dev = torch.device("cuda", 0)
    tile, voxel = (512, 512, 512), 0.05
    grid = tiling.tile_grid(8)
    whole = tuple(grid[i] * tile[i] for i in range(3))
    # two updates reach (64 + 2) x 66 x 66 blocks per tile (+ ghost layers): a pool of 340 000 blocks = 5.2 GB per mapper
    cfg = gie.make_config(voxel, tile, cutoff_dist=2.0, fast_mode=False, max_blocks=340000)
    ms = []
    try:
        for r in range(8):
            m = gie.Mapper(cfg)
            m.set_tile(tiling.tile_offset_voxels(r, 8, tile), whole)
            ms.append(m)
        bufs = {}
        for k in range(2):
            pos, q = scenes.pose(k, voxel, delta_vox=8, yaw_deg=2.0)
            pvw = np.array(scenes.local_pivot(pos, voxel, whole))
            pvts = []
            for r, m in enumerate(ms):
                pvt = scenes.local_pivot(pos, voxel, tile, tiling.tile_offset_voxels(r, 8, tile))
                m.set_pose(pos, q)
                assert tuple(m.pivot()) == tuple(pvt)
                lab = _labels_dev(torch, dev, pvt, tile, k).contiguous()
                torch.cuda.synchronize(dev)                # torch built the plane on ITS stream; the mapper reads it on its own
                m.ogm_labels_dev(lab.data_ptr())
                m.step_begin_tiled()
                m.sync()
                del lab
                pvts.append(np.array(pvt, dtype=np.int64))
            if k == 0:
                tiling.exchange_rounds_local_device(ms, grid, dev, rounds=2, bufs=bufs)
                for m in ms:
                    m.sync()
                rounds = -1
            else:
                rounds = tiling.exchange_until_stable_local_device(ms, grid, dev, bufs={})
                assert 1 <= rounds <= 12
            # the exact EDT of the whole world at this frame (only the obstacles inside the 1024^3 volume)
            world_lab = np.empty(whole[::-1], np.int8)
            for r in range(8):
                o = pvts[r] - pvw
                world_lab[o[2]:o[2] + 512, o[1]:o[1] + 512, o[0]:o[0] + 512] = ms[r].read_local(edt=False, dist_sq=False, coc=False)["type"]
            assert (world_lab != 0).all(), "unknown voxels after full observation: %s" % (
                [(r, int((world_lab[tuple(slice(int(v), int(v) + 512) for v in (pvts[r] - pvw)[::-1])] == 0).sum()),
                  int((ms[r].read_local(edt=False, dist_sq=False, coc=False)["type"] == 0).sum())) for r in range(8)],)
            exact, _ = edt_mt(world_lab, want_coc=False)
            faces = {}
            n_equal = n_total = n_outside = n_below = 0
            for r, m in enumerate(ms):
                rb = m.read_local(edt=False)
                st = m.stats()
                pv = pvts[r]
                o = pv - pvw
                ex = exact[o[2]:o[2] + 512, o[1]:o[1] + 512, o[0]:o[0] + 512]
                d, c = rb["dist_sq"], rb["coc"]                       # int32 throughout: coordinates and differences are far below 2^15 here
                ax = [np.arange(512, dtype=np.int32) + np.int32(pv[i]) for i in range(3)]
                w2 = (c[..., 0] - ax[0][None, None, :]) ** 2
                w2 += (c[..., 1] - ax[1][None, :, None]) ** 2
                w2 += (c[..., 2] - ax[2][:, None, None]) ** 2
                assert np.array_equal(w2, d), "tile %d update %d: a distance is not witnessed by its closest obstacle" % (r, k)
                del w2
                inside = np.ones(d.shape, bool)
                for i in range(3):
                    inside &= (c[..., i] >= pvw[i]) & (c[..., i] < pvw[i] + whole[i])
                below = inside & (d < ex)
                if below.any():
                    zz, yy, xx = np.nonzero(below)
                    print("tile", r, "update", k, "below exact:", int(below.sum()), "local bbox", (xx.min(), yy.min(), zz.min()), (xx.max(), yy.max(), zz.max()),
                          "examples", [(int(xx[i]), int(yy[i]), int(zz[i]), int(d[zz[i], yy[i], xx[i]]), int(ex[zz[i], yy[i], xx[i]]), (c[zz[i], yy[i], xx[i]] - pv).tolist(),
                                        int(world_lab[tuple((c[zz[i], yy[i], xx[i]] - pvw)[::-1])])) for i in range(0, len(xx), max(1, len(xx) // 6))][:6])
                n_below += int(below.sum())
                if k == 0:
                    assert inside.all()
                n_equal += int((d[inside] == ex[inside]).sum()); n_total += int(inside.sum()); n_outside += int((~inside).sum())
                # the closest obstacles are obstacles: of this frame inside the whole volume, remembered ones outside it
                sel = np.flatnonzero(inside.ravel())[::61]
                assert _occupied_at(c.reshape(-1, 3)[sel].astype(np.int64), k).all()
                if (~inside).any():
                    far = np.ascontiguousarray(c.reshape(-1, 3)[np.flatnonzero(~inside.ravel())[::97][:8192]], dtype=np.int32)
                    # (an obstacle the whole volume has left behind is remembered by the tile that last held it)
                    occ = np.zeros(len(far), bool)
                    for mm in ms:
                        occ |= mm.query_global(far)["vox_type"] == 2
                    assert occ.all()
                if k == 1:
                    assert st["visits_a"] + st["visits_b"] + st["visits_c"] > 0
                # the six face layers (distance + closest obstacle), kept for the agreement test
                faces[r] = {f: (np.take(d, -1 if f & 1 else 0, axis=2 - (f >> 1)).astype(np.int64), np.take(c, -1 if f & 1 else 0, axis=2 - (f >> 1)).astype(np.int64))
                            for f in range(6)}
                del rb, d, c, inside, ex
            assert n_below == 0, (k, n_below)
            assert n_equal >= 0.999 * n_total, (k, n_equal, n_total)
            if k == 1:
                assert 0 < n_outside < 0.05 * n_total
                # stable: across every cut, neither side can lower the other (obtainFrontiers' C-seed rule finds nothing)
                for r in range(8):
                    for f, nb in tiling.neighbours(r, 8).items():
                        dA, cA = faces[r][f]
                        dB, cB = faces[nb][f ^ 1]
                        axis = f >> 1
                        # global coordinates of my face voxels: the neighbour's face voxels sit one step across the cut
                        idx = np.stack(np.meshgrid(np.arange(512), np.arange(512), indexing="ij")[::-1], -1)     # (b, a) -> (a, b)
                        gA = np.zeros(idx.shape[:2] + (3,), np.int64)
                        rest = [i for i in range(3) if i != axis]
                        gA[..., rest[0]] = idx[..., 0] + pvts[r][rest[0]]
                        gA[..., rest[1]] = idx[..., 1] + pvts[r][rest[1]]
                        gA[..., axis] = pvts[r][axis] + (511 if f & 1 else 0)
                        cand = ((cB - gA) ** 2).sum(-1)          # the neighbour's closest obstacle seen from my voxel
                        lo = cB - pvts[r]
                        cB_outside_me = ~((lo >= 0) & (lo < 512)).all(-1)
                        assert (dA[cB_outside_me] <= cand[cB_outside_me]).all(), "tiles %d / %d disagree across face %d" % (r, nb, f)
            del world_lab, exact, faces
    finally:
        for m in ms:
            m.close()


def test_c3_512_cube_at_0p1_m_without_cutoff_full_waves(oracle_lib):
    from oracle_py import edt_mt
    size, voxel = (512, 512, 512), 0.1
    cfg = gie.make_config(voxel, size, cutoff_dist=100.0, fast_mode=False)
    assert cfg.cutoff_grids_sq == 1000000
    world = scenes.BoxWorld(3, extent=(22.0, 22.0, 5.0), n_boxes=300, toggle_frac=0.25, ground_z=-3.0, min_size=0.8, max_size=6.0)
    kw = dict(theta_inc=2.0 * np.pi / 440, theta_min=-np.pi, phi_inc=np.radians(2.0), phi_min=np.radians(-15.0))
    b = gie.Mapper(cfg)
    try:
        tot = [0, 0, 0]
        checked = 0
        for k in range(10):
            pos, q = scenes.pose(k, voxel, delta_vox=8, yaw_deg=2.0)
            pts, _ = scenes.lidar_frame(world, k, pos, q, rings=16, az=1800, phi_min_deg=-15.0, phi_inc_deg=2.0, max_range=100.0)
            img = scenes.range_image(pts)
            b.set_pose(pos, q)
            b.ogm_multiscan(img, **kw)
            b.step()
            st = b.stats()                       # raises on a capacity error (queues, levels, pool)
            for i, key in enumerate(("visits_a", "visits_b", "visits_c")):
                tot[i] += st[key]
            assert max(st["levels_a"], st["levels_b"], st["levels_c"]) < 4000
            if k not in (3, 6, 9):
                continue
            rb = b.read_local(edt=False)
            ty = rb["type"]
            known = (ty != 0) & (rb["dist_sq"] < 4000000)
            d_cpu, _ = edt_mt(ty, want_coc=False)
            pv = np.array(b.pivot(), dtype=np.int64)
            cl = rb["coc"].astype(np.int64) - pv
            inside = ((cl >= 0) & (cl < 512)).all(-1) & known
            outside = known & ~inside
            assert int(known.sum()) > 1000000
            assert np.array_equal(rb["dist_sq"][inside], d_cpu[inside]), "%d voxels differ from the exact EDT" % int((rb["dist_sq"][inside] != d_cpu[inside]).sum())
            ci = cl[inside]
            assert (ty[ci[:, 2], ci[:, 1], ci[:, 0]] == 2).all()
            if outside.any():
                assert (rb["dist_sq"][outside] <= d_cpu[outside]).all()
                far = np.ascontiguousarray(rb["coc"][outside][::997][:4096], dtype=np.int32)
                assert (b.query_global(far)["vox_type"] == 2).all()
            zz, yy, xx = np.nonzero(known)
            g = np.stack([xx, yy, zz], -1) + pv
            assert np.array_equal(((rb["coc"][known].astype(np.int64) - g) ** 2).sum(-1), rb["dist_sq"][known])
            checked += 1
            del rb, ty, known, d_cpu, cl, inside, outside, g
        assert checked == 3
        assert all(v > 0 for v in tot), tot          # all three waves ran; without a cut-off A and B walk as far as the map goes
    finally:
        b.close()
