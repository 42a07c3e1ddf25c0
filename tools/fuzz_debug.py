"""One scenario of tools/fuzz_parity.py stage by stage, with what differs spelled out (voxels, neighbours, stored records, wave
statistics):   python tools/fuzz_debug.py <seed> <number> [big]      environment: DBG_FOCUS=retain (the --focus of the fuzz run),
DBG_EMU=1 (the CPU emulation instead of the HIP library), DBG_VOX=x,y,z DBG_FRAME=k (one voxel's state before the merge of frame k)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gie-mapping_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import fuzz_parity, parity, gie
from oracle_py import OracleMapper
seed, only, big = int(sys.argv[1]), int(sys.argv[2]), (len(sys.argv) > 3 and sys.argv[3] == "big")
focus = os.environ.get("DBG_FOCUS")
emu = os.environ.get("DBG_EMU")
if emu:
    from emu_py import EmuMapper as Under
else:
    Under = gie.Mapper
rng = np.random.default_rng(seed)
for i in range(only + 1):
    sc = fuzz_parity.random_scenario(rng, i, big, focus)
cfg = sc.config()
print(sc.name, sc.size, "retain", sc.retain, "turn", sc.turn, "max_blocks", cfg.max_blocks, flush=True)
a, b = OracleMapper(cfg), Under(cfg)
X, Y, Z = sc.size
for k, (pos, q, kind, data, kw) in enumerate(sc.frames_iter()):
    for m in (a, b):
        m.set_pose(pos, q); parity._feed(m, kind, data, kw)
    pv = np.array(a.pivot())
    a.fuse(); b.fuse()
    ta = a.read_local(edt=False, dist_sq=False, coc=False)["type"]; tb = b.read_local(edt=False, dist_sq=False, coc=False)["type"]
    sa, sb = a.stats(), b.stats()
    nd = int((ta != tb).sum())
    print("frame", k, "pivot", pv.tolist(), "kind", kind, "fused type diff", nd, "blocks", sa["blocks_total"], sb["blocks_total"], flush=True)
    if nd:
        idx = np.argwhere(ta != tb)
        g = idx[:, ::-1] + pv
        blk = np.unique(g >> 3, axis=0)
        print("  differing voxels in", len(blk), "blocks; first blocks:", blk[:8].tolist())
        print("  oracle types", np.unique(ta[ta != tb], return_counts=True), "under", np.unique(tb[ta != tb], return_counts=True))
        xyz = g[:8].astype(np.int32)
        ga = a.query_global(xyz); print("  global a", ga["occ_val"].tolist(), ga["vox_type"].tolist())
        gb = b.query_global(xyz); print("  global b", gb["occ_val"].tolist(), gb["vox_type"].tolist())
        print("  robot block", ((pv + np.array(sc.size) // 2) >> 3).tolist(), " z range of diffs", int(idx[:, 0].min()), int(idx[:, 0].max()), "y", int(idx[:, 1].min()), int(idx[:, 1].max()), "x", int(idx[:, 2].min()), int(idx[:, 2].max()))
        break
    a.batch_edt(); b.batch_edt()
    if os.environ.get("DBG_VOX") and k == int(os.environ.get("DBG_FRAME", "-1")):
        vx, vy, vz = [int(v) for v in os.environ["DBG_VOX"].split(",")]
        ea, eb = a.read_batch_edt(), b.read_batch_edt()
        pa, pb = a.read_local(edt=False), b.read_local(edt=False)
        print("   before merge: batch dist", int(ea["dist_sq"][vz, vy, vx]), int(eb["dist_sq"][vz, vy, vx]), "coc", ea["coc"][vz, vy, vx].tolist(), eb["coc"][vz, vy, vx].tolist(),
              "| pair dist", int(pa["dist_sq"][vz, vy, vx]), int(pb["dist_sq"][vz, vy, vx]), "coc", pa["coc"][vz, vy, vx].tolist(), pb["coc"][vz, vy, vx].tolist(), "type", int(pa["type"][vz, vy, vx]), int(pb["type"][vz, vy, vx]))
    a.merge(); b.merge()
    sa, sb = a.stats(), b.stats()
    ra, rb = a.read_local(), b.read_local()
    print("   after merge: type diff", int((ra["type"] != rb["type"]).sum()), "dist diff", int((ra["dist_sq"] != rb["dist_sq"]).sum()), "blocks", sa["blocks_total"], sb["blocks_total"],
          "stats", [(kk, sa[kk], sb[kk]) for kk in ("seeds_a", "seeds_b", "seeds_c", "visits_a", "visits_b", "visits_c", "levels_a", "levels_b", "levels_c") if sa[kk] != sb[kk]], flush=True)
    dd = np.argwhere((ra["dist_sq"] != rb["dist_sq"]) | (ra["coc"] != rb["coc"]).any(-1))
    for (z, y, x) in dd[:6]:
        print("     voxel local", (int(x), int(y), int(z)), "global", (pv + np.array([x, y, z])).tolist(), "type", int(ra["type"][z, y, x]), int(rb["type"][z, y, x]),
              "dist", int(ra["dist_sq"][z, y, x]), int(rb["dist_sq"][z, y, x]), "coc", ra["coc"][z, y, x].tolist(), rb["coc"][z, y, x].tolist())
    if len(dd):
        z, y, x = [int(v) for v in dd[0]]
        for (ddx, ddy, ddz) in ((-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1)):
            nx, ny, nz = x + ddx, y + ddy, z + ddz
            if 0 <= nx < X and 0 <= ny < Y and 0 <= nz < Z:
                print("     nb", (ddx, ddy, ddz), "type", int(ra["type"][nz, ny, nx]), int(rb["type"][nz, ny, nx]), "dist", int(ra["dist_sq"][nz, ny, nx]), int(rb["dist_sq"][nz, ny, nx]),
                      "coc", ra["coc"][nz, ny, nx].tolist(), rb["coc"][nz, ny, nx].tolist())
            else:
                g = (pv + np.array([nx, ny, nz])).astype(np.int32).reshape(1, 3)
                qa, qb = a.query_global(g), b.query_global(g)
                print("     nb outside", (ddx, ddy, ddz), "stored: type", qa["vox_type"].tolist(), qb["vox_type"].tolist(), "dist", qa["dist_sq"].tolist(), qb["dist_sq"].tolist(), "coc", qa["coc"].tolist(), qb["coc"].tolist())
        print("     stats a", {kk: sa[kk] for kk in ("seeds_a", "seeds_b", "seeds_c", "visits_a", "visits_b", "visits_c", "levels_c")})
        xyz = (dd[:6, ::-1] + pv).astype(np.int32)
        ga, gb = a.query_global(xyz), b.query_global(xyz)
        print("     stored a dist", ga["dist_sq"].tolist(), "coc", ga["coc"].tolist()); print("     stored b dist", gb["dist_sq"].tolist(), "coc", gb["coc"].tolist())
        # the obstacle the two disagree about: what does each map hold there?
        for who, r in (("a", ra), ("b", rb)):
            cc = r["coc"][dd[0][0], dd[0][1], dd[0][2]].astype(np.int32).reshape(1, 3)
            qa, qb = a.query_global(cc), b.query_global(cc)
            print("     obstacle of", who, cc.tolist(), "oracle map: type", qa["vox_type"].tolist(), "occ", qa["occ_val"].tolist(), "| hip map: type", qb["vox_type"].tolist(), "occ", qb["occ_val"].tolist(),
                  "| inside volume:", bool(((cc[0] - pv >= 0) & (cc[0] - pv < np.array(sc.size))).all()))
        break
