"""Measurement aid: time stamps at every phase boundary of the waves kernel (build variant with -DGIE_WAVE_TIMING; the boss
thread writes them into the middle row of the edt plane).   python tools/wave_timing.py build | run [workload]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("GIE_WT_LIB") or os.path.join(ROOT, "tools", "ablate", "libgie_hip_wt.so")
if sys.argv[1] == "build":
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-no-stack-slot-sharing", "-shared", "-fPIC",
                           "-DGIE_WAVE_TIMING=1", "-DGIE_TEST_HOOKS", os.path.join(ROOT, "gie-mapping_amd", "csrc", "gie_hip.hip"), "-o", LIB])
    sys.exit(0)
sys.path[:0] = [ROOT, os.path.join(ROOT, "gie-mapping_amd")]
import numpy as np, torch, bench, gie
from gie import mapper, scenes
mapper.load_library(LIB)
wl = sys.argv[2] if len(sys.argv) > 2 else "c5"
dev = torch.device("cuda", 0)
P = bench.PRESETS[wl]
size = tuple(P["size"])
cfg = gie.make_config(P["voxel"], size, cutoff_dist=P["cutoff"], fast_mode=P["fast"])
cfg.wave_workgroups = 160                            # (bench.py's setting on a device of its own)
m = gie.Mapper(cfg)
feed = bench.make_feed(wl, torch, scenes, dev, P["voxel"], size, (0, 0, 0), 64)
NF = int(sys.argv[3]) if len(sys.argv) > 3 else 8
verbose = len(sys.argv) > 4
feed.prepare(0, NF)
names = {0: "start", 1: "A seeds -> blocks + barrier", 3: "A round (n = active blocks) + barrier", 4: "B seeds -> blocks + barrier", 6: "B round (n = active blocks) + barrier",
         8: "A done + grid barrier", 9: "B stores into the volume + grid barrier", 10: "C done", 11: "C round (n = active tiles) + barrier",
         12: "C seeds -> tiles + barrier"}
for i in range(NF):
    feed.step_input(m, i); m.step(); m.sync()
    st = m.stats()
    row = m.read_local(vtype=False, dist_sq=False, coc=False)["edt"][size[2] // 2, size[1] // 2, :]
    prev, tot, cnt, tiles = None, {}, {}, 0
    lines = []
    for k in range(0, 1000, 2):
        t, tag = float(row[k]), int(row[k + 1])
        tg, n = tag // 1000000, tag % 1000000
        if tg not in names or (k > 0 and tg == 0):
            break
        if prev is not None:
            dt = (t - prev) % 16777216.0 / 100.0
            lines.append("%-38s n %7d   %8.2f us" % (names[tg], n, dt))
            tot[tg] = tot.get(tg, 0.0) + dt; cnt[tg] = cnt.get(tg, 0) + 1
            if tg == 11:
                tiles += n
        prev = t
        if tg == 10:
            prof = []
            for k2 in range(k + 2, k + 2 + 64, 2):
                tag2 = int(row[k2 + 1])
                if tag2 // 1000000 < 20:
                    break
                prof.append((tag2 % 1000000) * 64)
            if len(prof) == 32:
                for nm, b in (("A", 0), ("B", 8), ("C", 16)):
                    nb = max(1, prof[b + 7])
                    if nm == "C":
                        lines.append("wave C tiles: %d tile-runs, %.1f levels inside each; per tile-run: load %.2f us, levels %.2f us, write-back %.2f us, activation %.2f us; the longest tile-run %.1f us, the busiest wave %.1f us" % (
                            prof[b + 7], prof[b + 6] / nb, prof[b] / nb / 100.0, prof[b + 1] / nb / 100.0, prof[b + 2] / nb / 100.0, prof[b + 3] / nb / 100.0, prof[b + 4] / 6400.0, prof[b + 5] / 6400.0))
                        lines.append("   inside the levels: merge %.2f us, expand %.2f us in %.1f passes, drain of the proposals across the border %.2f us" % (
                            prof[24] / nb / 100.0, prof[25] / nb / 100.0, prof[26] / nb, prof[27] / nb / 100.0))
                        continue
                    lines.append("wave %s blocks: %d block-runs, %.1f levels inside each; per block-run: lookups %.2f us, fill %.2f us, levels %.2f us, write-back %.2f us; the longest block-run %.1f us, the busiest wave %.1f us" % (
                        nm, prof[b + 7], prof[b + 6] / nb, prof[b] / nb / 100.0, prof[b + 1] / nb / 100.0, prof[b + 2] / nb / 100.0, prof[b + 3] / nb / 100.0, prof[b + 4] / 6400.0, prof[b + 5] / 6400.0))
            break
    print("frame %d: visits %d %d %d  levels %d %d %d | A %.0f us, B %.0f us, C %.0f us in %d rounds (%d tile-rounds)" % (
        i, st["visits_a"], st["visits_b"], st["visits_c"], st["levels_a"], st["levels_b"], st["levels_c"],
        sum(tot.get(k, 0) for k in (1, 2, 3, 8)), sum(tot.get(k, 0) for k in (4, 5, 6, 7, 9)), sum(tot.get(k, 0) for k in (10, 11, 12)), cnt.get(11, 0), tiles))
    if verbose:
        print("\n".join(lines))
