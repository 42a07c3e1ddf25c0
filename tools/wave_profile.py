"""Per-frame wave statistics on the dense-observation preset: visits, BFS levels, time of wave C."""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gie-mapping_amd")]
import torch, bench, gie
from gie import scenes

sensor = sys.argv[1] if len(sys.argv) > 1 else "vlp16_projective"
rings, az, phi_min, phi_inc, bins = bench.SENSORS[sensor]
frames = bench.make_frames(scenes, 0.05, 13, 5, sensor)
dev = torch.device("cuda", 0)
d_pts = [torch.from_numpy(f[2]).to(dev) for f in frames]
m = gie.Mapper(gie.make_config(0.05, (512, 512, 512), cutoff_dist=2.0, fast_mode=False))
m.profile_enable(True)
prev = {}
for i, (pos, q, pts, _) in enumerate(frames):
    m.set_pose(pos, q)
    if bins is None:
        m.ogm_pointcloud_dev(d_pts[i].data_ptr(), d_pts[i].shape[0])
    else:
        m.ogm_multiscan_dev(d_pts[i].data_ptr(), bins, rings, 2.0 * math.pi / bins, -math.pi, math.radians(phi_inc), math.radians(phi_min))
    m.step(); m.sync()
    st = m.stats(); prof = m.profile_read()
    row = {"frame": i, "seeds": [st["seeds_a"], st["seeds_b"], st["seeds_c"]], "visits": [st["visits_a"], st["visits_b"], st["visits_c"]],
           "levels": [st["levels_a"], st["levels_b"], st["levels_c"]]}
    for k in ("waves", "frontiers", "mark", "commit"):
        t = prof[k][0] - prev.get(k, 0.0); prev[k] = prof[k][0]
        row[k + "_ms"] = round(t, 3)
    if st["levels_c"]:
        row["us_per_level_c"] = round(1e3 * row["waves_ms"] / st["levels_c"], 2)
        row["visits_per_level_c"] = round(st["visits_c"] / st["levels_c"], 1)
    print(json.dumps(row))
