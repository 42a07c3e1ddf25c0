"""Placement experiment: the C5 Mark + commit sweep with all large planes in ONE arena (GIE_ARENA_MB) at a series of skews
(GIE_ARENA_SKEW, MiB per plane).  One mapper per configuration, each in a fresh process (the arena variables are read once).
    python tools/arena_probe.py            # the driver: runs the configurations below as subprocesses
"""
import os, subprocess, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "one":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "gie-mapping_amd"), os.path.join(ROOT, "tests")]
    import torch, gie, bench
    from gie import scenes
    from hooks_py import HooksMapper      # the arena / launch-parameter switches and the placement probe exist in the test build of the library only
    dev = torch.device("cuda", 0)
    size = tuple(int(v) for v in os.environ.get("PROBE_SIZE", "512,512,512").split(","))
    cfg = gie.make_config(0.05, size, cutoff_dist=2.0, fast_mode=False, max_blocks=bench.pool_blocks("c5", size, 40))
    for rep in range(int(os.environ.get("PROBE_REPS", "2"))):
        m = HooksMapper(cfg)
        probe_ms = -1.0
        if os.environ.get("PROBE_PLACE"):
            probe_ms = m.debug_place_probe(5)
        feed = bench.make_feed("c5", torch, scenes, dev, 0.05, size, (0, 0, 0), 10)
        feed.prepare(0, 8)
        for i in range(3):
            feed.step_input(m, i); m.step()
        m.sync(); m.profile_enable(True)
        for i in range(3, 8):
            feed.step_input(m, i); m.step()
        m.sync()
        pr = m.profile_read()
        print("RESULT probe=%.4f size=%s skew=%s rep=%d mark_commit/Gvox=%.4f mark_commit=%.4f fuse=%.4f z=%.4f x=%.4f total=%.4f" % (
            probe_ms, "x".join(map(str, size)), os.environ.get("GIE_ARENA_SKEW", "-"), rep, pr["mark_commit"][0] / max(1, pr["mark_commit"][1]) / (size[0] * size[1] * size[2] / 134217728.0),
            *[pr[k][0] / max(1, pr[k][1]) for k in ("mark_commit", "fuse", "edt_pass_z", "edt_pass_x")],
            sum(v[0] for v in pr.values()) / 5.0), flush=True)
        print("KERNELS rep=%d " % rep + " ".join("%s=%.4f" % (k, v[0] / max(1, v[1])) for k, v in sorted(pr.items())), flush=True)
        m.close()
    sys.exit(0)
random.seed(int(os.environ.get("PROBE_SEED", "1")))
configs = [None, ""]                       # None: separate allocations (the library's default); "": arena, no skew
if int(os.environ.get("PROBE_CONFIGS", "16")) == -2:
    configs = [None]
n = int(os.environ.get("PROBE_CONFIGS", "16"))
for _ in range(n):
    configs.append(",".join(str(random.choice([0, 2, 6, 14, 30, 62, 126, 254, 510])) for _ in range(14)))
for c in configs:
    env = dict(os.environ)
    if c is None:
        env.pop("GIE_ARENA_MB", None)
    else:
        env["GIE_ARENA_MB"] = os.environ.get("ARENA_MB", "40000"); env["GIE_ARENA_SKEW"] = c
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, capture_output=True, text=True)
    for ln in r.stdout.splitlines():
        if ln.startswith("KERNELS"):
            print(ln, flush=True)
        if ln.startswith("RESULT"):
            print(("separate " if c is None else "arena    ") + ln, flush=True)
    if r.returncode != 0:
        print("FAILED", c, r.stderr[-400:], flush=True)
