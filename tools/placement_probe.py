"""Measurement aid: Mark + commit (the HBM-bound sweep) of the C5 workload on several mapper instances of one process — created one
after the other, side by side, and after a large allocation was freed.  The kernel's time follows the PLACEMENT of a mapper's
planes (1.32 ... 1.63 ms for the same binary, DESIGN.md section 4 "Tried ... and dropped"); GIE_DEBUG_ALLOC=1 prints where they went.
    GIE_DEBUG_ALLOC=1 python tools/placement_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gie-mapping_amd")]
import torch, gie, bench
from gie import scenes
dev = torch.device("cuda", 0)
size = (512, 512, 512)
cfg = gie.make_config(0.05, size, cutoff_dist=2.0, fast_mode=False)
def run(tag, keep=None):
    print("== mapper", tag, flush=True); sys.stderr.flush()
    m = gie.Mapper(cfg)
    feed = bench.make_feed("c5", torch, scenes, dev, 0.05, size, (0, 0, 0), 8)
    feed.prepare(0, 6)
    for i in range(2):
        feed.step_input(m, i); m.step()
    m.sync(); m.profile_enable(True)
    for i in range(2, 6):
        feed.step_input(m, i); m.step()
    m.sync()
    pr = m.profile_read()
    print("   mark_commit %.4f ms, fuse %.4f, pass z %.4f" % tuple(pr[k][0] / max(1, pr[k][1]) for k in ("mark_commit", "fuse", "edt_pass_z")), flush=True)
    if keep is None:
        m.close()
    else:
        keep.append(m)
if os.environ.get("PROBE_N"):            # only that many mappers, one after the other
    for t in "ABCDEF"[:int(os.environ["PROBE_N"])]:
        run(t)
    sys.exit(0)
for t in "ABC":
    run(t)
held = []
run("D (kept)", held); run("E (while D lives)", held)
for m in held: m.close()
x = torch.empty(40 * 1024**3, dtype=torch.uint8, device=dev); del x; torch.cuda.empty_cache()
run("F (after a 40 GiB torch allocation was freed)")
