"""Which stream of the Mark + commit sweep carries the placement bands?  Fresh mappers, each probed with subsets of the
sweep's streams (1 type read, 2 batch-obstacle read, 4 pair write, 8 stored-obstacle write)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gie-mapping_amd"), os.path.join(ROOT, "tests")]
import torch, gie, bench
size = (512, 512, 512)
cfg = gie.make_config(0.05, size, cutoff_dist=2.0, fast_mode=False, max_blocks=bench.pool_blocks("c5", size, 40))
from hooks_py import HooksMapper        # gie_debug_place_probe exists in the test build of the library only
sets = [15, 12, 4, 8, 15 | 256, 12 | 256, 4 | 256, 8 | 256, 15 | 512, 12 | 512, 15 | 768, 12 | 768, 4 | 768]
print("streams: " + " ".join("%6d" % s for s in sets))
for rep in range(int(os.environ.get("PROBE_REPS", "8"))):
    m = HooksMapper(cfg)
    out = []
    for s in sets:
        out.append(m.debug_place_probe(5 | (s << 16)))
    print("mapper %d: " % rep + " ".join("%6.3f" % v for v in out), flush=True)
    m.close()
