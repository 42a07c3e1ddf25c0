import sys, time, os
sys.path[:0] = ['/root/repo', '/root/repo/gie-mapping_amd']
import torch, gie, bench
size = (512, 512, 512)
tries = int(os.environ.get("PLACE_TRIES", "0"))
cfg = gie.make_config(0.05, size, cutoff_dist=2.0, max_blocks=bench.pool_blocks("c5", size, 40), place_tries=tries)
for i in range(4):
    t0 = time.perf_counter(); m = gie.Mapper(cfg); m.sync(); t1 = time.perf_counter(); m.close()
    print("create %.1f ms (gie_config.place_tries = %d)" % (1e3 * (t1 - t0), tries), flush=True)
