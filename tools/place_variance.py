import sys, os, json, subprocess
# mark_commit time of several fresh mappers without the placement probe
import bench
bench.PLACE_TRIES = int(sys.argv[1])
sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--no-extras", "--no-cpu-baseline"]
bench.main()
