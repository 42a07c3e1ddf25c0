"""Measurement aid (GPU box): the C5 bench line of a fresh mapper with gie_config.place_tries = N (0: gie_create takes the planes as
the allocator hands them out).   python tools/place_variance.py N > line.json"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import bench  # noqa: E402
bench.PLACE_TRIES = int(sys.argv[1])
sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--no-extras", "--no-cpu-baseline"]
bench.main()
