#!/usr/bin/env python3
"""Merge the traffic_entry.json files tools/profile_round.sh leaves under gpurun_out/prof_<tag>/ into profiles/traffic_r06.json
(what bench.py's roofline.traffic reads):   python tools/merge_traffic.py <workload>_<X>x<Y>x<Z>=<entry.json> [...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

path = os.path.join(ROOT, "profiles", bench.TRAFFIC_FILE)
tab = json.load(open(path)) if os.path.exists(path) else {}
for arg in sys.argv[1:]:
    key, f = arg.split("=", 1)
    tab[key] = json.load(open(f))
json.dump(tab, open(path, "w"), indent=1)
print("wrote", path, {k: v.get("csrc_hash") for k, v in tab.items()}, "this build:", bench.csrc_hash())
