#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE of tools/pmc_calib.hip's streams of known size: python tools/pmc_calib_summary.py <fetch.db> <write.db>
Prints, per kernel, the KiB rocprofv3 reports, the true bytes, and true / reported = the factor to apply to that access width."""
import json
import sys
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from rocpd_summary import pmc_per_kernel

TRUE = 512 << 20
f = pmc_per_kernel(sys.argv[1], "FETCH_SIZE")
w = pmc_per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
print("%-28s %8s %16s %14s %10s" % ("kernel", "launches", "reported_KiB", "true_MiB", "factor"))
for name, tab in (("FETCH_SIZE", f), ("WRITE_SIZE", w)):
    for k in sorted(tab):
        if ("read" in k and name == "FETCH_SIZE") or ("write" in k and name == "WRITE_SIZE"):
            v, n = tab[k]
            fac = TRUE / (v * 1024.0) if v else float("nan")
            out[k.split("(")[0]] = round(fac, 3)
            print("%-28s %8d %16.0f %14.1f %10.3f   (%s)" % (k.split("(")[0], n, v, TRUE / 1048576.0, fac, name))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
