#!/usr/bin/env python3
"""Per-kernel means of arbitrary rocprofv3 PMC counters from a rocpd database:
   python tools/pmc_kernel.py <pmc.db> COUNTER [COUNTER ...]"""
import sys
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from rocpd_summary import pmc_per_kernel

db, counters = sys.argv[1], sys.argv[2:]
tab = {c: pmc_per_kernel(db, c) for c in counters}
kernels = sorted({k for t in tab.values() for k in t})
print("%-20s %8s " % ("kernel", "launches") + " ".join("%18s" % c for c in counters))
for k in kernels:
    n = max(tab[c].get(k, (0, 0))[1] for c in counters)
    print("%-20s %8d " % (k, n) + " ".join("%18.0f" % tab[c].get(k, (0.0, 0))[0] for c in counters))
