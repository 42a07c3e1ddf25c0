import sqlite3, sys, re
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
try: rows = db.execute("select name, start, end from kernels").fetchall()
except Exception: rows = db.execute("select kernel_name, start, end from kernels").fetchall()
agg = defaultdict(list)
for n, s, e in rows: agg[n[:70]].append(e - s)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print("%-72s %6d %10.1f us avg %10.1f max" % (k, len(v), sum(v) / len(v) / 1e3, max(v) / 1e3))
