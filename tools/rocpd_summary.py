#!/usr/bin/env python3
"""Turn rocprofv3 rocpd (.db) output into the text summaries committed under profiles/.

  kernel stats (the --stats table):  python tools/rocpd_summary.py stats  <kernel-trace.db>
  HBM bytes per launch (PMC passes): python tools/rocpd_summary.py pmc <fetch.db> <write.db>

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE counts 128-B
requests as 64 B for wide coalesced streams (MI355X_MICROARCH.md §HBM), so the read side is
doubled ("fetch_corrected"); narrower access patterns are uncalibrated, which is stated in the
output.  WRITE_SIZE is taken as reported.
"""
import json
import re
import sqlite3
import sys
from collections import defaultdict

SHORT = [
    (r"op_classify|k_labels16", "ogm_classify"), (r"op_register_point", "ray_register"), (r"k_free_rays|op_free_ray", "ray_free"),
    (r"op_raycast_finalize", "ray_finalize"), (r"op_fuse|k_fuse_rows", "fuse"), (r"k_edt_y", "edt_pass_y"), (r"k_edt_x", "edt_pass_x"),
    (r"k_edt_prep", "edt_prep"), (r"k_edt_z_direct", "edt_pass_z.direct"), (r"k_edt_z_stream", "edt_pass_z.stream"), (r"k_edt_z", "edt_pass_z.column"), (r"k_round_note", "halo.round_note"), (r"k_markc|op_markc|op_tile_oldskip", "mark_commit"), (r"op_pair_flush|op_evict|op_rehash", "block_alloc"), (r"op_mark", "mark"), (r"k_frontier_faces", "frontiers.faces"), (r"op_frontier|k_frontier_tiles", "frontiers.tiles"), (r"k_waves_ab", "waves.ab"), (r"k_waves_c", "waves.c"), (r"k_waves", "waves"), (r"op_commit", "commit"),
    (r"k_tile_oldskip", "block_alloc.oldskip"), (r"k_pair_lazy_list", "block_alloc.lazy_list"), (r"k_pair_lazy_run", "block_alloc.lazy_run"), (r"k_coc_catchup_list", "block_alloc.catchup_list"), (r"k_coc_catchup_run", "block_alloc.catchup_run"), (r"k_coc_catchup_new", "block_alloc.catchup_new"), (r"k_flush_clear", "block_alloc.flush_clear"), (r"k_cell_alloc", "block_alloc.cell_alloc"), (r"k_block_init", "block_alloc.block_init"), (r"op_stream", "stream_changed"), (r"k_place_probe", "create.place_probe"), (r"k_cell_alloc|k_block_init|k_clear|k_flush_clear|op_cell_|k_pool_advance|rocprim", "block_alloc"),

]


def short(name):
    for pat, s in SHORT:
        if re.search(pat, name):
            return s
    return name[:48]


def stats(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels").fetchall() if _has(db, "kernels", "name") else \
        db.execute("select kernel_name, start, end from kernels").fetchall()
    agg = defaultdict(list)
    for n, s, e in rows:
        agg[short(n)].append(e - s)
    tot = sum(sum(v) for v in agg.values())
    out = ["%-24s %8s %14s %12s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "median_ns", "min_ns", "max_ns", "pct")]
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        out.append("%-24s %8d %14d %12.0f %12d %12d %12d %6.2f%%" % (k, len(v), sum(v), sum(v) / len(v), sorted(v)[len(v) // 2], min(v), max(v), 100.0 * sum(v) / tot))
    return "\n".join(out)


def _has(db, view, col):
    try:
        db.execute("select %s from %s limit 1" % (col, view))
        return True
    except sqlite3.Error:
        return False


def pmc_per_kernel(path, counter):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    ncol = "kernel_name" if "kernel_name" in cols else "name"
    ccol = "counter_name" if "counter_name" in cols else "pmc_name"
    vcol = "value" if "value" in cols else "counter_value"
    dcol = "dispatch_id" if "dispatch_id" in cols else "id"
    rows = db.execute("select %s, %s, %s, %s from counters_collection" % (ncol, dcol, ccol, vcol)).fetchall()
    per = defaultdict(lambda: defaultdict(float))
    for n, d, c, v in rows:
        if c == counter:
            per[short(n)][d] += float(v)
    return {k: (sum(v.values()) / len(v), len(v)) for k, v in per.items()}


def pmc(fetch_db, write_db):
    f = pmc_per_kernel(fetch_db, "FETCH_SIZE")
    w = pmc_per_kernel(write_db, "WRITE_SIZE")
    out = ["%-16s %8s %16s %18s %16s %18s" % ("kernel", "launches", "FETCH_SIZE_KiB", "fetch_corrected_MB", "WRITE_SIZE_KiB", "hbm_bytes_MB/launch")]
    js = {}
    for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, (0, 0))[0] + w.get(k, (0, 0))[0])):
        fk, n = f.get(k, (0.0, 0))
        wk, _ = w.get(k, (0.0, 0))
        total = (2.0 * fk + wk) * 1024.0
        js[k] = round(total)
        out.append("%-16s %8d %16.0f %18.1f %16.0f %18.1f" % (k, n, fk, 2.0 * fk * 1024 / 1e6, wk, total / 1e6))
    # one launch of pass Y per map update; the two passes are separate runs of a command whose number of timed regions depends on
    # the clock, so each pass is normalised by its OWN number of map updates
    steps_f = f.get("edt_pass_y", (0, 0))[1] or w.get("edt_pass_y", (0, 0))[1]
    steps_w = w.get("edt_pass_y", (0, 0))[1] or steps_f
    steps = steps_f
    if steps:
        # bytes per MAP UPDATE under the names gie_profile_read / bench.py use (a stage may be several kernels)
        per_step = defaultdict(float)
        for k in set(f) | set(w):
            if k.startswith("__") or k.startswith("void "):
                continue
            fk, n = f.get(k, (0.0, 0)); wk, n2 = w.get(k, (0.0, 0))
            per_step[k.split(".")[0]] += (2.0 * fk * n / steps_f + wk * n2 / steps_w) * 1024.0
        js["_per_step_by_stage"] = {k: round(v) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])}
        js["_map_updates"] = steps
        out.append("")
        out.append("per MAP UPDATE and stage (the names of gie_profile_read; a stage may be several kernels), MB: "
                   + ", ".join("%s %.1f" % (k, v / 1e6) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])))
        # the map update's own kernels only: torch's label generation between the timed regions, the copies and fills of the harness
        # and gie_create's placement probe are not part of an update (83.8 GB "per map update")
        tot = sum(v for k, v in per_step.items() if k != "create")
        js["_per_step_total_bytes"] = round(tot)
        out.append("")
        out.append("the kernels of one map update together (without gie_create's probe, the harness's torch kernels and copies): %.1f MB (%d / %d map updates in the FETCH / WRITE pass)" % (tot / 1e6, steps_f, steps_w))
    out.append("")
    out.append("per launch = mean over the launches of the run; fetch_corrected = 2 x FETCH_SIZE: calibrated on this pool for coalesced streams of")
    out.append("1, 2, 4, 8 and 16 bytes per lane and for 8-byte records in 64-byte runs (profiles/r05_pmc_calibration.txt: factor 2.000, runs 1.99);")
    out.append("WRITE_SIZE is exact for 4 / 8 / 16-byte stores (1-byte stores: 0.955); hbm_bytes = fetch_corrected + WRITE_SIZE.")
    return "\n".join(out), js


def series(path, pat):
    """durations (us) of the launches of the kernels whose short name matches `pat`, in launch order"""
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall() if _has(db, "kernels", "name") else \
        db.execute("select kernel_name, start, end from kernels order by start").fetchall()
    return [round((e - s0) / 1e3, 1) for n, s0, e in rows if re.search(pat, short(n))]


if __name__ == "__main__":
    if sys.argv[1] == "series":
        print(" ".join(str(v) for v in series(sys.argv[2], sys.argv[3])))
    elif sys.argv[1] == "stats":
        print(stats(sys.argv[2]))
    else:
        txt, js = pmc(sys.argv[2], sys.argv[3])
        print(txt)
        if len(sys.argv) > 4:
            json.dump(js, open(sys.argv[4], "w"), indent=1)
