#!/usr/bin/env python3
"""Turn rocprofv3 rocpd SQLite outputs into the small text summaries committed here.

    python profiles/summarize_rocpd.py <tag> <kernel-trace.db> [<pmc.db> ...]

Writes profiles/<tag>_kernel_stats.csv (the `--stats` view: calls, total/avg ns, %)
and profiles/<tag>_pmc.csv (per kernel and counter: dispatches, summed value, per launch).
"""
import csv
import os
import sqlite3
import sys

HERE = os.environ.get("PROFILE_OUT") or os.path.dirname(os.path.abspath(__file__))  # PROFILE_OUT: write elsewhere


def kernel_stats(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    extra = {
        r[0]: r[1:]
        for r in cur.execute(
            "select name, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size),"
            " max(grid_x), max(workgroup_x) from kernels group by name"
        )
    }
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "pct", "vgpr_rocpd_raw", "agpr_rocpd_raw", "sgpr", "lds_bytes", "scratch",
                    "grid_x", "workgroup_x"])
        for name, calls, tot, avg, pct in rows:
            w.writerow([name, calls, f"{tot:.0f}", f"{avg:.0f}", f"{pct:.3f}", *extra.get(name, [""] * 7)])
    return rows


def pmc(dbs, out):
    acc = {}
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        q = ("select kernel_name, counter_name, count(distinct dispatch_id), sum(value) from counters_collection "
             "group by kernel_name, counter_name")
        for k, c, n, v in cur.execute(q):
            acc[(k, c)] = (n, v)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "sum", "per_launch"])
        for (k, c), (n, v) in sorted(acc.items()):
            w.writerow([k, c, n, f"{v:.1f}", f"{v / max(n, 1):.1f}"])
    return acc


if __name__ == "__main__":
    tag = sys.argv[1]
    rows = kernel_stats(sys.argv[2], os.path.join(HERE, f"{tag}_kernel_stats.csv"))
    for r in rows:
        print(r)
    if len(sys.argv) > 3:
        acc = pmc(sys.argv[3:], os.path.join(HERE, f"{tag}_pmc.csv"))
        for k, v in sorted(acc.items()):
            if "tfhe" in k[0]:
                print(k[0][:40], k[1], v)
