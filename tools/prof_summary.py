#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace rocpd database (results.db) into the per-kernel table that
`rocprofv3 --stats` prints: calls, total / average / min / max duration.  Usage:
    python tools/prof_summary.py gpurun_out/prof_x/p_results.db > profiles/rNN_x_kernel_stats.txt"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = (f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
         f"max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc")
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows) or 1
    print(f"# source: {path}")
    print(f"{'kernel':<58} {'calls':>8} {'total_us':>12} {'avg_us':>10} {'min_us':>9} {'max_us':>10} {'pct':>6}")
    for name, n, t, a, mn, mx in rows:
        name = name.replace("_ZN6hipkkt", "").replace(".kd", "")
        print(f"{name[:58]:<58} {n:>8d} {t / 1e3:>12.1f} {a / 1e3:>10.2f} {mn / 1e3:>9.2f} {mx / 1e3:>10.2f} {100.0 * t / tot:>6.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
