#!/usr/bin/env python
"""Developer tool: per-launch timeline of the LAST factorisation in a rocprofv3 --kernel-trace database (start relative to the
first k_init_panels / k_factor launch of that factorisation, duration, queue, grid).  usage: timeline.py <results.db> [max_rows]"""
import sqlite3
import sys


def main(path, nmax=400):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    gcol = "grid_size_x" if "grid_size_x" in cols else ("grid_x" if "grid_x" in cols else "0")
    wcol = "workgroup_size_x" if "workgroup_size_x" in cols else "1"
    rows = list(cur.execute(f"select s.kernel_name, d.start, d.end, d.{qcol}, d.{gcol}, d.{wcol} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    starts = [i for i, r in enumerate(rows) if "k_init_panels" in r[0]]
    if not starts:
        print("no factorisation found")
        return
    i0 = starts[-1]
    i1 = next((i for i in range(i0 + 1, len(rows)) if "k_permute_in" in rows[i][0]), len(rows))
    t0 = rows[i0][1]
    print(f"# last factorisation: launches {i0}..{i1}, span {(max(r[2] for r in rows[i0:i1]) - t0) / 1e3:.1f} us")
    prev_end = {}
    for name, st, en, q, g, w in rows[i0:i1][:nmax]:
        name = name.replace("_ZN6hipkkt", "").replace(".kd", "")[:34]
        gap = (st - prev_end.get(q, st)) / 1e3
        prev_end[q] = en
        print(f"{(st - t0) / 1e3:9.1f} +{(en - st) / 1e3:7.1f} us  q{q:<3} gap {gap:7.1f}  wgs {g // max(w, 1):>6}  {name}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 400)
