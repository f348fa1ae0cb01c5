#!/usr/bin/env python
"""Developer tool: print a window of the kernel timeline (start, duration, stream/queue, grid) from a rocprofv3
sqlite database, to see which kernels overlapped.  usage: trace_timeline.py <results.db> [first_kernel_substr] [count] [which_refactor]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
names = {r[0]: r[1] for r in c.execute(f"select id,kernel_name from {ks}")}
rows = list(c.execute(f"select kernel_id,start,end,grid_size_x,workgroup_size_x,queue_id,stream_id from {kd} order by start"))
sub = sys.argv[2] if len(sys.argv) > 2 else "k_update_dense"
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 60
def short(n):
    n = n.replace("hipkkt::", "").split("(")[0]
    return n[-40:]
# start at the LAST refactorisation: find last k_init_panels / first matching kernel after it
idx = [i for i, r in enumerate(rows) if "init_panels" in names[r[0]]]
which = int(sys.argv[4]) if len(sys.argv) > 4 else -1
i0 = idx[which] if idx else 0
big = [i for i in range(i0, len(rows)) if sub in names[rows[i][0]] and rows[i][3] // rows[i][4] >= 200]
i1 = max(i0, big[0] - 12) if big else i0
t0 = rows[i1][1]
prev_end = t0
for r in rows[i1:i1 + cnt]:
    k, s, e, g, w, q, st = r
    print("%9.1f us  dur %7.1f  gap %6.1f  q%-2s s%-2s  wgs %5d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, q, st, g // w, short(names[k])))
    prev_end = max(prev_end, e)
