#!/bin/bash
# Round-6 developer loop: kernel timeline (both streams) of the solves of one KKT iteration unit of cfg 2a
tag=${1:-r6st}
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$tag
timeout 400 rocprofv3 --kernel-trace -d gpurun_out/prof_$tag -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_$tag.log 2>&1
db=$(ls gpurun_out/prof_$tag/*results.db | head -1)
python - $db <<'PY' > gpurun_out/${tag}_solve_timeline.txt
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
names = {r[0]: r[1] for r in c.execute(f"select id,kernel_name from {ks}")}
rows = list(c.execute(f"select kernel_id,start,end,grid_size_x,workgroup_size_x,queue_id,stream_id from {kd} order by start"))
# the LAST unit: from the last k_invert_super to the end
idx = [i for i, r in enumerate(rows) if "invert_super" in names[r[0]]]
# skip the profiling-mode refactorisations at the very end: take the last one that is followed by >= 60 solve kernels
for i0 in reversed(idx):
    nxt = [i for i in idx if i > i0]
    i1 = nxt[0] if nxt else len(rows)
    if sum(1 for r in rows[i0:i1] if "k_permute_in" in names[r[0]]) >= 6:
        break
t0 = rows[i0][2]
for r in rows[i0:i1]:
    k, s, e, g, w, q, st = r
    n = names[k].replace("hipkkt::", "").split("(")[0][-34:]
    print("%9.1f +%7.1f us  q%-2s wgs %5d x%4d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, g // w, w, n))
PY
find gpurun_out -name "*.db" -delete
head -120 gpurun_out/${tag}_solve_timeline.txt
