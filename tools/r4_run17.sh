#!/bin/bash
tag=${1:-r04t}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in 0.25 0; do
HIPKKT_PAD_HELD=$v timeout 900 python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_5_${tag}_$v.log 2>&1
tail -1 gpurun_out/bench_5_${tag}_$v.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; k = r['kernels']['dense_update']
print('pad $v value', d['value'], 'factor_ms', d['kkt_factor_ms'], 'solve', d['kkt_solve_ms_per_call'], 'whole', r['achieved'], 'dense', k['achieved'], 'all_upd', r['kernels']['all_update_kernels'])
for l in k.get('launches', []): print('   ', l)
"
done
