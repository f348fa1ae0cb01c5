#!/bin/bash
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocm-smi --showclocks --showperflevel 2>&1 | grep -E "sclk|fclk|mclk|Perf" | head
for e in "HIPKKT_VERBOSE=0" "HIPKKT_FB_V2=0"; do
env $e timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); fb = d['roofline']['kernels']['front_block']
print('$e', d['value'], 'factor', d['kkt_factor_ms'], 'solve', d['kkt_solve_ms_per_call'], 'fb us/launch', round(1e3 * fb['ms_per_refactor'] / 17, 1), 'frac', d['roofline']['frac'], d.get('box_probe'))"
done 2>&1 | cut -c1-600 | tee gpurun_out/${1:-r5box}.txt
rocm-smi --showclocks 2>&1 | grep -E "sclk|fclk" | head -4
