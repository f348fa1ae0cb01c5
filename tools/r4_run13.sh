#!/bin/bash
tag=${1:-r04p}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for c in 5 3; do
  rm -rf gpurun_out/prof_${c}_$tag
  timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${c}_$tag -o p -- python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_${c}_$tag.log 2>&1
  python tools/prof_summary.py $(ls gpurun_out/prof_${c}_$tag/*results.db | head -1) > gpurun_out/prof_summary_${c}_$tag.txt 2>&1
  head -24 gpurun_out/prof_summary_${c}_$tag.txt | cut -c1-150
  tail -1 gpurun_out/prof_${c}_$tag.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('cfg', '$c', 'value', d['value'], 'factor_ms', d['kkt_factor_ms'], 'solve_ms', d['kkt_solve_ms_per_call'], 'levels', d['config']['levels'], 'nnzL', d['config']['nnzL'], 'roofline', d['roofline']['achieved'])"
done
find gpurun_out -name "*.db" -size +30M -delete
