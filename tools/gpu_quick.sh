#!/bin/bash
# quick GPU check: parity suite (stop at first failure) + one bench line per config given:  bash tools/gpu_quick.sh <tag> <cfg> [<cfg> ...]
tag=$1; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for c in "$@"; do
  timeout 400 python bench.py --config $c --no-cpu-baseline > gpurun_out/bench_${c}_$tag.log 2>&1
  tail -1 gpurun_out/bench_${c}_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg', '$c', 'units/s', d['value'], 'factor_ms', d['kkt_factor_ms'], 'solve_ms', d['kkt_solve_ms_per_call'], 'e2e it/s', d['ipm_iterations_per_s_end_to_end'], 'roofline', d['roofline']['achieved'], d['roofline'].get('all_update_kernels', {}).get('achieved'))" || tail -5 gpurun_out/bench_${c}_$tag.log
done
