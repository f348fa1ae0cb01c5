#!/bin/bash
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in 2a 3 5 1 2b; do
timeout 400 python bench.py --config $c --no-cpu-baseline --steps 10 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('cfg $c', d['value'], 'factor', d['kkt_factor_ms'], 'solve', d['kkt_solve_ms_per_call'], 'e2e', d.get('ipm_iterations_per_s_end_to_end'), d.get('refined_block_solves', {}).get('factorisations_with_some'), d.get('refined_block_solves', {}).get('blocks_last_factorisation'))"
done 2>&1 | cut -c1-300 | tee gpurun_out/${1:-r5cfgs}.txt
