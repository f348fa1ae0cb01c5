#!/bin/bash
tag=${1:-r04m}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -s -k "twin_factorisation or (ipm_matches_oracle and (cfg3 or cfg2a)) or (batch_config and (324 or 179 or 208))" > gpurun_out/pytest_new_$tag.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_new_$tag.log); grep -h "full-size-ipm\|batch-parity\|passed\|failed\|rc=\|SKIP\|skipped" gpurun_out/pytest_new_$tag.log | cut -c1-400
timeout 900 python bench.py > gpurun_out/bench_2a_$tag.log 2>&1; tail -1 gpurun_out/bench_2a_$tag.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('value', d['value'], 'factor_ms', d['kkt_factor_ms'], 'solve_ms', d['kkt_solve_ms_per_call'], 'e2e', d['ipm_iterations_per_s_end_to_end'], {k: v['iterations_per_s_runs'] for k, v in d['end_to_end']['runs'].items()})
print('roofline', r['achieved'], r['frac'], 'dense', r['kernels']['dense_update']['achieved'], r['kernels']['dense_update']['frac'], 'fb', r['kernels']['front_block']['ms_per_refactor'], 'solves', r['solves']['achieved'], r['solves']['frac'])
print('parity', d.get('parity'))
print('cpu', d.get('cpu_baseline', {}).get('value'), d.get('speedup_vs_cpu_baseline'))
" || tail -5 gpurun_out/bench_2a_$tag.log
timeout 900 python bench.py --config 4 --warmup 4 > gpurun_out/bench_4_$tag.log 2>&1; tail -1 gpurun_out/bench_4_$tag.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('cfg4 value', d['value'], 'cpu', d['cpu_baseline']['value'], 'parity', json.dumps(d['parity'])[:1500])
" || tail -5 gpurun_out/bench_4_$tag.log
