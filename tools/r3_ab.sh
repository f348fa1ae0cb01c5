#!/bin/bash
# Round-3 A/B run (via gpurun): GPU parity suite on the defaults, factor / solve device times of the given configs under the
# HIPKKT_* variants listed in $VARIANTS ("label:ENV=val,ENV=val ..."), end-to-end rates, kernel trace.
#   usage: VARIANTS="..." bash tools/r3_ab.sh <tag> [cfgs...]
tag=${1:-r03b}; shift
cfgs=${@:-2a}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
if [ -z "$SKIP_TESTS" ]; then
  (timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$tag.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.log)
  tail -5 gpurun_out/pytest_gpu_$tag.log
fi
out=gpurun_out/ab_$tag.txt; : > $out
VARIANTS=${VARIANTS:-"default: superhop_off:HIPKKT_SUPERHOP=0"}
for c in $cfgs; do
  for v in $VARIANTS; do
    label=${v%%:*}; envs=${v#*:}
    env $(echo $envs | tr ',' ' ') timeout 300 python tools/ab_variant.py $c $label 2>&1 | grep -E "^AB|rror|timed out" >> $out
  done
done
cat $out
if [ -n "$E2E" ]; then timeout 600 python tools/e2e_opts.py $E2E 2>&1 | grep -E "^E2E|rror" | tee gpurun_out/e2e_$tag.txt; fi
if [ -n "$PROF" ]; then
  rm -rf gpurun_out/prof_$tag
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o p -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_$tag.log 2>&1
  python tools/prof_summary.py $(ls gpurun_out/prof_$tag/*results.db gpurun_out/prof_$tag/*/*results.db 2>/dev/null | head -1) > gpurun_out/prof_summary_$tag.txt 2>&1
  head -24 gpurun_out/prof_summary_$tag.txt
  find gpurun_out/prof_$tag -name "*.db" -size +40M -delete
fi
[ -n "$SKIP_BENCH" ] && exit 0
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/bench_2a_$tag.log 2>&1; tail -1 gpurun_out/bench_2a_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('BENCH 2a units/s', d['value'], 'ms/step', d['ms_per_step'], 'factor', d['kkt_factor_ms'], 'solve/call', d['kkt_solve_ms_per_call'], 'e2e', d['ipm_iterations_per_s_end_to_end'], 'frac', r['frac'], 'fullK', r.get('full_K_launches'))" || tail -3 gpurun_out/bench_2a_$tag.log
