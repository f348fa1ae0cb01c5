#!/bin/bash
# Round-6 developer loop on the GPU box: parity of the factorisation path, bench line of cfg 2a, per-launch timeline of one
# refactorisation.   usage (via gpurun): bash tools/r6_quick.sh <tag> [tests: kkt|none|full] [configs for extra bench lines]
tag=${1:-r6}; tests=${2:-kkt}; cfgs=${3:-}; kexpr=${4:-}
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
if [ "$tests" = "kkt" ]; then
(timeout 900 python -m pytest tests/test_gpu_kkt.py -x -q -m gpu > gpurun_out/${tag}_pytest_kkt.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_pytest_kkt.txt); tail -2 gpurun_out/${tag}_pytest_kkt.txt
(timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "full_size_matches_oracle or full_size_solve_properties or second_form" > gpurun_out/${tag}_pytest_fs.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_pytest_fs.txt); tail -2 gpurun_out/${tag}_pytest_fs.txt
if [ -n "$kexpr" ]; then
(timeout 900 python -m pytest tests -x -q -m gpu -k "$kexpr" > gpurun_out/${tag}_pytest_k.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_pytest_k.txt); tail -4 gpurun_out/${tag}_pytest_k.txt
fi
fi
if [ "$tests" = "full" ]; then
(timeout 2400 python -m pytest tests -x -q -m "gpu and not slow" > gpurun_out/${tag}_pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_pytest_gpu.txt); tail -3 gpurun_out/${tag}_pytest_gpu.txt
fi
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench_2a.json 2> gpurun_out/${tag}_bench_2a.err; python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${tag}_bench_2a.json").read().strip().splitlines()[-1])
    print("cfg2a", d["value"], "units/s factor", d.get("kkt_factor_ms"), "solve", d.get("kkt_solve_ms"), "frac", d["roofline"]["frac"], "parity", d.get("parity"))
except Exception as e:
    print("bench failed", e)
PY
for c in $cfgs; do
  extra=""; [ $c = 3 -o $c = 5 ] && extra="--device-scaling"; [ $c = 5 ] && extra="$extra --steps 10 --warmup 2"
  timeout 500 python bench.py --config $c --no-cpu-baseline $extra > gpurun_out/${tag}_bench_$c.json 2> gpurun_out/${tag}_bench_$c.err
  python -c "
import json
d = json.loads(open('gpurun_out/${tag}_bench_$c.json').read().strip().splitlines()[-1]); print('cfg$c', d['value'], 'factor', d.get('kkt_factor_ms'), 'solve', d.get('kkt_solve_ms'))"
done
rm -rf gpurun_out/prof_$tag
timeout 400 rocprofv3 --kernel-trace -d gpurun_out/prof_$tag -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_$tag.log 2>&1
db=$(ls gpurun_out/prof_$tag/*results.db | head -1)
python tools/prof_summary.py $db > gpurun_out/${tag}_kernel_stats.txt 2>&1
python tools/timeline.py $db 200 > gpurun_out/${tag}_factor_timeline.txt 2>&1
find gpurun_out -name "*.db" -delete
head -40 gpurun_out/${tag}_kernel_stats.txt | cut -c1-150
