#!/bin/bash
# Final evidence of a round: full GPU test suite, bench lines for every config, rocprofv3 kernel stats and
# the two PMC passes for the headline config.  usage (via gpurun): bash tools/final_round.sh <tag>
tag=${1:-final}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_$tag.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.log); tail -3 gpurun_out/pytest_gpu_$tag.log
timeout 600 python bench.py > gpurun_out/bench_2a_$tag.log 2>&1; tail -1 gpurun_out/bench_2a_$tag.log | cut -c1-200
for c in 2b 1; do timeout 400 python bench.py --config $c > gpurun_out/bench_${c}_$tag.log 2>&1; tail -1 gpurun_out/bench_${c}_$tag.log | cut -c1-160; done
timeout 400 python bench.py --config 3 --device-scaling > gpurun_out/bench_3_$tag.log 2>&1; tail -1 gpurun_out/bench_3_$tag.log | cut -c1-160
timeout 400 python bench.py --config 5 --steps 10 --warmup 2 --cpu-budget 60 --device-scaling > gpurun_out/bench_5_$tag.log 2>&1; tail -1 gpurun_out/bench_5_$tag.log | cut -c1-160
timeout 600 python bench.py --config 4 --steps 128 --warmup 4 > gpurun_out/bench_4_$tag.log 2>&1; tail -1 gpurun_out/bench_4_$tag.log | cut -c1-300
for c in 2a 2b; do
  rm -rf gpurun_out/prof_${c}_$tag
  timeout 500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${c}_$tag -o p -- python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_${c}_$tag.log 2>&1
  python tools/prof_summary.py $(ls gpurun_out/prof_${c}_$tag/*results.db | head -1) > gpurun_out/prof_summary_${c}_$tag.txt 2>&1
done
rm -rf gpurun_out/pmcf_$tag gpurun_out/pmcw_$tag
timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmcf_$tag -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmcf_$tag.log 2>&1
timeout 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmcw_$tag -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmcw_$tag.log 2>&1
python tools/pmc_summary.py gpurun_out/pmcf_$tag/p_results.db gpurun_out/pmcw_$tag/p_results.db > gpurun_out/pmc_summary_$tag.txt 2>&1
head -8 gpurun_out/pmc_summary_$tag.txt
find gpurun_out -name "*.db" -size +30M -delete
