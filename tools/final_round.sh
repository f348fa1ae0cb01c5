#!/bin/bash
# Final evidence of a round: GPU test suite, bench lines for every config, rocprofv3 kernel stats and the hardware-counter passes
# of the headline config (FETCH_SIZE, WRITE_SIZE, SQ matrix-core counters, GRBM_GUI_ACTIVE -- each in its OWN pass, with
# --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).   usage (via gpurun): bash tools/final_round.sh <tag> [quick]
tag=${1:-final}; mode=${2:-full}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
sel="gpu"; [ "$mode" = "quick" ] && sel="gpu and not slow"
(timeout 1800 python -m pytest tests -m "$sel" -q > gpurun_out/pytest_gpu_$tag.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.log); tail -3 gpurun_out/pytest_gpu_$tag.log
grep -h "batch-parity\|order-parity" gpurun_out/pytest_gpu_$tag.log > gpurun_out/parity_causes_$tag.txt
# ---- counter passes of the headline config first: bench.py then finds a counter file stamped with the current kernel sources
pass() { rm -rf gpurun_out/$1_$tag; timeout 500 rocprofv3 --pmc $2 --kernel-trace -d gpurun_out/$1_$tag -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$1_$tag.log 2>&1; }
pass pmcf FETCH_SIZE
pass pmcw WRITE_SIZE
pass pmcm "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES"
pass pmcg GRBM_GUI_ACTIVE
python tools/pmc_summary.py gpurun_out/pmcf_$tag/p_results.db gpurun_out/pmcw_$tag/p_results.db > gpurun_out/pmc_summary_$tag.txt 2>&1
python tools/pmc_counters.py gpurun_out/pmcm_$tag/p_results.db gpurun_out/pmcg_$tag/p_results.db > gpurun_out/mfma_util_$tag.txt 2>&1
python tools/pmc_to_json.py 2a gpurun_out/counters_2a_$tag.json gpurun_out/pmcf_$tag/p_results.db gpurun_out/pmcw_$tag/p_results.db gpurun_out/pmcm_$tag/p_results.db gpurun_out/pmcg_$tag/p_results.db > gpurun_out/counters_2a_$tag.log 2>&1
tail -1 gpurun_out/counters_2a_$tag.log | cut -c1-400
cp gpurun_out/counters_2a_$tag.json profiles/${tag}_cfg2a_counters.json 2>/dev/null     # (the GPU box's copy: bench.py below reads it)
# ---- bench lines
timeout 900 python bench.py > gpurun_out/bench_2a_$tag.log 2>&1; tail -1 gpurun_out/bench_2a_$tag.log | cut -c1-200
for c in 2b 1; do timeout 400 python bench.py --config $c > gpurun_out/bench_${c}_$tag.log 2>&1; tail -1 gpurun_out/bench_${c}_$tag.log | cut -c1-160; done
timeout 400 python bench.py --config 3 --device-scaling > gpurun_out/bench_3_$tag.log 2>&1; tail -1 gpurun_out/bench_3_$tag.log | cut -c1-160
timeout 400 python bench.py --config 5 --steps 10 --warmup 2 --device-scaling > gpurun_out/bench_5_$tag.log 2>&1; tail -1 gpurun_out/bench_5_$tag.log | cut -c1-160
timeout 900 python bench.py --config 4 --warmup 4 > gpurun_out/bench_4_$tag.log 2>&1; tail -1 gpurun_out/bench_4_$tag.log | cut -c1-300
timeout 600 python tools/e2e_opts.py 2a 3 2>&1 | grep -E "^E2E|rror" > gpurun_out/e2e_$tag.txt; cat gpurun_out/e2e_$tag.txt | cut -c1-200
# ---- kernel stats
for c in 2a 2b; do
  rm -rf gpurun_out/prof_${c}_$tag
  timeout 500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${c}_$tag -o p -- python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_${c}_$tag.log 2>&1
  python tools/prof_summary.py $(ls gpurun_out/prof_${c}_$tag/*results.db | head -1) > gpurun_out/prof_summary_${c}_$tag.txt 2>&1
done
head -8 gpurun_out/pmc_summary_$tag.txt
find gpurun_out -name "*.db" -size +30M -delete
