#!/bin/bash
# Final evidence of a round: GPU test suite, bench lines for every config, rocprofv3 kernel stats and the hardware-counter passes of
# cfg 2a / 3 / 5 (FETCH_SIZE, WRITE_SIZE, SQ matrix-core counters, GRBM_GUI_ACTIVE -- each in its OWN pass, with --kernel-trace only,
# as /opt/skills/guides/MI355X_MICROARCH.md prescribes).   usage (via gpurun): bash tools/final_round.sh <tag> [quick|full|notests|testsonly] [nocounters]
tag=${1:-final}; mode=${2:-full}; ctrs=${3:-counters}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
if [ "$mode" != "notests" ]; then
sel="gpu"; [ "$mode" = "quick" ] && sel="gpu and not slow"
# THE DRIVER'S COMMAND, one process, file order (round 4's red suite was a leak between two tests that six xdist workers hid):
(timeout 2400 python -m pytest tests -x -q -m "$sel" > gpurun_out/pytest_gpu_$tag.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.log); tail -3 gpurun_out/pytest_gpu_$tag.log
grep -h "batch-parity\|order-parity\|full-size-ipm\|twin-parity" gpurun_out/pytest_gpu_$tag.log > gpurun_out/parity_causes_$tag.txt
fi
[ "$mode" = "testsonly" ] && exit 0
# ---- counter passes first: bench.py then finds counter files stamped with the current kernel sources
if [ "$ctrs" = "counters" ]; then
for c in ${PMC_CONFIGS:-2a 3 5 2b 1 4}; do
  # cfg 2a / 2b / 1: the bench command itself; cfg 3 / 5: refactor + refined-solve loops without the end-to-end runs (tools/ab_variant.py);
  # cfg 4: a ONE-process sample of the batch (24 problems; rocprofv3 follows one process)
  cmd="python tools/ab_variant.py $c pmc 3"; [ $c = 2a ] && cmd="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
  [ $c = 2b -o $c = 1 ] && cmd="python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline"
  [ $c = 4 ] && cmd="python bench.py --config 4 --workers 1 --steps 24 --warmup 2 --no-cpu-baseline"
  export PMC_COMMAND="$cmd"
  pass() { rm -rf gpurun_out/$1_${c}_$tag; timeout 600 rocprofv3 --pmc $2 --kernel-trace -d gpurun_out/$1_${c}_$tag -o p -- $cmd > gpurun_out/$1_${c}_$tag.log 2>&1; }
  pass pmcf FETCH_SIZE
  pass pmcw WRITE_SIZE
  pass pmcm "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES"
  pass pmcg GRBM_GUI_ACTIVE
  python tools/pmc_summary.py gpurun_out/pmcf_${c}_$tag/p_results.db gpurun_out/pmcw_${c}_$tag/p_results.db > gpurun_out/pmc_summary_${c}_$tag.txt 2>&1
  python tools/pmc_counters.py gpurun_out/pmcm_${c}_$tag/p_results.db gpurun_out/pmcg_${c}_$tag/p_results.db > gpurun_out/mfma_util_${c}_$tag.txt 2>&1
  python tools/pmc_to_json.py $c gpurun_out/counters_${c}_$tag.json gpurun_out/pmcf_${c}_$tag/p_results.db gpurun_out/pmcw_${c}_$tag/p_results.db gpurun_out/pmcm_${c}_$tag/p_results.db gpurun_out/pmcg_${c}_$tag/p_results.db > gpurun_out/counters_${c}_$tag.log 2>&1
  tail -1 gpurun_out/counters_${c}_$tag.log | cut -c1-300
  cp gpurun_out/counters_${c}_$tag.json profiles/${tag}_cfg${c}_counters.json 2>/dev/null     # (the GPU box's copy: bench.py below reads it)
done
fi
# ---- bench lines
timeout 900 python bench.py > gpurun_out/bench_2a_$tag.log 2>&1; tail -1 gpurun_out/bench_2a_$tag.log | cut -c1-200
for c in 2b 1; do timeout 400 python bench.py --config $c > gpurun_out/bench_${c}_$tag.log 2>&1; tail -1 gpurun_out/bench_${c}_$tag.log | cut -c1-160; done
timeout 400 python bench.py --config 3 --device-scaling > gpurun_out/bench_3_$tag.log 2>&1; tail -1 gpurun_out/bench_3_$tag.log | cut -c1-160
timeout 600 python bench.py --config 5 --steps 10 --warmup 2 --device-scaling > gpurun_out/bench_5_$tag.log 2>&1; tail -1 gpurun_out/bench_5_$tag.log | cut -c1-160
timeout 900 python bench.py --config 4 --warmup 4 > gpurun_out/bench_4_$tag.log 2>&1; tail -1 gpurun_out/bench_4_$tag.log | cut -c1-300
# ---- kernel stats (refactor + refined solve loops without the end-to-end runs: tools/ab_variant.py; cfg 5's profile would otherwise
#      be dominated by the one factorisation on the robust-order twin)
for c in 2a 2b 3 5; do
  rm -rf gpurun_out/prof_${c}_$tag
  cmd="python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline"; [ $c = 5 ] && cmd="python tools/ab_variant.py 5 prof 8"
  timeout 500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${c}_$tag -o p -- $cmd > gpurun_out/prof_${c}_$tag.log 2>&1
  python tools/prof_summary.py $(ls gpurun_out/prof_${c}_$tag/*results.db | head -1) > gpurun_out/prof_summary_${c}_$tag.txt 2>&1
  grep "^AB" gpurun_out/prof_${c}_$tag.log >> gpurun_out/prof_summary_${c}_$tag.txt
done
timeout 300 python tools/fb2_trace.py > gpurun_out/fbtrace_$tag.txt 2>&1
python -c "import clarabel_jl_amd; from clarabel_jl_amd import hipkkt; print(hipkkt.box_probe(0))" > gpurun_out/box_probe_$tag.txt 2>&1
rocm-smi --showclocks --showperflevel --showcomputepartition --showmemorypartition >> gpurun_out/box_probe_$tag.txt 2>&1
head -8 gpurun_out/pmc_summary_2a_$tag.txt
find gpurun_out -name "*.db" -delete      # (only the summaries travel back: gpurun merges at most 64 MiB)
