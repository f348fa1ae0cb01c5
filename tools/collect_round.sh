#!/bin/bash
# Copies the summaries of a tools/final_round.sh run from gpurun_out/ (scratch) into profiles/ (tracked).  usage: tools/collect_round.sh <tag>
tag=$1
for c in 2a 2b 1 3 5 4; do tail -1 gpurun_out/bench_${c}_$tag.log > profiles/${tag}_bench_cfg$c.json; done
cp gpurun_out/prof_summary_2a_$tag.txt profiles/${tag}_cfg2a_kernel_stats.txt
cp gpurun_out/prof_summary_2b_$tag.txt profiles/${tag}_cfg2b_kernel_stats.txt
cp gpurun_out/pmc_summary_$tag.txt profiles/${tag}_cfg2a_pmc_hbm_traffic.txt
cp gpurun_out/mfma_util_$tag.txt profiles/${tag}_cfg2a_mfma_util.txt
cp gpurun_out/counters_2a_$tag.json profiles/${tag}_cfg2a_counters.json
cp gpurun_out/e2e_$tag.txt profiles/${tag}_e2e_hooks.txt
cp gpurun_out/parity_causes_$tag.txt profiles/${tag}_parity_causes.txt
tail -3 gpurun_out/pytest_gpu_$tag.log > profiles/${tag}_pytest_gpu_summary.txt
