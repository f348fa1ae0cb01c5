#!/bin/bash
# Copies the summaries of a tools/final_round.sh run from gpurun_out/ (scratch) into profiles/ (tracked).  usage: tools/collect_round.sh <tag>
tag=$1
for c in 2a 2b 1 3 5 4; do tail -1 gpurun_out/bench_${c}_$tag.log > profiles/${tag}_bench_cfg$c.json; done
for c in 2a 2b 3 5; do cp gpurun_out/prof_summary_${c}_$tag.txt profiles/${tag}_cfg${c}_kernel_stats.txt; done
for c in 2a 3 5 2b 1 4; do
  [ -f gpurun_out/counters_${c}_$tag.json ] || continue
  cp gpurun_out/pmc_summary_${c}_$tag.txt profiles/${tag}_cfg${c}_pmc_hbm_traffic.txt
  cp gpurun_out/mfma_util_${c}_$tag.txt profiles/${tag}_cfg${c}_mfma_util.txt
  cp gpurun_out/counters_${c}_$tag.json profiles/${tag}_cfg${c}_counters.json
done
cp gpurun_out/parity_causes_$tag.txt profiles/${tag}_parity_causes.txt
cp gpurun_out/fbtrace_$tag.txt profiles/${tag}_cfg2a_front_block_stamps.txt
tail -3 gpurun_out/pytest_gpu_$tag.log > profiles/${tag}_pytest_gpu_summary.txt
