#!/bin/bash
# Round-3 measurement probe (via gpurun): counter list, FETCH/WRITE_SIZE calibration on the kernels' access shapes,
# matrix-core busy counters of the headline bench.  usage: bash tools/r3_probe.sh <tag>
tag=${1:-r03a}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 -L > gpurun_out/counters_all_$tag.txt 2>&1
grep -i -E "^\s*(Name|name)?.*(MFMA|SQ_BUSY|GRBM_GUI_ACTIVE|SQ_ACTIVE_INST_VALU|SQ_WAVE_CYCLES|SQ_INSTS_VALU )" gpurun_out/counters_all_$tag.txt | sort -u | head -60 > gpurun_out/counters_mfma_$tag.txt
have() { grep -q -w "$1" gpurun_out/counters_all_$tag.txt; }
# ---- calibration
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/cal_${c}_$tag
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/cal_${c}_$tag -o p -- tools/bin/ubench_traffic > gpurun_out/cal_${c}_$tag.log 2>&1
done
python tools/pmc_summary.py gpurun_out/cal_FETCH_SIZE_$tag/p_results.db gpurun_out/cal_WRITE_SIZE_$tag/p_results.db > gpurun_out/cal_summary_$tag.txt 2>&1
cat gpurun_out/cal_FETCH_SIZE_$tag.log | tail -4; cat gpurun_out/cal_summary_$tag.txt
# ---- matrix-core counters of the headline bench
sel=""
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F64; do
  if have $c; then sel="$sel $c"; fi
done
echo "SQ counters selected:$sel"
rm -rf gpurun_out/pmcm_$tag
timeout 400 rocprofv3 --pmc $sel --kernel-trace -d gpurun_out/pmcm_$tag -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmcm_$tag.log 2>&1
tail -2 gpurun_out/pmcm_$tag.log | cut -c1-300
dbs=gpurun_out/pmcm_$tag/p_results.db
if have GRBM_GUI_ACTIVE; then
  rm -rf gpurun_out/pmcg_$tag
  timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/pmcg_$tag -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmcg_$tag.log 2>&1
  dbs="$dbs gpurun_out/pmcg_$tag/p_results.db"
fi
python tools/pmc_counters.py $dbs > gpurun_out/mfma_util_$tag.txt 2>&1
head -14 gpurun_out/mfma_util_$tag.txt | cut -c1-260
find gpurun_out -name "*.db" -size +30M -delete
