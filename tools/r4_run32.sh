#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "dense_triangle or (test_full_size_matches_oracle and cfg5) or (solve_properties and cfg5)" > gpurun_out/r32_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r32_pytest.log); tail -15 gpurun_out/r32_pytest.log
HIPKKT_DENSE_TRI=0 python tools/ab_variant.py 5 view 4 | grep "^AB"
python tools/ab_variant.py 5 densetri 4 | grep "^AB"
python tools/ab_variant.py 2a base 4 | grep "^AB"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_r32; timeout 500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r32 -o p -- python tools/ab_variant.py 5 prof 6 > gpurun_out/prof_r32.log 2>&1
python tools/prof_summary.py $(ls gpurun_out/prof_r32/*results.db | head -1) 2>&1 | head -24
find gpurun_out -name "*.db" -size +30M -delete
