cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m "gpu and not slow" -k "super_block or solve_properties or (matches_oracle and cfg)" 2>&1 | tail -2
VARIANTS="default:" SKIP_TESTS=1 SKIP_BENCH=1 bash tools/r3_ab.sh r03q 2a 3 2>&1 | grep -E "^AB|BENCH"
rm -rf gpurun_out/tl_x
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/tl_x -o p -- python tools/ab_variant.py 2a x 3 > gpurun_out/tl_x.log 2>&1
python tools/timeline.py gpurun_out/tl_x/p_results.db 2>&1 | grep -E "invert|span"
find gpurun_out -name "*.db" -size +30M -delete
