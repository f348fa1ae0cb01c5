cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_r03m.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_r03m.log); tail -6 gpurun_out/pytest_gpu_r03m.log
