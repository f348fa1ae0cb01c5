#!/bin/bash
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_kkt.py -x -q -m gpu > gpurun_out/r5_panel_pytest.txt 2>&1; echo "rc=$?" >> gpurun_out/r5_panel_pytest.txt); tail -2 gpurun_out/r5_panel_pytest.txt
bash tools/r5_cfgs.sh ${1:-r5panel}
