#!/bin/bash
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_kkt.py -x -q -m gpu -k "developer_switches" > gpurun_out/r5_v1check.txt 2>&1; echo "rc=$?" >> gpurun_out/r5_v1check.txt); tail -2 gpurun_out/r5_v1check.txt
(timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "second_form or streamed_pivot" >> gpurun_out/r5_v1check.txt 2>&1; echo "rc=$?" >> gpurun_out/r5_v1check.txt); tail -2 gpurun_out/r5_v1check.txt
python -c "import __graft_entry__ as g; g.smoke()"
