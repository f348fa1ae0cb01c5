#!/bin/bash
tag=${1:-r5t}
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python tools/fb2_trace.py > gpurun_out/${tag}_fb2trace.txt 2>&1; grep -A7 "batch 8" gpurun_out/${tag}_fb2trace.txt | cut -c1-400; tail -1 gpurun_out/${tag}_fb2trace.txt
