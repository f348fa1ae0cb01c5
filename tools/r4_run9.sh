#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python tools/fb_trace.py > gpurun_out/fbtrace_c.txt 2>&1; grep -A5 "wg 1 streamed" gpurun_out/fbtrace_c.txt | head -14; grep "hand-off" gpurun_out/fbtrace_c.txt; grep chain gpurun_out/fbtrace_c.txt
