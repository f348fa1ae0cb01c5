#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python tools/fb_trace.py > gpurun_out/fbtrace_blocks.txt 2>&1; grep -A12 "wg 0 pivot loop" gpurun_out/fbtrace_blocks.txt | head -45; grep chain gpurun_out/fbtrace_blocks.txt
HIPKKT_FB_STREAM=0 timeout 300 python tools/fb_trace.py 2>&1 | grep -A12 "wg 0 pivot loop" | head -14
