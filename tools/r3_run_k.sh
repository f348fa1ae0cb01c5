cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
VARIANTS="default: laA:HIPKKT_LOOKAHEAD=1,HIPKKT_LA_SCHED=0 laA6:HIPKKT_LOOKAHEAD=1,HIPKKT_LA_SCHED=0,HIPKKT_LA_MAXGF=6 laA4:HIPKKT_LOOKAHEAD=1,HIPKKT_LA_SCHED=0,HIPKKT_LA_MAXGF=4 laA3:HIPKKT_LOOKAHEAD=1,HIPKKT_LA_SCHED=0,HIPKKT_LA_MAXGF=3" SKIP_TESTS=1 SKIP_BENCH=1 bash tools/r3_ab.sh r03l 2a 3 2>&1 | grep -E "^AB|BENCH"
rm -rf gpurun_out/tl_la
HIPKKT_LOOKAHEAD=1 HIPKKT_LA_SCHED=0 timeout 300 rocprofv3 --kernel-trace -d gpurun_out/tl_la -o p -- python tools/ab_variant.py 2a la 3 > gpurun_out/tl_la.log 2>&1
python tools/timeline.py gpurun_out/tl_la/p_results.db > gpurun_out/timeline_la.txt 2>&1
find gpurun_out -name "*.db" -size +30M -delete
