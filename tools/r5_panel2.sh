#!/bin/bash
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_kkt.py -x -q -m gpu > gpurun_out/r5_panel2_pytest.txt 2>&1; echo "rc=$?" >> gpurun_out/r5_panel2_pytest.txt); tail -2 gpurun_out/r5_panel2_pytest.txt
(timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "full_size_solve_properties or split_k or dense_triangle or super_block" > gpurun_out/r5_panel2_full.txt 2>&1; echo "rc=$?" >> gpurun_out/r5_panel2_full.txt); tail -3 gpurun_out/r5_panel2_full.txt
HIPKKT_VERBOSE=1 timeout 300 python bench.py --config 5 --no-cpu-baseline --steps 3 2>&1 | grep "single 64-column" | head -2
HIPKKT_VERBOSE=1 timeout 300 python bench.py --config 2b --no-cpu-baseline --steps 3 2>&1 | grep "single 64-column" | head -2
bash tools/r5_cfgs.sh ${1:-r5panel2}
for c in 5 2b; do
HIPKKT_PANEL_V2=0 timeout 400 python bench.py --config $c --no-cpu-baseline --steps 10 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('PANEL_V2=0 cfg $c', d['value'], 'factor', d['kkt_factor_ms'], 'solve', d['kkt_solve_ms_per_call'])"
done
