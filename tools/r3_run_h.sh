cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_kkt.py -q -m gpu -k "update_scaling_dev" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "matches_oracle and (324 or 208 or 179)" -s 2>&1 | grep -E "batch-parity|passed|failed" | cut -c1-400
echo "---- lookahead correctness"
HIPKKT_LOOKAHEAD=1 HIPKKT_VERBOSE=1 timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m "gpu and not slow" -k "cfg2a or cfg3" 2>&1 | grep -E "look-ahead|passed|failed|Error" | sort | uniq -c | tail -8
echo "---- lookahead timing"
VARIANTS="default: nograph:HIPKKT_NO_GRAPH=1 lookahead:HIPKKT_LOOKAHEAD=1" SKIP_TESTS=1 bash tools/r3_ab.sh r03h 2a 3 2>&1 | grep -E "^AB|BENCH"
