#!/bin/bash
# round 6, call 15: dQ kernel form A/B: isolated backward (no profiler, two alternations) and inside the step
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
for i in 1 2; do timeout 300 python tools/check_dq64.py 2>/dev/null | head -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('isolated bwd ms: dq32', d['bwd_ms_dq32'], 'dq64', d['bwd_ms_dq64'], 'bit-equal', d['dQ_bit_equal'])" >> $O/r06_c15_ab_dq64.txt; done
bash tools/ab_env_bench.sh TR1_DQ64 1 0 1 0 >> $O/r06_c15_ab_dq64.txt 2>&1
cat $O/r06_c15_ab_dq64.txt
