#!/bin/bash
# round 6, call 28: the vector-only interval mask in attn_fwd32_kernel (vision towers, short key ranges): tests + timing
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_edge_cases_gpu.py -m gpu -x -q -k "attention or attn" 2>&1 | tail -3 > $O/r06_c28.txt
for i in 1 2; do timeout 300 python tools/bench_attn_vit.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read())['vit_attention']; print({k: v['us'] for k, v in d.items() if isinstance(v, dict)})" >> $O/r06_c28.txt; done
TR1_FWD64=0 timeout 300 python tools/bench_attn.py --no-check --iters 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fwd32 forced at the config-3 call:', d['fwd_ms'])" >> $O/r06_c28.txt
timeout 300 python tools/sweep_fwd64.py 2>/dev/null | head -5 >> $O/r06_c28.txt
cat $O/r06_c28.txt
