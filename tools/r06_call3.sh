#!/bin/bash
# round 6, call 3: delta / lse2 / qmeta folded into the dQ kernel's prologue, the live-96 vision attention launch; tests + A/B
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py tests/test_edge_cases_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -5 > $O/r06_c3_tests.txt
timeout 300 python tools/bench_attn_vit.py > $O/r06_c3_vit.json 2> $O/r06_c3_vit.err
for rep in 1 2; do
  for v in 1 0; do
    TR1_BWD_FUSE_DELTA=$v timeout 300 python tools/bench_attn.py --no-check --iters 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('FUSE_DELTA=$v', d['fwd_ms'], d['bwd_ms'])" >> $O/r06_c3_attn.txt
  done
done
timeout 900 python -m pytest tests/test_trainer_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | tail -4 >> $O/r06_c3_tests.txt
cat $O/r06_c3_tests.txt $O/r06_c3_vit.json $O/r06_c3_attn.txt; tail -3 $O/r06_c3_vit.err
