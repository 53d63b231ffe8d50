#!/bin/bash
# round 6, call 10: attn_fwd64 as the default head-dim-128 forward: whole -m gpu suite, isolated attention timing, driver-args bench
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/r06_c10_tests.txt
timeout 300 python tools/check_fwd64.py 2>/dev/null | grep -v "true, \"nan\": false}$" > $O/r06_c10_fwd64.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_c10_bench.json 2> $O/r06_c10_bench.err
cat $O/r06_c10_tests.txt $O/r06_c10_fwd64.txt; tail -3 $O/r06_c10_bench.err; cat $O/r06_c10_bench.json | cut -c1-1500
