#!/bin/bash
# round 5: fused merge + o projection: test, then A/B of the decode step (TR1_O_FUSED=0/1, two alternations)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out/r05_ofused_ab.txt; : > $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "merge_fused or decode" 2>&1 | tail -15 >> $O
for rep in 1 2; do for v in 0 1; do echo "== TR1_O_FUSED=$v" >> $O; TR1_O_FUSED=$v timeout 400 python tools/decode_steps_probe.py 2>&1 | grep "rep" >> $O; done; done
cat $O
