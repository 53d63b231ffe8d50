#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05_fp8down.txt; : > $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py tests/test_trainer_gpu.py -x -q -k "fp8 or w8 or drift" 2>&1 | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl" | tail -4 >> $O
X="--no-cpu-baseline --no-roofline --no-peak-probe --no-engine-leg"
for lib in tools/_var_ref.so "" tools/_var_ref.so ""; do echo "== TR1_HIP_LIB=$lib" >> $O; TR1_HIP_LIB=$lib timeout 600 python bench.py --rollout-fp8 --steps 8 --warmup 2 $X 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['ms_per_step'], 1), d.get('phases_ms_per_step'), round(d.get('rollout_tokens_per_sec') or 0))" >> $O 2>&1; done
cat $O
