#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05_down32.txt; : > $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_configs_gpu.py tests/test_fullsize_gpu.py -x -q -k "fixup or skinny or config4 or decode" 2>&1 | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl" | tail -4 >> $O
A="--model qwen2.5-vl-7b --frames 64 --G 16 --C 1024 --beta 0 --clip-loss --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-peak-probe --no-engine-leg"
for lib in tools/_var_ref.so "" tools/_var_ref.so ""; do echo "== TR1_HIP_LIB=$lib" >> $O; TR1_HIP_LIB=$lib timeout 900 python bench.py $A 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['ms_per_step'], 1), d.get('phases_ms_per_step'), round(d.get('rollout_tokens_per_sec') or 0))" >> $O 2>&1; done
cat $O
