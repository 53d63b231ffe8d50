#!/bin/bash
# round 6, call 16: BASELINE configs[0] on the host cores of the GPU box with the CPU oracle, fp32 and bf16 (review item 7 / SURVEY 8d); transposes in isolation
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
python - > $O/r06_c16_transpose_isolated.txt 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, ".")
import time_r1_amd
from time_r1_amd.ops import HipOps
ops = HipOps("cuda:0")
for (R, C) in ((5074, 3584), (5074, 4608), (1600, 152064)):
    x = torch.randn(R, C, device="cuda").to(torch.bfloat16)
    for _ in range(3): ops.transpose(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.transpose(x)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 50
    print("transpose %d x %d bf16: %.1f us, %.2f TB/s (read + write)" % (R, C, t * 1e3, 2 * R * C * 2 / t / 1e9))
PY
cat $O/r06_c16_transpose_isolated.txt
timeout 1200 python bench.py --cpu-config1 --cpu-config1-dtype fp32 > $O/r06_cpu_config1_fp32.json 2> $O/r06_cpu_config1_fp32.err
timeout 2400 python bench.py --cpu-config1 --cpu-config1-dtype bf16 > $O/r06_cpu_config1_bf16.json 2> $O/r06_cpu_config1_bf16.err
cat $O/r06_cpu_config1_fp32.json $O/r06_cpu_config1_bf16.json | cut -c1-900; tail -2 $O/r06_cpu_config1_bf16.err
