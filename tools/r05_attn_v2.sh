#!/bin/bash
# round 5: attn_dec32 (coalesced partial stores, plan loads behind the first DMA) + one-wave-per-row merge: tests, per-layer times, per-kernel split, in-step A/B against tools/_var_ref.so
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out/r05_attn_v2.txt; : > $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_configs_gpu.py -x -q -k "attention or attn or decode" 2>&1 | tail -4 >> $O
C3="3474 8 200 2"; C4="3266 16 1024 2"; C2="2522 8 200 2"
run() { echo "== $*" >> $O; env "$@" 2>&1 | grep -v "^$\|amdgpu.ids" >> $O; }
for lib in tools/_var_ref.so ""; do
run TR1_HIP_LIB=$lib PLAN=1 STEPS=100 timeout 120 python tools/bench_attn_decode.py $C3
run TR1_HIP_LIB=$lib PLAN=1 STEPS=1,512,1023 TR1_DECODE_NSPLIT=16 timeout 120 python tools/bench_attn_decode.py $C4
run TR1_HIP_LIB=$lib PLAN=1 STEPS=100 NH=12 NKV=2 timeout 120 python tools/bench_attn_decode.py $C2
done
echo "#### per-kernel (new)" >> $O
tools/stats_cmd.sh c3 "attn_dec32|attn_combine" PLAN=1 STEPS=100 -- python tools/bench_attn_decode.py $C3 >> $O 2>&1
tools/stats_cmd.sh c4 "attn_dec32|attn_combine" PLAN=1 STEPS=512 TR1_DECODE_NSPLIT=16 -- python tools/bench_attn_decode.py $C4 >> $O 2>&1
tools/stats_cmd.sh c2 "attn_dec32|attn_combine" PLAN=1 STEPS=100 NH=12 NKV=2 -- python tools/bench_attn_decode.py $C2 >> $O 2>&1
echo "#### in-step" >> $O
for lib in tools/_var_ref.so "" tools/_var_ref.so ""; do echo "== TR1_HIP_LIB=$lib" >> $O; TR1_HIP_LIB=$lib timeout 400 python tools/decode_steps_probe.py 2>&1 | grep "rep 1" >> $O; done
cat $O
