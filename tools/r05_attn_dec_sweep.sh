#!/bin/bash
# round 5: decode attention pair (attn_dec32 + combine) at the three regimes: nsplit sweep, per-kernel split, block timeline
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out/r05_attn_sweep.txt; : > $O
C3="3474 8 200 2"; C4="3266 16 1024 2"; C2="2522 8 200 2"
run() { echo "== $*" >> $O; env "$@" 2>&1 | grep -v "^$" >> $O; }
for ns in 10 14 20 28 40 56; do run PLAN=1 STEPS=100 TR1_DECODE_NSPLIT=$ns timeout 120 python tools/bench_attn_decode.py $C3; done
for ns in 14 28 42 56; do run PLAN=1 STEPS=1,512,1023 TR1_DECODE_NSPLIT=$ns timeout 120 python tools/bench_attn_decode.py $C4; done
for ns in 10 21 32 42 64; do run PLAN=1 STEPS=100 NH=12 NKV=2 TR1_DECODE_NSPLIT=$ns timeout 120 python tools/bench_attn_decode.py $C2; done
echo "#### per-kernel" >> $O
tools/stats_cmd.sh c3 "attn_dec32|attn_combine" PLAN=1 STEPS=100 -- python tools/bench_attn_decode.py $C3 >> $O 2>&1
tools/stats_cmd.sh c4 "attn_dec32|attn_combine" PLAN=1 STEPS=512 -- python tools/bench_attn_decode.py $C4 >> $O 2>&1
tools/stats_cmd.sh c2 "attn_dec32|attn_combine" PLAN=1 STEPS=100 NH=12 NKV=2 -- python tools/bench_attn_decode.py $C2 >> $O 2>&1
echo "#### timelines" >> $O
run PLAN=1 PROBE=1 STEPS=100 TR1_HIP_LIB=tools/_probe_lib.so timeout 120 python tools/bench_attn_decode.py $C3
run PLAN=1 PROBE=1 STEPS=512 TR1_HIP_LIB=tools/_probe_lib.so timeout 120 python tools/bench_attn_decode.py $C4
run PLAN=1 PROBE=1 STEPS=100 NH=12 NKV=2 TR1_HIP_LIB=tools/_probe_lib.so timeout 120 python tools/bench_attn_decode.py $C2
cat $O
