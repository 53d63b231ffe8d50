#!/bin/bash
# Is the MFMA phases' time sensitive to where the buffers land?  Shift every address by an early ballast of various sizes.  usage (inside gpurun): bash tools/ab_layout.sh
mkdir -p gpurun_out
B="bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-peak-probe --no-engine-leg"
show() { python -c "
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[2], round(d['ms_per_step'],1), d['phases_ms_per_step'])" $1 "$2"; }
for gb in 0 0.002 0.033 0.26 1.001 3.3; do
python $B --ballast-gb $gb --ballast-early > gpurun_out/lay.json 2>/dev/null; show gpurun_out/lay.json early_$gb
done
