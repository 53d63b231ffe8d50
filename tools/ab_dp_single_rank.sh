#!/bin/bash
# What does the data-parallel path cost per rank, measured on ONE GPU?  TR1_DIST_FORCE=1 builds a single-rank process group on "nccl" (RCCL) and runs every
# collective of the path.  usage (inside gpurun): bash tools/ab_dp_single_rank.sh
mkdir -p gpurun_out
B="bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-peak-probe --no-engine-leg"
show() { python -c "
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[2], round(d['ms_per_step'],1), d['phases_ms_per_step'], d['distributed']['hbm_gb_allocated_peak_per_rank'], d['distributed'].get('grad_exchange_exposed_ms_per_optimizer_step'))" $1 "$2"; }
python $B > gpurun_out/ab1.json 2>/dev/null; show gpurun_out/ab1.json no_group
TR1_DIST_FORCE=1 python $B --replicated-optimizer > gpurun_out/ab2.json 2>/dev/null; show gpurun_out/ab2.json group_replicated
TR1_DIST_FORCE=1 python $B > gpurun_out/ab3.json 2>/dev/null; show gpurun_out/ab3.json group_sharded
TR1_DIST_FORCE=1 TR1_MAIN_PRIO=1 python $B --replicated-optimizer > gpurun_out/ab4.json 2>/dev/null; show gpurun_out/ab4.json group_replicated_priority_stream
