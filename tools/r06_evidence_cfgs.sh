#!/bin/bash
# Round-6 evidence for BASELINE configs 2, 4, 5 on one box: bench JSON + kernel table each.  outputs under gpurun_out/evc_*
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
X="--no-cpu-baseline"
T="--no-cpu-baseline --no-roofline --no-peak-probe --no-engine-leg"
# config 5 first: HBM read traffic per launch of the fp8 decode kernels (own pass), so that the bench line below can attach roofline.traffic
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf $O/evc_fetch8
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/evc_fetch8 -- python bench.py --rollout-fp8 --steps 2 --warmup 0 --C 12 $T > /dev/null 2> $O/evc_fetch8.err </dev/null
db=$(ls $O/evc_fetch8/*/*_results.db 2>/dev/null | head -1)
if [ -n "$db" ]; then
  timeout 120 python tools/pmc_to_json.py "$db" $O/evc_pmc_traffic_fp8.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --rollout-fp8 --steps 2 --warmup 0 --C 12 $T (round 6)" > $O/evc_pmc_traffic_fp8.txt 2>&1 </dev/null
  cp $O/evc_pmc_traffic_fp8.json profiles/r06_pmc_traffic_fp8.json
  rm -rf $O/evc_fetch8
else echo "no fp8 fetch db"; tail -3 $O/evc_fetch8.err; fi
timeout 600 python bench.py --rollout-fp8 --steps 8 --warmup 2 $X > $O/evc_cfg5.json 2> $O/evc_cfg5.err </dev/null
bash tools/trace_bench.sh evc_cfg5 --rollout-fp8 --steps 4 --warmup 2 $T > /dev/null 2>&1 </dev/null
timeout 600 python bench.py --model qwen2-vl-2b --frames 16 --steps 8 --warmup 2 $X > $O/evc_cfg2.json 2> $O/evc_cfg2.err </dev/null
bash tools/trace_bench.sh evc_cfg2 --model qwen2-vl-2b --frames 16 --steps 4 --warmup 2 $T > /dev/null 2>&1 </dev/null
timeout 900 python bench.py --model qwen2.5-vl-7b --frames 64 --G 16 --C 1024 --beta 0 --clip-loss --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-peak-probe --no-engine-leg > $O/evc_cfg4.json 2> $O/evc_cfg4.err </dev/null
bash tools/trace_bench.sh evc_cfg4 --model qwen2.5-vl-7b --frames 64 --G 16 --C 1024 --beta 0 --clip-loss --steps 2 --warmup 0 $T > /dev/null 2>&1 </dev/null
python - <<PY
import json
for c in ("cfg5", "cfg2", "cfg4"):
    try:
        d = json.loads(open("$O/evc_%s.json" % c).read().strip().splitlines()[-1])
        print(c, round(d["value"], 4), round(d["ms_per_step"], 1), d.get("rollout_tokens_per_sec"), d.get("phases_ms_per_step"), d.get("hbm_gb"))
    except Exception as e:
        print(c, "failed", e)
PY
