#!/bin/bash
# Round-6 evidence on one box: PMC traffic pass -> driver-args bench (traffic attached) -> kernel trace -> MFMA / LDS PMC pass.
# usage (inside gpurun): bash tools/r06_evidence.sh [tag]      outputs under gpurun_out/ev_<tag>_*
tag=${1:-a}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
SHORT="--steps 2 --warmup 0 --C 12 --no-cpu-baseline --no-roofline --no-peak-probe --no-engine-leg"
# 1. HBM read traffic per launch (own pass)
rm -rf $O/ev_${tag}_fetch
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/ev_${tag}_fetch -- python bench.py $SHORT > /dev/null 2> $O/ev_${tag}_fetch.err </dev/null
db=$(ls $O/ev_${tag}_fetch/*/*_results.db 2>/dev/null | head -1)
if [ -n "$db" ]; then
  timeout 120 python tools/pmc_to_json.py "$db" $O/ev_${tag}_pmc_traffic.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py $SHORT (round 6)" > $O/ev_${tag}_pmc_traffic.txt 2>&1 </dev/null
  timeout 120 python tools/rocpd_pmc.py "$db" $O/ev_${tag}_pmc_fetch_size.md > /dev/null 2>&1 </dev/null
  cp $O/ev_${tag}_pmc_traffic.json profiles/r06_pmc_traffic.json
  rm -rf $O/ev_${tag}_fetch
else echo "no fetch db"; tail -3 $O/ev_${tag}_fetch.err; fi
# 2. the driver's command
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/ev_${tag}_bench.json 2> $O/ev_${tag}_bench.err </dev/null
python - <<PY
import json
d = json.loads(open("$O/ev_${tag}_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("bench", d["value"], d["ms_per_step"], d.get("rollout_tokens_per_sec"), d.get("phases_ms_per_step"), "roofline", round(r["achieved"]), round(r["frac"], 3), r.get("traffic"), r.get("traffic_guard"))
PY
# 3. kernel trace of the same workload (even step count: every window batches two prompts)
bash tools/trace_bench.sh ev_${tag} --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-peak-probe --no-engine-leg > $O/ev_${tag}_trace.txt 2>&1 </dev/null
# 4. MFMA busy / wave stalls, then LDS counters (own passes)
for pass in mfma lds; do
  if [ $pass = mfma ]; then C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"; else C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS"; fi
  rm -rf $O/ev_${tag}_$pass
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d $O/ev_${tag}_$pass -- python bench.py $SHORT > /dev/null 2> $O/ev_${tag}_$pass.err </dev/null
  db=$(ls $O/ev_${tag}_$pass/*/*_results.db 2>/dev/null | head -1)
  if [ -n "$db" ]; then
    if [ $pass = mfma ]; then timeout 120 python tools/pmc_mfma.py "$db" $O/ev_${tag}_pmc_mfma.md > /dev/null 2>&1 </dev/null; fi
    timeout 120 python tools/pmc_dump.py "$db" > $O/ev_${tag}_pmc_${pass}_raw.txt 2>&1 </dev/null
    rm -rf $O/ev_${tag}_$pass
  else echo "no $pass db"; tail -3 $O/ev_${tag}_$pass.err; fi
done
ls -la $O | grep ev_${tag} | awk '{print $5, $9}'
