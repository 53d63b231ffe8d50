#!/bin/bash
# tools/ab_env_bench.sh with the allocated-peak HBM of every run.  usage: tools/ab_env_bench_mem.sh VAR val1 val2 ... [-- bench args]
var=$1; shift
vals=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "$1" = "--" ] && shift
for v in "${vals[@]}"; do
  out=$(env $var="$v" timeout 600 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline --no-peak-probe --no-engine-leg "$@" 2>/dev/null </dev/null | tail -1)
  echo "$var=$v $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_step %.1f" % d["ms_per_step"], {k: round(v,1) for k,v in d.get("phases_ms_per_step",{}).items()}, d.get("hbm_gb"))' 2>/dev/null)"
done
