#!/bin/bash
# A/B of library variants on tools/bench_attn.py: tools/ab_attn.sh <lib-or-"-"> ...   ("-" = the product library); prints fwd / bwd ms per variant, twice
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = "-" ]; then r=$(timeout 200 python tools/bench_attn.py --no-check --iters 30 2>/dev/null </dev/null); else r=$(TR1_HIP_LIB=$v timeout 200 python tools/bench_attn.py --no-check --iters 30 2>/dev/null </dev/null); fi
  echo "$v $(echo $r | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["fwd_ms"], d["bwd_ms"])')"
done; done
