#!/bin/bash
# A/B of environment switches on tools/bench_attn.py: tools/ab_env_attn.sh "ENV=.. ENV2=.." "..." ; prints fwd / bwd ms (+ errors with --check as first arg)
cd "$GRAFT_REPO_ROOT"
chk="--no-check"; if [ "$1" = "--check" ]; then chk=""; shift; fi
for rep in 1 2; do for v in "$@"; do
  r=$(env $v timeout 250 python tools/bench_attn.py $chk --iters 30 2>/dev/null </dev/null)
  echo "[$v] $(echo $r | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["fwd_ms"], d["bwd_ms"], d.get("rel_l2",""))')"
done; done
