#!/bin/bash
# round 6, call 6: attn_fwd64 timing ablations (wrong results by design): per-phase cycles of the tile body
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
for v in lib abl1 abl2 abl36; do
  echo "== $v" >> $O/r06_c6_probe.txt
  TR1_HIP_LIB=tools/_probe_$v.so timeout 300 python tools/check_fwd64.py --probe 2>/dev/null | sed -n 10,14p >> $O/r06_c6_probe.txt
done
cat $O/r06_c6_probe.txt
