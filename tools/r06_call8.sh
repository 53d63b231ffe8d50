#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
for v in lib abl128; do
  echo "== $v" >> $O/r06_c8_probe.txt
  TR1_HIP_LIB=tools/_probe_$v.so timeout 300 python tools/check_fwd64.py --probe 2>/dev/null | sed -n 1,2p >> $O/r06_c8_probe.txt
  TR1_HIP_LIB=tools/_probe_$v.so timeout 300 python tools/check_fwd64.py --probe 2>/dev/null | sed -n 20,23p >> $O/r06_c8_probe.txt
done
cat $O/r06_c8_probe.txt
timeout 300 python tools/check_fwd64.py 2>/dev/null | grep -v "true, \"nan\": false}$" 
