#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 300 python tools/check_fwd64.py 2>/dev/null | grep -v "true, \"nan\": false}$" > $O/r06_c9.txt
TR1_HIP_LIB=tools/_probe_lib.so timeout 300 python tools/check_fwd64.py --probe 2>/dev/null | sed -n 20,23p >> $O/r06_c9.txt
cat $O/r06_c9.txt
