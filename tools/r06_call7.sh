#!/bin/bash
# round 6, call 7: attn_fwd64 after the permlane exchange / vector-only mask / V fragments read ahead of the barrier: bit identity, timing, timeline
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 600 python tools/check_fwd64.py > $O/r06_c7_fwd64.txt 2> $O/r06_c7_fwd64.err
TR1_HIP_LIB=tools/_probe_lib.so timeout 300 python tools/check_fwd64.py --probe > $O/r06_c7_probe.txt 2>/dev/null
tail -3 $O/r06_c7_fwd64.err; cat $O/r06_c7_fwd64.txt; sed -n 1,3p $O/r06_c7_probe.txt; sed -n 20,24p $O/r06_c7_probe.txt; sed -n 52,62p $O/r06_c7_probe.txt
