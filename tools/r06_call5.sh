#!/bin/bash
# round 6, call 5: wave timeline of attn_fwd64_kernel's tile bodies
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
TR1_HIP_LIB=tools/_probe_lib.so timeout 600 python tools/check_fwd64.py --probe > $O/r06_c5_probe.txt 2> $O/r06_c5_probe.err
tail -3 $O/r06_c5_probe.err; head -70 $O/r06_c5_probe.txt
