#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 600 python tools/sweep_fwd64.py > $O/r06_c18_sweep_fwd64.txt 2>/dev/null; cat $O/r06_c18_sweep_fwd64.txt
