#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 600 python tools/sweep_dq64.py > $O/r06_c19_sweep_dq64.txt 2>$O/r06_c19.err; cat $O/r06_c19_sweep_dq64.txt; tail -3 $O/r06_c19.err
