#!/bin/bash
# round 6, call 11: A/B of the head-dim-128 forward kernel inside the step (same box, two alternations)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
bash tools/ab_env_bench.sh TR1_FWD64 1 0 1 0 > $O/r06_c11_ab_fwd64.txt 2>&1
cat $O/r06_c11_ab_fwd64.txt
