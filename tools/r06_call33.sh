#!/bin/bash
# round 6, call 33: the permlane / DPP wave reductions inside the step: current library against the previous commit's (tools/_var_ref.so), three alternations
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
bash tools/ab_env_bench.sh TR1_HIP_LIB time-r1_amd/lib/libtimer1_hip.so tools/_var_ref.so time-r1_amd/lib/libtimer1_hip.so tools/_var_ref.so time-r1_amd/lib/libtimer1_hip.so tools/_var_ref.so > $O/r06_c33_ab_wave_reductions.txt 2>&1
cat $O/r06_c33_ab_wave_reductions.txt
