#!/bin/bash
# round 6, call 12: attn_bwd_dq64_kernel: bit identity against the 32-row dQ kernel + backward timing
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 600 python tools/check_dq64.py > $O/r06_c12_dq64.txt 2> $O/r06_c12_dq64.err
tail -5 $O/r06_c12_dq64.err; cat $O/r06_c12_dq64.txt
