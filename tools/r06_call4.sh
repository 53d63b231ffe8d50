#!/bin/bash
# round 6, call 4: attn_fwd64_kernel (64 rows per wave, one wave per SIMD): bit identity against the 32-row kernel + timing
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 600 python tools/check_fwd64.py > $O/r06_c4_fwd64.txt 2> $O/r06_c4_fwd64.err
tail -5 $O/r06_c4_fwd64.err; cat $O/r06_c4_fwd64.txt
