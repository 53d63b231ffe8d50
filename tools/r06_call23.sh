#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 900 python tools/check_gemm4w.py > $O/r06_c23_gemm4w.txt 2> $O/r06_c23.err; tail -3 $O/r06_c23.err; cat $O/r06_c23_gemm4w.txt
