#!/bin/bash
# rocprofv3 kernel trace as CSV of a bench.py run.  usage: tools/trace_csv.sh <tag> <bench args...>  -> gpurun_out/csv_<tag>_kernel_trace.csv
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/csv_$tag
timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/csv_$tag -- python bench.py "$@" > gpurun_out/csv_$tag.json 2> gpurun_out/csv_$tag.err </dev/null
f=$(ls gpurun_out/csv_$tag/*/*kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then mv "$f" gpurun_out/csv_${tag}_kernel_trace.csv; rm -rf gpurun_out/csv_$tag; else echo "no csv"; tail -5 gpurun_out/csv_$tag.err; fi
