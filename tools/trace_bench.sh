#!/bin/bash
# rocprofv3 kernel trace of a bench.py run -> per-kernel table + gap table.  usage: tools/trace_bench.sh <tag> <bench args...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/trace_$tag
timeout 900 rocprofv3 --kernel-trace -d gpurun_out/trace_$tag -- python bench.py "$@" > gpurun_out/trace_$tag.json 2> gpurun_out/trace_$tag.err </dev/null
db=$(ls gpurun_out/trace_$tag/*/*_results.db 2>/dev/null | head -1)
if [ -n "$db" ]; then
  timeout 120 python tools/rocpd_stats.py "$db" gpurun_out/trace_${tag}_kernel_stats.md > /dev/null 2>&1
  timeout 120 python tools/rocpd_gaps.py "$db" > gpurun_out/trace_${tag}_gaps.md 2>&1
  timeout 120 python tools/rocpd_streams.py "$db" > gpurun_out/trace_${tag}_streams.md 2>&1
  head -40 gpurun_out/trace_${tag}_kernel_stats.md | cut -c1-160
  rm -rf gpurun_out/trace_$tag
else echo "no db"; tail -5 gpurun_out/trace_$tag.err; fi
