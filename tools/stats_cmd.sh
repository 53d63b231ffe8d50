#!/bin/bash
# per-kernel times of an arbitrary command under rocprofv3: tools/stats_cmd.sh <tag> <grep-pattern> ENV=.. -- cmd...
tag=$1; pat=$2; shift 2
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/stats_$tag
env "${envs[@]}" timeout 300 rocprofv3 --kernel-trace -d gpurun_out/stats_$tag -- "$@" > gpurun_out/stats_$tag.out 2> gpurun_out/stats_$tag.err </dev/null
db=$(ls gpurun_out/stats_$tag/*/*_results.db 2>/dev/null | head -1)
if [ -n "$db" ]; then timeout 60 python tools/rocpd_stats.py "$db" gpurun_out/stats_$tag.md > /dev/null 2>&1; grep -E "$pat" gpurun_out/stats_$tag.md | cut -c1-150; rm -rf gpurun_out/stats_$tag; else echo "no db"; tail -3 gpurun_out/stats_$tag.err; fi
