#!/bin/bash
# per-kernel times of tools/bench_attn.py under rocprofv3 (kernel trace):  tools/stats_attn.sh <tag> [ENV=VAL ...] -> gpurun_out/stats_attn_<tag>.md
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/stats_attn_$tag
env "$@" timeout 300 rocprofv3 --kernel-trace -d gpurun_out/stats_attn_$tag -- python tools/bench_attn.py --no-check --iters 10 > /dev/null 2> gpurun_out/stats_attn_$tag.err </dev/null
db=$(ls gpurun_out/stats_attn_$tag/*/*_results.db 2>/dev/null | head -1)
if [ -n "$db" ]; then timeout 60 python tools/rocpd_stats.py "$db" gpurun_out/stats_attn_$tag.md > /dev/null 2>&1; grep -E "attn|reduce" gpurun_out/stats_attn_$tag.md | cut -c1-150; rm -rf gpurun_out/stats_attn_$tag; else echo "no db"; tail -3 gpurun_out/stats_attn_$tag.err; fi
