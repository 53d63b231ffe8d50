#!/bin/bash
# round 6, call 13: per-kernel times (rocprofv3 --kernel-trace --stats) and PMC passes of the attention kernels, 32-row forms (before) against 64-row forms (after)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
for tag in before after; do
  if [ $tag = before ]; then E="TR1_FWD64=0 TR1_DQ64=0"; else E="TR1_FWD64=1 TR1_DQ64=1"; fi
  rm -rf $O/ks_$tag
  env $E timeout 300 rocprofv3 --kernel-trace --stats -d $O/ks_$tag -- python tools/bench_attn.py --no-check --iters 20 > $O/r06_c13_bench_$tag.json 2> $O/r06_c13_$tag.err
  f=$(ls $O/ks_$tag/*/*kernel_stats.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then grep -i "attn\|Name" "$f" > $O/r06_c13_kernel_stats_$tag.csv; fi
  rm -rf $O/ks_$tag
  bash tools/pmc_attn.sh r06$tag $E > /dev/null 2>&1
done
python tools/pmc_attn_md.py r06before r06after > $O/r06_pmc_mfma_attn.md 2>&1
cat $O/r06_c13_kernel_stats_before.csv $O/r06_c13_kernel_stats_after.csv; cat $O/r06_pmc_mfma_attn.md
