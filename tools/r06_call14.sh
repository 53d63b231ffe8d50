#!/bin/bash
# round 6, call 14: per-kernel times (rocprofv3 --kernel-trace --stats, csv) of the attention kernels, 32-row forms (before) against 64-row forms (after)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
for tag in before after before after; do
  if [ $tag = before ]; then E="TR1_FWD64=0 TR1_DQ64=0"; else E="TR1_FWD64=1 TR1_DQ64=1"; fi
  rm -rf $O/ks_$tag
  env $E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$tag -- python tools/bench_attn.py --no-check --iters 20 > $O/r06_c14_bench_$tag.json 2> $O/r06_c14_$tag.err
  f=$(find $O/ks_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag $f" >> $O/r06_c14_kernel_stats.txt
  if [ -n "$f" ]; then grep -i "attn\|Name" "$f" >> $O/r06_c14_kernel_stats.txt; else find $O/ks_$tag | head >> $O/r06_c14_kernel_stats.txt; fi
  rm -rf $O/ks_$tag
done
cat $O/r06_c14_kernel_stats.txt
