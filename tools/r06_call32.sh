#!/bin/bash
# round 6, call 32: wave_sum / wave_max and the merge kernel's closing sums on v_permlane16/32_swap + DPP row rotations (tr1_common.h): output hashes against the previous
# library, whole -m gpu suite, decode step time (no profiler) under both libraries
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 300 python tools/hash_attn.py > $O/r06_c32_hash_new.txt 2>$O/r06_c32_hash.err
TR1_HIP_LIB=tools/_var_ref.so timeout 300 python tools/hash_attn.py > $O/r06_c32_hash_ref.txt 2>/dev/null
if diff -q $O/r06_c32_hash_new.txt $O/r06_c32_hash_ref.txt > /dev/null; then echo "HASHES IDENTICAL ($(wc -l < $O/r06_c32_hash_new.txt) lines)" > $O/r06_c32.txt; else echo "HASHES DIFFER" > $O/r06_c32.txt; diff $O/r06_c32_hash_new.txt $O/r06_c32_hash_ref.txt >> $O/r06_c32.txt; fi
for lib in new ref new ref; do
  if [ $lib = ref ]; then E="TR1_HIP_LIB=tools/_var_ref.so"; else E="X=1"; fi
  env $E timeout 300 python tools/decode_steps_probe.py 2>/dev/null | tail -1 | sed "s/^/$lib: /" >> $O/r06_c32.txt
done
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -2 >> $O/r06_c32.txt
cat $O/r06_c32.txt; tail -2 $O/r06_c32_hash.err
