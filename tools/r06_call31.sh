#!/bin/bash
# round 6, call 31: vector-only mask selects in attn_bwd_dkdv32_kernel: output hashes against the previous library, backward tests, timing (two alternations)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 300 python tools/hash_attn.py > $O/r06_c31_hash_new.txt 2>/dev/null
TR1_HIP_LIB=tools/_var_ref.so timeout 300 python tools/hash_attn.py > $O/r06_c31_hash_ref.txt 2>/dev/null
if diff -q $O/r06_c31_hash_new.txt $O/r06_c31_hash_ref.txt > /dev/null; then echo "HASHES IDENTICAL ($(wc -l < $O/r06_c31_hash_new.txt) lines)" > $O/r06_c31.txt; else echo "HASHES DIFFER" > $O/r06_c31.txt; diff $O/r06_c31_hash_new.txt $O/r06_c31_hash_ref.txt >> $O/r06_c31.txt; fi
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_edge_cases_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "attention or attn" 2>&1 | tail -2 >> $O/r06_c31.txt
for lib in new ref new ref; do
  if [ $lib = ref ]; then E="TR1_HIP_LIB=tools/_var_ref.so"; else E="X=1"; fi
  env $E timeout 300 python tools/bench_attn.py --no-check --iters 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib fwd', d['fwd_ms'], 'bwd', d['bwd_ms'])" >> $O/r06_c31.txt
done
cat $O/r06_c31.txt
