#!/bin/bash
# round 6, call 30: vector-only interval mask + v_permlane32_swap exchanges in attn_dec32_kernel (and the exchanges in attn_fwd32_kernel): parity tests, output hashes, decode attention per layer, vision-tower attention, against the previous library
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_edge_cases_gpu.py tests/test_fullsize_gpu.py tests/test_configs_gpu.py tests/test_engine_gpu.py -m gpu -x -q -k "decode or attn or attention or rollout or generate" 2>&1 | tail -2 > $O/r06_c30.txt
for lib in new ref new ref; do
  if [ $lib = ref ]; then E="TR1_HIP_LIB=tools/_var_ref.so"; else E="X=1"; fi
  echo "== $lib" >> $O/r06_c30.txt
  env $E PLAN=1 timeout 300 python tools/bench_attn_decode.py 2>/dev/null | tail -4 >> $O/r06_c30.txt
  env $E PLAN=1 STEPS=100 timeout 300 python tools/bench_attn_decode.py 3474 8 200 2 2>/dev/null | tail -1 >> $O/r06_c30.txt
done
cat $O/r06_c30.txt
timeout 300 python tools/hash_attn.py > $O/r06_c30_hash_new.txt 2>/dev/null
TR1_HIP_LIB=tools/_var_ref.so timeout 300 python tools/hash_attn.py > $O/r06_c30_hash_ref.txt 2>/dev/null
if diff -q $O/r06_c30_hash_new.txt $O/r06_c30_hash_ref.txt > /dev/null; then echo "HASHES IDENTICAL ($(wc -l < $O/r06_c30_hash_new.txt) lines)"; else echo "HASHES DIFFER"; fi
for i in 1 2; do timeout 300 python tools/bench_attn_vit.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read())['vit_attention']; print({k: v['us'] for k, v in d.items() if isinstance(v, dict)})"; done
