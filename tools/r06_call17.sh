#!/bin/bash
# round 6, call 17: attn_fwd64_kernel<6> (the vision towers' live-96 launch) + attn_bwd_dq64 with separate K / V rings (two tiles of DMA lead): tests, timing
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "attention" 2>&1 | tail -4 > $O/r06_c17_tests.txt
for f in 1 0 1 0; do TR1_FWD64=$f timeout 300 python tools/bench_attn_vit.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read())['vit_attention']; print('TR1_FWD64=$f', {k: v['us'] for k, v in d.items() if isinstance(v, dict)}, d.get('max_abs_diff_vs_all_128'))" >> $O/r06_c17_vit.txt; done
for i in 1 2; do
  for lib in default lead1; do
    if [ $lib = default ]; then E=""; else E="TR1_HIP_LIB=tools/_var_lead1.so"; fi
    env $E timeout 300 python tools/check_dq64.py 2>/dev/null | head -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib: isolated bwd ms: dq32', d['bwd_ms_dq32'], 'dq64', d['bwd_ms_dq64'], 'bit-equal', d['dQ_bit_equal'], d['dK_bit_equal'])" >> $O/r06_c17_dq64_lead.txt
  done
done
timeout 300 python tools/check_dq64.py 2>/dev/null | tail -1 >> $O/r06_c17_dq64_lead.txt
cat $O/r06_c17_tests.txt $O/r06_c17_vit.txt $O/r06_c17_dq64_lead.txt
