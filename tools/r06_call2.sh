#!/bin/bash
# round 6, call 2: 2B decode with the fragment-major down projection (csrc/oproj.hip long-K form) against the split-K + fixup form, A/B in one process layout
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "fragment_major or down_projection" 2>&1 | tail -5 > $O/r06_c2_tests.txt
for rep in 1 2; do
  for v in 1 0; do
    TR1_DOWN_FRAG=$v MODEL=2b timeout 300 python tools/decode_steps_probe.py 2>&1 | grep "rep 1" | sed "s/^/DOWN_FRAG=$v /" >> $O/r06_c2_probe.txt
  done
done
timeout 600 python -m pytest tests/test_configs_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -4 >> $O/r06_c2_tests.txt
cat $O/r06_c2_tests.txt $O/r06_c2_probe.txt
