#!/bin/bash
# PMC passes over tools/bench_attn.py (attention fwd / bwd at the config-3 shape): MFMA busy / wave stalls, then LDS conflicts.
# usage: tools/pmc_attn.sh <tag> [ENV=VAL ...]   -> gpurun_out/pmc_attn_<tag>_{a,b}.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for pass in a b; do
  if [ $pass = a ]; then C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"; else C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; fi
  rm -rf gpurun_out/pmc_attn_${tag}_$pass
  env "$@" timeout 300 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/pmc_attn_${tag}_$pass -- python tools/bench_attn.py --no-check --iters 3 > /dev/null 2> gpurun_out/pmc_attn_${tag}_$pass.err
  db=$(ls gpurun_out/pmc_attn_${tag}_$pass/*/*_results.db 2>/dev/null | head -1)
  if [ -n "$db" ]; then timeout 60 python tools/pmc_dump.py "$db" attn > gpurun_out/pmc_attn_${tag}_$pass.txt 2>&1; cat gpurun_out/pmc_attn_${tag}_$pass.txt; rm -rf gpurun_out/pmc_attn_${tag}_$pass; else echo "no db for $tag $pass"; tail -5 gpurun_out/pmc_attn_${tag}_$pass.err; fi
done
