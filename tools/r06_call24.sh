#!/bin/bash
# PMC pass: the 8192^3 NT GEMM on the default (8-wave phased) form and on the four-wave form
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=gpurun_out
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"
for f in 0 1; do
  rm -rf $O/pmc_g4_$f
  TR1_GEMM4W=$f timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/pmc_g4_$f -- python tools/bench_gemm_one.py 8192 8192 8192 5 > /dev/null 2> $O/pmc_g4_$f.err
  db=$(ls $O/pmc_g4_$f/*/*_results.db 2>/dev/null | head -1)
  echo "== TR1_GEMM4W=$f" >> $O/r06_c24_pmc_gemm.txt
  if [ -n "$db" ]; then timeout 60 python tools/pmc_dump.py "$db" gemm_nt >> $O/r06_c24_pmc_gemm.txt 2>&1; rm -rf $O/pmc_g4_$f; else tail -3 $O/pmc_g4_$f.err >> $O/r06_c24_pmc_gemm.txt; fi
done
cat $O/r06_c24_pmc_gemm.txt
