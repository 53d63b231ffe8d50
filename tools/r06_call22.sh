#!/bin/bash
# round 6, call 22: whole -m gpu suite + smoke at the round's last kernel sources
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/r06_c22_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_c22_smoke.txt 2>&1
cat $O/r06_c22_tests.txt; tail -3 $O/r06_c22_smoke.txt
