#!/bin/bash
# A/B helper: build the C-ABI library of another commit (default HEAD~1) into tools/_var_ref.so (load with TR1_HIP_LIB=tools/_var_ref.so; same header required)
set -e
REV=${1:-HEAD~1}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/tr1_ref_tree; rm -rf $W; mkdir -p $W/obj
git -C "$ROOT" archive "$REV" time-r1_amd/csrc include | tar -x -C $W
cd $W/time-r1_amd/csrc
ls *.hip | xargs -P 8 -I{} sh -c '/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -c {} -o '$W'/obj/{}.o 2>/dev/null'
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/_var_ref.so" $W/obj/*.o
echo built tools/_var_ref.so from $REV
