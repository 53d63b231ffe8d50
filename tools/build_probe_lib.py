"""Build tools/_probe_lib.so: the same C-ABI library compiled with -DTR1_PROBE (in-kernel s_memtime stamps; see BWD_STAMP in csrc/attn_bwd.hip).
Load it with TR1_HIP_LIB=tools/_probe_lib.so (e.g. `python tools/bench_attn.py --probe`).  Not part of the product build."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "time-r1_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_probe_lib.so")
OBJ = "/tmp/tr1_probe_obj"
os.makedirs(OBJ, exist_ok=True)
srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
def comp(f):
    o = os.path.join(OBJ, f[:-4] + ".o")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value", "-DTR1_PROBE"] + os.environ.get("TR1_PROBE_EXTRA", "").split() + [
                        "-c", os.path.join(CSRC, f), "-o", o], capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr)
    return o
with ThreadPoolExecutor(8) as ex:
    objs = list(ex.map(comp, srcs))
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs, capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr)
print("built", OUT)
