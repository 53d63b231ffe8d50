"""Build an A/B variant of the C-ABI library with extra compiler flags: `python tools/build_variant.py <name> [-DFOO=1 ...]` -> tools/_var_<name>.so
(load with TR1_HIP_LIB=tools/_var_<name>.so).  Only csrc files that mention one of the -D macro names are recompiled; the rest reuse the product objects."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "time-r1_amd", "csrc")
name, flags = sys.argv[1], sys.argv[2:]
OUT = os.path.join(ROOT, "tools", "_var_%s.so" % name)
OBJ = "/tmp/tr1_var_obj_%s" % name
BASE = os.path.join(ROOT, "time-r1_amd", "build")
os.makedirs(OBJ, exist_ok=True)
macros = [f[2:].split("=")[0] for f in flags if f.startswith("-D")]
all_files = any(not f.startswith("-D") for f in flags)      # a code-generation flag (-mllvm ...) concerns every file
def comp(f):
    src = os.path.join(CSRC, f)
    txt = open(src).read()
    base_o = os.path.join(BASE, f[:-4] + ".o")
    if not all_files and not any(m in txt for m in macros) and os.path.exists(base_o) and os.path.getmtime(base_o) >= os.path.getmtime(src):
        return base_o
    o = os.path.join(OBJ, f[:-4] + ".o")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value"] + flags +
                       ["-c", src, "-o", o], capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr)
    return o
with ThreadPoolExecutor(8) as ex:
    objs = list(ex.map(comp, sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))))
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs, capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr)
print("built", OUT)
