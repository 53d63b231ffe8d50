"""Fused rmsnorm + gate/up + SwiGLU decode GEMM at 17..32 rows (config 4 decodes 2 prompts x 16 rollouts = 32 rows per step), rotating over
weight copies so every launch streams its 271 MB from HBM.  Checks the result against the unfused chain first.
   python tools/bench_glu32.py [M]          (TR1_GLU32_CFG selects the kernel form; run once per value)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time_r1_amd  # noqa
from time_r1_amd.ops import HipOps
ops = HipOps("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K, I, NC = 3584, 18944, 6
torch.manual_seed(0)
x = torch.randn(M, K, device="cuda").bfloat16()
lnw = (1 + 0.1 * torch.randn(K, device="cuda")).bfloat16()
ws = [(torch.randn(2 * I, K, device="cuda") * 0.02).bfloat16() for _ in range(NC)]
ref = ops.swiglu_fwd(ops.gemm_nt(ops.rmsnorm_fwd(x, lnw, 1e-6)[0], ws[0]))
got = ops.norm_gemm(x, lnw, 1e-6, ws[0], glu=True)
print("cfg %s  max |fused - unfused| = %.4g (ref max %.3g)" % (os.environ.get("TR1_GLU32_CFG", "0"), float((got.float() - ref.float()).abs().max()), float(ref.float().abs().max())))
i = [0]
def run():
    i[0] = (i[0] + 1) % NC
    return ops.norm_gemm(x, lnw, 1e-6, ws[i[0]], glu=True)
for _ in range(10): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 5
print("M=%d: %6.1f us  -> %5.0f GB/s of weights" % (M, us, 2 * I * K * 2 / us / 1e3))
if os.environ.get("PROBE"):      # TR1_HIP_LIB=tools/_probe_lib.so PROBE=1: block timeline of ONE launch (within-block differences; XCD clocks are not aligned)
    import ctypes
    import numpy as np
    from time_r1_amd import hip
    buf = torch.zeros(256 * 2 * 8, dtype=torch.int64, device="cuda")
    assert hip.lib().cdll.probe_glu_set_ptr(ctypes.c_void_p(buf.data_ptr())) == 0
    run(); torch.cuda.synchronize()
    hip.lib().cdll.probe_glu_set_ptr(ctypes.c_void_p(0))
    st = buf.cpu().view(256, 2, 8).numpy().astype(np.float64)
    names = ("prologue (pointers + first DMA)", "x' fragments built", "barrier (sum x^2)", "first column pair", "remaining pairs")
    for w, wn in ((0, "wave 0"), (1, "wave 7")):
        d = st[:, w, 1:6] - st[:, w, 0:5]
        print(" %s, median cycles per phase: " % wn + "  ".join("%s %.0f" % (n, np.median(d[:, i])) for i, n in enumerate(names)), " total %.0f" % np.median(st[:, w, 5] - st[:, w, 0]))
