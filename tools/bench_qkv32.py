"""Fused rmsnorm + QKV projection + M-RoPE + cache append, and the o projection, at M decode rows (config 4: 32), rotating weight copies.
   python tools/bench_qkv32.py [M]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time_r1_amd  # noqa
from time_r1_amd.ops import HipOps
ops = HipOps("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K, nh, nkv, hd, NC = 3584, 28, 4, 128, 12
torch.manual_seed(0)
x = torch.randn(M, K, device="cuda").bfloat16()
lnw = (1 + 0.1 * torch.randn(K, device="cuda")).bfloat16()
ws = [(torch.randn((nh + 2 * nkv) * hd, K, device="cuda") * 0.02).bfloat16() for _ in range(NC)]
wo = [(torch.randn(K, K, device="cuda") * 0.02).bfloat16() for _ in range(NC)]
b = torch.randn((nh + 2 * nkv) * hd, device="cuda").bfloat16()
cos, sin = torch.randn(M, hd // 2, device="cuda"), torch.randn(M, hd // 2, device="cuda")
S = 4096
kc, vt = torch.zeros(S, nkv * hd, device="cuda", dtype=torch.bfloat16), torch.zeros(nkv * hd, S, device="cuda", dtype=torch.bfloat16)
slots = torch.arange(M, device="cuda", dtype=torch.int32) * 64
res = torch.randn(M, K, device="cuda").bfloat16()
def timeit(fn, reps=300):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
i = [0]
def qkv():
    i[0] = (i[0] + 1) % NC
    return ops.norm_gemm_qkv(x, lnw, 1e-6, ws[i[0]], b, cos, sin, kc, vt, slots, nh, nkv, hd)
def oproj():
    i[0] = (i[0] + 1) % NC
    return ops.gemm_nt(x, wo[i[0]], residual=res)
xfrag = torch.randn((M + 15) // 16 * 16 * K, device="cuda").bfloat16()
def oproj_frag():      # round 5: all weight stages in flight, fragment-major activation, no fixup (csrc/oproj.hip)
    i[0] = (i[0] + 1) % NC
    return ops.gemm_oproj_frag(xfrag, wo[i[0]], M, residual=res)
print("M=%d: fused qkv %6.2f us   o projection %6.2f us   o projection (fragment-major x, csrc/oproj.hip) %6.2f us" %
      (M, timeit(qkv), timeit(oproj), timeit(oproj_frag)))

if os.environ.get("PROBE"):      # TR1_HIP_LIB=tools/_probe_lib.so PROBE=1: block timeline of ONE fused QKV launch
    import ctypes
    from time_r1_amd import hip
    buf = torch.zeros(160 * 2 * 8, dtype=torch.int64, device="cuda")
    assert hip.lib().cdll.probe_qkv_set_ptr(ctypes.c_void_p(buf.data_ptr())) == 0
    qkv(); torch.cuda.synchronize()
    hip.lib().cdll.probe_qkv_set_ptr(ctypes.c_void_p(0))
    st = buf.cpu().view(160, 2, 8).numpy()
    nb = int((st[:, 0, 0] > 0).sum())
    t0 = int(st[:nb, :, 0].min())
    import numpy as np
    rel = (st[:nb, :, :5] - t0).astype(np.float64)
    names = ("entry", "first loads issued", "stream consumed", "partials reduced", "epilogue stored")
    print("blocks", nb, "(s_memtime ticks; 100 ticks = 1 us at the 100 MHz reference clock)")
    for w in (0, 1):
        print(" wave %s:" % ("0" if w == 0 else "last"), "  ".join("%s min %.0f med %.0f max %.0f" % (names[i], rel[:, w, i].min(), np.median(rel[:, w, i]), rel[:, w, i].max()) for i in range(5)))
    d = rel[:, 0, 1:] - rel[:, 0, :-1]
    print(" wave 0 phase durations (median ticks):", "  ".join("%s %.0f" % (n, np.median(d[:, i])) for i, n in enumerate(("prologue", "stream", "reduce+barrier", "epilogue"))))
