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
print("cfg %s M=%d: fused qkv %6.2f us   o projection %6.2f us" % (os.environ.get("TR1_NG32_CFG", "0"), M, timeit(qkv), timeit(oproj)))
