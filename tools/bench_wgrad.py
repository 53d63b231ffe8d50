"""A/B of the two weight-gradient forms at the 7B shapes: transpose(x) + NT GEMM vs the K-major (NN) GEMM that reads x as stored."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time_r1_amd  # noqa
from time_r1_amd.ops import HipOps
ops = HipOps("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 5074
g = torch.Generator(device="cuda").manual_seed(0)
def rnd(*s): return (torch.randn(*s, generator=g, device="cuda") * 0.05).bfloat16()
def t(fn, n=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
for name, N, K in (("down", 3584, 18944), ("gate/up", 37888, 3584), ("o", 3584, 3584), ("qkv", 4608, 3584), ("lm_head(1600 rows)", 152064, 3584)):
    m = 1600 if "lm_head" in name else M
    dy, x = rnd(m, N), rnd(m, K)
    gw = torch.zeros(N, K, device="cuda")
    dyt = ops.transpose(dy)
    for acc in (False, True):
        a = t(lambda: ops.gemm_nt(dyt, ops.transpose(x), out_f32=True, out=gw, accumulate=acc))
        xt = ops.transpose(x)
        a2 = t(lambda: ops.gemm_nt(dyt, xt, out_f32=True, out=gw, accumulate=acc))
        b = t(lambda: ops.wgrad_nn(dyt, x, gw, acc))
        tr = t(lambda: ops.transpose(dy))
        fl = 2.0 * m * N * K
        print("%-20s acc=%d  transpose(x)+NT %7.1f us (NT alone %7.1f = %5.0f TF)   NN %7.1f us (%5.0f TF)   transpose(dy) %6.1f us" % (name, acc, a, a2, fl / a2 / 1e6, b, fl / b / 1e6, tr))
