"""Weight gradient with both operands as stored (csrc/gemm_tn.hip) against the transposed-copy forms, at the 7B shapes; checks the result too.
usage: bench_wgrad_tn.py [tokens]   (TR1_TN_NBT=2|4 selects the 8-wave / 4-wave form)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time_r1_amd  # noqa
from time_r1_amd.ops import HipOps
ops = HipOps("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 5074
g = torch.Generator(device="cuda").manual_seed(0)
def rnd(*s): return (torch.randn(*s, generator=g, device="cuda") * 0.05).bfloat16()
def t(fn, n=10):
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
    gw2 = torch.zeros(N, K, device="cuda")
    assert ops.wgrad_tn(dy, x, gw, False)
    ops.gemm_nt(ops.transpose(dy), ops.transpose(x), out_f32=True, out=gw2, accumulate=False)
    err = (gw - gw2).abs().max().item(); ref = gw2.abs().max().item()
    assert ops.wgrad_tn(dy, x, gw, True)
    err2 = (gw - 2 * gw2).abs().max().item()
    fl = 2.0 * m * N * K
    for acc in (False, True):
        old = t(lambda: ops.gemm_nt(ops.transpose(dy), ops.transpose(x), out_f32=True, out=gw2, accumulate=acc)) if K < 2 * N else \
            t(lambda: ops.wgrad_nn(ops.transpose(dy), x, gw2, acc))
        new = t(lambda: ops.wgrad_tn(dy, x, gw, acc))
        print("%-20s acc=%d  copies + GEMM %7.1f us   TN %7.1f us (%5.0f TF)   max err %.3g / %.3g (ref max %.3g)" % (name, acc, old, new, fl / new / 1e6, err, err2, ref), flush=True)
