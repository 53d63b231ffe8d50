"""This library's bf16 GEMM against hipBLASLt (torch.matmul) on the training shapes of config 3 (M = 5074 packed rows / 1600 continuation rows)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time_r1_amd  # noqa
from time_r1_amd.ops import HipOps
ops = HipOps("cuda:0")
def rnd(*s): return (torch.randn(*s, device="cuda") * 0.05).bfloat16()
def t(fn, n=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
for name, M, N, K in (("gate/up fwd", 5074, 37888, 3584), ("down fwd", 5074, 3584, 18944), ("qkv fwd", 5074, 4608, 3584), ("o fwd", 5074, 3584, 3584),
                      ("gate/up cont", 1600, 37888, 3584), ("down cont", 1600, 3584, 18944), ("lm_head", 1600, 152064, 3584), ("prefill gate/up", 3474, 37888, 3584),
                      ("ViT fc1", 13376, 5120, 1280), ("ViT qkv", 13376, 3840, 1280), ("8192^3", 8192, 8192, 8192)):
    a, b = rnd(M, K), rnd(N, K)
    own = t(lambda: ops.gemm_nt(a, b))
    lib = t(lambda: torch.matmul(a, b.t()))
    fl = 2.0 * M * N * K
    print("%-16s M=%5d N=%6d K=%5d   own %7.1f us %5.0f TF   hipBLASLt %7.1f us %5.0f TF   own/lib time %.2f" % (name, M, N, K, own, fl / own / 1e6, lib, fl / lib / 1e6, own / lib))
