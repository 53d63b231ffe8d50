"""Are the ~4 us idle gaps in front of the decode GEMMs GPU-side or host-side?  Queue a long kernel first so that the whole decode-layer
sequence is already enqueued when the GPU reaches it, then read the gaps with tools/rocpd_gaps.py."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import time_r1_amd  # noqa: E402,F401
from time_r1_amd.ops import HipOps  # noqa: E402

ops = HipOps("cuda:0")
BF = torch.bfloat16
r = lambda *s: (torch.randn(*s, device="cuda") * 0.1).to(BF)
M, d, I = 16, 3584, 18944
h, ln = r(M, d), r(d)
wq, bq, wo, wgu, wd = r(4608, d), r(4608), r(d, d), r(2 * I, d), r(d, I)
big_a, big_b = r(8192, 8192), r(8192, 8192)
for rep in range(3):
    for _ in range(4):
        ops.gemm_nt(big_a, big_b)          # ~4 x 1 ms of GPU work: the host runs ahead
    x = h
    for layer in range(8):
        qkv = ops.norm_gemm(x, ln, 1e-6, wq, bias=bq)
        o = qkv[:, :d].contiguous()
        h2 = ops.gemm_nt(o, wo, residual=x)
        a = ops.norm_gemm(h2, ln, 1e-6, wgu, glu=True)
        x = ops.gemm_nt(a, wd, residual=h2)
    torch.cuda.synchronize()
