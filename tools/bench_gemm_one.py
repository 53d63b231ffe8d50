"""One NT GEMM shape, a few launches (for rocprofv3 --pmc passes): python tools/bench_gemm_one.py M N K [iters]   (TR1_GEMM4W=1 selects the four-wave form)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time_r1_amd  # noqa
from time_r1_amd.ops import HipOps
M, N, K = (int(x) for x in sys.argv[1:4]); it = int(sys.argv[4]) if len(sys.argv) > 4 else 5
ops = HipOps("cuda:0"); HipOps.SPLITK = False
g = torch.Generator(device="cuda").manual_seed(7)
A = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16); B = torch.randn(N, K, generator=g, device="cuda").to(torch.bfloat16)
for _ in range(it): ops.gemm_nt(A, B)
torch.cuda.synchronize()
