"""Is the training GEMM clock / power bound?  The same launch on random, small-magnitude and all-zero operands (round 3: a weight-gradient GEMM on
zero-filled inputs ran ~25 % faster inside the step, which made a 'skip the transposes' experiment read 25 ms where the truth was 7 ms)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time_r1_amd  # noqa
from time_r1_amd.ops import HipOps
ops = HipOps("cuda:0")
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, N, K in ((5074, 37888, 3584), (5074, 3584, 18944), (8192, 8192, 8192)):
    fl = 2.0 * M * N * K
    for name, mk in (("randn", lambda *s: torch.randn(*s, device="cuda").bfloat16()), ("randn x 1e-3", lambda *s: (torch.randn(*s, device="cuda") * 1e-3).bfloat16()),
                     ("constant 1.0", lambda *s: torch.ones(*s, device="cuda", dtype=torch.bfloat16)), ("zeros", lambda *s: torch.zeros(*s, device="cuda", dtype=torch.bfloat16))):
        a, b = mk(M, K), mk(N, K)
        us = t(lambda: ops.gemm_nt(a, b))
        lib = t(lambda: torch.matmul(a, b.t()))
        print("%5d x %5d x %5d  %-13s own %7.1f us = %5.0f TFLOP/s   hipBLASLt %7.1f us = %5.0f TFLOP/s" % (M, N, K, name, us, fl / us / 1e6, lib, fl / lib / 1e6), flush=True)
