"""Per-tile fixed cost of the training GEMM: one round of 256 tiles (4096 x 4096 outputs, 256-row tiles) and ten rounds (5120 x 32768), K swept; time = a + b K.
usage: python tools/probe_gemm_fixed_cost.py [zeros|randn]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from time_r1_amd.ops import HipOps

kind = sys.argv[1] if len(sys.argv) > 1 else "zeros"
ops = HipOps("cuda:0")
for (M, N) in ((4096, 4096), (5120, 32768)):
    pts = []
    for K in (512, 1024, 2048, 3584, 4096, 8192):
        mk = (lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device="cuda")) if kind == "zeros" else (lambda *s: torch.randn(*s, dtype=torch.bfloat16, device="cuda"))
        a, w = mk(M, K), mk(N, K)
        for _ in range(3):
            ops.gemm_nt(a, w)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 20
        for _ in range(n):
            ops.gemm_nt(a, w)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / n
        pts.append((K, us))
        print("%s M %d N %d K %5d: %8.1f us  %7.1f TFLOP/s" % (kind, M, N, K, us, 2.0 * M * N * K / us / 1e6), flush=True)
    (k0, t0), (k1, t1) = pts[1], pts[-1]
    b = (t1 - t0) / (k1 - k0)
    rounds = ((M + 255) // 256) * (N // 256) / 256.0
    print("   slope %.4f us per k (= %.0f TFLOP/s asymptotic), intercept %.1f us over %.1f rounds = %.2f us per tile round" % (b, 2.0 * M * N / b / 1e6, t0 - b * k0, rounds, (t0 - b * k0) / rounds))
