#!/usr/bin/env python
"""Kernel micro-benchmarks on one MI355X (HIP events on the launch stream). Usage: python tools/microbench.py <what> [...]
   what: skinny | fixup | fused | gemm | attn_decode | attn_train | small | sampler | optim | all"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import time_r1_amd  # noqa: E402,F401
from time_r1_amd.ops import HipOps  # noqa: E402

ops = HipOps("cuda:0")
BF = torch.bfloat16


def timeit(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def rnd(*s):
    return (torch.randn(*s, device="cuda") * 0.5).to(BF)


def skinny():
    print("== decode GEMM (M rows x W[N,K]) : us, GB/s")
    for M in (8, 16, 32, 64):
        for N, K in [(4608, 3584), (3584, 3584), (37888, 3584), (3584, 18944), (152064, 3584), (2048, 1536), (17920, 1536), (1536, 8960)]:
            # rotate over several weight copies so the stream really comes from HBM, not the 256 MB Infinity Cache
            ncopy = max(1, int(600e6 // (N * K * 2)))
            ws = [rnd(N, K) for _ in range(min(ncopy, 8))]
            x = rnd(M, K)
            i = [0]

            def f():
                ops.gemm_nt(x, ws[i[0] % len(ws)])
                i[0] += 1
            us = timeit(f, reps=40)
            print("M=%2d N=%6d K=%6d  %8.1f us  %7.1f GB/s" % (M, N, K, us, (N * K * 2 + M * K * 2 + M * N * 2) / us / 1e3))


def fixup():
    import ctypes
    print("== narrow decode projections: plain skinny vs split-K + in-kernel fixup (raw C-ABI calls): us")
    f0, f1 = ops.L.raw("tr1_gemm_nt_bf16"), ops.L.raw("tr1_gemm_skinny_fixup")
    for M in (8, 16, 32):
        for N, K in [(3584, 18944), (3584, 3584), (1536, 8960), (1536, 1536)]:
            ws_ = [rnd(N, K) for _ in range(max(1, min(8, int(600e6 // (N * K * 2)))))]
            x, res, out = rnd(M, K), rnd(M, N), torch.empty(M, N, device="cuda", dtype=BF)
            n = int(ops.L.raw("tr1_gemm_skinny_fixup_workspace_floats")(M, N, K))
            wsf = torch.zeros(n, device="cuda")
            P = lambda t: ctypes.c_void_p(t.data_ptr())
            a0 = [(P(x), P(w), P(out), None, P(res), M, N, K, K, K, N, N, 0, 0, None) for w in ws_]
            a1 = [(P(x), P(w), P(out), None, P(res), M, N, K, K, K, N, N, P(wsf), n, None) for w in ws_]
            i = [0]

            def g0():
                f0(*a0[i[0] % len(a0)]); i[0] += 1

            def g1():
                f1(*a1[i[0] % len(a1)]); i[0] += 1
            print("M=%2d N=%6d K=%6d   plain %6.1f   fixup %6.1f" % (M, N, K, timeit(g0, reps=100), timeit(g1, reps=100)))


def w8():
    import ctypes
    print("== decode GEMMs with fp8 weights (W8A16) vs bf16, raw C-ABI calls: us")
    f8, fb, fn = ops.L.raw("tr1_gemm_skinny_w8"), ops.L.raw("tr1_gemm_nt_bf16"), ops.L.raw("tr1_norm_gemm_skinny")
    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    for M in (16, 32):
        for N, K, mode in [(4608, 3584, "norm"), (3584, 3584, "res"), (18944, 3584, "glu"), (3584, 18944, "res"), (152064, 3584, "norm")]:
            nw = 2 * N if mode == "glu" else N
            nc = max(1, min(6, int(600e6 // (nw * K))))
            ws_ = [rnd(nw, K) for _ in range(nc)]
            qs = [ops.quantize_fp8_rows(w) for w in ws_]
            x, lnw, res, out = rnd(M, K), rnd(K), rnd(M, N), torch.empty(M, N, device="cuda", dtype=BF)
            a8 = [(P(x), P(lnw) if mode != "res" else None, P(q), P(sc), None, P(res) if mode == "res" else None, P(out), M, N, K, K, K, N, N, 1e-6,
                   int(mode == "glu"), None) for q, sc in qs]
            if mode == "res":
                ab = [(P(x), P(w), P(out), None, P(res), M, N, K, K, K, N, N, 0, 0, None) for w in ws_]
                fbase = fb
            else:
                ab = [(P(x), P(lnw), P(w), None, P(out), M, N, K, K, K, N, 1e-6, int(mode == "glu"), None) for w in ws_]
                fbase = fn
            i = [0]

            def g8():
                f8(*a8[i[0] % len(a8)]); i[0] += 1

            def gb():
                fbase(*ab[i[0] % len(ab)]); i[0] += 1
            tb, t8 = timeit(gb, reps=60), timeit(g8, reps=60)
            print("M=%2d N=%6d K=%6d %-4s  bf16 %7.1f  fp8 %7.1f   (fp8 %.2f TB/s)" % (M, N, K, mode, tb, t8, nw * K / t8 / 1e6))


def transpose():
    print("== transpose (dy^T / x^T / W^T operands of the backward GEMMs): us, TB/s (read + write)")
    for R, C in [(5074, 37888), (5074, 18944), (5074, 3584), (37888, 3584), (3584, 18944), (4608, 3584)]:
        x = rnd(R, C)
        us = timeit(lambda: ops.transpose(x), reps=20, warm=3)
        print("R=%6d C=%6d  %8.1f us  %5.2f TB/s" % (R, C, us, 2.0 * R * C * 2 / us / 1e6))


def fused():
    import ctypes
    print("== decode layer pieces, raw C-ABI calls: us  (unfused rmsnorm + gemm [+ swiglu]  vs  norm_gemm)")
    L = ops.L
    for M in (16, 32):
        for N, K, glu in [(4608, 3584, False), (18944, 3584, True)]:
            nw = 2 * N if glu else N
            ws = [rnd(nw, K) for _ in range(max(1, min(8, int(600e6 // (nw * K * 2)))))]
            x, lnw, bias = rnd(M, K), rnd(K), rnd(N)
            i = [0]

            def unf():
                w = ws[i[0] % len(ws)]; i[0] += 1
                xn, _, _ = ops.rmsnorm_fwd(x, lnw, 1e-6, need_rstd=False)
                y = ops.gemm_nt(xn, w, bias=None if glu else bias)
                if glu:
                    ops.swiglu_fwd(y)

            def fu():
                w = ws[i[0] % len(ws)]; i[0] += 1
                ops.norm_gemm(x, lnw, 1e-6, w, bias=None if glu else bias, glu=glu)
            print("M=%2d N=%6d K=%6d glu=%d   unfused %7.1f   fused %7.1f" % (M, N, K, glu, timeit(unf, reps=60), timeit(fu, reps=60)))


def gemm():
    print("== training GEMM: us, TFLOP/s")
    for M, N, K in [(5074, 4608, 3584), (5074, 3584, 3584), (5074, 37888, 3584), (5074, 3584, 18944), (3474, 37888, 3584), (1600, 152064, 3584),
                    (1600, 37888, 3584), (1600, 3584, 18944), (1600, 4608, 3584), (3474, 3584, 18944), (3474, 4608, 3584), (3584, 18944, 5120), (37888, 3584, 5120), (13376, 3840, 1280), (13376, 5120, 1280), (13376, 1280, 5120), (4096, 4096, 4096), (8192, 8192, 8192)]:
        a, b = rnd(M, K), rnd(N, K)
        us = timeit(lambda: ops.gemm_nt(a, b), reps=10, warm=2)
        print("M=%6d N=%6d K=%6d  %9.1f us  %7.1f TF" % (M, N, K, us, 2.0 * M * N * K / us / 1e6))
    print("== dgrad dX = dY @ W: transpose(W) + NT   vs   NN (K-major B): us")
    for M, N, K in [(5074, 3584, 18944), (5074, 18944, 3584), (5074, 3584, 37888), (5074, 3584, 4608), (5074, 3584, 3584)]:
        dy, w = rnd(M, K), rnd(K, N)
        t0 = timeit(lambda: ops.gemm_nt(dy, ops.transpose(w)), reps=10, warm=2)
        t1 = timeit(lambda: ops.gemm_nn(dy, w), reps=10, warm=2)
        print("M=%6d N=%6d K=%6d  transpose+NT %8.1f   NN %8.1f  (%.0f TF)" % (M, N, K, t0, t1, 2.0 * M * N * K / t1 / 1e6))
    a, b = rnd(3584, 5120), rnd(18944, 5120)
    out = torch.zeros(3584, 18944, device="cuda")
    us = timeit(lambda: ops.gemm_nt(a, b, out_f32=True, out=out, accumulate=True), reps=10, warm=2)
    print("wgrad f32 accumulate 3584x18944x5120  %9.1f us  %7.1f TF" % (us, 2.0 * 3584 * 18944 * 5120 / us / 1e6))


def attn_decode():
    import numpy as np
    print("== decode attention over the KV cache (7B: 28 q heads / 4 kv, hd 128, P=3474, G=8, C=200): us per layer")
    P, G, C, nh, nkv, hd = 3474, 8, 200, 28, 4, 128
    S = P + G * C
    Scap = (S + 63) // 64 * 64
    k, vt, q = rnd(Scap, nkv * hd), rnd(nkv * hd, Scap), rnd(G, nh * hd)
    step = 100
    pre = torch.full((G,), P, dtype=torch.int32, device="cuda")
    lo = (P + torch.arange(G) * C).int().cuda()
    hi = (lo + step).int()
    for ns in [int(x) for x in os.environ.get('TR1_MB_NSPLIT', '1,2,4,8,14,28,57').split(',')]:
        us = timeit(lambda: ops.attn_fwd(q, k, vt, pre, lo, hi, nh, nkv, S, hd, hd ** -0.5, nsplit=ns, need_lse=False), reps=100)
        print("nsplit=%2d  %7.1f us" % (ns, us))


def attn_train():
    print("== packed training attention fwd/bwd (7B config 3): ms")
    from time_r1_amd.positions import PackedLayout
    P, G, C, nh, nkv, hd = 3474, 8, 200, 28, 4, 128
    lay = PackedLayout(P, G, C)
    M = lay.M
    pre, lo, hi = [torch.tensor(a).cuda() for a in lay.masks()]
    q, k, v, do = rnd(M, nh * hd), rnd(M, nkv * hd), rnd(M, nkv * hd), rnd(M, nh * hd)
    vt = ops.pack_transpose(v, nkv, nkv, hd)
    us = timeit(lambda: ops.attn_fwd(q, k, vt, pre, lo, hi, nh, nkv, M, hd, hd ** -0.5), reps=10, warm=2)
    o, lse = ops.attn_fwd(q, k, vt, pre, lo, hi, nh, nkv, M, hd, hd ** -0.5)
    vis = float((P * (P + 1) / 2 + G * (C * P + C * (C + 1) / 2)))
    fl = 4.0 * vis * hd * nh
    print("fwd %8.2f ms  %6.1f TF (visible pairs only)" % (us / 1e3, fl / us / 1e6))
    us = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, pre, lo, hi, nh, nkv, M, hd, hd ** -0.5), reps=5, warm=1)
    print("bwd %8.2f ms  %6.1f TF (2.5x fwd flops)" % (us / 1e3, 2.5 * fl / us / 1e6))


def small():
    print("== small decode-step kernels (rows = 8): us")
    x, w = rnd(8, 3584), rnd(3584)
    print("rmsnorm_fwd [8,3584]      %6.1f" % timeit(lambda: ops.rmsnorm_fwd(x, w, 1e-6, need_rstd=False), reps=200))
    gu = rnd(8, 2 * 18944)
    print("swiglu_fwd  [8,2x18944]   %6.1f" % timeit(lambda: ops.swiglu_fwd(gu), reps=200))
    qkv = rnd(8, 4608)
    cos, sin = torch.rand(8, 64, device="cuda"), torch.rand(8, 64, device="cuda")
    print("rope_apply q [8,28x128]   %6.1f" % timeit(lambda: ops.rope_apply(qkv[:, :3584], 28, 128, cos, sin), reps=200))
    xx, ww = rnd(5074, 3584), rnd(3584)
    us = timeit(lambda: ops.rmsnorm_fwd(xx, ww, 1e-6), reps=50)
    print("rmsnorm_fwd [5074,3584]   %6.1f us  %6.1f GB/s" % (us, 2 * 5074 * 3584 * 2 / us / 1e3))
    big = rnd(29376, 3584)
    us = timeit(lambda: ops.rmsnorm_fwd(big, ww, 1e-6), reps=20)
    print("rmsnorm_fwd [29376,3584]  %6.1f us  %6.1f GB/s" % (us, 2 * 29376 * 3584 * 2 / us / 1e3))
    t = rnd(5074, 18944)
    xb, dyb = rnd(5074, 3584), rnd(5074, 3584)
    _, rs, _ = ops.rmsnorm_fwd(xb, ww, 1e-6)
    dwb = torch.zeros(3584, device="cuda")
    us = timeit(lambda: ops.rmsnorm_bwd(dyb, xb, ww, rs, dres=dyb, dw=dwb), reps=20)
    print("rmsnorm_bwd [5074,3584] (+dres, dw) %6.1f us  %6.0f GB/s (4 row passes)" % (us, 4 * 5074 * 3584 * 2 / us / 1e3))
    us = timeit(lambda: ops.rmsnorm_bwd(dyb, xb, ww, rs, dres=dyb, dw=None), reps=20)
    print("rmsnorm_bwd [5074,3584] (+dres, no dw) %6.1f us" % us)
    us = timeit(lambda: ops.transpose(t), reps=10)
    print("transpose [5074,18944]    %6.1f us  %6.1f GB/s" % (us, 2 * 5074 * 18944 * 2 / us / 1e3))


def optim():
    n = 2 * 1000 * 1000 * 1000
    p, m, v, g = (torch.zeros(n, device="cuda") for _ in range(4))
    p16 = torch.zeros(n, dtype=BF, device="cuda")
    ss = torch.zeros(1, device="cuda")
    us = timeit(lambda: ops.sumsq_accum(g, ss), reps=5, warm=1)
    print("sumsq  n=2e9  %8.1f us  %6.0f GB/s" % (us, 4 * n / us / 1e3))
    us = timeit(lambda: ops.adamw_step(p, m, v, g, p16, 1e-6, 0.9, 0.999, 1e-8, 0.0, 1, ss, 1.0, 1.0, True), reps=5, warm=1)
    print("adamw  n=2e9  %8.1f us  %6.0f GB/s (34 B per parameter)" % (us, 34 * n / us / 1e3))


def sampler():
    logits = rnd(8, 152064)
    tok = torch.zeros(8, 4, dtype=torch.int32, device="cuda")
    st = torch.zeros(1, dtype=torch.int32, device="cuda")
    for k in (0, 50):
        print("sample top_k=%2d [8,152064]  %6.1f us" % (k, timeit(lambda: ops.sample_tokens(logits, 1.0, k, 1, st, tok, None, 1, 0, False), reps=50)))


if __name__ == "__main__":
    what = sys.argv[1:] or ["all"]
    for w in what:
        for name in (["skinny", "fixup", "w8", "transpose", "fused", "gemm", "attn_decode", "attn_train", "small", "sampler", "optim"] if w == "all" else [w]):
            globals()[name]()
