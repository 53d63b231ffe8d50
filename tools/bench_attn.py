"""Packed shared-prefix attention forward / backward at the config-3 shape (P = 3474, G = 8, C = 200, 28 heads / 4 kv heads, head dim 128):
per-kernel times from HIP events and relative L2 error against the masked fp32 attention on the GPU (same reference as
tests/test_fullsize_gpu.py).  (Rounds 3-4 selected kernel variants by environment - TR1_DKDV32, TR1_DQ32, TR1_FWD32 - one process per variant; round 5 removed those switches: the 32x32x16 kernels are the only head-dim-128 forms.  A/B runs now go through tools/build_ref_lib.sh.)

    python tools/bench_attn.py [--iters 20] [--no-check] [--P 3474 --G 8 --C 200]
"""
import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time_r1_amd  # noqa: E402,F401
from time_r1_amd.ops import HipOps  # noqa: E402
from time_r1_amd.positions import PackedLayout  # noqa: E402

BF16 = torch.bfloat16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--probe-fwd", action="store_true", help="needs TR1_HIP_LIB=tools/_probe_lib.so: s_memtime stamps of the heaviest forward block (tile top, after barrier, after S, after softmax, after PV, end)")
    ap.add_argument("--probe", action="store_true", help="needs TR1_HIP_LIB=tools/_probe_lib.so: dump the s_memtime stamps of one dK/dV block")
    ap.add_argument("--yardstick", action="store_true", help="also time torch's scaled_dot_product_attention (vendor flash backend) fwd / bwd on the same head geometry")
    ap.add_argument("--P", type=int, default=3474)
    ap.add_argument("--G", type=int, default=8)
    ap.add_argument("--C", type=int, default=200)
    ap.add_argument("--H", type=int, default=28)
    ap.add_argument("--NKV", type=int, default=4)
    ap.add_argument("--HD", type=int, default=128)
    a = ap.parse_args()
    H, NKV, HD = a.H, a.NKV, a.HD
    ops = HipOps("cuda:0")
    lay = PackedLayout(a.P, a.G, a.C)
    M = lay.M
    pre, lo, hi = [torch.tensor(x).cuda() for x in lay.masks()]
    g = torch.Generator(device="cuda").manual_seed(1)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device="cuda") * sc).to(BF16)
    q, k, v, do = rnd(M, H * HD), rnd(M, NKV * HD), rnd(M, NKV * HD), rnd(M, H * HD, sc=0.1)
    scale = HD ** -0.5
    vt = ops.pack_transpose(v, NKV, NKV, HD)
    o, lse = ops.attn_fwd(q, k, vt, pre, lo, hi, H, NKV, M, HD, scale, v_rows=v)
    dq, dk, dv = ops.attn_bwd(q, k, v, o, do, lse, pre, lo, hi, H, NKV, M, HD, scale)
    torch.cuda.synchronize()
    out = {"shape": dict(P=a.P, G=a.G, C=a.C, M=M, H=H, NKV=NKV, HD=HD), "env": {k_: v_ for k_, v_ in os.environ.items() if k_.startswith("TR1_")}}
    pairs = a.P * (a.P + 1) / 2 + a.G * (a.C * a.P + a.C * (a.C + 1) / 2)
    fl_fwd = 4.0 * pairs * HD * H

    def timed(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters
    t_f = timed(lambda: ops.attn_fwd(q, k, vt, pre, lo, hi, H, NKV, M, HD, scale, v_rows=v))
    t_b = timed(lambda: ops.attn_bwd(q, k, v, o, do, lse, pre, lo, hi, H, NKV, M, HD, scale))
    out["fwd_ms"], out["fwd_TFLOPs_visible"] = round(t_f, 4), round(fl_fwd / t_f / 1e9, 1)
    out["bwd_ms"], out["bwd_TFLOPs_2p5x"] = round(t_b, 4), round(2.5 * fl_fwd / t_b / 1e9, 1)
    if a.yardstick:
        # Vendor yardstick (checker only, never product): torch's scaled_dot_product_attention (the ROCm flash backend) on the same head geometry.
        # (a) one causal sequence of M tokens (the packed call's nearest dense-mask relative: rate compared in TFLOP/s over VISIBLE pairs);
        # (b) what the reference executes for this step: G sequences of P + C tokens, causal (flash_attn_varlen semantics, the prefix G times).
        import torch.nn.functional as F
        from torch.nn.attention import SDPBackend, sdpa_kernel
        ys = {}
        for name, B, L in (("causal_1xM", 1, M), ("reference_GxPC", a.G, a.P + a.C)):
            qy = torch.randn(B, H, L, HD, generator=g, device="cuda").to(BF16).requires_grad_(True)
            ky = torch.randn(B, NKV, L, HD, generator=g, device="cuda").to(BF16).requires_grad_(True)
            vy = torch.randn(B, NKV, L, HD, generator=g, device="cuda").to(BF16).requires_grad_(True)
            doy = (torch.randn(B, H, L, HD, generator=g, device="cuda") * 0.1).to(BF16)
            fl = 4.0 * B * (L * (L + 1) / 2) * HD * H
            for backend in ("FLASH_ATTENTION", "EFFICIENT_ATTENTION"):
                try:
                    with sdpa_kernel(getattr(SDPBackend, backend)):
                        fwd = lambda: F.scaled_dot_product_attention(qy, ky, vy, is_causal=True, enable_gqa=True)
                        with torch.no_grad():
                            t_fy = timed(fwd)

                        def fb():
                            o_ = fwd()
                            o_.backward(doy)
                            qy.grad = ky.grad = vy.grad = None
                        t_fb = timed(fb)
                    ys["%s/%s" % (name, backend)] = dict(fwd_ms=round(t_fy, 4), fwd_TFLOPs_visible=round(fl / t_fy / 1e9, 1), bwd_ms=round(t_fb - t_fy, 4),
                                                         bwd_TFLOPs_2p5x=round(2.5 * fl / max(t_fb - t_fy, 1e-6) / 1e9, 1))
                except Exception as e:      # a backend that refuses the shape is part of the answer
                    ys["%s/%s" % (name, backend)] = "unavailable: %s" % (str(e).splitlines()[0][:160],)
        out["yardstick_sdpa"] = ys
    if not a.no_check:
        T, S = M, M
        kvi = torch.arange(S, device="cuda")[None, :]
        vis = (kvi < pre[:, None]) | ((kvi >= lo[:, None]) & (kvi <= hi[:, None]))
        qf, kf, vf = [t.float().requires_grad_(True) for t in (q, k, v)]
        qh = qf.view(T, H, HD).transpose(0, 1)
        kh = kf.view(S, NKV, HD).transpose(0, 1).repeat_interleave(H // NKV, 0)
        vh = vf.view(S, NKV, HD).transpose(0, 1).repeat_interleave(H // NKV, 0)
        outs = []
        for h0 in range(0, H, 4):
            s = (qh[h0:h0 + 4] @ kh[h0:h0 + 4].transpose(1, 2)) * scale
            s = s.masked_fill(~vis[None], float("-inf"))
            outs.append(torch.softmax(s, -1) @ vh[h0:h0 + 4])
        ref = torch.cat(outs, 0).transpose(0, 1).reshape(T, H * HD)
        ref.backward(do.float())
        rel = lambda x, y: float((x.float() - y.float()).norm() / y.float().norm().clamp(min=1e-20))
        out["rel_l2"] = dict(o=rel(o, ref.detach()), dq=rel(dq, qf.grad), dk=rel(dk, kf.grad), dv=rel(dv, vf.grad))
        out["nan"] = bool(torch.isnan(dq.float()).any() or torch.isnan(dk.float()).any() or torch.isnan(dv.float()).any())
    if a.probe_fwd:
        from time_r1_amd import hip
        buf = torch.zeros(8 * 48 * 8, dtype=torch.int64, device="cuda")
        rc = hip.lib().cdll.probe_fwd_set_ptr(ctypes.c_void_p(buf.data_ptr()))
        assert rc == 0, rc
        ops.attn_fwd(q, k, vt, pre, lo, hi, H, NKV, M, HD, scale, v_rows=v)
        torch.cuda.synchronize()
        hip.lib().cdll.probe_fwd_set_ptr(ctypes.c_void_p(0))
        st = buf.cpu().view(8, 48, 8).numpy()
        t0 = int(st[st > 0].min())
        lines = []
        for w in range(8):
            for it in range(48):
                if st[w, it].max() == 0:
                    break
                lines.append("w%d it%02d " % (w, it) + " ".join("%7d" % (int(x) - t0 if x > 0 else -1) for x in st[w, it][:6]))
        out["probe_fwd"] = lines
    if a.probe:
        from time_r1_amd import hip
        buf = torch.zeros(16 * 64 * 8, dtype=torch.int64, device="cuda")
        rc = hip.lib().cdll.probe_bwd_set_ptr(ctypes.c_void_p(buf.data_ptr()))
        assert rc == 0, rc
        ops.attn_bwd(q, k, v, o, do, lse, pre, lo, hi, H, NKV, M, HD, scale)
        torch.cuda.synchronize()
        hip.lib().cdll.probe_bwd_set_ptr(ctypes.c_void_p(0))
        st = buf.cpu().view(16, 64, 8).numpy()
        t0 = int(st[st > 0].min())
        lines = []
        for w in range(16):
            if st[w].max() == 0:
                continue
            for it in range(64):
                if st[w, it].max() == 0:
                    break
                lines.append("w%d it%02d " % (w, it) + " ".join("%7d" % (int(x) - t0 if x > 0 else -1) for x in st[w, it]))
        out["probe"] = lines
    print(json.dumps(out))


if __name__ == "__main__":
    main()
