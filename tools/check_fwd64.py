"""attn_fwd64_kernel (64 query rows per wave, one wave per SIMD) against attn_fwd32_kernel on the same inputs: O and LSE must agree BIT FOR BIT (same 32-row softmax
groups, same lazy-maximum decisions, same summation orders), then both are timed.  TR1_FWD64 is read per call by tr1_attn_fwd_rows, so one process runs both.

    python tools/check_fwd64.py [--iters 30]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time_r1_amd  # noqa: E402,F401
from time_r1_amd.ops import HipOps  # noqa: E402
from time_r1_amd.positions import PackedLayout  # noqa: E402

BF16 = torch.bfloat16


def run(ops, q, k, v, pre, lo, hi, H, NKV, M, HD, scale, form):
    os.environ["TR1_FWD64"] = form
    o, lse = ops.attn_fwd(q, k, None, pre, lo, hi, H, NKV, M, HD, scale, v_rows=v)
    torch.cuda.synchronize()
    return o, lse


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--probe", action="store_true", help="needs TR1_HIP_LIB=tools/_probe_lib.so: s_memtime stamps of the heaviest block's tile bodies (top, barrier, mask, PV phase, S phase, end)")
    a = ap.parse_args()
    ops = HipOps("cuda:0")
    if a.probe:
        import ctypes
        from time_r1_amd import hip
        from time_r1_amd.positions import PackedLayout as PL
        P, G, C, H, NKV, HD = 3474, 8, 200, 28, 4, 128
        lay = PL(P, G, C); M = lay.M
        pre, lo, hi = [torch.tensor(x).cuda() for x in lay.masks()]
        g = torch.Generator(device="cuda").manual_seed(3)
        rnd = lambda *s: torch.randn(*s, generator=g, device="cuda").to(BF16)
        q, k, v = rnd(M, H * HD), rnd(M, NKV * HD), rnd(M, NKV * HD)
        os.environ["TR1_FWD64"] = "1"
        for _ in range(3):
            ops.attn_fwd(q, k, None, pre, lo, hi, H, NKV, M, HD, HD ** -0.5, v_rows=v)
        buf = torch.zeros(4 * 64 * 8, dtype=torch.int64, device="cuda")
        assert hip.lib().cdll.probe_fwd64_set_ptr(ctypes.c_void_p(buf.data_ptr())) == 0
        ops.attn_fwd(q, k, None, pre, lo, hi, H, NKV, M, HD, HD ** -0.5, v_rows=v)
        torch.cuda.synchronize()
        hip.lib().cdll.probe_fwd64_set_ptr(ctypes.c_void_p(0))
        st = buf.cpu().view(4, 64, 8).numpy()
        t0 = int(st[st > 0].min())
        for w in range(4):
            for it in range(64):
                if st[w, it].max() == 0:
                    break
                r = [int(x) - t0 for x in st[w, it][:6]]
                print("w%d it%02d top %7d | wait+barrier %5d mask %5d PV %5d S %5d tail %5d | body %5d" % (w, it, r[0], r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], r[5] - r[0]))
        return 0
    g = torch.Generator(device="cuda").manual_seed(3)
    out = []
    ok_all = True
    for (P, G, C, H, NKV, sc) in ((3474, 8, 200, 28, 4, 1.0), (100, 4, 30, 28, 4, 1.0), (7, 2, 5, 12, 2, 1.0), (64, 1, 1, 4, 4, 1.0), (700, 8, 50, 12, 2, 3.0),
                                  (333, 3, 77, 6, 3, 1.0), (1000, 2, 100, 16, 16, 1.0), (1, 1, 1, 28, 4, 1.0), (2500, 16, 64, 28, 4, 6.0)):
        HD = 128
        lay = PackedLayout(P, G, C)
        M = lay.M
        pre, lo, hi = [torch.tensor(x).cuda() for x in lay.masks()]
        rnd = lambda *s, sc_=1.0: (torch.randn(*s, generator=g, device="cuda") * sc_).to(BF16)
        q, k, v = rnd(M, H * HD, sc_=sc), rnd(M, NKV * HD, sc_=sc), rnd(M, NKV * HD)
        scale = HD ** -0.5
        o32, l32 = run(ops, q, k, v, pre, lo, hi, H, NKV, M, HD, scale, "0")
        o64, l64 = run(ops, q, k, v, pre, lo, hi, H, NKV, M, HD, scale, "1")
        same_o = bool(torch.equal(o32.view(torch.int16), o64.view(torch.int16)))
        same_l = bool(torch.equal(l32.view(torch.int32), l64.view(torch.int32)))
        nan = bool(torch.isnan(o64.float()).any())
        rec = dict(P=P, G=G, C=C, H=H, NKV=NKV, q_scale=sc, M=M, O_bit_equal=same_o, LSE_bit_equal=same_l, nan=nan)
        if not same_o:
            d = (o32.float() - o64.float()).abs()
            rec["max_abs_diff"] = float(d.max()); rec["n_diff"] = int((d > 0).sum())
            bad = torch.nonzero(d.max(dim=1).values > 0).flatten()
            rec["first_bad_rows"] = bad[:8].tolist(); rec["n_bad_rows"] = int(bad.numel())
        if not same_l:
            rec["lse_max_abs_diff"] = float((l32 - l64).abs().nan_to_num(0).max())
        ok_all &= same_o and same_l and not nan
        if P >= 2500:
            pairs = P * (P + 1) / 2 + G * (C * P + C * (C + 1) / 2)
            fl = 4.0 * pairs * HD * H
            for form in ("0", "1"):
                os.environ["TR1_FWD64"] = form
                fn = lambda: ops.attn_fwd(q, k, None, pre, lo, hi, H, NKV, M, HD, scale, v_rows=v)
                fn(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    fn()
                e1.record(); torch.cuda.synchronize()
                t = e0.elapsed_time(e1) / a.iters
                rec["fwd%s_ms" % ("64" if form == "1" else "32")] = round(t, 4)
                rec["fwd%s_TFLOPs" % ("64" if form == "1" else "32")] = round(fl / t / 1e9, 1)
        out.append(rec)
        print(json.dumps(rec), flush=True)
    os.environ.pop("TR1_FWD64", None)
    print("ALL_BIT_EQUAL" if ok_all else "MISMATCH")
    return 0 if ok_all else 1


if __name__ == "__main__":
    sys.exit(main())
