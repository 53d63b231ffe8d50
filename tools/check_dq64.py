"""attn_bwd_dq64_kernel (64 query rows per wave, one wave per SIMD) against attn_bwd_dq32_kernel through tr1_attn_bwd: dQ, dK and dV (the dK/dV kernel reads the
delta / log2-LSE / mask summary the dQ kernel's prologue writes) must agree BIT FOR BIT, then the whole backward is timed under both.  TR1_DQ64 is read per call.

    python tools/check_dq64.py [--iters 30]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    ops = HipOps("cuda:0")
    g = torch.Generator(device="cuda").manual_seed(5)
    ok_all = True
    for (P, G, C, H, NKV, sc) in ((3474, 8, 200, 28, 4, 1.0), (100, 4, 30, 28, 4, 1.0), (7, 2, 5, 12, 2, 1.0), (64, 1, 1, 4, 4, 1.0), (700, 8, 50, 12, 2, 3.0),
                                  (333, 3, 77, 6, 3, 1.0), (1000, 2, 100, 16, 16, 1.0), (1, 1, 1, 28, 4, 1.0)):
        HD = 128
        lay = PackedLayout(P, G, C)
        M = lay.M
        pre, lo, hi = [torch.tensor(x).cuda() for x in lay.masks()]
        rnd = lambda *s, sc_=1.0: (torch.randn(*s, generator=g, device="cuda") * sc_).to(BF16)
        q, k, v, do = rnd(M, H * HD, sc_=sc), rnd(M, NKV * HD, sc_=sc), rnd(M, NKV * HD), rnd(M, H * HD, sc_=0.1)
        scale = HD ** -0.5
        o, lse = ops.attn_fwd(q, k, None, pre, lo, hi, H, NKV, M, HD, scale, v_rows=v)
        res = {}
        for form in ("0", "1"):
            os.environ["TR1_DQ64"] = form
            res[form] = [t.clone() for t in ops.attn_bwd(q, k, v, o, do, lse, pre, lo, hi, H, NKV, M, HD, scale)]
            torch.cuda.synchronize()
        same = [bool(torch.equal(x.view(torch.int16), y.view(torch.int16))) for x, y in zip(res["0"], res["1"])]
        nan = any(bool(torch.isnan(t.float()).any()) for t in res["1"])
        rec = dict(P=P, G=G, C=C, H=H, NKV=NKV, q_scale=sc, M=M, dQ_bit_equal=same[0], dK_bit_equal=same[1], dV_bit_equal=same[2], nan=nan)
        if not same[0]:
            d = (res["0"][0].float() - res["1"][0].float()).abs()
            bad = torch.nonzero(d.max(dim=1).values > 0).flatten()
            rec["dq_max_abs_diff"] = float(d.max()); rec["n_bad_rows"] = int(bad.numel()); rec["first_bad_rows"] = bad[:8].tolist()
        ok_all &= all(same) and not nan
        if P >= 2500:
            for form in ("0", "1"):
                os.environ["TR1_DQ64"] = form
                fn = lambda: ops.attn_bwd(q, k, v, o, do, lse, pre, lo, hi, H, NKV, M, HD, scale)
                fn(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    fn()
                e1.record(); torch.cuda.synchronize()
                rec["bwd_ms_dq%s" % ("64" if form == "1" else "32")] = round(e0.elapsed_time(e1) / a.iters, 4)
        print(json.dumps(rec), flush=True)
    os.environ.pop("TR1_DQ64", None)
    print("ALL_BIT_EQUAL" if ok_all else "MISMATCH")
    return 0 if ok_all else 1


if __name__ == "__main__":
    sys.exit(main())
