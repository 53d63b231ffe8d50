"""gemm_nt4w_kernel (256 x 256 tile on four waves, one per SIMD; TR1_GEMM4W=1) against the default NT GEMM dispatch on the step's shapes: outputs must agree
BIT FOR BIT (same MFMA k order per output element), then both are timed (HIP events) and priced in TFLOP/s.
    python tools/check_gemm4w.py [--iters 20]"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time_r1_amd  # noqa
from time_r1_amd.ops import HipOps

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=20); a = ap.parse_args()
ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(7)
ok_all = True
for (M, N, K, f32) in ((5074, 37888, 3584, False), (5074, 3584, 18944, False), (5074, 4608, 3584, False), (5074, 3584, 3584, False), (1600, 37888, 3584, False),
                      (3474, 37888, 3584, False), (8192, 8192, 8192, False), (37888, 3584, 5120, True), (777, 520, 192, False)):
    A = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
    B = torch.randn(N, K, generator=g, device="cuda").to(torch.bfloat16)
    SK = ops.SPLITK; HipOps.SPLITK = False
    res, tm = {}, {}
    for form in ("0", "1", "0", "1"):
        os.environ["TR1_GEMM4W"] = form
        fn = lambda: ops.gemm_nt(A, B, out_f32=f32)
        res[form] = fn().clone(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters): fn()
        e1.record(); torch.cuda.synchronize()
        tm.setdefault(form, []).append(e0.elapsed_time(e1) / a.iters)
    HipOps.SPLITK = SK
    same = bool(torch.equal(res["0"].view(torch.int32 if f32 else torch.int16), res["1"].view(torch.int32 if f32 else torch.int16)))
    ok_all &= same
    fl = 2.0 * M * N * K
    rec = dict(M=M, N=N, K=K, out="f32" if f32 else "bf16", bit_equal=same, default_us=[round(t * 1e3, 1) for t in tm["0"]], g4w_us=[round(t * 1e3, 1) for t in tm["1"]],
               default_TFLOPs=round(fl / min(tm["0"]) / 1e9, 1), g4w_TFLOPs=round(fl / min(tm["1"]) / 1e9, 1))
    if not same:
        d = (res["0"].float() - res["1"].float()).abs(); rec["max_abs_diff"] = float(d.max()); rec["n_diff"] = int((d > 0).sum())
    print(json.dumps(rec), flush=True)
os.environ.pop("TR1_GEMM4W", None)
print("ALL_BIT_EQUAL" if ok_all else "MISMATCH")
