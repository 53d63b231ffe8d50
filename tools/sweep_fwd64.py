"""attn_fwd64_kernel against attn_fwd32_kernel over the prompt length (tiles per block): where the one-wave-per-SIMD pipeline's fill / drain (one body of 64 MFMAs per block) stops paying.
    python tools/sweep_fwd64.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time_r1_amd  # noqa
from time_r1_amd.ops import HipOps
from time_r1_amd.positions import PackedLayout
ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(3)
for (H, NKV) in ((28, 4), (12, 2)):
    for P in (256, 512, 768, 1024, 1536, 2048, 3072):
        G, C, HD = 8, 200, 128
        lay = PackedLayout(P, G, C); M = lay.M
        pre, lo, hi = [torch.tensor(x).cuda() for x in lay.masks()]
        rnd = lambda *s: torch.randn(*s, generator=g, device="cuda").to(torch.bfloat16)
        q, k, v = rnd(M, H * HD), rnd(M, NKV * HD), rnd(M, NKV * HD)
        rec = dict(H=H, NKV=NKV, P=P, M=M)
        for form in ("0", "1", "0", "1"):
            os.environ["TR1_FWD64"] = form
            fn = lambda: ops.attn_fwd(q, k, None, pre, lo, hi, H, NKV, M, HD, HD ** -0.5, v_rows=v)
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): fn()
            e1.record(); torch.cuda.synchronize()
            rec.setdefault("fwd64_us" if form == "1" else "fwd32_us", []).append(round(e0.elapsed_time(e1) / 30 * 1e3, 1))
        print(json.dumps(rec), flush=True)
