"""Vision-tower attention at the config-3 shape (32 frames: 16 temporal patches x 836 tokens = 13 376 tokens, 16 heads of 80 on 128-wide padded heads, per-frame segments):
the head-dim-128 row-major kernel with all 16 + 16 MFMAs per tile against the live-96 launch (round 6: 12 + 12).  HIP events, --iters launches each."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time_r1_amd  # noqa: E402,F401
from time_r1_amd.ops import HipOps  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--segs", type=int, default=16)
ap.add_argument("--seg", type=int, default=836)
a = ap.parse_args()
ops = HipOps("cuda:0")
H, N = 16, a.segs * a.seg
g = torch.Generator(device="cuda").manual_seed(1)
live = (torch.arange(128, device="cuda") % 128 < 96) & ((torch.arange(128, device="cuda") < 40) | ((torch.arange(128, device="cuda") >= 48) & (torch.arange(128, device="cuda") < 88)))
mk = lambda: (torch.randn(N, H, 128, generator=g, device="cuda") * live).to(torch.bfloat16).view(N, H * 128)
q, k, v = mk(), mk(), mk()
t = torch.arange(N, device="cuda", dtype=torch.int32)
pre = torch.zeros(N, dtype=torch.int32, device="cuda")
lo = (t // a.seg) * a.seg
hi = lo + a.seg - 1
o = torch.zeros(N, H * 128, dtype=torch.bfloat16, device="cuda")
out = {"shape": dict(tokens=N, heads=H, segment=a.seg)}
fl = 4.0 * N * a.seg * 80 * H
res = {}
for name, l96 in (("all_128", False), ("live_96", True), ("all_128_again", False), ("live_96_again", True)):
    fn = lambda: ops.attn_fwd(q, k, None, pre, lo, hi, H, H, N, 128, 80 ** -0.5, need_lse=False, v_rows=v, out=o, live96=l96)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    res[name] = dict(us=round(ms * 1e3, 1), TFLOPs_head_dim_80=round(fl / ms / 1e9, 1))
    if name == "all_128":
        ref = o.clone()
    elif name == "live_96":
        res["max_abs_diff_vs_all_128"] = float((o.float() - ref.float()).abs().max())
out["vit_attention"] = res
print(json.dumps(out))
