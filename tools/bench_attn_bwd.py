"""Attention backward at the packed config-3 / config-4 shapes: time per call and parity of the dK/dV kernel (rounds 2-4: the form selected by TR1_DKDV_DMA, removed in round 5) against the
register-staged form is checked by the tests; this prints the timings (HIP events around tr1_attn_bwd = delta + dQ + dK/dV + reduce)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time_r1_amd  # noqa
from time_r1_amd.ops import HipOps
from time_r1_amd.positions import PackedLayout
ops = HipOps("cuda:0")
P, G, C = (3474, 8, 200) if len(sys.argv) < 4 else map(int, sys.argv[1:4])
H, NKV, HD = 28, 4, 128
lay = PackedLayout(P, G, C)
M = lay.M
pre, lo, hi = [torch.tensor(a).cuda() for a in lay.masks()]
g = torch.Generator(device="cuda").manual_seed(0)
def rnd(*s, scale=1.0): return (torch.randn(*s, generator=g, device="cuda") * scale).bfloat16()
q, k, v, do = rnd(M, H * HD), rnd(M, NKV * HD), rnd(M, NKV * HD), rnd(M, H * HD, scale=0.1)
o, lse = ops.attn_fwd(q, k, ops.pack_transpose(v, NKV, NKV, HD), pre, lo, hi, H, NKV, M, HD, HD ** -0.5)
def run(): return ops.attn_bwd(q, k, v, o, do, lse, pre, lo, hi, H, NKV, M, HD, HD ** -0.5)
out = run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
pairs = float(((torch.arange(M, device="cuda")[None] < pre[:, None]) | ((torch.arange(M, device="cuda")[None] >= lo[:, None]) & (torch.arange(M, device="cuda")[None] <= hi[:, None]))).sum())
fl = pairs * H * HD * 2 * 5
ms = e0.elapsed_time(e1) / 10
print("M=%d  attn_bwd %.3f ms  (%.0f TFLOP/s algorithmic)  checksum %.6f %.6f %.6f" % (M, ms, fl / ms / 1e9,
      float(out[0].float().abs().mean()), float(out[1].float().abs().mean()), float(out[2].float().abs().mean())))
