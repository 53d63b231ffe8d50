"""sha256 of the attention forward / backward / split-KV merge outputs and of the row-reduction kernels (rmsnorm, log-prob + entropy) on fixed seeded inputs (several mask families and sizes): run under two builds (TR1_HIP_LIB=...) and diff the
lines to show that a kernel edit left every bit where it was.    python tools/hash_attn.py"""
import hashlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time_r1_amd  # noqa
from time_r1_amd.ops import HipOps
from time_r1_amd.positions import PackedLayout
ops = HipOps("cuda:0")
h = lambda t: hashlib.sha256(t.contiguous().view(torch.int16 if t.dtype == torch.bfloat16 else torch.int32).cpu().numpy().tobytes()).hexdigest()[:16]
def seg_masks(lengths):
    lo, hi, a = [], [], 0
    for n in lengths:
        lo += [a] * n; hi += [a + n - 1] * n; a += n
    return torch.zeros(a, dtype=torch.int32), torch.tensor(lo, dtype=torch.int32), torch.tensor(hi, dtype=torch.int32)
cases = [("prefix-shared", 28, 4, [torch.tensor(x) for x in PackedLayout(700, 8, 50).masks()]), ("prefix-shared-2b", 12, 2, [torch.tensor(x) for x in PackedLayout(333, 3, 77).masks()]),
         ("segments", 4, 4, seg_masks([70, 3, 130, 64, 200])), ("big", 28, 4, [torch.tensor(x) for x in PackedLayout(3474, 8, 200).masks()])]
for name, H, NKV, (pre, lo, hi) in cases:
    pre, lo, hi = pre.int().cuda(), lo.int().cuda(), hi.int().cuda()
    S = pre.numel(); g = torch.Generator(device="cuda").manual_seed(11)
    rnd = lambda *s: torch.randn(*s, generator=g, device="cuda").to(torch.bfloat16)
    q, k, v, do = rnd(S, H * 128), rnd(S, NKV * 128), rnd(S, NKV * 128), rnd(S, H * 128)
    for f64 in ("0", "1"):
        os.environ["TR1_FWD64"] = f64
        o, lse = ops.attn_fwd(q, k, None, pre, lo, hi, H, NKV, S, 128, 128 ** -0.5, v_rows=v)
        print(name, "fwd64=" + f64, "O", h(o), "lse", h(lse))
    dq, dk, dv = ops.attn_bwd(q, k, v, o, do, lse, pre, lo, hi, H, NKV, S, 128, 128 ** -0.5)
    print(name, "dq", h(dq), "dk", h(dk), "dv", h(dv))
    if S <= 1200:      # the split-KV path (V^T operand, 5 splits): partials + the one-wave-per-row merge (wave reductions, closing sums)
        vt = ops.pack_transpose(v, NKV, NKV, 128)
        o5, l5 = ops.attn_fwd(q, k, vt, pre, lo, hi, H, NKV, S, 128, 128 ** -0.5, nsplit=5)
        print(name, "split-KV O", h(o5), "lse", h(l5))
# row reductions (wave_sum / wave_max of tr1_common.h): rmsnorm forward / backward, log-prob + entropy
g = torch.Generator(device="cuda").manual_seed(12)
x = torch.randn(777, 3584, generator=g, device="cuda").to(torch.bfloat16); w = (1 + 0.1 * torch.randn(3584, generator=g, device="cuda")).to(torch.bfloat16)
y, rstd, _ = ops.rmsnorm_fwd(x, w, 1e-6, need_rstd=True)
print("rmsnorm fwd", h(y), h(rstd))
dy = torch.randn(777, 3584, generator=g, device="cuda").to(torch.bfloat16)
dw = torch.zeros(3584, dtype=torch.float32, device="cuda")
dx = ops.rmsnorm_bwd(dy, x, w, rstd, dw=dw)
print("rmsnorm bwd", h(dx), h(dw))
lg = torch.randn(64, 152064, generator=g, device="cuda").to(torch.bfloat16); tg = torch.randint(0, 152064, (64,), generator=g, device="cuda", dtype=torch.int32)
lp, ent, lse = ops.logp_entropy_fwd(lg, tg)
print("logp / entropy / lse", h(lp), h(ent), h(lse))
