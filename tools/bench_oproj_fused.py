"""Round 5: split-KV decode attention + merge + o projection per layer - the three launches (attn_dec32, attn_combine, gemm_skinny) against the two
(attn_dec32, merge fused into the o projection, csrc/oproj_fused.hip) - over rotating HBM-resident caches and weights.
   python tools/bench_oproj_fused.py [P G C B]   (default config 3 = 3474 8 200 2; NH / NKV / HID env for the 2B model: NH=12 NKV=2 HID=1536)
   PROBE=1 TR1_HIP_LIB=tools/_probe_lib.so: block timeline of the fused launch (s_memtime stamps)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time_r1_amd  # noqa
from time_r1_amd.ops import HipOps
ops = HipOps("cuda:0")
P, G, C, B = (3474, 8, 200, 2) if len(sys.argv) < 5 else map(int, sys.argv[1:5])
nh, nkv, hd = int(os.environ.get("NH", 28)), int(os.environ.get("NKV", 4)), 128
hid = int(os.environ.get("HID", 3584))
step = int(os.environ.get("STEP", C // 2))
S = P + G * C
scap = (S + 63) // 64 * 64
NC = 12
ks = [(torch.randn(B * scap, nkv * hd, device="cuda") * 0.5).bfloat16() for _ in range(NC)]
vts = [(torch.randn(nkv * hd, B * scap, device="cuda") * 0.5).bfloat16() for _ in range(NC)]
ws = [(torch.randn(hid, nh * hd, device="cuda") * (nh * hd) ** -0.5).bfloat16() for _ in range(NC)]
q = (torch.randn(B * G, nh * hd, device="cuda") * 0.5).bfloat16()
res = (torch.randn(B * G, hid, device="cuda") * 0.5).bfloat16()
nsplit = int(os.environ.get("TR1_DECODE_NSPLIT", max(1, min(28, ((P + 63) // 64 + 3) // 2))))
pre = torch.full((B * G,), P, dtype=torch.int32, device="cuda")
lo = (P + torch.arange(G) * C).int().repeat(B).cuda()
hi = (lo + step).int()
plan = ops.attn_plan(G, nh, nkv, B)
ops.attn_fwd(q, ks[0], vts[0], pre, lo, hi, nh, nkv, scap, hd, hd ** -0.5, nsplit=nsplit, need_lse=False, n_batch=B, kv_batch_slots=scap, plan=plan, plan_mode=1)
i = [0]
def base():
    i[0] = (i[0] + 1) % NC
    o, _ = ops.attn_fwd(q, ks[i[0]], vts[i[0]], pre, lo, hi, nh, nkv, scap, hd, hd ** -0.5, nsplit=nsplit, need_lse=False, n_batch=B, kv_batch_slots=scap, plan=plan, plan_mode=2)
    return ops.gemm_nt(o, ws[i[0]], residual=res)
def fused():
    i[0] = (i[0] + 1) % NC
    return ops.attn_combine_oproj(q, ks[i[0]], vts[i[0]], pre, lo, hi, nh, nkv, scap, hd, hd ** -0.5, nsplit, ws[i[0]], residual=res, n_batch=B, kv_batch_slots=scap, plan=plan, plan_mode=2)[1]
def t(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for rep in range(2):
    print("P=%d G=%d C=%d B=%d nsplit=%d step %d hidden %d: attention + merge + o projection %6.1f us (three launches)  %6.1f us (merge inside the projection)"
          % (P, G, C, B, nsplit, step, hid, t(base), t(fused)))
i[0] = 0; a = base().float(); i[0] = 0; b = fused().float()
print("max |difference| of the two forms: %.4g (|c| max %.3g)" % ((a - b).abs().max().item(), a.abs().max().item()))
if os.environ.get("PROBE") == "1":
    import ctypes, numpy as np
    from time_r1_amd import hip
    nblk = hid // 16
    buf = torch.zeros(nblk * 2 * 8, dtype=torch.int64, device="cuda")
    assert hip.lib().cdll.probe_opf_set_ptr(ctypes.c_void_p(buf.data_ptr())) == 0
    fused(); torch.cuda.synchronize()
    hip.lib().cdll.probe_opf_set_ptr(ctypes.c_void_p(0))
    st = buf.cpu().view(nblk, 2, 8).numpy().astype(np.int64)
    for w, names in ((0, ["start", "dma_issued", "-", "-", "at_barrier", "past_barrier", "x_landed", "end"]), (1, ["start", "merged", "stored", "published", "swept", "past_barrier", "-", "end"])):
        for stat, f in (("median", np.median), ("max", np.max)):
            print("wave %s %s ticks since the block's start:" % ("stream0" if w == 0 else "merge", stat),
                  " ".join("%s=%d" % (n, int(f(st[:, w, j] - st[:, w, 0]))) for j, n in enumerate(names) if n != "-"))
