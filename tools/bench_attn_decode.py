"""Split-KV decode attention (+ combine) per layer at a given rollout step: B prompts x G rows over [prefix | G x C suffix slots] caches, rotating
over several cache copies so the K / V^T tiles come from HBM (one decoder layer's cache is never re-read within a step).
   python tools/bench_attn_decode.py [P G C B]      (default: config 4 = 3266 16 1024 2; config 3 = 3474 8 200 2)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time_r1_amd  # noqa
from time_r1_amd.ops import HipOps
ops = HipOps("cuda:0")
P, G, C, B = (3266, 16, 1024, 2) if len(sys.argv) < 5 else map(int, sys.argv[1:5])
nh, nkv, hd = int(os.environ.get("NH", 28)), int(os.environ.get("NKV", 4)), 128      # NH=12 NKV=2: Qwen2-VL-2B (config 2)
S = P + G * C
scap = (S + 63) // 64 * 64
NC = 12
ks = [(torch.randn(B * scap, nkv * hd, device="cuda") * 0.5).bfloat16() for _ in range(NC)]
vts = [(torch.randn(nkv * hd, B * scap, device="cuda") * 0.5).bfloat16() for _ in range(NC)]
q = (torch.randn(B * G, nh * hd, device="cuda") * 0.5).bfloat16()
nsplit = int(os.environ.get("TR1_DECODE_NSPLIT", max(1, min(28, ((P + 63) // 64 + 3) // 2))))
for step in ([int(s) for s in os.environ["STEPS"].split(",")] if os.environ.get("STEPS") else (1, C // 4, C // 2, C - 1)):
    pre = torch.full((B * G,), P, dtype=torch.int32, device="cuda")
    lo = (P + torch.arange(G) * C).int().repeat(B).cuda()
    hi = (lo + step).int()
    i = [0]
    plan = ops.attn_plan(G, nh, nkv, B) if os.environ.get("PLAN") == "1" else None      # PLAN=1: the plan-reading launches of layers 1.. of a decode step
    if plan is not None:
        ops.attn_fwd(q, ks[0], vts[0], pre, lo, hi, nh, nkv, scap, hd, hd ** -0.5, nsplit=nsplit, need_lse=False, n_batch=B, kv_batch_slots=scap, plan=plan, plan_mode=1)
    def run():
        i[0] = (i[0] + 1) % NC
        kw = dict(plan=plan, plan_mode=2) if plan is not None else {}
        return ops.attn_fwd(q, ks[i[0]], vts[i[0]], pre, lo, hi, nh, nkv, scap, hd, hd ** -0.5, nsplit=nsplit, need_lse=False, n_batch=B, kv_batch_slots=scap, **kw)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    kvb = B * (P + G * (step + 1)) * 2 * nkv * hd * 2
    print("P=%d G=%d C=%d B=%d nsplit=%d step %4d: %6.1f us per layer (attention + combine), visible KV %5.1f MB -> %5.0f GB/s" % (P, G, C, B, nsplit, step, us, kvb / 1e6, kvb / us / 1e3))

if os.environ.get("PROBE") == "1":      # TR1_HIP_LIB=tools/_probe_lib.so PLAN=1 PROBE=1: block timelines of attn_dec32_kernel (s_memtime stamps)
    import ctypes
    from time_r1_amd import hip
    nblk = 1 * (nkv * B) * nsplit
    buf = torch.zeros(nblk * 4 * 12, dtype=torch.int64, device="cuda")
    assert hip.lib().cdll.probe_dec_set_ptr(ctypes.c_void_p(buf.data_ptr())) == 0
    run(); torch.cuda.synchronize()
    hip.lib().cdll.probe_dec_set_ptr(ctypes.c_void_p(0))
    st = buf.cpu().view(nblk, 4, 12).numpy()
    t0 = st[st > 0].min()
    names = ["start", "spec_dma", "q_loads", "plan_n", "late_dma", "landed", "tile0", "tile1", "tile2", "loop_end", "stored"]
    import numpy as np
    for w in range(4):
        rel = np.where(st[:, w, :11] > 0, st[:, w, :11] - st[:, w, :1], -1)
        print("wave %d median ticks since block start:" % w, " ".join("%s=%d" % (n, int(np.median(rel[:, i]))) for i, n in enumerate(names)))
    print("block start spread (ticks): min %d median %d max %d" % (0, int(np.median(st[:, 0, 0] - t0)), int((st[:, 0, 0] - t0).max())),
          " block end max:", int((st[:, :, 10].max() - t0)))
