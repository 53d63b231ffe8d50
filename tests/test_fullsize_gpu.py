"""Parity at BASELINE.json's full sizes (Qwen2-VL-7B widths, config[1]: P = 3474 prompt tokens, G = 8, C = 200).

The CPU oracle cannot run these shapes in seconds, so each kernel family is checked against a plain torch fp32 computation ON THE GPU of
the same op (floating-point kernels: tolerance written next to each check) and through size-independent properties:
linearity of the GEMM, the packed training attention == the replicated causal attention, decode logits == training-forward logits.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
P, G, C = 3474, 8, 200
H, NKV, HD, HID, INTER = 28, 4, 128, 3584, 18944


def dev_rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(*shape, generator=g, device="cuda") * scale).to(BF16)


def rel_l2(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp(min=1e-20))


@pytest.mark.parametrize("M,N,K", [(P + G * C, 2 * INTER, HID), (P + G * C, HID, INTER), (P + G * C, H * HD + 2 * NKV * HD, HID), (G * C, 152064, HID)])
def test_gemm_full_shapes_vs_torch_fp32(hip_ops, M, N, K):
    """Training-forward GEMM shapes of the 7B step (256x256-tile and 128x128-tile paths).  bf16 output of an fp32 accumulation:
    per-element error <= 1 bf16 ulp of the result (2^-8 relative) + accumulation-order noise; checked as relative L2 < 3e-3."""
    a, b = dev_rnd(M, K, seed=1), dev_rnd(N, K, seed=2, scale=1.0 / math.sqrt(K))
    out = hip_ops.gemm_nt(a, b)
    ref = a.float() @ b.float().t()
    assert rel_l2(out, ref) < 3e-3
    # linearity (size-independent property): gemm(a1 + a2) == gemm(a1) + gemm(a2) up to bf16 rounding of the three outputs
    a2 = dev_rnd(M, K, seed=3)
    s = hip_ops.gemm_nt((a.float() + a2.float()).to(BF16), b).float()
    t = out.float() + hip_ops.gemm_nt(a2, b).float()
    assert rel_l2(s, t) < 8e-3


def test_wgrad_accumulate_full_shape_vs_torch_fp32(hip_ops):
    """dW[N, K] (fp32) += dy[M, N]^T x[M, K] at the gate/up shape, two accumulation passes (gradient accumulation)."""
    M, N, K = P + G * C, HID, 2 * INTER // 4          # a quarter of the gate/up rows keeps the fp32 reference at 270 MB
    dy, x = dev_rnd(M, N, seed=4, scale=0.05), dev_rnd(M, K, seed=5)
    out = torch.zeros(N, K, device="cuda")
    for _ in range(2):
        hip_ops.gemm_nt(hip_ops.transpose(dy), hip_ops.transpose(x), out_f32=True, out=out, accumulate=True)
    ref = 2.0 * (dy.float().t() @ x.float())
    assert rel_l2(out, ref) < 2e-3


@pytest.mark.parametrize("M,N,K", [(P + G * C, HID, INTER), (P + G * C, 2 * INTER // 4, HID), (1600, H * HD + 2 * NKV * HD, HID), (700, 512, 256)])
def test_wgrad_k_major_form_reads_activation_as_stored(hip_ops, M, N, K):
    """dW[N, K] (fp32) (+)= dy[M, N]^T x[M, K] with x read AS STORED (K-major B operand, transposing LDS reads): equal to the transposed-copy
    path bit for bit (same tiles, same accumulation order), right against fp32 torch, for token counts that are not multiples of 64 (the
    rows past M are re-reads of the last row against the zero padding of dy^T) and with x an exact-size allocation (nothing readable behind it)."""
    dy, x = dev_rnd(M, N, seed=4, scale=0.05), dev_rnd(M, K, seed=5)
    dyt = hip_ops.transpose(dy)
    out = torch.zeros(N, K, device="cuda")
    assert hip_ops.wgrad_nn(dyt, x, out, accumulate=False)
    assert hip_ops.wgrad_nn(dyt, x, out, accumulate=True)
    want = torch.zeros(N, K, device="cuda")
    for acc in (False, True):
        hip_ops.gemm_nt(dyt, hip_ops.transpose(x), out_f32=True, out=want, accumulate=acc)
    assert torch.equal(out, want)
    ref = 2.0 * (dy.float().t() @ x.float())
    assert rel_l2(out, ref) < 2e-3


def _replicated_attention(q, k, v, pre, lo, hi, scale):
    """fp32 reference of the two-interval mask on the GPU: softmax over visible keys, GQA by head repetition."""
    T, S = q.shape[0], k.shape[0]
    kv = torch.arange(S, device="cuda")[None, :]
    vis = (kv < pre[:, None]) | ((kv >= lo[:, None]) & (kv <= hi[:, None]))
    qh = q.float().view(T, H, HD).transpose(0, 1)
    kh = k.float().view(S, NKV, HD).transpose(0, 1).repeat_interleave(H // NKV, 0)
    vh = v.float().view(S, NKV, HD).transpose(0, 1).repeat_interleave(H // NKV, 0)
    out = torch.empty(H, T, HD, device="cuda")
    for h0 in range(0, H, 4):       # 4 heads at a time: [4, 5074, 5074] fp32 scores = 412 MB
        s = (qh[h0:h0 + 4] @ kh[h0:h0 + 4].transpose(1, 2)) * scale
        s = s.masked_fill(~vis[None], float("-inf"))
        out[h0:h0 + 4] = torch.softmax(s, -1) @ vh[h0:h0 + 4]
    return out.transpose(0, 1).reshape(T, H * HD)


def test_packed_attention_fwd_bwd_full_size(hip_ops):
    """Shared-prefix packed attention (P + G*C = 5074 rows) forward and backward vs the masked fp32 attention, and the property the
    packing relies on: completion rows equal the rows of an ordinary causal attention over [prompt | own completion]."""
    from time_r1_amd.positions import PackedLayout
    lay = PackedLayout(P, G, C)
    M = lay.M
    pre, lo, hi = [torch.tensor(a).cuda() for a in lay.masks()]
    q, k, v = dev_rnd(M, H * HD, seed=1), dev_rnd(M, NKV * HD, seed=2), dev_rnd(M, NKV * HD, seed=3)
    do = dev_rnd(M, H * HD, seed=4, scale=0.1)
    scale = HD ** -0.5
    vt = hip_ops.pack_transpose(v, NKV, NKV, HD)
    o, lse = hip_ops.attn_fwd(q, k, vt, pre, lo, hi, H, NKV, M, HD, scale)
    qf, kf, vf = [t.float().requires_grad_(True) for t in (q, k, v)]
    ref = _replicated_attention(qf, kf, vf, pre, lo, hi, scale)
    assert rel_l2(o, ref.detach()) < 6e-3                      # bf16 P and O rounding
    ref.backward(do.float())
    dq, dk, dv = hip_ops.attn_bwd(q, k, v, o, do, lse, pre, lo, hi, H, NKV, M, HD, scale)
    assert rel_l2(dq, qf.grad) < 1.5e-2 and rel_l2(dk, kf.grad) < 1.5e-2 and rel_l2(dv, vf.grad) < 1.5e-2
    # property: group g's rows == causal attention over the sequence [prompt, completion g]
    g = 5
    rows = torch.cat([torch.arange(P), P + g * C + torch.arange(C)]).cuda()
    L = rows.numel()
    z = torch.zeros(L, dtype=torch.int32, device="cuda")
    ar = torch.arange(L, dtype=torch.int32, device="cuda")
    k_g, v_g = k[rows].contiguous(), v[rows].contiguous()
    o_g, _ = hip_ops.attn_fwd(q[rows].contiguous(), k_g, hip_ops.pack_transpose(v_g, NKV, NKV, HD), z, z, ar, H, NKV, L, HD, scale)
    assert rel_l2(o_g[P:], o[P + g * C: P + (g + 1) * C]) < 4e-3


def test_decode_attention_matches_training_rows_full_size(hip_ops):
    """Split-KV decode over the cache (the G new tokens of a step) == the same rows of the packed training attention."""
    from time_r1_amd.positions import PackedLayout
    lay = PackedLayout(P, G, C)
    M, S = lay.M, lay.S_cap
    pre, lo, hi = [torch.tensor(a).cuda() for a in lay.masks()]
    q, k, v = dev_rnd(M, H * HD, seed=11), dev_rnd(S, NKV * HD, seed=12), dev_rnd(S, NKV * HD, seed=13)
    scale = HD ** -0.5
    vt = hip_ops.pack_transpose(v, NKV, NKV, HD)
    o_full, _ = hip_ops.attn_fwd(q, k[:M].contiguous(), hip_ops.pack_transpose(v[:M].contiguous(), NKV, NKV, HD), pre, lo, hi, H, NKV, M, HD, scale)
    for step in (0, 63, 199):
        rows = torch.tensor(lay.completion_slots(step)).long().cuda()
        pd, ld, hd_ = [torch.tensor(a).cuda() for a in lay.decode_masks(step)]
        for nsplit in (8, 28):
            o_dec, _ = hip_ops.attn_fwd(q[rows].contiguous(), k, vt, pd, ld, hd_, H, NKV, S, HD, scale, nsplit=nsplit, need_lse=False)
            assert rel_l2(o_dec, o_full[rows]) < 4e-3, (step, nsplit)


def test_decode_step_logits_match_training_forward_full_width(hip_ops):
    """Teacher-forced property at 7B width (2 decoder layers, 1 ViT block, full prompt): the logits the sampler sees at every decode step
    equal the packed training forward's logits for the same tokens (fused decode kernels vs training kernels), bf16 tolerance."""
    import time_r1_amd  # noqa: F401
    from time_r1_amd.config import qwen2_vl_7b
    from time_r1_amd.params import ModelParams
    from time_r1_amd.model import Engine
    from time_r1_amd.grpo import GRPOCore
    from time_r1_amd.synthetic import synthetic_prompt
    cfg = qwen2_vl_7b()
    cfg.text.n_layers, cfg.vision.depth, cfg.text.vocab_size = 2, 1, 32768
    ops = hip_ops
    params = ModelParams(cfg, ops, init="none", optimizer_state=False)
    params.init_random_device(3)
    eng = Engine(cfg, ops, params)
    Cs = 6
    core = GRPOCore(eng, None, G, Cs, beta=0.0, seed=3, rope_index_mode="hf4")
    ids, pix, grid = synthetic_prompt(cfg, (16, 22, 38), 20, 30, seed=2, text_vocab=30000)
    rec = []
    orig = ops.sample_tokens

    def spy(logits, *a, **k):
        rec.append(logits.float().clone())
        return orig(logits, *a, **k)
    ops.sample_tokens = spy
    try:
        st = core.prepare(ids, pix, grid)
        core.rollout(st)
    finally:
        ops.sample_tokens = orig
    core.forward_logps(st)
    hl = st.head_ctx["logits"].float()
    scale = float(hl.abs().max())
    assert torch.allclose(hl[:G], rec[0][0][None].expand(G, -1), atol=0.02 * scale, rtol=0.03)
    for s in range(1, Cs):
        rows = torch.tensor([G + g * (Cs - 1) + (s - 1) for g in range(G)]).cuda()
        assert torch.allclose(hl[rows], rec[s], atol=0.02 * scale, rtol=0.03), "decode step %d" % s


class _unfused_ops:
    def __init__(self, ops):
        self.ops = ops

    def __enter__(self):
        self.ops.FUSE_EPI = False
        return self.ops

    def __exit__(self, *a):
        del self.ops.FUSE_EPI


def test_fused_epilogue_gemms_at_full_7b_shapes(hip_ops):
    """Round 4: the training GEMMs that carry their elementwise neighbours in the epilogue, at the config-3 shapes (M = P + G*C = 5074 packed rows, hidden 3584,
    intermediate 18944, 28 / 4 heads of 128; ViT 13376 tokens x 16 heads of 80): bit-identical to GEMM + elementwise kernel, and right against fp32 torch."""
    M = P + G * C
    x = dev_rnd(M, HID, seed=1)
    # gate/up + SwiGLU
    wgu = dev_rnd(2 * INTER, HID, seed=2, scale=1.0 / math.sqrt(HID))
    a, gu = hip_ops.gemm_glu(x, wgu, save_gu=True)
    with _unfused_ops(hip_ops) as o:
        a0, gu0 = o.gemm_glu(x, wgu, save_gu=True)
    assert torch.equal(a, a0) and torch.equal(gu, gu0)
    g32 = x.float() @ wgu.float().t()
    ref = torch.nn.functional.silu(g32[:, :INTER]) * g32[:, INTER:]
    assert rel_l2(a, ref) < 8e-3
    del g32, ref
    # down-projection dgrad + SwiGLU backward
    dh, wd = dev_rnd(M, HID, seed=3, scale=0.05), dev_rnd(HID, INTER, seed=4, scale=1.0 / math.sqrt(INTER))
    dgu, dgut = hip_ops.dgrad_glu_bwd(dh, wd, gu, want_t=True)
    with _unfused_ops(hip_ops) as o:
        dgu0 = o.dgrad_glu_bwd(dh, wd, gu)
    assert torch.equal(dgu, dgu0)
    assert dgut is not None and torch.equal(dgut, hip_ops.transpose(dgu0)), "dgu^T from the dgrad epilogue (the gate/up weight gradient's operand)"
    del dgu, dgu0, dgut, a, a0, gu0
    # q|k|v + bias + M-RoPE, k into a cache buffer at a row offset
    qd, kvd = H * HD, NKV * HD
    wq, bq = dev_rnd(qd + 2 * kvd, HID, seed=5, scale=1.0 / math.sqrt(HID)), dev_rnd(qd + 2 * kvd, seed=6)
    ang = torch.rand(M, HD // 2, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)) * 6.28
    cos, sin = torch.cos(ang).to(BF16).float(), torch.sin(ang).to(BF16).float()
    kc = torch.zeros(M + 64, kvd, dtype=BF16, device="cuda")
    q, k, v = hip_ops.gemm_qkv_rope(x, wq, bq, cos, sin, H, NKV, HD, k_out=kc[:M])
    with _unfused_ops(hip_ops) as o:
        q0, k0, v0 = o.gemm_qkv_rope(x, wq, bq, cos, sin, H, NKV, HD)
    assert torch.equal(q, q0) and torch.equal(k, k0) and torch.equal(v, v0) and float(kc[M:].abs().max()) == 0.0
    # vision q|k|v + bias + 2-D rotary on 128-wide padded heads
    Nv, Hv, Ev = 13376, 16, 1280
    xv, wv, bv = dev_rnd(Nv, Ev, seed=8), dev_rnd(3 * Ev, Ev, seed=9, scale=1.0 / math.sqrt(Ev)), dev_rnd(3 * Ev, seed=10)
    angv = torch.rand(Nv, 40, device="cuda", generator=torch.Generator(device="cuda").manual_seed(11)) * 6.28
    cv, sv = torch.cos(angv).contiguous(), torch.sin(angv).contiguous()
    bufs = [torch.zeros(Nv, Hv * 128, dtype=BF16, device="cuda") for _ in range(3)]
    hip_ops.gemm_qkv_rope_vit(xv, wv, bv, cv, sv, Hv, 40, *bufs)
    with _unfused_ops(hip_ops) as o:
        qkv = o.gemm_nt(xv, wv, bias=bv)
        want = [o.rope_apply(qkv[:, :Ev], Hv, 80, cv, sv), o.rope_apply(qkv[:, Ev:2 * Ev], Hv, 80, cv, sv), qkv[:, 2 * Ev:]]
    for got, w_ in zip(bufs, want):
        g3, w3 = got.view(Nv, Hv, 128), w_.reshape(Nv, Hv, 80)
        assert torch.equal(g3[:, :, :40], w3[:, :, :40]) and torch.equal(g3[:, :, 48:88], w3[:, :, 40:])
        assert float(g3[:, :, 40:48].abs().max()) == 0.0 and float(g3[:, :, 88:].abs().max()) == 0.0


def test_split_k_and_lm_head_dgrad_at_full_7b_shapes(hip_ops):
    """Round 4: the continuation forward's down projection (1600 x 3584 x 18944) and the lm_head's data gradient (1600 x 3584 over K = 152064, weight as stored)
    on the deterministic split-K forms, against fp32 torch."""
    R = G * C
    a, w, res = dev_rnd(R, INTER, seed=1), dev_rnd(HID, INTER, seed=2, scale=1.0 / math.sqrt(INTER)), dev_rnd(R, HID, seed=3)
    assert hip_ops._splitk_ok(R, HID, INTER)
    y = hip_ops.gemm_nt(a, w, residual=res)
    assert rel_l2(y, a.float() @ w.float().t() + res.float()) < 3e-3
    V = 152064
    dl, wl = dev_rnd(R, V, seed=4, scale=0.01), dev_rnd(V, HID, seed=5, scale=0.02)
    assert hip_ops._splitk_ok(R, HID, V)
    d = hip_ops.gemm_nn(dl, wl)
    assert rel_l2(d, dl.float() @ wl.float()) < 3e-3
    assert torch.equal(d, hip_ops.gemm_nn(dl, wl)), "deterministic"
