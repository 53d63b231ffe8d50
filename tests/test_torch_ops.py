"""`torch.ops.timer1.*`: dispatcher registration (CPU part) and numerics + autograd on the GPU against fp32 torch references, including the
ops dropped into a transformers Qwen2 decoder (patch_hf_model + the registered attention implementation)."""
import math

import pytest
import torch

import time_r1_amd  # noqa: F401
import time_r1_amd.torch_ops as T

BF16 = torch.bfloat16


def test_ops_are_registered_with_schemas_and_fail_loudly_on_cpu():
    for n in T.OP_NAMES:
        op = getattr(torch.ops.timer1, n)
        assert str(op.default._schema).startswith("timer1::" + n)
    assert "Tensor(a0!) p32" in str(torch.ops.timer1.adamw_step.default._schema)            # in-place ops declare their mutation
    x = torch.randn(4, 128).to(BF16)
    with pytest.raises(NotImplementedError):                                                   # HIP only: no CPU fallback behind the op
        T.rmsnorm(x, torch.ones(128, dtype=BF16))
    with pytest.raises(NotImplementedError):
        T.linear(x, torch.randn(64, 128).to(BF16))


def test_fake_kernels_give_shapes_for_tracing():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        x = torch.empty(2, 8, 128, dtype=BF16, device="cuda")
        w = torch.empty(128, dtype=BF16, device="cuda")
        assert T.rmsnorm(x, w).shape == (2, 8, 128)
        assert T.swiglu(torch.empty(16, 512, dtype=BF16, device="cuda")).shape == (16, 256)
        assert T.linear(x, torch.empty(192, 128, dtype=BF16, device="cuda")).shape == (2, 8, 192)
        q = torch.empty(16, 4 * 32, dtype=BF16, device="cuda")
        kv = torch.empty(16, 2 * 32, dtype=BF16, device="cuda")
        m = torch.empty(16, dtype=torch.int32, device="cuda")
        assert T.attention(q, kv, kv, m, m, m, 4, 2, 32).shape == (16, 128)


def _close(a, b, tol, what):
    err = float((a.float() - b.float()).abs().max())
    ref = float(b.float().abs().max())
    assert err <= tol * max(1.0, ref), (what, err, ref)


@pytest.mark.gpu
def test_op_numerics_and_gradients_vs_torch_fp32():
    g = torch.Generator(device="cuda").manual_seed(0)

    def rnd(*s, scale=1.0):
        return (torch.randn(*s, generator=g, device="cuda") * scale).to(BF16)
    # ---- rmsnorm
    x, w = rnd(3, 40, 256).requires_grad_(True), (1 + 0.1 * rnd(256).float()).to(BF16).requires_grad_(True)
    y = T.rmsnorm(x, w, 1e-6)
    xf, wf = x.detach().float().requires_grad_(True), w.detach().float().requires_grad_(True)
    yr = wf * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6))
    dy = rnd(3, 40, 256)
    y.backward(dy); yr.backward(dy.float())
    _close(y, yr, 0.02, "rmsnorm"); _close(x.grad, xf.grad, 0.02, "rmsnorm dx"); _close(w.grad, wf.grad, 0.03, "rmsnorm dw")
    # ---- swiglu
    gu = rnd(64, 512).requires_grad_(True)
    a = T.swiglu(gu)
    guf = gu.detach().float().requires_grad_(True)
    ar = torch.nn.functional.silu(guf[:, :256]) * guf[:, 256:]
    da = rnd(64, 256)
    a.backward(da); ar.backward(da.float())
    _close(a, ar, 0.02, "swiglu"); _close(gu.grad, guf.grad, 0.02, "swiglu bwd")
    # ---- linear (forward GEMM, NN dgrad / transposed dgrad for small shapes, fp32 wgrad, bias)
    for M, N, K in ((96, 192, 128), (1024, 512, 256)):
        x, w, b = rnd(M, K).requires_grad_(True), rnd(N, K, scale=1 / math.sqrt(K)).requires_grad_(True), rnd(N).requires_grad_(True)
        y = T.linear(x, w, b)
        xf, wf, bf = [t.detach().float().requires_grad_(True) for t in (x, w, b)]
        yr = xf @ wf.t() + bf
        dy = rnd(M, N, scale=0.1)
        y.backward(dy); yr.backward(dy.float())
        _close(y, yr, 0.02, "linear"); _close(x.grad, xf.grad, 0.02, "linear dx"); _close(w.grad, wf.grad, 0.02, "linear dw"); _close(b.grad, bf.grad, 0.02, "linear db")
    # ---- rope: gradient = inverse rotation
    T_, nh, hd = 50, 4, 64
    pos = torch.arange(T_, device="cuda", dtype=torch.int32)[None].repeat(3, 1).contiguous()
    cos, sin = torch.ops.timer1.mrope_table(pos, hd, 8, 12, 12, 10000.0)
    x = rnd(T_, nh * hd).requires_grad_(True)
    y = T.rope(x, cos, sin, nh, hd)
    xf = x.detach().float().view(T_, nh, hd).requires_grad_(True)
    c, s = torch.cat([cos, cos], -1)[:, None], torch.cat([sin, sin], -1)[:, None]
    rot = torch.cat([-xf[..., hd // 2:], xf[..., :hd // 2]], -1)
    yr = (xf * c.to(BF16).float() + rot * s.to(BF16).float()).reshape(T_, nh * hd)
    dy = rnd(T_, nh * hd)
    y.backward(dy); yr.backward(dy.float())
    _close(y, yr, 0.02, "rope"); _close(x.grad, xf.grad.reshape(T_, nh * hd), 0.02, "rope bwd")
    # ---- attention: causal GQA and varlen segments, forward + backward
    for causal in (True, False):
        Tn, H, KV, D = 200, 4, 2, 64
        q, k, v = rnd(Tn, H * D).requires_grad_(True), rnd(Tn, KV * D).requires_grad_(True), rnd(Tn, KV * D).requires_grad_(True)
        if causal:
            pre, lo, hi = T.causal_masks(Tn, "cuda")
        else:
            pre, lo, hi = T.varlen_masks(torch.tensor([0, 70, 71, 200], device="cuda"))
        o = T.attention(q, k, v, pre, lo, hi, H, KV, D)
        qf, kf, vf = [t.detach().float().requires_grad_(True) for t in (q, k, v)]
        kvi = torch.arange(Tn, device="cuda")[None]
        vis = (kvi < pre[:, None]) | ((kvi >= lo[:, None]) & (kvi <= hi[:, None]))
        qh = qf.view(Tn, H, D).transpose(0, 1)
        kh = kf.view(Tn, KV, D).transpose(0, 1).repeat_interleave(H // KV, 0)
        vh = vf.view(Tn, KV, D).transpose(0, 1).repeat_interleave(H // KV, 0)
        sc = (qh @ kh.transpose(1, 2)) * D ** -0.5
        orf = (torch.softmax(sc.masked_fill(~vis[None], float("-inf")), -1) @ vh).transpose(0, 1).reshape(Tn, H * D)
        do = rnd(Tn, H * D, scale=0.2)
        o.backward(do); orf.backward(do.float())
        _close(o, orf, 0.02, "attn"); _close(q.grad, qf.grad, 0.03, "attn dq"); _close(k.grad, kf.grad, 0.03, "attn dk"); _close(v.grad, vf.grad, 0.03, "attn dv")
    # ---- logp/entropy + grpo loss through autograd: d loss / d logits
    R, V, G = 24, 1000, 4
    logits = rnd(R, V, scale=2.0).requires_grad_(True)
    tg = torch.randint(0, V, (R,), device="cuda", generator=g)
    lp, ent = T.logp_entropy(logits, tg)
    adv = torch.tensor([0.5, -1.0, 1.5, -1.0], device="cuda")
    mask = torch.ones(G, R // G, dtype=torch.int32, device="cuda"); mask[1, 4:] = 0
    ref_lp = (lp.detach() + 0.1 * torch.randn(R, device="cuda", generator=g)).view(G, -1)
    loss, kl = T.grpo_loss(lp.view(G, -1), ref_lp, mask, adv, 0.04, True)
    loss.backward()
    lf = logits.detach().float().requires_grad_(True)
    lsm = torch.log_softmax(lf, -1)
    lpr = lsm.gather(1, tg[:, None].long())[:, 0].view(G, -1)
    klr = torch.exp(ref_lp - lpr) - (ref_lp - lpr) - 1
    ptl = -(torch.exp(lpr - lpr.detach()) * adv[:, None] - 0.04 * klr)
    mk = mask.float()
    lossr = ((ptl * mk).sum(1) / mk.sum(1)).mean()
    lossr.backward()
    assert abs(float(loss) - float(lossr)) < 1e-4 and float((ent - (-(lsm.exp() * lsm).sum(-1))).abs().max()) < 0.02
    _close(logits.grad, lf.grad, 0.02, "dlogits through grpo_loss + logp")


@pytest.mark.gpu
def test_ops_inside_a_transformers_qwen2_decoder():
    """The one-liner swap the integration doc promises: HF model (bf16, on the GPU) with RMSNorm / SwiGLU / Linear routed through timer1 ops
    and attention through the registered `timer1_hip` implementation == the unpatched eager model, forward logits and parameter gradients."""
    transformers = pytest.importorskip("transformers")
    from transformers import Qwen2Config, Qwen2ForCausalLM
    T.register_hf_attention()
    cfg = Qwen2Config(vocab_size=512, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                      max_position_embeddings=512, rms_norm_eps=1e-6, tie_word_embeddings=False)
    torch.manual_seed(0)
    base = Qwen2ForCausalLM(cfg).to("cuda", BF16)
    base.config._attn_implementation = "eager"
    ids = torch.randint(0, 512, (2, 64), device="cuda")
    out = {}
    for name in ("eager", "timer1"):
        m = Qwen2ForCausalLM(cfg).to("cuda", BF16)
        m.load_state_dict(base.state_dict())
        if name == "timer1":
            m.config._attn_implementation = "timer1_hip"
            counts = T.patch_hf_model(m, linears=True)
            assert counts["rmsnorm"] == 5 and counts["mlp"] == 2 and counts["linear"] >= 8
        else:
            m.config._attn_implementation = "eager"
        logits = m(input_ids=ids).logits
        loss = torch.nn.functional.cross_entropy(logits[:, :-1].float().reshape(-1, 512), ids[:, 1:].reshape(-1))
        loss.backward()
        out[name] = (logits.float(), {k: p.grad.float() for k, p in m.named_parameters()})
    _close(out["timer1"][0], out["eager"][0], 0.03, "logits")
    for k, gr in out["eager"][1].items():
        rel = float((out["timer1"][1][k] - gr).norm() / gr.norm().clamp(min=1e-12))
        assert rel < 0.08, (k, rel)


@pytest.mark.gpu
def test_hf_attention_seam_honours_is_causal_cu_seqlens_and_kv_cache():
    """transformers routes the Qwen2-VL vision blocks through the same registered function with is_causal=False and
    flash-style cu_seq_lens; `generate` calls it with S > T.  Each form against an fp32 softmax reference; a padding mask is refused."""
    import types
    g = torch.Generator(device="cuda").manual_seed(3)
    B, H, KV, D = 1, 4, 2, 64

    def ref(q, k, v, vis):
        kh, vh = k[0].float().repeat_interleave(H // KV, 0), v[0].float().repeat_interleave(H // KV, 0)
        sc = (q[0].float() @ kh.transpose(1, 2)) * D ** -0.5
        return (torch.softmax(sc.masked_fill(~vis[None], float("-inf")), -1) @ vh).transpose(0, 1)     # [T, H, D]

    def mk(T_, S_):
        return [(torch.randn(B, h, n, D, device="cuda", generator=g) * 0.5).to(BF16) for h, n in ((H, T_), (KV, S_), (KV, S_))]
    mod = types.SimpleNamespace(is_causal=True)
    # (1) vision form: bidirectional inside cu_seqlens segments
    Tn = 96
    q, k, v = mk(Tn, Tn)
    cu = torch.tensor([0, 40, 64, 96], device="cuda", dtype=torch.int32)
    o, _ = T.hf_attention_forward(mod, q, k, v, None, scaling=D ** -0.5, is_causal=False, cu_seq_lens_q=cu, cu_seq_lens_k=cu)
    seg = torch.searchsorted(cu[1:].long(), torch.arange(Tn, device="cuda"), right=True)
    _close(o[0], ref(q, k, v, seg[:, None] == seg[None, :]), 0.02, "vision varlen")
    # (2) is_causal=False without segments: full attention; module default is_causal is NOT used when the kwarg says False
    o, _ = T.hf_attention_forward(mod, q, k, v, None, scaling=D ** -0.5, is_causal=False)
    _close(o[0], ref(q, k, v, torch.ones(Tn, Tn, dtype=torch.bool, device="cuda")), 0.02, "full")
    # (3) generate with a KV cache: T new queries are the last T of S keys
    Tq, S = 16, 80
    q, k, v = mk(Tq, S)
    o, _ = T.hf_attention_forward(mod, q, k, v, None, scaling=D ** -0.5)
    vis = torch.arange(S, device="cuda")[None, :] <= (S - Tq + torch.arange(Tq, device="cuda"))[:, None]
    _close(o[0], ref(q, k, v, vis), 0.02, "suffix queries")
    # (4) the plain causal float mask transformers builds is accepted, a left-padding mask is refused loudly
    q, k, v = mk(32, 32)
    tri = torch.zeros(1, 1, 32, 32, device="cuda").masked_fill(~torch.ones(32, 32, dtype=torch.bool, device="cuda").tril(), float("-inf"))
    T.hf_attention_forward(mod, q, k, v, tri, scaling=D ** -0.5)
    pad = tri.clone(); pad[..., :4] = float("-inf")
    with pytest.raises(NotImplementedError):
        T.hf_attention_forward(mod, q, k, v, pad, scaling=D ** -0.5)
    # (5) ADVICE r4: an explicit ALL-KEEP mask over T > 1 queries replaces is_causal in transformers' sdpa path = bidirectional attention
    allkeep = torch.zeros(1, 1, 32, 32, device="cuda")
    o, _ = T.hf_attention_forward(mod, q, k, v, allkeep, scaling=D ** -0.5)
    _close(o[0], ref(q, k, v, torch.ones(32, 32, dtype=torch.bool, device="cuda")), 0.02, "explicit all-keep mask")
