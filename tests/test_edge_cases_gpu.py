"""Edge cases through the C ABI on the GPU: empty inputs, single rows / single keys, ragged tails, argument checks that must fail
loudly (no silent fallback), extreme sampler settings.  Checked against the CPU oracle where there is something to compare."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
I32 = torch.int32


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF16)


def test_empty_inputs_are_noops(hip_ops):
    ops = hip_ops
    w = rnd(64, 128).cuda()
    assert ops.gemm_nt(torch.empty(0, 128, dtype=BF16, device="cuda"), w).shape == (0, 64)
    assert ops.gather_rows(w, torch.empty(0, dtype=I32, device="cuda")).shape == (0, 128)
    y, rstd, _ = ops.rmsnorm_fwd(torch.empty(0, 128, dtype=BF16, device="cuda"), rnd(128).cuda(), 1e-6)
    assert y.shape == (0, 128) and rstd.shape == (0,)
    z = torch.empty(0, dtype=I32, device="cuda")
    o, lse = ops.attn_fwd(torch.empty(0, 64, dtype=BF16, device="cuda"), rnd(64, 32).cuda(), ops.pack_transpose(rnd(64, 32).cuda(), 1, 1, 32), z, z, z,
                          2, 1, 64, 32, 0.17)
    assert o.shape == (0, 64)
    torch.cuda.synchronize()


def test_argument_checks_fail_loudly(hip_ops):
    from time_r1_amd.hip import HipError
    ops = hip_ops
    with pytest.raises(HipError, match="multiple of 64"):
        ops.gemm_nt(rnd(4, 100).cuda(), rnd(8, 100).cuda())                      # K not a multiple of 64
    with pytest.raises(HipError, match="head dim"):
        z = torch.zeros(4, dtype=I32, device="cuda")
        ops.attn_fwd(rnd(4, 2 * 20).cuda(), rnd(64, 20).cuda(), rnd(20, 64).cuda(), z, z, z, 2, 1, 64, 20, 0.2)     # head_dim 20: not a multiple of 8
    with pytest.raises(HipError, match="decode rows"):
        ops.norm_gemm(rnd(65, 128).cuda(), rnd(128).cuda(), 1e-6, rnd(64, 128).cuda())     # fused decode GEMM: at most 64 rows
    with pytest.raises(AssertionError):
        ops.gemm_nt(rnd(4, 128), rnd(8, 128))                                    # CPU tensors: the product path has no CPU fallback


@pytest.mark.parametrize("T,S", [(1, 1), (1, 63), (1, 64), (1, 65), (3, 130), (70, 70)])
def test_attention_tiny_and_ragged_shapes(hip_ops, ref_ops, T, S):
    """Single query / single key, sequence ends on and around the 64-key tile boundary (stale cache slots past S must not leak)."""
    nh, nkv, hd = 4, 2, 32
    q, k, v = rnd(T, nh * hd, seed=1), rnd(S, nkv * hd, seed=2), rnd(S, nkv * hd, seed=3)
    Scap = (S + 63) // 64 * 64
    kc = torch.full((Scap, nkv * hd), 77.0, dtype=BF16)       # poison beyond S
    vtc = torch.full((nkv * hd, Scap), 77.0, dtype=BF16)
    kc[:S] = k
    vtc[:, :S] = ref_ops.pack_transpose(v.float(), nkv, nkv, hd).to(BF16)[:, :S]
    pre = torch.zeros(T, dtype=I32)
    lo = torch.zeros(T, dtype=I32)
    hi = torch.tensor([min(S - 1, S - T + t) for t in range(T)], dtype=I32)      # causal tail
    o, lse = hip_ops.attn_fwd(q.cuda(), kc.cuda(), vtc.cuda(), pre.cuda(), lo.cuda(), hi.cuda(), nh, nkv, S, hd, hd ** -0.5)
    ro, rlse = ref_ops.attn_fwd(q.float(), k.float(), ref_ops.pack_transpose(v.float(), nkv, nkv, hd), pre, lo, hi, nh, nkv, S, hd, hd ** -0.5)
    assert torch.isfinite(o.float()).all()
    assert (o.float().cpu() - ro.float()).abs().max() < 0.03
    assert (lse.cpu() - rlse).abs().max() < 0.02
    for nsplit in (2, 5):      # split-KV with more splits than tiles: empty splits must contribute nothing
        o2, _ = hip_ops.attn_fwd(q.cuda(), kc.cuda(), vtc.cuda(), pre.cuda(), lo.cuda(), hi.cuda(), nh, nkv, S, hd, hd ** -0.5, nsplit=nsplit, need_lse=False)
        assert (o2.float().cpu() - ro.float()).abs().max() < 0.03, nsplit


def test_attention_row_with_no_visible_key_gives_zero(hip_ops):
    """pre = 0 and an empty interval (lo > hi): the row's output is 0 and its lse is -inf (never NaN)."""
    nh, nkv, hd, S = 2, 1, 32, 64
    q, k, v = rnd(2, nh * hd, seed=1).cuda(), rnd(S, nkv * hd, seed=2).cuda(), rnd(S, nkv * hd, seed=3).cuda()
    pre = torch.tensor([0, 0], dtype=I32).cuda()
    lo = torch.tensor([5, 0], dtype=I32).cuda()
    hi = torch.tensor([4, 9], dtype=I32).cuda()          # row 0: empty
    o, lse = hip_ops.attn_fwd(q, k, hip_ops.pack_transpose(v, nkv, nkv, hd), pre, lo, hi, nh, nkv, S, hd, 0.2)
    assert torch.equal(o[0].float(), torch.zeros(nh * hd, device="cuda")) and torch.isinf(lse[:, 0]).all() and (lse[:, 0] < 0).all()
    assert torch.isfinite(o[1].float()).all() and torch.isfinite(lse[:, 1]).all()


def test_sampler_extremes(hip_ops):
    """top_k = 1 is greedy; top_k = 0 (disabled) samples from the full softmax; a tiny temperature is greedy; finished rows emit pad."""
    ops = hip_ops
    V, G = 1024, 8
    logits = rnd(G, V, seed=5, scale=3.0).cuda()
    steps = torch.zeros(1, dtype=I32, device="cuda")
    greedy = logits.float().argmax(-1).int()

    def run(temp, top_k, fin=None, stop=False, seed=3):
        toks = torch.zeros(G, 4, dtype=I32, device="cuda")
        finished = fin.clone() if fin is not None else torch.zeros(G, dtype=I32, device="cuda")
        ops.sample_tokens(logits, temp, top_k, seed, steps, toks, finished, 1, 0, stop)
        return toks[:, 0], finished
    t, _ = run(1.0, 1)
    assert torch.equal(t, greedy)
    t, _ = run(1e-4, 50)
    assert torch.equal(t, greedy)
    t, _ = run(1.0, 0)
    assert (t >= 0).all() and (t < V).all()
    fin = torch.tensor([1, 0, 1, 0, 0, 0, 0, 0], dtype=I32, device="cuda")
    t, f2 = run(1.0, 50, fin=fin, stop=True)
    assert t[0] == 0 and t[2] == 0                      # already finished rows emit the pad id
    # forcing EOS everywhere: every live row finishes
    eos_logits = torch.full((G, V), -30.0, dtype=BF16, device="cuda")
    eos_logits[:, 1] = 30.0
    toks = torch.zeros(G, 4, dtype=I32, device="cuda")
    finished = torch.zeros(G, dtype=I32, device="cuda")
    ops.sample_tokens(eos_logits, 1.0, 50, 3, steps, toks, finished, 1, 0, True)
    assert (toks[:, 0] == 1).all() and (finished == 1).all()


@pytest.mark.parametrize("G,C", [(1, 1), (1, 5), (3, 2)])
def test_rollout_and_update_degenerate_group_sizes(hip_ops, G, C):
    """One completion per prompt / one generated token: the packed layout, decode loop, logprob head and backward still line up
    (for G = 1 the group statistics degenerate exactly as in the reference)."""
    import time_r1_amd  # noqa: F401
    from time_r1_amd.config import tiny_test
    from time_r1_amd.params import ModelParams
    from time_r1_amd.model import Engine
    from time_r1_amd.grpo import GRPOCore, eos_mask, group_advantages
    from time_r1_amd.synthetic import synthetic_prompt
    cfg = tiny_test(n_layers=2)
    ops = hip_ops
    params = ModelParams(cfg, ops, seed=4)
    eng = Engine(cfg, ops, params)
    core = GRPOCore(eng, params.train.clone_weights_only(), G, C, beta=0.04, seed=9, rope_index_mode="hf4")
    ids, pix, grid = synthetic_prompt(cfg, (2, 4, 6), 5, 6, seed=1, text_vocab=400)
    st = core.prepare(ids, pix, grid)
    core.rollout(st)
    assert st.completion_ids.shape == (G, C)
    core.forward_logps(st)
    assert st.logp.shape == (G, C) and torch.isfinite(st.logp).all() and (st.logp <= 0).all() and torch.isfinite(st.ref_logp).all()
    mask = ops.tensor(eos_mask(st.completion_ids.cpu().numpy(), cfg.eos_token_id).astype(np.int32), I32)
    _, adv, _ = group_advantages(torch.arange(G, dtype=torch.float32)[:, None] * 0.5, G)
    if G == 1:          # unbiased std of one sample is NaN in the reference too (timer1_trainer.py:706): a group of one carries no signal
        assert torch.isnan(adv).all()
        adv = torch.zeros(G)
    out3, row_len = core.loss_backward(st, mask, adv.to(ops.device))
    assert torch.isfinite(out3).all() and torch.isfinite(params.train.grad).all()


@pytest.mark.parametrize("T,nh,nkv,empty_row", [(1, 2, 2, None), (70, 4, 2, 3), (129, 14, 2, None), (64, 2, 1, 0)])
def test_attention_backward_head_dim_128_edges(hip_ops, ref_ops, T, nh, nkv, empty_row):
    """The LDS-DMA staged dK/dV form (head dim 128): a single query / key, a query-tile count that leaves a ragged last tile (rows past the
    end are clamped re-reads that must drop out), group size 1 and 7, and a row that sees no key at all (its LSE is -inf: the log2-scaled copy
    carries +inf so p = 0, never NaN)."""
    hd, S = 128, T
    q, k, v, do = rnd(T, nh * hd, seed=1), rnd(S, nkv * hd, seed=2), rnd(S, nkv * hd, seed=3), rnd(T, nh * hd, seed=4, scale=0.3)
    pre = torch.zeros(T, dtype=I32)
    lo = torch.zeros(T, dtype=I32)
    hi = torch.arange(T, dtype=I32)
    if empty_row is not None:
        lo[empty_row], hi[empty_row] = 5, 4
    scale = hd ** -0.5
    o_r, lse_r = ref_ops.attn_fwd(q.float(), k.float(), ref_ops.pack_transpose(v.float(), nkv, nkv, hd), pre, lo, hi, nh, nkv, S, hd, scale)
    dq_h, dk_h, dv_h = hip_ops.attn_bwd(q.cuda(), k.cuda(), v.cuda(), o_r.to(BF16).cuda(), do.cuda(), lse_r.cuda(), pre.cuda(), lo.cuda(), hi.cuda(),
                                        nh, nkv, S, hd, scale)
    dq_r, dk_r, dv_r = ref_ops.attn_bwd(q.float(), k.float(), v.float(), o_r, do.float(), lse_r, pre, lo, hi, nh, nkv, S, hd, scale)
    for a, b, name in ((dq_h, dq_r, "dQ"), (dk_h, dk_r, "dK"), (dv_h, dv_r, "dV")):
        a = a.float().cpu()
        assert torch.isfinite(a).all(), name
        assert (a - b.float()).abs().max() <= 0.03 * math.sqrt(nh // nkv) + 0.02 + 0.03 * b.float().abs().max(), name
    if empty_row is not None:
        assert float(dq_h[empty_row].float().abs().max()) == 0.0
