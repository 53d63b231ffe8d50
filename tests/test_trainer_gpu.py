"""GPU parity proper: the drop-in trainer on the HIP path (C ABI -> gfx950 kernels) against (a) the golden micro-steps captured from
the unmodified reference and (b) the CPU oracle at a larger size.  bf16 activations vs an fp32 reference: logp within 0.06
(SURVEY section 7 hard part 3), loss/KL within 5e-3 absolute, gradients within 6 % relative L2 error per tensor."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import frames_for, CASES, load_case, golden_params, HF_GRAD_KEYS, pick_grad, grads_cleared  # noqa: E402


def _make(fx, ops):
    import time_r1_amd  # noqa: F401
    from time_r1_amd.trainer import TimeR1_Trainer, GRPOConfig
    from time_r1_amd import rewards as R
    from oracle.text import FakeProcessor
    cfg, pol, ref = golden_params(ops, fx)
    args = GRPOConfig(output_dir="/tmp/tr1_gpu", num_generations=fx["G"], max_completion_length=fx["C"], beta=fx["beta"], use_grpo=fx["use_grpo"],
                      rope_index_mode="hf5", temperature=1.0, save_strategy="no")
    tr = TimeR1_Trainer(pol, [R.iou_timestamp_reward_v2, R.format_reward], [], args=args, processing_class=FakeProcessor(cfg), ops=ops)
    if fx["beta"] != 0:
        tr.ref_model.w16.copy_(ref.train.w16)
    return cfg, tr


@pytest.mark.parametrize("case", CASES)
def test_hip_trainer_vs_reference_golden(hip_ops, case):
    fx = load_case(case)
    cfg, tr = _make(fx, hip_ops)
    frames = frames_for(fx)
    tr._video_inputs = lambda ex: ([frames], [2.0])
    row = dict(fx["row"])
    row["_forced_completion_ids"] = fx["completion_ids"].numpy()
    loss = tr.compute_loss(tr.model, [row])
    assert abs(float(loss) - float(fx["loss"])) < 5e-3
    for k, v in fx["metrics"].items():
        tol = 5e-3 if k in ("kl",) else (0.05 if k == "generation_entropy" else 1e-6)
        assert abs(tr._metrics[k][0] - v[0]) <= tol, (k, tr._metrics[k], v)
    assert tr.last_completions == fx["completions"]            # decode + rewards: exact
    g = tr.params.train
    for hk, gold in fx["grads"].items():
        if hk in HF_GRAD_KEYS:
            mine = pick_grad(cfg, g.g, hk).float().cpu()
            rel = (mine - gold).norm() / gold.norm().clamp(min=1e-12)
            assert rel < 0.06, (hk, float(rel))


def test_hip_ft_trainer_vs_reference_ft_golden(hip_ops):
    """`TimeR1_Trainer_ft` on the HIP path against the step captured from the unmodified reference `_ft.compute_loss`
    (timer1_trainer_ft.py:536-852): EVERY metric value (shaping metrics and clip ratios included), loss, decoded strings, gradients."""
    from test_trainer_host_logic import make_ft_trainer
    fx = load_case("ft_clip_nobeta_ragged_v2")
    cfg, tr, proc, row = make_ft_trainer(fx, hip_ops)
    loss = tr.compute_loss(tr.model, [row])
    assert proc.seen_conv == fx["conversation"] and proc.seen_fps == fx["fps_seen"]
    assert abs(float(loss) - float(fx["loss"])) < 5e-3
    assert set(tr._metrics) == set(fx["metrics"])
    for k, v in fx["metrics"].items():
        tol = 0.05 if k == "generation_entropy" else 1e-6
        assert abs(tr._metrics[k][0] - v[0]) <= tol, (k, tr._metrics[k], v)
    assert tr.last_completions == fx["completions"]
    g = tr.params.train
    for hk, gold in fx["grads"].items():
        if hk in HF_GRAD_KEYS:
            mine = pick_grad(cfg, g.g, hk).float().cpu()
            rel = (mine - gold).norm() / gold.norm().clamp(min=1e-12)
            assert rel < 0.06, (hk, float(rel))


def test_hip_full_step_with_rollout_and_optimizer(hip_ops):
    """Sampling + update end to end on the GPU: tokens valid, metrics finite, weights move, grads zeroed; two identical seeds agree."""
    fx = load_case("grpo_beta")
    outs = []
    for _ in range(2):
        cfg, tr = _make(fx, hip_ops)
        frames = torch.randint(0, 256, (4, 3, 56, 84), generator=torch.Generator().manual_seed(3), dtype=torch.uint8).float()
        tr._video_inputs = lambda ex: ([frames], [2.0])
        w0 = tr.params.train.w16.clone()
        loss = tr.compute_loss(tr.model, [dict(fx["row"])])
        assert np.isfinite(float(loss)) and all(np.isfinite(v[0]) for v in tr._metrics.values())
        assert len(tr.last_completions) == fx["G"]
        gn = tr.optimizer.step(lr=1e-3)
        assert float(gn) > 0 and grads_cleared(tr)
        assert not torch.equal(w0, tr.params.train.w16)
        outs.append((tr.last_completions, tr.params.train.w16.clone()))
    assert outs[0][0] == outs[1][0], "same seed -> same sampled completions (Philox keyed by seed,row,step)"
    assert torch.equal(outs[0][1], outs[1][1]) or torch.allclose(outs[0][1].float(), outs[1][1].float(), atol=1e-2)


def test_trainer_path_gradients_equal_engine_path_gradients(hip_ops):
    """the drop-in class adds nothing to the arithmetic.  The same sampled tokens pushed through
    (a) TimeR1_Trainer.accumulation_window (loader rows, fused preprocessing, deferred metrics) and (b) a bare GRPOCore loop give
    identical losses and the same gradient arena (up to fp32 atomic-add order)."""
    from time_r1_amd.grpo import eos_mask, group_advantages
    from time_r1_amd import rewards as R
    from time_r1_amd import vision_process as VP
    fx = load_case("grpo_beta")
    G, C = fx["G"], fx["C"]
    rows = []
    for i in range(2):
        r = dict(fx["row"])
        r["problem"] = "event %d" % i
        r["video_frames"] = torch.randint(0, 256, (4, 3, 72, 96), generator=torch.Generator().manual_seed(40 + i), dtype=torch.uint8)
        rows.append(r)
    # (a) trainer path: sampled rollout, window of 2
    cfg, tr = _make(fx, hip_ops)
    tr.args.gradient_accumulation_steps = 2
    losses_a = [float(x) for x in tr.accumulation_window([[dict(r)] for r in rows])]
    toks = None
    grad_a = tr.params.train.grad.clone()
    m = tr._metrics
    assert len(m["reward"]) == 2
    # the tokens the trainer sampled: replay them through the bare engine loop on a fresh, identically initialised model
    cfg_b, tr_b = _make(fx, hip_ops)
    core, ops, v = tr_b.core, hip_ops, cfg_b.vision
    core.roll.calls = 0
    states = []
    for r in rows:
        T, _, H, W = r["video_frames"].shape
        th, tw = VP.video_target_size({"total_pixels": 3584 * 28 * 28, "min_pixels": 16 * 28 * 28}, T, H, W)
        pix, g = ops.video_preprocess(r["video_frames"].to(ops.device), (th, tw), v.patch_dim_padded, v.patch_size, v.temporal_patch_size, v.spatial_merge_size)
        ids = tr_b.processing_class.prompt_ids("PROMPT", g[0] * g[1] * g[2] // v.merge_unit)
        states.append(core.prepare(ids, pix, np.asarray([g])))
    core.rollout_many(states)
    losses_b = []
    for st in states:
        th_ = st.completion_ids.cpu().numpy()
        core.forward_logps(st)
        comps = tr_b.processing_class.batch_decode(torch.as_tensor(th_), skip_special_tokens=True)
        mask = eos_mask(th_, tr_b.processing_class.eos_token_id)
        rew = torch.zeros(G, 2)
        kw = dict(solution=[fx["row"]["solution"]] * G, durations=[fx["row"]["durations"]] * G)
        for j, fn in enumerate([R.iou_timestamp_reward_v2, R.format_reward]):
            rew[:, j] = torch.tensor(fn(prompts=None, completions=comps, **kw), dtype=torch.float32)
        _, adv, _ = group_advantages(rew, G)
        out3, _ = core.loss_backward(st, ops.tensor(mask, torch.int32), ops.tensor(adv.numpy(), torch.float32), 0.5)
        losses_b.append(float(out3.float()[0]))
    assert losses_a == losses_b                                    # forward + loss: bit-identical
    grad_b = tr_b.params.train.grad
    # same kernels on the same inputs; the only freedom is the order of the fp32 atomic adds in the embedding / norm-weight gradient kernels
    rel = float((grad_a - grad_b).norm() / grad_b.norm())
    assert rel < 1e-5, rel


@pytest.mark.parametrize("case", ["grpo_beta", "q25_grpo_beta"])
def test_hip_path_is_as_close_to_fp32_as_the_reference_in_bf16(hip_ops, case):
    """the 0.06 / 6 % tolerances above are assertions; this turns them into a measurement.  `grpo_step_<case>_bf16.pt` is the
    SAME micro-step (weights, frames, completion ids) run by the unmodified reference with its model in bf16 - what every reference script
    does (timer1_trainer.py:244-246, :469).  Its distance from the fp32 capture is the reference's own bf16 noise; the HIP path (bf16 storage,
    fp32 accumulation / softmax / statistics) must not be further from fp32 than 1.5 x that noise (+ a small absolute floor)."""
    fx, fx16 = load_case(case), load_case(case + "_bf16")
    assert fx16["dtype"] == "bfloat16" and torch.equal(fx16["completion_ids"], fx["completion_ids"])
    cfg, tr = _make(fx, hip_ops)
    frames = frames_for(fx)
    tr._video_inputs = lambda ex: ([frames], [2.0])
    seen = {}
    orig = tr.core.loss_backward

    def spy(st, *a, **k):
        seen["logp"], seen["ent"] = st.logp.float().cpu(), st.entropy.float().cpu()
        seen["ref_logp"] = st.ref_logp.float().cpu() if st.ref_logp is not None else None
        return orig(st, *a, **k)
    tr.core.loss_backward = spy
    row = dict(fx["row"])
    row["_forced_completion_ids"] = fx["completion_ids"].numpy()
    loss = float(tr.compute_loss(tr.model, [row]))
    report = {}
    for key in ("logp", "ref_logp"):
        e_ref = float((fx16[key] - fx[key]).abs().max())
        e_hip = float((seen[key] - fx[key]).abs().max())
        report[key] = (e_hip, e_ref)
        assert e_hip <= 1.5 * e_ref + 2e-3, (key, e_hip, e_ref)
    e_ref, e_hip = abs(float(fx16["loss"]) - float(fx["loss"])), abs(loss - float(fx["loss"]))
    assert e_hip <= 1.5 * e_ref + 2e-4, ("loss", e_hip, e_ref)
    g = tr.params.train
    for hk, gold in fx["grads"].items():
        if hk in HF_GRAD_KEYS:
            mine = pick_grad(cfg, g.g, hk).float().cpu()
            r_hip = float((mine - gold).norm() / gold.norm().clamp(min=1e-12))
            r_ref = float((fx16["grads"][hk] - gold).norm() / gold.norm().clamp(min=1e-12))
            report[hk] = (r_hip, r_ref)
            assert r_hip <= 1.5 * r_ref + 5e-3, (hk, r_hip, r_ref)
    print("HIP-vs-fp32 error against the reference's own bf16-vs-fp32 error:", {k: (round(a, 5), round(b, 5)) for k, (a, b) in report.items()})


@pytest.mark.parametrize("wdtype", ["bf16", "fp8-mfma"])
def test_rollout_logp_drift_is_logged_on_hip(hip_ops, wdtype):
    """Config 5 (fp8 sampling policy): the drift of the sampled tokens' log-probs against the bf16 update policy is a logged metric; with bf16
    rollout weights (decode kernels vs training kernels, same weights) it is numerical noise."""
    from time_r1_amd.trainer import TimeR1_Trainer, GRPOConfig
    from time_r1_amd import rewards as R
    from time_r1_amd.config import tiny_test
    from time_r1_amd.params import ModelParams
    from oracle.text import FakeProcessor
    cfg = tiny_test(n_layers=3)
    args = GRPOConfig(output_dir="/tmp/tr1_gpu_drift", num_generations=8, max_completion_length=10, beta=0.0, use_grpo=True, temperature=1.0,
                      save_strategy="no", rollout_weight_dtype=wdtype, log_rollout_drift=True, disable_log_print=True)
    tr = TimeR1_Trainer(ModelParams(cfg, hip_ops, seed=1), [R.format_reward], [], args=args, processing_class=FakeProcessor(cfg), ops=hip_ops)
    frames = torch.randint(0, 256, (8, 3, 84, 112), generator=torch.Generator().manual_seed(3), dtype=torch.uint8)
    row = {"problem": "person sits down", "video_path": "x.mp4", "video_frames": frames, "solution": (2.0, 12.0), "durations": 30.0}
    tr.compute_loss(tr.model, [row])
    d = tr._metrics["rollout_logp_drift"][0]
    assert np.isfinite(d) and d >= 0
    assert d < (0.02 if wdtype == "bf16" else 0.2), d
    if wdtype != "bf16":
        assert d > 1e-5, "an fp8 sampling policy cannot reproduce the bf16 log-probs exactly"


@pytest.mark.parametrize("wdtype,n_layers", [("bf16", 8), ("fp8-mfma", 8), ("fp8-mfma", 28)], ids=["bf16-8-layers", "fp8-mfma-8-layers", "fp8-mfma-28-layers"])
def test_rollout_logp_drift_bound_at_7b_width(hip_ops, wdtype, n_layers):
    """the drift bound of the sampling policy stated in DESIGN section 5, asserted at the width config 5 names (Qwen2-VL-7B: hidden 3584,
    28 / 4 heads of 128, intermediate 18944, V = 152064) and 8 decoder layers, random-init weights (the worst case: near-uniform next-token distributions).
    bf16 sampling policy = decode kernels vs training kernels on the SAME weights: the yardstick, < 0.05 nat (0.028 at 28 layers on the bench).
    fp8 (W8A8) sampling policy: e4m3 weights alone cost ~0.2 nat at 28 layers (3-bit mantissa, bench `--rollout-fp8-w8a16`), block-scaled e4m3
    activations bring it to 0.30: bound 0.40 nat, and the update corrects for it with truncated importance weights (cap 2) by default.
    Round 6: the fp8 bound is also asserted at the FULL depth of config 5 (28 decoder layers; the bench line of that config logs 0.244)."""
    from time_r1_amd.trainer import TimeR1_Trainer, GRPOConfig, FP8_IMPORTANCE_CAP
    from time_r1_amd import rewards as R
    from time_r1_amd.config import qwen2_vl_7b
    from time_r1_amd.params import ModelParams
    from time_r1_amd.synthetic import SyntheticProcessor
    cfg = qwen2_vl_7b()
    cfg.text.n_layers = n_layers
    cfg.vision.depth = 2
    params = ModelParams(cfg, hip_ops, init="none")
    params.init_random_device(seed=0)
    args = GRPOConfig(output_dir="/tmp/tr1_gpu_drift7b", num_generations=8, max_completion_length=24, beta=0.0, use_grpo=True, temperature=1.0, top_k=50,
                      save_strategy="no", rollout_weight_dtype=wdtype, log_rollout_drift=True, disable_log_print=True, rope_index_mode="hf4")
    tr = TimeR1_Trainer(params, [R.format_reward], [], args=args, processing_class=SyntheticProcessor(cfg), ops=hip_ops)
    assert tr._is_cap == (None if wdtype == "bf16" else FP8_IMPORTANCE_CAP)
    frames = torch.randint(0, 256, (8, 3, 112, 168), generator=torch.Generator().manual_seed(3), dtype=torch.uint8)
    row = {"problem": "person sits down", "video_path": "x.mp4", "video_frames": frames, "solution": (2.0, 12.0), "durations": 30.0}
    loss = tr.compute_loss(tr.model, [row])
    d = tr._metrics["rollout_logp_drift"][0]
    assert np.isfinite(float(loss)) and np.isfinite(d)
    if wdtype == "bf16":
        assert d < 0.05, d
    else:
        assert 1e-3 < d < 0.40, d
    del tr, params
    torch.cuda.empty_cache()


def test_grad_norm_from_weight_gradient_epilogues(hip_ops):
    """The last micro-step's weight-gradient GEMMs leave the squared norm of the FINAL gradient of the decoder layers' large matrices (tr1_wgrad_f32_sumsq);
    AdamWFlat.step adds the rest of the arena.  Same norm as a full pass over the gradient arena, same gradients as the plain GEMM path."""
    from time_r1_amd.trainer import TimeR1_Trainer, GRPOConfig
    from time_r1_amd import rewards as R
    from time_r1_amd.config import tiny_test, TextConfig
    from time_r1_amd.params import ModelParams
    from oracle.text import FakeProcessor
    cfg = tiny_test()
    cfg.text = TextConfig(vocab_size=512, hidden=512, intermediate=1024, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=128, mrope_section=(16, 24, 24))
    cfg.vision.out_hidden = 512
    norms, grads = [], []
    for use_sink in (True, False):
        args = GRPOConfig(output_dir="/tmp/tr1_gpu_sink", num_generations=4, max_completion_length=8, beta=0.04, use_grpo=True, temperature=1.0,
                          save_strategy="no", disable_log_print=True, gradient_accumulation_steps=2, lazy_grad_zero=use_sink)
        tr = TimeR1_Trainer(ModelParams(cfg, hip_ops, seed=1), [R.format_reward], [], args=args, processing_class=FakeProcessor(cfg), ops=hip_ops)
        rows = []
        for i in range(2):
            frames = torch.randint(0, 256, (4, 3, 84, 112), generator=torch.Generator().manual_seed(3 + i), dtype=torch.uint8)
            rows.append([{"problem": "event %d" % i, "video_path": "x.mp4", "video_frames": frames, "solution": (2.0, 12.0), "durations": 30.0}])
        tr.accumulation_window(rows)
        g = tr.params.train.grad.clone()
        want = float(g.double().norm())
        gn = float(tr.optimizer.step())
        assert tr.optimizer.norm_from_sink == use_sink, "lazy_grad_zero=False has no periodic layout plan: the full pass runs"
        assert abs(gn - want) < 2e-4 * want, (gn, want)
        norms.append(gn); grads.append({n: tr.params.train.view(g, n).clone() for n in ("l0.qkv.w", "l0.o.w", "l1.gu.w", "l1.down.w")})
    # (the norm / embedding gradients use fp32 atomics and are not bit-reproducible between two runs; the GEMM outputs are)
    for n in grads[0]:
        assert torch.equal(grads[0][n], grads[1][n]), "the sum-of-squares epilogue must not change the gradient: " + n


def test_grad_norm_sink_is_dropped_when_another_backward_follows_the_armed_one(hip_ops):
    """accumulation_window([b]) arms the weight-gradient sink; a further compute_loss(b2) accumulates into the arena WITHOUT the sink, so step()
    must fall back to the full pass over the gradient arena (the sink only describes the older backward)."""
    from time_r1_amd.trainer import TimeR1_Trainer, GRPOConfig
    from time_r1_amd import rewards as R
    from time_r1_amd.config import tiny_test, TextConfig
    from time_r1_amd.params import ModelParams
    from oracle.text import FakeProcessor
    cfg = tiny_test()
    cfg.text = TextConfig(vocab_size=512, hidden=512, intermediate=1024, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=128, mrope_section=(16, 24, 24))
    cfg.vision.out_hidden = 512
    args = GRPOConfig(output_dir="/tmp/tr1_gpu_sink2", num_generations=4, max_completion_length=8, beta=0.04, use_grpo=True, temperature=1.0,
                      save_strategy="no", disable_log_print=True, gradient_accumulation_steps=1)
    tr = TimeR1_Trainer(ModelParams(cfg, hip_ops, seed=1), [R.format_reward], [], args=args, processing_class=FakeProcessor(cfg), ops=hip_ops)
    rows = []
    for i in range(2):
        frames = torch.randint(0, 256, (4, 3, 84, 112), generator=torch.Generator().manual_seed(30 + i), dtype=torch.uint8)
        rows.append([{"problem": "event %d" % i, "video_path": "x.mp4", "video_frames": frames, "solution": (2.0, 12.0), "durations": 30.0}])
    tr.accumulation_window(rows[:1])          # arms the sink (last micro-step of its window)
    tr.compute_loss(tr.model, rows[1])        # a second backward, not armed
    want = float(tr.params.train.grad.double().norm())
    gn = float(tr.optimizer.step())
    assert not tr.optimizer.norm_from_sink, "a stale sink must not be used"
    assert abs(gn - want) < 2e-4 * want, (gn, want)


@pytest.mark.parametrize("shard", [False, True])
def test_wire_copies_from_the_weight_gradient_epilogues_equal_the_staged_gradient(hip_ops, shard, monkeypatch):
    """in a data-parallel window the weight-gradient epilogues of the last micro-step write the bf16 wire copy of the large matrices and
    GradSync / ShardSync stage only the gaps.  With a one-rank group (TR1_DIST_FORCE=1, torch.distributed "nccl" = RCCL) the exchanged arena must then
    equal bf16(fp32 gradient) EVERYWHERE - a wrong or unwritten wire tile shows up as a mismatch inside a marked range."""
    import torch.distributed as dist
    from time_r1_amd.trainer import TimeR1_Trainer, GRPOConfig
    from time_r1_amd import rewards as R
    from time_r1_amd.config import tiny_test, TextConfig
    from time_r1_amd.params import ModelParams
    from oracle.text import FakeProcessor
    monkeypatch.setenv("TR1_DIST_FORCE", "1")
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
    assert not dist.is_initialized()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, world_size=1, rank=0, device_id=torch.device("cuda:0"))
    try:
        cfg = tiny_test()
        cfg.text = TextConfig(vocab_size=512, hidden=512, intermediate=1024, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=128, mrope_section=(16, 24, 24))
        cfg.vision.out_hidden = 512
        args = GRPOConfig(output_dir="/tmp/tr1_gpu_wire", num_generations=4, max_completion_length=8, beta=0.04, use_grpo=True, temperature=1.0,
                          save_strategy="no", disable_log_print=True, gradient_accumulation_steps=2, shard_optimizer=shard)
        tr = TimeR1_Trainer(ModelParams(cfg, hip_ops, seed=1), [R.format_reward], [], args=args, processing_class=FakeProcessor(cfg), ops=hip_ops)
        assert tr.dp.enabled and tr.dp.world == 1
        rows = []
        for i in range(2):
            frames = torch.randint(0, 256, (4, 3, 84, 112), generator=torch.Generator().manual_seed(3 + i), dtype=torch.uint8)
            rows.append([{"problem": "event %d" % i, "video_path": "x.mp4", "video_frames": frames, "solution": (2.0, 12.0), "durations": 30.0}])
        tr.accumulation_window(rows)
        sync = tr.optimizer.sync
        wired = sorted(sync.wired)
        assert len(wired) >= 8, "the four large matrices of both layers come out of their epilogues in wire format: %r" % (wired,)
        if shard:
            sync.finish()
        else:
            sync.finish(copy_back=False)
        torch.cuda.synchronize()
        g = tr.params.train.grad
        want = g.to(torch.bfloat16)
        for lo, hi in wired:
            assert float(g[lo:hi].abs().max()) > 0.0
            assert torch.equal(sync.stage[lo:hi], want[lo:hi]), "wire copy written by the epilogue differs from bf16(gradient) in [%d, %d)" % (lo, hi)
        assert torch.equal(sync.stage, want), "staged gaps + epilogue-written ranges must tile the whole arena"
        if shard:
            assert sync.gshard.numel() == want.numel() and torch.equal(sync.gshard, want.float()), "one rank: the reduce-scattered shard is the whole (bf16-rounded) gradient"
    finally:
        dist.destroy_process_group()
