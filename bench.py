#!/usr/bin/env python
"""GRPO throughput benchmark (BASELINE.json metric): video-query samples/s and rollout tokens/s for Qwen2-VL GRPO post-training.

    python bench.py --gpus N --steps K --warmup W
        N > 1: one rank per GPU.  Under torch.distributed.run (WORLD_SIZE set) this process is one of the N ranks; started plainly,
        bench.py re-launches itself under torch.distributed.run with N ranks.  WORLD_SIZE != N is an error, never a silent 1-GPU run.

A "step" is one GRPO micro-step on one synthetic prompt per GPU: uint8 frames -> fused resize / normalise / patchify kernel ->
vision tower -> G sampled completions (shared-prefix rollout) -> policy + reference log-probs -> rewards / group advantages ->
loss gradient -> backward; the AdamW step (with the RCCL gradient exchange for N > 1) closes every window of `--ga` steps inside
the timed region, like the reference's gradient_accumulation_steps=2 (scripts/posttrain/train_rl.sh:27).  ANY --steps / --warmup
is accepted: K steps are K micro-steps in windows of `ga`, the last window holding the remainder (it still ends in an optimizer
step), so exactly K steps are timed.  Inputs (token ids, decoded uint8 frames - what the reference's video reader hands over,
src/utils/vision_process.py:467-472) are resident in HBM before the timed region.
Weights are random-init of the exact architecture; data is synthetic (no checkpoints / videos exist offline).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import time_r1_amd  # noqa: E402,F401
from time_r1_amd.config import PRESETS  # noqa: E402
from time_r1_amd.params import ModelParams  # noqa: E402
from time_r1_amd.model import Engine  # noqa: E402
from time_r1_amd.grpo import GRPOCore, eos_mask, group_advantages  # noqa: E402
from time_r1_amd.synthetic import synthetic_prompt, piece_decode  # noqa: E402
from time_r1_amd import rewards as R  # noqa: E402
from time_r1_amd import vision_process as VP  # noqa: E402
from time_r1_amd.dist import init_from_env, DataParallel, shard_world_ok  # noqa: E402

# (frames -> video_grid_thw) for a 360x640 source under the reference's pixel budget
GRIDS = {8: (4, 26, 46), 16: (8, 26, 46), 32: (16, 22, 38), 64: (32, 14, 28)}
PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA, vendor nominal (MI355X_MICROARCH.md); `peak_measured` is probed on the box beside it
PEAK_HBM_GBS = 8000.0       # HBM3E spec
SRC_HW = (360, 640)         # synthetic source resolution
VIDEO_ELE = {"total_pixels": 3584 * 28 * 28, "min_pixels": 16 * 28 * 28}      # reference timer1_trainer.py:503-509

def window_plan(n_steps, ga):
    """K micro-steps -> window sizes: full windows of `ga`, then the remainder (every window ends in an optimizer step)."""
    n_steps, ga = int(n_steps), max(1, int(ga))
    plan = [ga] * (n_steps // ga)
    if n_steps % ga:
        plan.append(n_steps % ga)
    return plan


class Workload:
    """The benchmark's job: a `TimeR1_Trainer` (the drop-in class the reference's users call, main.py:573-625) on synthetic dataset rows.
    `window()` = `trainer.optimizer_window(batches)`: the body of `trainer.train()`'s loop, fed by the trainer's own sampler + prefetch
    thread.  `engine_window()` drives the same engine objects (GRPOCore / AdamWFlat) from a bare loop without the trainer class - the
    round-1/2 measurement path, kept as a cross-check that the class costs nothing (`engine_path` in the JSON)."""

    def __init__(self, args, ops, device, rank):
        from time_r1_amd.trainer import TimeR1_Trainer, GRPOConfig
        from time_r1_amd.synthetic import SyntheticProcessor, SyntheticClips
        self.cfg = PRESETS[args.model]()
        self.args, self.ops, self.rank = args, ops, rank
        dp = DataParallel()
        # N > 1 defaults to the sharded optimizer (every reference multi-GPU script runs ZeRO, scripts/zero3.json): master / m / v are allocated as
        # 1/world shards only (AdamWFlat -> Arena.set_shard); --replicated-optimizer opts out, world sizes other than 2 / 4 / 8 fall back
        self.shard = dp.enabled and shard_world_ok(dp.world) and not args.replicated_optimizer
        self.params = ModelParams(self.cfg, ops, init="none", optimizer_state=not self.shard)
        if hasattr(self.params, "init_random_device"):
            self.params.init_random_device(seed=0)
        tiny = args.model.startswith("tiny")
        grid = GRIDS[args.frames] if not tiny else (2, 4, 6)
        self.grid = grid
        v = self.cfg.vision
        # decoded source frames (uint8, what the reference's video reader returns) and the size plan of the reference's fetch_video_v3
        n_frames = grid[0] * v.temporal_patch_size
        src_hw = SRC_HW if not tiny else (72, 96)
        self.src_hw = src_hw
        self.target = VP.video_target_size(VIDEO_ELE, n_frames, *src_hw) if not tiny else (56, 84)
        assert (self.target[0] // v.patch_size, self.target[1] // v.patch_size) == tuple(grid[1:]), (self.target, grid)
        # every rank holds the whole (tiny) synthetic dataset; the trainer's sampler deals rows perm[rank::world] (staged before the timed region)
        self.dataset = SyntheticClips(args.n_prompts * dp.world, n_frames, src_hw, device=device, pin=args.host_frames)
        self.reward_funcs = [R.iou_timestamp_reward_v2, R.format_reward]
        targs = GRPOConfig(output_dir="/tmp/tr1_bench", num_generations=args.G, max_completion_length=args.C, beta=args.beta, use_grpo=not args.clip_loss,
                           temperature=1.0, top_k=50, seed=1234, rope_index_mode="hf4", gradient_accumulation_steps=args.ga, learning_rate=1e-6,
                           lr_scheduler_type="constant", logging_steps=1, save_strategy="no", disable_log_print=True, shard_optimizer=self.shard,
                           rollout_batching=not args.no_rollout_batching,
                           rollout_weight_dtype="fp8" if args.rollout_fp8_w8a16 else ("fp8-mfma" if args.rollout_fp8 else "bf16"),
                           rollout_fp8_keep_bf16=None if args.rollout_fp8_keep_bf16 == "auto" else tuple(x for x in args.rollout_fp8_keep_bf16.split(",") if x and x != "none"),
                           rollout_importance_cap=args.rollout_importance_cap, grad_wire_dtype=args.grad_wire)
        self.ballast = torch.empty(int(args.ballast_gb * (1 << 30)), dtype=torch.uint8, device=device) if (args.ballast_gb > 0 and not args.ballast_early) else None
        self.trainer = TimeR1_Trainer(self.params, self.reward_funcs, [], args=targs, train_dataset=self.dataset,
                                      processing_class=SyntheticProcessor(self.cfg), ops=ops)
        tr = self.trainer
        self.eng, self.core, self.opt = tr.engine, tr.core, tr.optimizer
        self._feed = self._batches()
        self.P = None
        self.micro = 0

    def _batches(self):
        """Endless stream of batches through the trainer's own data path: rank-sharded sampler -> prefetch thread (host half of the
        preparation: chat template, size plan, tokenisation) -> identity collation."""
        tr = self.trainer
        while True:
            for b in tr._prefetching(tr.get_train_dataloader()):
                yield b

    # ------------------------------------------------------------------------------------------------------------ trainer path
    def window(self, n=None):
        """`n` (default `ga`) micro-steps + one optimizer step through `TimeR1_Trainer.optimizer_window`."""
        n = self.args.ga if n is None else n
        tr = self.trainer
        batches = [next(self._feed) for _ in range(n)]
        if self.args.ragged_eos:
            self._ragged_hook()
        tr.optimizer_window(batches)
        if self.P is None:
            self.P = int(tr.core.last_P)
        self.micro += n

    def _ragged_hook(self):
        """SURVEY 8d "ragged case": an EOS at a uniform position in [C/2, C) of every row (seed 1); the decode ran all C steps (the
        reference's generation config has no EOS either), what changes is the mask: loss weights, lengths, counted tokens.  Installed
        once as a wrapper of the processor's batch_decode input: the trainer's EOS mask is computed from the same host token array."""
        tr, a = self.trainer, self.args
        if getattr(tr, "_ragged_installed", False):
            return
        tr._ragged_installed = True
        rng = np.random.default_rng(1 + 7919 * self.rank)
        core = tr.core
        orig_many, orig_one = core.rollout_many, core.rollout

        def inject(states):
            for st in states:
                toks = st.completion_ids.cpu().numpy().copy()
                toks[np.arange(a.G), rng.integers(a.C // 2, max(a.C // 2 + 1, a.C), size=a.G)] = self.cfg.eos_token_id
                st.completion_ids = self.ops.tensor(toks.astype(np.int32), torch.int32)

        def rollout_many(states):
            out = orig_many(states)
            inject(states)
            return out

        def rollout(st):
            out = orig_one(st)
            inject([st])
            return out
        core.rollout_many, core.rollout = rollout_many, rollout

    # ------------------------------------------------------------------------------------------------------------ bare engine loop
    def _finish(self, st, last, n_in_window):
        """Policy / reference log-probs, host rewards, loss gradient and backward of one prompt (no trainer class)."""
        a, core = self.args, self.core
        toks_host = st.completion_ids_host
        core.forward_logps(st)
        completions = [piece_decode(r) for r in toks_host]
        mask = eos_mask(toks_host, self.cfg.eos_token_id)
        rew = torch.zeros(a.G, len(self.reward_funcs))
        kw = dict(solution=[(2.0, 12.0)] * a.G, durations=[30.0] * a.G)
        for j, fn in enumerate(self.reward_funcs):
            rew[:, j] = torch.tensor(fn(prompts=None, completions=completions, **kw), dtype=torch.float32)
        _, adv, _ = group_advantages(rew, a.G)
        sync = None
        if last and self.opt.dp.enabled and not a.no_grad_overlap:
            sync = self.opt.sync
            sync.begin()                # last micro-step of the window: overlap the RCCL gradient exchange with its backward
        if last:
            self.eng.norm_sink = self.opt.norm_sink_begin(self.eng)
        core.loss_backward(st, self.ops.tensor(mask, torch.int32), self.ops.tensor(adv.numpy(), torch.float32), 1.0 / a.ga, grad_sync=sync)
        self.eng.norm_sink = None

    def engine_window(self, n=None):
        a, core, v, tr = self.args, self.core, self.cfg.vision, self.trainer
        n = a.ga if n is None else n
        states = []
        for j in range(n):
            row = self.dataset[(self.rank + (self.micro + j) * self.opt.dp.world) % len(self.dataset)]
            frames = row["video_frames"].to(self.ops.device, non_blocking=True)
            pix, g = self.ops.video_preprocess(frames, self.target, v.patch_dim_padded, v.patch_size, v.temporal_patch_size, v.spatial_merge_size)
            n_tok = g[0] * g[1] * g[2] // v.merge_unit
            ids = tr.processing_class.prompt_ids(tr.processing_class.apply_chat_template(tr.make_conversation_video(row)), n_tok)
            states.append(core.prepare(ids, pix, np.asarray([g])))
        if a.no_rollout_batching:
            for j, st in enumerate(states):
                core.rollout(st)
                st.completion_ids_host = st.completion_ids.cpu().numpy()
                self._finish(st, j == n - 1, n)
        else:
            core.rollout_many(states)
            for st in states:
                st.completion_ids_host = st.completion_ids.cpu().numpy()     # one wait for the decode loop, ahead of every update
            for si, st in enumerate(states):
                self._finish(st, si == n - 1, n)
        self.opt.step()
        self.micro += n


def measure_peaks(ops, device):
    """On-box probes printed beside the vendor nominals: HBM read-stream rate (the library's own row-contiguous read kernel over 2 GiB - larger
    than the 256 MB Infinity Cache), the device-to-device copy rate (read + write streams, what round 2 reported), and the bf16 GEMM rate of
    this library's own kernel and of hipBLASLt (torch.matmul) on 8192^3.  Runs before the workload is built; the buffers are freed again."""
    out = {}
    n = 1 << 31
    a = torch.empty(n, dtype=torch.uint8, device=device)
    a.view(torch.int64).random_()       # arbitrary bit patterns: a zero-filled buffer toggles no data lines and reads optimistically
    sink = torch.zeros(4, dtype=torch.int32, device=device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.probe_hbm_read(a, sink)
    e0.record()
    for _ in range(10):
        ops.probe_hbm_read(a, sink)
    e1.record(); torch.cuda.synchronize()
    out["hbm_read_stream_GBs"] = float(n) * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    b = torch.empty(n // 2, dtype=torch.uint8, device=device)
    b.copy_(a[: n // 2])
    e0.record()
    for _ in range(10):
        b.copy_(a[: n // 2])
    e1.record(); torch.cuda.synchronize()
    out["hbm_copy_GBs"] = 2.0 * (n // 2) * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b
    M = 8192
    x = torch.randn(M, M, device=device, dtype=torch.bfloat16) * 0.05
    w = torch.randn(M, M, device=device, dtype=torch.bfloat16) * 0.05
    for name, fn in (("gemm_bf16_own_TFLOPs", lambda: ops.gemm_nt(x, w)), ("gemm_bf16_hipblaslt_TFLOPs", lambda: torch.matmul(x, w.t()))):
        try:
            fn(); fn()
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); torch.cuda.synchronize()
            out[name] = 2.0 * M ** 3 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e12
        except Exception as e:          # the probe is informative only
            out[name] = None
            out[name + "_error"] = repr(e)[:200]
    del x, w
    torch.cuda.empty_cache()
    return {k: (round(v, 1) if isinstance(v, float) else v) for k, v in out.items()}


def instrument_gemms(ops):
    """Wrap ops.gemm_nt with HIP events (torch events on the current stream = the stream the kernels are launched on)."""
    rec = []
    orig = ops.gemm_nt

    orig_fx = ops.gemm_skinny_fixup

    def timed_fx(a, b, bias=None, residual=None):               # decode down projection (split-K + in-kernel fixup): same weight stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig_fx(a, b, bias=bias, residual=residual)
        e1.record()
        rec.append((True, a.shape[0], b.shape[0], a.shape[1], e0, e1))
        return r
    ops.gemm_skinny_fixup = timed_fx

    def timed(a, b, bias=None, residual=None, out_f32=False, out=None, accumulate=False):
        M, K = a.shape
        N = b.shape[0]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(a, b, bias=bias, residual=residual, out_f32=out_f32, out=out, accumulate=accumulate)
        e1.record()
        skinny = M <= 64 and not accumulate and K >= 256
        rec.append((skinny, M, N, K, e0, e1))
        return r

    orig_ng = ops.norm_gemm

    def timed_ng(x, lnw, eps, w, bias=None, glu=False):      # decode-step fused rmsnorm + projection (+ SwiGLU): same weight stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig_ng(x, lnw, eps, w, bias=bias, glu=glu)
        e1.record()
        rec.append((True, x.shape[0], w.shape[0], x.shape[1], e0, e1))
        return r
    orig_qkv = ops.norm_gemm_qkv

    def timed_qkv(x, lnw, eps, wqkv, *a, **k):                 # decode qkv projection with fused norm / RoPE / KV append: same weight stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig_qkv(x, lnw, eps, wqkv, *a, **k)
        e1.record()
        rec.append((True, x.shape[0], wqkv.shape[0], x.shape[1], e0, e1))
        return r
    ops.gemm_nt = timed
    ops.norm_gemm = timed_ng
    ops.norm_gemm_qkv = timed_qkv
    return rec, (orig, orig_ng, orig_fx, orig_qkv)


def cpu_baseline(args, budget_note=True):
    """Reference CPU path: the same engine driven by the CPU oracle ops (oracle/ref_ops.py, proven equal to the imported
    reference on the golden fixtures) on the host cores; kind = "port".  Bounded sample of the SAME workload: the full-width
    architecture, same prompt and G, ONE decoder layer and ONE ViT block, 2 decode steps, beta = 0.  The per-layer pieces
    (Engine.vit_features / llm_fwd / llm_bwd, timed individually) are scaled to the full depth, everything else in a phase is a
    fixed cost; the reference-policy forward is priced as one more policy forward (identical computation, other weights) and
    the decode steps at the measured host stream bandwidth over the fp32 weights."""
    from oracle.ref_ops import RefOps  # noqa: checker/baseline only
    import copy
    full = PRESETS[args.model]()
    ops = RefOps(act_dtype=torch.float32)
    C_cpu = min(args.C, 2)
    grid = GRIDS[args.frames] if args.model != "tiny" else (2, 4, 6)
    t_all0 = time.time()
    cfg = copy.deepcopy(full)
    cfg.text.n_layers = 1
    cfg.vision.depth = 1
    params = ModelParams(cfg, ops, init="none", optimizer_state=False)
    chunk = torch.empty(1 << 24).uniform_(-0.03, 0.03)
    for arena in (params.train, params.frozen):      # timing only: tile one random chunk instead of drawing billions of normals
        flat = arena.w16
        for a0 in range(0, flat.numel(), chunk.numel()):
            b0 = min(flat.numel(), a0 + chunk.numel())
            flat[a0:b0].copy_(chunk[: b0 - a0])
        for name, _ in arena.specs:
            if name.endswith("ln1") or name.endswith("ln2") or name == "norm" or name.endswith("ln.w") or name.endswith("n1.w") or name.endswith("n2.w"):
                arena.w(name).fill_(1.0)
    eng = Engine(cfg, ops, params)
    layer_t = {"vit_features": 0.0, "llm_fwd": 0.0, "llm_bwd": 0.0}

    def timed(name):
        orig = getattr(eng, name)

        def f(*a, **k):
            t0 = time.time()
            r = orig(*a, **k)
            layer_t[name] += time.time() - t0
            return r
        setattr(eng, name, f)
    for n in layer_t:
        timed(n)
    core = GRPOCore(eng, None, args.G, C_cpu, beta=0.0, use_grpo=not args.clip_loss, seed=1, rope_index_mode="hf4", reuse_prefill=False)
    ids, pix, g = synthetic_prompt(cfg, grid, 64, 64, seed=0)
    ph = {}
    t0 = time.time(); st = core.prepare(ids, pix, g); ph["vision"] = time.time() - t0
    vit_layer = layer_t["vit_features"]
    t0 = time.time(); core.rollout(st); ph["rollout"] = time.time() - t0
    prefill_layer = layer_t["llm_fwd"]
    t0 = time.time(); core.forward_logps(st); ph["logps"] = time.time() - t0
    fwd_layer = layer_t["llm_fwd"] - prefill_layer
    mask = torch.ones(args.G, C_cpu, dtype=torch.int32)
    adv = torch.randn(args.G)
    t0 = time.time(); core.loss_backward(st, mask, adv, 1.0); ph["backward"] = time.time() - t0
    bwd_layer = layer_t["llm_bwd"]
    nl, nv = full.text.n_layers, full.vision.depth
    est = (ph["vision"] - vit_layer) + nv * vit_layer
    est += nl * prefill_layer                                                   # prompt prefill
    fwd = (ph["logps"] - fwd_layer) + nl * fwd_layer
    est += fwd * (2.0 if args.beta != 0.0 else 1.0)                             # policy forward (+ reference-policy forward)
    est += (ph["backward"] - bwd_layer) + nl * bwd_layer
    w_bytes = 4.0 * (nl * (full.text.hidden * (full.text.qkv_dim + full.text.q_dim) + 3 * full.text.hidden * full.text.intermediate)
                     + full.text.vocab_size * full.text.hidden)
    x = torch.empty(64 * 1024 * 1024)
    t0 = time.time(); y = x * 2.0; bw = 2 * x.numel() * 4 / (time.time() - t0)
    est += args.C * w_bytes / bw                                                # decode: every step streams all weights once
    return {"value": 1.0 / est, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "CPU oracle ops fp32, %s width, 1 decoder layer + 1 ViT block, grid %s, G=%d, %d of %d decode steps, beta=0; measured "
                      "phases %s, per-layer parts %s scaled to depth %d/%d, ref forward priced as a second policy forward, decode steps priced at measured "
                      "host stream bandwidth (%.1f GB/s over %.1f GB fp32 weights); %.0f s of CPU work"
                      % (full.name, str(grid), args.G, C_cpu, args.C, {k: round(v, 2) for k, v in ph.items()},
                         {"vit_block": round(vit_layer, 2), "prefill_layer": round(prefill_layer, 2), "fwd_layer": round(fwd_layer, 2),
                          "bwd_layer": round(bwd_layer, 2)}, nl, nv, bw / 1e9, w_bytes / 1e9, time.time() - t_all0),
            "seconds_per_sample_est": est}


def cpu_config1(n_prompts=1, dtype="fp32"):
    """BASELINE.json configs[0] timed for real on the host cores (no extrapolation): Qwen2-VL-2B architecture at FULL depth, 8 frames of
    360x640 (grid 4x26x46, 1196 video tokens), G = 4, C = 64, beta = 0.04, fp32, the CPU oracle ops (kind "port").  One micro-step per
    prompt: preprocessing, vision tower, prefill + 64 decode steps, policy and reference log-probs, loss, backward; the optimizer step is
    timed once.  `python bench.py --cpu-config1 [--cpu-config1-prompts N]` prints this leg only (about 1 minute per prompt on 128 cores)."""
    from oracle.ref_ops import RefOps  # noqa: checker/baseline only
    cfg = PRESETS["qwen2-vl-2b"]()
    ops = RefOps(act_dtype=torch.bfloat16 if dtype == "bf16" else torch.float32)      # bf16: activations / weights rounded to bf16 like the GPU path (the reference's --bf16 run on CPU)
    G, C, grid = 4, 64, GRIDS[8]
    t_all = time.time()
    params = ModelParams(cfg, ops, init="none")
    chunk = torch.empty(1 << 24).uniform_(-0.03, 0.03)
    for arena in (params.train, params.frozen):
        flat = arena.w16
        for a0 in range(0, flat.numel(), chunk.numel()):
            b0 = min(flat.numel(), a0 + chunk.numel())
            flat[a0:b0].copy_(chunk[: b0 - a0])
        for name, _ in arena.specs:
            if name.endswith("ln1") or name.endswith("ln2") or name == "norm" or name.endswith("ln.w") or name.endswith("n1.w") or name.endswith("n2.w"):
                arena.w(name).fill_(1.0)
    params.train.sync_master_from_w16()
    eng = Engine(cfg, ops, params)
    ref = params.train.clone_weights_only()
    core = GRPOCore(eng, ref, G, C, beta=0.04, use_grpo=True, seed=1, rope_index_mode="hf4")
    from time_r1_amd.optim import AdamWFlat
    opt = AdamWFlat(params, ops, lr=1e-6)
    v = cfg.vision
    ph = {"preprocess": 0.0, "vision": 0.0, "rollout": 0.0, "logps": 0.0, "backward": 0.0}
    toks = 0
    for i in range(n_prompts):
        ids, _, g = synthetic_prompt(cfg, grid, 64, 64, seed=i)
        frames = torch.randint(0, 256, (8, 3) + SRC_HW, generator=torch.Generator().manual_seed(7 + i), dtype=torch.uint8)
        t0 = time.time(); pix, gg = ops.video_preprocess(frames, VP.video_target_size(VIDEO_ELE, 8, *SRC_HW), v.patch_dim_padded); ph["preprocess"] += time.time() - t0
        t0 = time.time(); st = core.prepare(ids, pix[:, : v.patch_dim], g); ph["vision"] += time.time() - t0
        t0 = time.time(); core.rollout(st); ph["rollout"] += time.time() - t0
        t0 = time.time(); core.forward_logps(st); ph["logps"] += time.time() - t0
        mask = torch.ones(G, C, dtype=torch.int32)
        t0 = time.time(); core.loss_backward(st, mask, torch.linspace(-1, 1, G), 1.0); ph["backward"] += time.time() - t0
        toks += G * C
    t0 = time.time(); opt.step(); t_opt = time.time() - t0
    per = sum(ph.values()) / n_prompts + t_opt / 2.0          # gradient_accumulation_steps = 2: half an optimizer step per micro-step
    return {"value": 1.0 / per, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "generated_tokens_per_sec": toks / ph["rollout"],
            "sample": "BASELINE configs[0] measured, not extrapolated: Qwen2-VL-2B full depth, 8 frames (grid %s), G=4, C=64, beta=0.04, %s CPU oracle ops, "
                      "%d prompt(s); seconds per prompt by phase %s, optimizer step %.1f s (amortised over GA=2); %.0f s of CPU work in total"
                      % (str(grid), dtype, n_prompts, {k: round(x / n_prompts, 1) for k, x in ph.items()}, t_opt, time.time() - t_all),
            "seconds_per_sample": per, "dtype": dtype}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="qwen2-vl-7b", choices=list(PRESETS))
    ap.add_argument("--frames", type=int, default=32, choices=list(GRIDS))
    ap.add_argument("--G", type=int, default=8)
    ap.add_argument("--C", type=int, default=200)
    ap.add_argument("--beta", type=float, default=0.04)
    ap.add_argument("--ga", type=int, default=2)
    ap.add_argument("--clip-loss", action="store_true", help="PPO-clip branch (use_grpo=False) instead of the sequence-mean GRPO loss")
    ap.add_argument("--n-prompts", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-config1", action="store_true", help="only time BASELINE configs[0] (2B, 8 frames, G=4, C=64) on the host cores with the CPU oracle and print it")
    ap.add_argument("--cpu-config1-prompts", type=int, default=1)
    ap.add_argument("--cpu-config1-dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--roofline-op-by-op", action="store_true", help="time the decode GEMMs from Python torch events around op-by-op launches (rounds 1-2 method)")
    ap.add_argument("--no-peak-probe", action="store_true")
    ap.add_argument("--host-frames", action="store_true", help="copy the uint8 frames from pinned host memory inside the timed step (PCIe-inclusive rate; default: frames resident in HBM)")
    ap.add_argument("--ragged-eos", action="store_true", help="inject an EOS at a uniform position in [C/2, C) of every completion (seed 1): the ragged-length case of SURVEY 8d")
    ap.add_argument("--no-rollout-batching", action="store_true", help="decode each prompt of the accumulation window separately")
    ap.add_argument("--rollout-fp8", action="store_true", help="BASELINE config 'fp8 weights': decode GEMMs read e4m3 weight copies (sampling policy only), fp8 MFMA (W8A8)")
    ap.add_argument("--rollout-fp8-w8a16", action="store_true", help="fp8 weight copies converted to bf16 in registers (bf16 MFMA) instead of the fp8 MFMA")
    ap.add_argument("--rollout-fp8-keep-bf16", default="auto", help="comma list of matrices of the fp8 sampling policy that stay bf16: qkv,o,gu,down,lm_head; 'auto' = the trainer's "
                    "default (qkv,o for the fp8-MFMA policy: faster and less drift than all-fp8, DESIGN 7d), 'none' = every matrix fp8")
    ap.add_argument("--rollout-importance-cap", type=float, default=None, help="truncated importance weight min(exp(policy logp - sampling logp), c) on the advantage term")
    ap.add_argument("--engine-path", action="store_true", help="time the bare engine loop (GRPOCore + AdamWFlat, no TimeR1_Trainer) instead of the trainer class")
    ap.add_argument("--grad-wire", default="bf16", choices=["bf16", "fp32"], help="N > 1: wire format of the gradient exchange (bf16: a 2 B / parameter staging arena)")
    ap.add_argument("--ballast-gb", type=float, default=0.0, help="diagnostic: hold this many GiB of untouched HBM (footprint / address-placement A/B runs)")
    ap.add_argument("--ballast-early", action="store_true", help="diagnostic: take the ballast before anything else is allocated (shifts every later address)")
    ap.add_argument("--no-engine-leg", action="store_true", help="skip the short bare-engine-loop cross-check that follows the timed region")
    ap.add_argument("--no-grad-overlap", action="store_true", help="N > 1: exchange the gradient arena after backward instead of during it")
    ap.add_argument("--shard-optimizer", action="store_true", help="DEPRECATED, no effect: ZeRO-style optimizer sharding (reduce-scatter grads, AdamW on the local 1/N "
                    "shard of master/m/v, all-gather bf16 weights; reference scripts/zero3.json) has been the DEFAULT for N in {2, 4, 8} since round 3; "
                    "--replicated-optimizer opts out")
    ap.add_argument("--replicated-optimizer", action="store_true", help="N > 1: keep master / m / v whole on every rank and all-reduce the gradient")
    args = ap.parse_args(argv)
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0 or args.ga < 1:
        ap.error("--gpus/--steps/--ga must be >= 1 and --warmup >= 0")
    return args


def resolve_launch(args, env, device_count):
    """-> ("run", None) when this process is a rank (or the only process), ("spawn", argv) when it must re-launch itself under
    torch.distributed.run with one rank per GPU.  Raises SystemExit on any mismatch: a run never silently uses fewer GPUs than --gpus."""
    world = env.get("WORLD_SIZE")
    if world is not None:
        if int(world) != args.gpus:
            raise SystemExit("bench.py: WORLD_SIZE=%s but --gpus %d; launch with torch.distributed.run --nproc-per-node %d" % (world, args.gpus, args.gpus))
        return "run", None
    if args.gpus == 1:
        return "run", None
    if env.get("TR1_FORCE_DEVICE") is None and device_count < args.gpus:
        raise SystemExit("bench.py: --gpus %d but only %d HIP device(s) are visible" % (args.gpus, device_count))
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return "spawn", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
                     "--master-port", str(port), os.path.abspath(__file__)]


def main(argv=None):
    args = parse_args(argv)
    if args.cpu_config1:
        print(json.dumps({"metric": "grpo_samples_per_sec", "config": {"workload": "BASELINE configs[0], CPU only"}, "cpu_baseline": cpu_config1(args.cpu_config1_prompts, args.cpu_config1_dtype)}))
        return
    mode, cmd = resolve_launch(args, os.environ, torch.cuda.device_count())
    if mode == "spawn":
        import subprocess
        raise SystemExit(subprocess.call(cmd + list(sys.argv[1:] if argv is None else argv)))

    rank, local, world = init_from_env("cuda")
    if world != args.gpus:
        raise SystemExit("bench.py: world size %d != --gpus %d" % (world, args.gpus))
    if os.environ.get("TR1_FORCE_DEVICE") is not None:      # test hook: several ranks on one GPU (gloo backend)
        local = int(os.environ["TR1_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    device = "cuda:%d" % local
    early_ballast = torch.empty(int(args.ballast_gb * (1 << 30)), dtype=torch.uint8, device=device) if (args.ballast_gb > 0 and args.ballast_early) else None  # noqa: F841
    from time_r1_amd.ops import HipOps
    ops = HipOps(device)
    ops.use_priority_stream()            # main chain ahead of the weight-gradient side stream in the dispatcher (same call as the trainer)
    peaks = measure_peaks(ops, device) if (rank == 0 and not args.no_peak_probe) else None
    from time_r1_amd.dist import dist_diagnostics, exposed_ms
    diag = dist_diagnostics(device)          # collectives: every rank; N = 1: a local stub.  A hang here ends in the process-group timeout.
    wl = Workload(args, ops, device, rank)
    dp = DataParallel()
    tr = wl.trainer
    run_window = wl.engine_window if args.engine_path else wl.window

    for n in window_plan(args.warmup, args.ga):
        run_window(n=n)
    tr._flush_metrics()
    torch.cuda.synchronize(); dp.barrier(); torch.cuda.synchronize()
    tok0, ph0 = tr.generated_tokens, dict(tr.phase_ms_total)
    exposed_ms(wl.opt.sync)                  # drop the warm-up's exchange waits
    timed_plan = window_plan(args.steps, args.ga)
    t0 = time.perf_counter()
    for n in timed_plan:
        run_window(n=n)
    torch.cuda.synchronize(); dp.barrier(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dp.enabled:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    tr._flush_metrics()
    exch_ms = exposed_ms(wl.opt.sync) if dp.enabled else 0.0     # compute-stream time spent waiting on the gradient exchange in the timed windows
    if dp.enabled:
        et = torch.tensor([exch_ms], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(et, op=torch.distributed.ReduceOp.MAX)
        exch_ms = float(et.item())
    if args.engine_path:
        gen_tokens = args.steps * world * args.G * args.C        # the bare loop keeps no token counter; all C tokens are generated (EOS suppressed)
        phases = {}
    else:
        gen_tokens = int(tr.generated_tokens - tok0)              # gathered completion lengths of the timed micro-steps, all ranks
        phases = {k: (v - ph0.get(k, 0.0)) / args.steps for k, v in tr.phase_ms_total.items()}   # per micro-step (rollout / vision are per window / ga)

    # cross-check: the same engine objects driven by a bare loop without the trainer class (a few windows, after the timed region)
    engine_leg = None
    if not args.engine_path and not args.no_engine_leg:
        n_leg = max(args.ga, min(args.steps, 4 * args.ga) // args.ga * args.ga)
        wl.engine_window()
        torch.cuda.synchronize(); dp.barrier(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for n in window_plan(n_leg, args.ga):
            wl.engine_window(n=n)
        torch.cuda.synchronize(); dp.barrier(); torch.cuda.synchronize()
        engine_leg = {"ms_per_step": 1000.0 * (time.perf_counter() - t1) / n_leg, "steps": n_leg,
                      "what": "GRPOCore + AdamWFlat driven by a bare loop (no TimeR1_Trainer, no metrics, no log): the round-1/2 measurement path"}

    # per-rank peak HBM (every rank takes part in the gather): the optimizer-state sharding shows up here as bytes, not as a claim
    mem_here = round(torch.cuda.max_memory_allocated() / 1e9, 3)
    mem_all = [mem_here]
    if dp.enabled:
        mem_all = [None] * world
        torch.distributed.all_gather_object(mem_all, mem_here)
    out = None
    if rank == 0:
        cfg = wl.cfg
        value = args.steps * world / dt
        logs = [h for h in tr.state.log_history if "samples_per_sec" in h]
        out = {
            "metric": "grpo_samples_per_sec", "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "path": "bare engine loop (--engine-path)" if args.engine_path else "TimeR1_Trainer.optimizer_window (the loop body of trainer.train(): sampler + prefetch thread, "
                    "compute path, reward callbacks, metrics, AdamW step, on_step_end, log() every optimizer step)",
            "rollout_tokens_per_sec": (gen_tokens / world / args.steps) / (max(phases.get("rollout", 0.0), 1e-9) / 1000.0) * world if phases else None,
            "generated_tokens_per_sec_end_to_end": gen_tokens / dt,
            "phases_ms_per_step": {k: round(v, 2) for k, v in phases.items()},
            "engine_path": engine_leg,
            "distributed": {**diag, "grad_exchange_exposed_ms_per_optimizer_step": round(exch_ms / max(len(timed_plan), 1), 3),
                            "optimizer_sharded": bool(wl.shard), "grad_wire_dtype": args.grad_wire, "hbm_gb_allocated_peak_per_rank": mem_all,
                            "process_group_timeout_s": float(os.environ.get("TR1_DIST_TIMEOUT_S", "600"))},
            "trainer_log_last": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in (logs[-1] if logs else {}).items()},
            "optimizer_steps": len(timed_plan), "windows": "%d x %d" % (args.steps // args.ga, args.ga) + (" + 1 x %d" % (args.steps % args.ga) if args.steps % args.ga else ""),
            "hbm_gb": {"allocated_peak": round(torch.cuda.max_memory_allocated() / 1e9, 1), "reserved_peak": round(torch.cuda.max_memory_reserved() / 1e9, 1),
                       "device_mallocs": int(torch.cuda.memory_stats().get("num_device_alloc", 0)), "alloc_retries": int(torch.cuda.memory_stats().get("num_alloc_retries", 0))},
            # (the driver keeps the first 120 characters of this string: model, frames, G, C, beta, loss, GA and P come first)
            "config": {"workload": "%s %df G=%d C=%d beta=%g loss=%s GA=%d P=%d grid=%s; GRPO micro-step, 1 prompt/GPU/step; "
                                   "uint8 %dx%d frames -> fused resize/normalise/patchify inside the step%s"
                                   % (cfg.name, args.frames, args.G, args.C, args.beta, "ppo-clip" if args.clip_loss else "grpo", args.ga, wl.P or 0,
                                      str(wl.grid).replace(" ", ""), wl.src_hw[0], wl.src_hw[1],
                                      ("; fp8 (e4m3) weights for the SAMPLING policy only (fp8: %s; bf16: %s) - prefill, log-probs, KL and the update read bf16; advantage term weighted by "
                                       "the truncated importance ratio min(p_update / p_sampling, %g)"
                                       % (",".join(n for n in ("qkv", "o", "gu", "down", "lm_head") if n not in tr.core.roll.fp8_keep_bf16),
                                          ",".join(tr.core.roll.fp8_keep_bf16) or "-", tr._is_cap or 0.0)) if (args.rollout_fp8 or args.rollout_fp8_w8a16) else ""),
                       "frames": "pinned host memory, copied to HBM inside the timed step" if args.host_frames else "resident in HBM before the timed region",
                       "completion_lengths": "ragged: EOS injected at uniform[C/2, C) per row, seed 1" if args.ragged_eos else "all C tokens (EOS suppressed)",
                       "parallelism": "dp%d" % world, "weights": "random-init", "rollout_prompts_in_flight": 1 if args.no_rollout_batching else args.ga,
                       "rollout_weight_dtype": ("fp8-e4m3 weights x e4m3 block-scaled activations, fp8 MFMA (sampling policy only)" if args.rollout_fp8 else
                                                "fp8-e4m3 weights -> bf16 in registers, bf16 MFMA (sampling policy only)" if args.rollout_fp8_w8a16 else "bf16"),
                       "optimizer": "zero-sharded (reduce-scatter / local AdamW / all-gather)" if wl.shard else "replicated AdamW + gradient all-reduce"},
        }
    # ---- roofline of the dominant kernel, measured live with HIP events in one extra (untimed) step
    if not args.no_roofline:
        # every rank runs the extra window (it contains the gradient all-reduce); only rank 0 records events
        rec, orig = instrument_gemms(ops) if rank == 0 else ([], None)
        # the timed windows drive decode through ONE native call per step (csrc/decode.hip).  That driver records a pair of HIP events around
        # each projection GEMM it launches while tr1_decode_profile_begin() is in effect, so the decode family is timed with its launches back to
        # back exactly as in the timed windows.  (--roofline-op-by-op: rounds 1-2 method, the same kernels launched op by op from Python with
        # torch events - every pair then also brackets the interpreter time between two enqueues, which reads ~6% low.)
        wl.core.roll.native_decode = not args.roofline_op_by_op
        overlap0 = wl.eng.overlap_wgrad
        wl.eng.overlap_wgrad = False        # ... and the weight-gradient GEMMs stay on the main stream: a launch timed while another GEMM
        if rank == 0 and wl.core.roll.native_decode:
            ops.decode_profile_begin()      # the C driver brackets its own GEMM launches (back to back, as in the timed windows)
        # the same window also records, per drawn token, its log-prob under the logits it was SAMPLED from (one more pass over the step's logits -
        # outside the timed region): |that - the update policy's log-prob| is the drift of the sampling policy.  For the bf16 policy it is the
        # yardstick (decode kernels vs training kernels on the same weights) the fp8 policies of config 5 are read against.
        track0 = wl.core.roll.track_logp
        wl.core.roll.track_logp = True
        n_hist0 = len(tr.state.log_history)
        run_window()                        # shares the GPU would be charged the other kernel's time
        torch.cuda.synchronize()
        wl.core.roll.track_logp = track0
        drift_vals = [h["rollout_logp_drift"] for h in tr.state.log_history[n_hist0:] if "rollout_logp_drift" in h]      # log() runs every optimizer step
        if rank == 0 and drift_vals and not args.engine_path:
            out["rollout_logp_drift"] = {"mean_abs_nat": sum(drift_vals) / len(drift_vals), "optimizer_steps": len(drift_vals),
                                         "what": "mean over completion tokens of |log p_sampling(token) - log p_update(token)| in the instrumented window: "
                                                 "sampling policy = the decode kernels' logits (%s), update policy = the bf16 training forward on the same weights"
                                                 % ("bf16 weights" if not (args.rollout_fp8 or args.rollout_fp8_w8a16) else
                                                    ("fp8 W8A8" if args.rollout_fp8 else "fp8 W8A16") + (", bf16 kept for " + ",".join(tr.core.roll.fp8_keep_bf16) if tr.core.roll.fp8_keep_bf16 else ""))}
        dec_prof = ops.decode_profile_end() if rank == 0 and wl.core.roll.native_decode else None
        wl.core.roll.native_decode = True
        wl.eng.overlap_wgrad = overlap0
        if rank == 0:
            ops.gemm_nt, ops.norm_gemm, ops.gemm_skinny_fixup, ops.norm_gemm_qkv = orig
    if rank == 0 and not args.no_roofline:
        nstep = float(args.ga)
        big_ms = sum(e0.elapsed_time(e1) for s, M, N, K, e0, e1 in rec if not s)
        big_fl = sum(2.0 * M * N * K for s, M, N, K, e0, e1 in rec if not s)
        big_n = sum(1 for r in rec if not r[0])
        sk_ms = sum(e0.elapsed_time(e1) for s, M, N, K, e0, e1 in rec if s)
        sk_by = sum(2.0 * (N * K + M * K + M * N) for s, M, N, K, e0, e1 in rec if s)
        sk_n = sum(1 for r in rec if r[0])
        sk_best_native = 0.0
        if dec_prof:
            c = wl.cfg.text
            R_ = args.G * (1 if args.no_rollout_batching else args.ga)
            qd, kvd = c.n_heads * c.head_dim, c.n_kv_heads * c.head_dim
            nk = {"qkv": (qd + 2 * kvd, c.hidden), "o": (c.hidden, qd), "gate_up": (2 * c.intermediate, c.hidden), "down": (c.hidden, c.intermediate),
                  "lm_head": (c.vocab_size, c.hidden)}
            fp8_run_ = bool(args.rollout_fp8 or args.rollout_fp8_w8a16)
            keep_ = set(tr.core.roll.fp8_keep_bf16) if fp8_run_ else set()
            short_ = {"qkv": "qkv", "o": "o", "gate_up": "gu", "down": "down", "lm_head": "lm_head"}
            for k_, (ms_, mn_, n_) in dec_prof.items():
                N_, K_ = nk[k_]
                wb = 1.0 if (fp8_run_ and short_[k_] not in keep_) else 2.0      # bytes per weight element of THIS matrix of the sampling policy
                by = wb * N_ * K_ + 2.0 * (R_ * K_ + R_ * N_)
                sk_ms += ms_; sk_by += by * n_; sk_n += n_
                if n_ and N_ * K_ >= (1 << 26) and mn_ > 0:
                    sk_best_native = max(sk_best_native, by / (mn_ * 1e-3) / 1e9)
        # the best single launch of each family in this window: a kernel of this library demonstrably sustains that rate on this box, so the
        # "measured peak" the fractions are read against is never below it (frac_of_measured <= 1 by construction)
        best_sk = max([2.0 * (N * K + M * K + M * N) / (e0.elapsed_time(e1) * 1e-3) / 1e9 for s_, M, N, K, e0, e1 in rec if s_ and N * K >= (1 << 26)] or [0.0])
        best_sk = max(best_sk, sk_best_native)
        best_big = max([2.0 * M * N * K / (e0.elapsed_time(e1) * 1e-3) / 1e12 for s_, M, N, K, e0, e1 in rec if not s_ and M * N * K >= (1 << 36)] or [0.0])
        mfma = {"kernel": "gemm_nt_kernel+gemm_nt8p_kernel", "bound": "mfma", "achieved": big_fl / (big_ms * 1e-3) / 1e12 if big_ms else 0.0, "peak": PEAK_BF16_TFLOPS,
                "unit": "TFLOP/s", "launches": big_n, "avg_launch_us": 1000.0 * big_ms / max(big_n, 1), "ms_per_step": big_ms / nstep, "traffic": None}
        mfma["frac"] = mfma["achieved"] / PEAK_BF16_TFLOPS
        hbm = {"kernel": "decode GEMM family: gemm_skinny + norm_gemm_skinny + norm_glu_lds + gemm_skinny_lds_fix + oproj_frag kernels (fp8 runs: their _w8 / _f8 twins)", "bound": "hbm", "achieved": sk_by / (sk_ms * 1e-3) / 1e9 if sk_ms else 0.0, "peak": PEAK_HBM_GBS,
               "unit": "GB/s", "launches": sk_n, "avg_launch_us": 1000.0 * sk_ms / max(sk_n, 1), "ms_per_step": sk_ms / nstep, "traffic": None}
        hbm["frac"] = hbm["achieved"] / PEAK_HBM_GBS
        # HBM traffic per launch from the PMC pass committed under profiles/ (rocprofv3 --pmc FETCH_SIZE in its own run, gfx950 x2 read
        # correction; same workload shapes) - PMC counters cannot be collected inside this un-profiled run
        try:
            fp8_run = bool(args.rollout_fp8 or args.rollout_fp8_w8a16)
            # (round 5: the fp8 sampling policy has a PMC pass of its own - its decode kernels stream other bytes than the bf16 family's)
            cands = ("r06_pmc_traffic_fp8.json", "r05_pmc_traffic_fp8.json") if fp8_run else ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")
            pmc_name = [f for f in cands if os.path.exists(os.path.join(ROOT, "profiles", f))][0]
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_name)))
            # guard: the counters are only quoted while the kernels they were collected on are the kernels that just ran - the PMC pass records
            # a fingerprint of the GEMM sources (tools/pmc_to_json.py); any edit since then detaches `traffic` until the pass is re-run
            import hashlib
            csrc = os.path.join(ROOT, "time-r1_amd", "csrc")
            now = {f: hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest()[:16] for f in (pmc.get("_source_sha16") or {"gemm.hip": 0, "decode.hip": 0})}
            fresh = pmc.get("_source_sha16") == now
            for r_ in (hbm, mfma):
                r_["traffic_source"] = "profiles/" + pmc_name
                r_["traffic_guard"] = ("kernel sources unchanged since the PMC pass" if fresh else
                                       "STALE: csrc/gemm.hip or decode.hip changed since the PMC pass (or the file predates the guard) - traffic withheld")
            if fresh and args.model == "qwen2-vl-7b" and args.G == 8 and args.ga == 2:
                def fam(*names):      # launch-weighted mean over the kernel families that make up one roofline entry
                    n = sum(pmc[k]["launches"] for k in names if k in pmc)
                    return sum(pmc[k]["launches"] * pmc[k]["fetch_bytes_per_launch_corrected"] for k in names if k in pmc) / max(n, 1)
                hbm["traffic"] = fam("gemm_skinny_kernel", "norm_gemm_skinny_kernel", "norm_glu_lds_kernel", "gemm_skinny_lds_fix_kernel", "oproj_frag_kernel",
                                     "gemm_skinny_w8_kernel", "gemm_skinny_w8a8_kernel", "norm_glu_lds_f8_kernel", "gemm_skinny_lds_fix_f8_kernel")
                mfma["traffic"] = fam("gemm_nt_kernel", "gemm_nt8p_kernel", "gemm_nt256_kernel")
            hbm["algorithmic_bytes_per_launch"] = sk_by / max(sk_n, 1)
            mfma["algorithmic_flops_per_launch"] = big_fl / max(big_n, 1)
        except Exception:
            pass
        if peaks:                 # measured on this box beside the vendor nominal `peak` (frac stays against the nominal)
            own, lib = peaks.get("gemm_bf16_own_TFLOPs"), peaks.get("gemm_bf16_hipblaslt_TFLOPs")
            mfma["peak_measured"] = max([x for x in (own, lib, best_big) if x] or [0.0]) or None
            hbm["peak_measured"] = max([x for x in (peaks.get("hbm_read_stream_GBs"), peaks.get("hbm_copy_GBs"), best_sk) if x] or [0.0]) or None
            mfma["peak_measured_from"] = "max(8192^3 own, 8192^3 hipBLASLt, best own GEMM launch of the instrumented window = %.0f TFLOP/s)" % best_big
            hbm["peak_measured_from"] = "max(read-stream probe, d2d copy probe, best decode-GEMM launch of the instrumented window = %.0f GB/s)" % best_sk
            for r in (mfma, hbm):
                r["frac_of_measured"] = (r["achieved"] / r["peak_measured"]) if r.get("peak_measured") else None
        dominant, other = (mfma, hbm) if big_ms >= sk_ms else (hbm, mfma)
        out["roofline"] = dominant
        out["roofline_secondary"] = other
    if rank == 0 and peaks:
        out["peak_probe"] = peaks
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(args)
        except Exception as e:  # the GPU numbers above stay valid; say why the baseline is missing
            out["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port", "sample": "failed: %r" % (e,)}
    if rank == 0:
        print(json.dumps(out))
    dp.barrier()
    # Memory headroom is a tested property (round 5): an allocator retry inside the run means the caching allocator had to free and re-allocate its blocks in
    # the middle of a step (config 4 once went from 0.64 to 2.2 s per backward that way, unnoticed) - the line above is then not a steady-state number.
    retries = int(torch.cuda.memory_stats().get("num_alloc_retries", 0))
    if retries > 0:
        sys.stderr.write("bench.py: %d caching-allocator retries during the run (HBM headroom exhausted): the measurement is invalid\n" % retries)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
