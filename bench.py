#!/usr/bin/env python
"""GRPO throughput benchmark (BASELINE.json metric): video-query samples/s and rollout tokens/s for Qwen2-VL GRPO post-training.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one GRPO micro-step on one synthetic prompt per GPU: vision tower -> G sampled completions (shared-prefix rollout)
-> policy + reference log-probs -> rewards / group advantages -> loss gradient -> backward; the AdamW step (with the RCCL gradient
average for N > 1) runs every `--ga` steps inside the timed region, exactly like the reference's gradient_accumulation_steps=2
(scripts/posttrain/train_rl.sh:27).  Inputs (token ids, normalised patches) are resident in HBM before the timed region.
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
from time_r1_amd.synthetic import synthetic_prompt  # noqa: E402
from time_r1_amd import rewards as R  # noqa: E402
from time_r1_amd.dist import init_from_env, DataParallel  # noqa: E402

# (frames -> video_grid_thw) for a 360x640 source under the reference's pixel budget
GRIDS = {8: (4, 26, 46), 16: (8, 26, 46), 32: (16, 22, 38), 64: (32, 14, 28)}
PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0       # HBM3E spec

_PIECES = ["<think>", "</think>", "<answer>", "</answer>", " to ", " and ", "1", "2", "3", "4", "5", "6", "7", "8", "9", "0", ".", " ", "the", "person",
           "because", "\n", "step", "observe", "<timestep>", "</timestep>"]


def fake_decode(ids_row):
    """No tokenizer files exist offline: a fixed id -> text piece map so that the real reward callbacks run on real strings."""
    return "".join(_PIECES[int(i) % len(_PIECES)] for i in ids_row)


class Workload:
    def __init__(self, args, ops, device, rank):
        self.cfg = PRESETS[args.model]()
        self.args, self.ops, self.rank = args, ops, rank
        self.params = ModelParams(self.cfg, ops, init="none")
        if hasattr(self.params, "init_random_device"):
            self.params.init_random_device(seed=0)
        self.eng = Engine(self.cfg, ops, self.params)
        self.ref = self.params.train.clone_weights_only() if args.beta != 0.0 else None
        self.core = GRPOCore(self.eng, self.ref, args.G, args.C, beta=args.beta, use_grpo=not args.clip_loss, temperature=1.0, top_k=50,
                             seed=1234 + rank, rope_index_mode="hf4")
        if args.rollout_fp8:
            self.core.roll.weight_dtype = "fp8"
        from time_r1_amd.optim import AdamWFlat
        self.opt = AdamWFlat(self.params, ops, lr=1e-6, dp=DataParallel())
        grid = GRIDS[args.frames] if args.model != "tiny" else (2, 4, 6)
        self.grid = grid
        self.prompts = []
        v = self.cfg.vision
        for i in range(args.n_prompts):
            ids, pix, g = synthetic_prompt(self.cfg, grid, 64, 64, seed=100 * rank + i)
            pp = ops.zeros(pix.shape[0], v.patch_dim_padded)
            pp[:, : v.patch_dim] = pix.to(pp.device).to(pp.dtype)   # staged in HBM before the timed region
            self.prompts.append((ids, pp, g))
        self.P = len(self.prompts[0][0])
        self.reward_funcs = [R.iou_timestamp_reward_v2, R.format_reward]
        self.micro = 0
        self.ev = []

    def window(self, timing=None):
        """`ga` micro-steps = one optimizer step. The rollouts of the window are decoded together (weights are constant inside an
        accumulation window, so this is the reference's sequence of micro-steps with the decode GEMMs amortised over ga*G rows)."""
        a, core = self.args, self.core

        def mark(name):
            if timing is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                timing.append((name, e))
        mark("start")
        states = []
        for j in range(a.ga):
            ids, pix, grid = self.prompts[(self.micro + j) % len(self.prompts)]
            states.append(core.prepare(ids, pix, grid))
        mark("vision")
        if a.no_rollout_batching:
            for st in states:
                core.rollout(st)
        else:
            core.rollout_many(states)
        mark("rollout")
        self.last_tokens = 0
        for si, st in enumerate(states):
            core.forward_logps(st)          # enqueued asynchronously; the host work below overlaps with it
            toks_host = st.completion_ids.cpu().numpy()
            completions = [fake_decode(r) for r in toks_host]
            mask = eos_mask(toks_host, self.cfg.eos_token_id)
            rew = torch.zeros(a.G, len(self.reward_funcs))
            kw = dict(solution=[(2.0, 12.0)] * a.G, durations=[30.0] * a.G)
            for j, fn in enumerate(self.reward_funcs):
                rew[:, j] = torch.tensor(fn(prompts=None, completions=completions, **kw), dtype=torch.float32)
            _, adv, _ = group_advantages(rew, a.G)
            mark("logps")
            sync = None
            if si == len(states) - 1 and self.opt.dp.enabled and not a.no_grad_overlap:
                sync = self.opt.sync
                sync.begin()                # last micro-step of the window: overlap the RCCL gradient exchange with its backward
            core.loss_backward(st, self.ops.tensor(mask, torch.int32), self.ops.tensor(adv.numpy(), torch.float32), 1.0 / a.ga, grad_sync=sync)
            mark("backward")
            self.micro += 1
            self.last_tokens += int(mask.sum())
        self.opt.step()
        mark("optimizer")


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


def main():
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
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-rollout-batching", action="store_true", help="decode each prompt of the accumulation window separately")
    ap.add_argument("--rollout-fp8", action="store_true", help="BASELINE config 'fp8 weights': decode GEMMs read e4m3 weight copies (sampling policy only)")
    ap.add_argument("--no-grad-overlap", action="store_true", help="N > 1: all-reduce the gradient arena after backward instead of during it")
    args = ap.parse_args()

    rank, local, world = init_from_env("cuda")
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    if os.environ.get("TR1_FORCE_DEVICE") is not None:      # test hook: several ranks on one GPU (gloo backend)
        local = int(os.environ["TR1_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    device = "cuda:%d" % local
    from time_r1_amd.ops import HipOps
    ops = HipOps(device)
    ops.use_priority_stream()            # main chain ahead of the weight-gradient side stream in the dispatcher (same call as the trainer)
    wl = Workload(args, ops, device, rank)
    dp = DataParallel()

    assert args.steps % args.ga == 0 and args.warmup % args.ga == 0, "--steps and --warmup must be multiples of --ga (whole optimizer steps)"
    for _ in range(args.warmup // args.ga):
        wl.window()
    torch.cuda.synchronize(); dp.barrier(); torch.cuda.synchronize()
    timing = []
    gen_tokens = 0
    t0 = time.perf_counter()
    for _ in range(args.steps // args.ga):
        wl.window(timing)
        gen_tokens += wl.last_tokens
    torch.cuda.synchronize(); dp.barrier(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dp.enabled:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
        gt = torch.tensor([gen_tokens], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(gt)
        gen_tokens = int(gt.item())

    phases = {}
    for (n0, e0), (n1, e1) in zip(timing[:-1], timing[1:]):
        if n1 != "start":
            phases[n1] = phases.get(n1, 0.0) + e0.elapsed_time(e1)
    phases = {k: v / args.steps for k, v in phases.items()}     # per micro-step (the rollout / vision phases are per window / ga)

    out = None
    if rank == 0:
        cfg = wl.cfg
        value = args.steps * world / dt
        out = {
            "metric": "grpo_samples_per_sec", "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "rollout_tokens_per_sec": (gen_tokens / world / args.steps) / (phases.get("rollout", 1e-9) / 1000.0) * world,
            "generated_tokens_per_sec_end_to_end": gen_tokens / dt,
            "phases_ms_per_step": {k: round(v, 2) for k, v in phases.items()},
            "hbm_gb": {"allocated_peak": round(torch.cuda.max_memory_allocated() / 1e9, 1), "reserved_peak": round(torch.cuda.max_memory_reserved() / 1e9, 1),
                       "device_mallocs": int(torch.cuda.memory_stats().get("num_device_alloc", 0)), "alloc_retries": int(torch.cuda.memory_stats().get("num_alloc_retries", 0))},
            "config": {"workload": "%s GRPO micro-step: %d frames (grid %s), prompt P=%d tokens, G=%d completions x C=%d tokens, beta=%g, "
                                   "loss=%s, grad-accum %d, 1 prompt/GPU/step" % (cfg.name, args.frames, str(wl.grid), wl.P, args.G, args.C, args.beta,
                                                                                    "ppo-clip" if args.clip_loss else "grpo", args.ga),
                       "parallelism": "dp%d" % world, "weights": "random-init", "rollout_prompts_in_flight": 1 if args.no_rollout_batching else args.ga,
                       "rollout_weight_dtype": "fp8-e4m3 (sampling policy only)" if args.rollout_fp8 else "bf16"},
        }
    # ---- roofline of the dominant kernel, measured live with HIP events in one extra (untimed) step
    if not args.no_roofline:
        # every rank runs the extra window (it contains the gradient all-reduce); only rank 0 records events
        rec, orig = instrument_gemms(ops) if rank == 0 else ([], None)
        # the timed windows drive decode through ONE native call per step (csrc/decode.hip); for this window the same kernels are launched
        # op by op from the host so that every GEMM launch can be bracketed by its own pair of HIP events
        wl.core.roll.native_decode = False
        wl.eng.overlap_wgrad = False        # ... and the weight-gradient GEMMs stay on the main stream: a launch timed while another GEMM
        wl.window()                         # shares the GPU would be charged the other kernel's time
        torch.cuda.synchronize()
        wl.core.roll.native_decode = True
        wl.eng.overlap_wgrad = True
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
        mfma = {"kernel": "gemm_nt_kernel+gemm_nt8p_kernel", "bound": "mfma", "achieved": big_fl / (big_ms * 1e-3) / 1e12 if big_ms else 0.0, "peak": PEAK_BF16_TFLOPS,
                "unit": "TFLOP/s", "launches": big_n, "avg_launch_us": 1000.0 * big_ms / max(big_n, 1), "ms_per_step": big_ms / nstep, "traffic": None}
        mfma["frac"] = mfma["achieved"] / PEAK_BF16_TFLOPS
        hbm = {"kernel": "decode GEMM family: gemm_skinny + norm_gemm_skinny + norm_glu_lds + gemm_skinny_lds_fix kernels", "bound": "hbm", "achieved": sk_by / (sk_ms * 1e-3) / 1e9 if sk_ms else 0.0, "peak": PEAK_HBM_GBS,
               "unit": "GB/s", "launches": sk_n, "avg_launch_us": 1000.0 * sk_ms / max(sk_n, 1), "ms_per_step": sk_ms / nstep, "traffic": None}
        hbm["frac"] = hbm["achieved"] / PEAK_HBM_GBS
        # HBM traffic per launch from the PMC pass committed under profiles/ (rocprofv3 --pmc FETCH_SIZE in its own run, gfx950 x2 read
        # correction; same workload shapes) - PMC counters cannot be collected inside this un-profiled run
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if args.model == "qwen2-vl-7b" and args.G == 8 and args.ga == 2:
                def fam(*names):      # launch-weighted mean over the kernel families that make up one roofline entry
                    n = sum(pmc[k]["launches"] for k in names if k in pmc)
                    return sum(pmc[k]["launches"] * pmc[k]["fetch_bytes_per_launch_corrected"] for k in names if k in pmc) / max(n, 1)
                hbm["traffic"] = fam("gemm_skinny_kernel", "norm_gemm_skinny_kernel", "norm_glu_lds_kernel", "gemm_skinny_lds_fix_kernel")
                hbm["algorithmic_bytes_per_launch"] = sk_by / max(sk_n, 1)
                mfma["traffic"] = fam("gemm_nt_kernel", "gemm_nt8p_kernel", "gemm_nt256_kernel")
                mfma["algorithmic_flops_per_launch"] = big_fl / max(big_n, 1)
                hbm["traffic_source"] = mfma["traffic_source"] = "profiles/r01_pmc_traffic.json"
        except Exception:
            pass
        dominant, other = (mfma, hbm) if big_ms >= sk_ms else (hbm, mfma)
        out["roofline"] = dominant
        out["roofline_secondary"] = other
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(args)
        except Exception as e:  # the GPU numbers above stay valid; say why the baseline is missing
            out["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port", "sample": "failed: %r" % (e,)}
    if rank == 0:
        print(json.dumps(out))
    dp.barrier()


if __name__ == "__main__":
    main()
