"""Shared helpers for the golden-vector tests."""
import os

import torch

import time_r1_amd  # noqa: F401
from time_r1_amd.config import tiny_test, tiny_test_25
from time_r1_amd.params import ModelParams
from time_r1_amd import vision_process as VP
from time_r1_amd import rewards as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["grpo_beta", "clip_beta", "grpo_nobeta_ragged", "clip_nobeta", "q25_grpo_beta", "q25_clip_beta_ragged"]   # q25_*: Qwen2.5-VL tower


def cfg_for(fx):
    return tiny_test_25() if fx.get("model") == "qwen2_5_vl" else tiny_test()


def frames_for(fx):
    shape = tuple(fx.get("frames_shape", (4, 3, 56, 84)))
    return torch.randint(0, 256, shape, generator=torch.Generator().manual_seed(fx["frames_seed"]), dtype=torch.uint8).float()


def load_case(name):
    return torch.load(os.path.join(GOLDEN, "grpo_step_%s.pt" % name), weights_only=False)


def golden_params(ops, fx):
    """(policy params, reference-policy params) exactly as tests/golden/gen_grpo_golden.py built them."""
    cfg = cfg_for(fx)
    pol = ModelParams(cfg, ops, seed=fx["param_seed"])
    from oracle.ref_ops import RefOps
    base = ModelParams(cfg, RefOps(), seed=fx["param_seed"]).export_hf_state_dict()
    g = torch.Generator().manual_seed(fx["ref_noise_seed"])
    sd = {k: (v.float() + 0.02 * torch.randn(v.shape, generator=g) * (0 if "visual.blocks" in k or "patch_embed" in k else 1)) for k, v in base.items()}
    ref = ModelParams(cfg, ops, init="none")
    ref.load_hf_state_dict(sd)
    return cfg, pol, ref


def golden_inputs(fx):
    pv, grid = VP.patchify(frames_for(fx))
    return pv, [grid]


def golden_rewards(fx):
    G = fx["G"]
    row = fx["row"]
    kw = dict(solution=[row["solution"]] * G, durations=[row["durations"]] * G)
    fns = [R.iou_timestamp_reward_v2, R.format_reward]
    rew = torch.zeros(G, len(fns))
    for j, fn in enumerate(fns):
        rew[:, j] = torch.tensor(fn(prompts=None, completions=fx["completions"], **kw), dtype=torch.float32)
    return rew, fns


HF_GRAD_KEYS = {
    "lm_head.weight": ("lm_head", None),
    "model.language_model.norm.weight": ("norm", None),
    "model.language_model.layers.0.self_attn.q_proj.weight": ("l0.qkv.w", "q"),
    "model.language_model.layers.0.self_attn.v_proj.bias": ("l0.qkv.b", "v"),
    "model.language_model.layers.1.mlp.down_proj.weight": ("l1.down.w", None),
    "model.visual.merger.mlp.2.weight": ("merger.fc2.w", None),
    "model.visual.merger.mlp.2.bias": ("merger.fc2.b", None),
    "model.visual.merger.ln_q.weight": ("merger.ln.w", None),
    "model.visual.merger.ln_q.bias": ("merger.ln.b", None),
}


def pick_grad(cfg, get, hf_key):
    """Map an HF gradient key to the matching slice of this repo's fused parameter gradient (`get(name)` returns the fused grad)."""
    name, part = HF_GRAD_KEYS[hf_key]
    g = get(name)
    t = cfg.text
    if part == "q":
        return g[: t.q_dim]
    if part == "v":
        return g[t.q_dim + t.kv_dim:]
    return g


def grads_cleared(tr):
    """After an optimizer step: everything that ACCUMULATES from zero is zero again.  The decoder layers' large matrices may keep the last window's values -
    the first weight-gradient GEMM of the next window overwrites them (Engine.lazy_zero_plan / AdamWFlat.lazy_zero)."""
    g = tr.params.train.grad.clone()
    lz = tr.optimizer.lazy_zero
    if lz:
        for l in range(lz["count"]):
            for a, b in lz["keep"]:
                g[lz["base"] + l * lz["stride"] + a: lz["base"] + l * lz["stride"] + b] = 0
    return float(g.abs().max()) == 0.0
