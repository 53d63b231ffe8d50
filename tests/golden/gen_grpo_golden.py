"""Golden GRPO micro-steps captured by running the reference's UNMODIFIED `TimeR1_Trainer.compute_loss`
(/root/reference/src/time_r1/rl/timer1_trainer.py:512-782) on a tiny random-init Qwen2-VL (transformers 5.15.0, fp32, CPU).

Run here:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_grpo_golden.py   ->  tests/golden/grpo_step_*.pt

What is captured per case: the sampled completion ids, decoded strings, per-token policy/ref log-probs and entropies (via a pass-through
wrapper around `_get_per_token_logps`), the loss, every metric the reference logs, and the parameter gradients of loss.backward().
Weights are NOT stored: they are re-created from `ModelParams(tiny_test(), seed)` (deterministic CPU generator) by both sides.
"""
import collections
import copy
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from _ref_harness import load_ref_main, load_ref_trainers  # noqa: E402

import time_r1_amd  # noqa: E402,F401
from time_r1_amd.config import tiny_test, tiny_test_25  # noqa: E402
from time_r1_amd.params import ModelParams  # noqa: E402
from time_r1_amd import vision_process as VP  # noqa: E402
from oracle.ref_ops import RefOps  # noqa: E402
from oracle.text import fake_decode  # noqa: E402


def hf_tiny(cfg):
    from transformers import Qwen2VLConfig, Qwen2VLForConditionalGeneration
    t, v = cfg.text, cfg.vision
    if v.variant == "qwen2_5_vl":
        from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration
        hc = Qwen2_5_VLConfig(
            text_config=dict(vocab_size=t.vocab_size, hidden_size=t.hidden, intermediate_size=t.intermediate, num_hidden_layers=t.n_layers,
                             num_attention_heads=t.n_heads, num_key_value_heads=t.n_kv_heads, max_position_embeddings=4096,
                             rope_parameters={"rope_type": "default", "rope_theta": t.rope_theta, "mrope_section": list(t.mrope_section)},
                             rms_norm_eps=t.rms_eps, pad_token_id=cfg.pad_token_id, eos_token_id=cfg.eos_token_id, bos_token_id=None,
                             tie_word_embeddings=False),
            vision_config=dict(depth=v.depth, hidden_size=v.embed_dim, hidden_act="silu", intermediate_size=v.mlp_dim, num_heads=v.num_heads,
                               in_channels=3, patch_size=14, spatial_merge_size=2, temporal_patch_size=2, tokens_per_second=int(cfg.tokens_per_second),
                               window_size=v.window_size, out_hidden_size=v.out_hidden, fullatt_block_indexes=list(v.fullatt_block_indexes)),
            image_token_id=cfg.image_token_id, video_token_id=cfg.video_token_id, vision_start_token_id=cfg.vision_start_token_id,
            vision_end_token_id=cfg.vision_end_token_id)
        hc._attn_implementation = "eager"
        return Qwen2_5_VLForConditionalGeneration(hc).float()
    hc = Qwen2VLConfig(
        text_config=dict(vocab_size=t.vocab_size, hidden_size=t.hidden, intermediate_size=t.intermediate, num_hidden_layers=t.n_layers,
                         num_attention_heads=t.n_heads, num_key_value_heads=t.n_kv_heads, max_position_embeddings=4096,
                         rope_parameters={"rope_type": "default", "rope_theta": t.rope_theta, "mrope_section": list(t.mrope_section)},
                         rms_norm_eps=t.rms_eps, pad_token_id=cfg.pad_token_id, eos_token_id=cfg.eos_token_id, bos_token_id=None,
                         tie_word_embeddings=False),
        vision_config=dict(depth=v.depth, embed_dim=v.embed_dim, num_heads=v.num_heads, hidden_size=v.out_hidden, mlp_ratio=v.mlp_dim // v.embed_dim,
                           patch_size=14, temporal_patch_size=2, spatial_merge_size=2, in_channels=3),
        image_token_id=cfg.image_token_id, video_token_id=cfg.video_token_id, vision_start_token_id=cfg.vision_start_token_id,
        vision_end_token_id=cfg.vision_end_token_id)
    hc._attn_implementation = "eager"
    return Qwen2VLForConditionalGeneration(hc).float()


class Shim(torch.nn.Module):
    """transformers 4.51 -> 5.15 API drift only: mm_token_type_ids is mandatory in 5.x, use_model_defaults is gone.
    `force_eos` post-edits the sampled completions (an INPUT of the loss algebra) to exercise ragged EOS masks."""

    def __init__(self, m, video_token_id, force_eos=None, eos=1, force_rows=None):
        super().__init__()
        self.m, self.config, self.vid = m, m.config, video_token_id
        self.force_eos, self.eos, self.force_rows = force_eos, eos, force_rows

    def forward(self, input_ids, **kw):
        kw.setdefault("mm_token_type_ids", (input_ids == self.vid).int() * 2)
        return self.m(input_ids=input_ids, **kw)

    def generate(self, **kw):
        kw.pop("use_model_defaults", None)
        kw.setdefault("mm_token_type_ids", (kw["input_ids"] == self.vid).int() * 2)
        out = self.m.generate(**kw)
        P = kw["input_ids"].shape[1]
        if self.force_rows:
            for g, ids in self.force_rows.items():
                out[g, P:P + len(ids)] = torch.tensor(ids)
        if self.force_eos:
            for g, pos in self.force_eos.items():
                out[g, P + pos] = self.eos
        return out


class FakeProc:
    """Stand-in for the HF processor (no tokenizer files offline): fixed prompt ids around the expanded video pads, the
    HF patchify layout, and a deterministic id -> text map for batch_decode."""
    eos_token_id, pad_token_id = 1, 0

    def __init__(self, cfg):
        self.cfg = cfg

    def apply_chat_template(self, conv, tokenize=False, add_generation_prompt=True):
        return "PROMPT"

    def __call__(self, text=None, images=None, videos=None, fps=None, **kw):
        from transformers import BatchFeature
        pv, grid = VP.patchify(videos[0])
        n_tok = grid[0] * grid[1] * grid[2] // 4
        c = self.cfg
        ids = torch.tensor([[5, 6, 7, c.vision_start_token_id] + [c.video_token_id] * n_tok + [c.vision_end_token_id, 8, 9, 10]])
        return BatchFeature({"input_ids": ids, "attention_mask": torch.ones_like(ids), "pixel_values_videos": pv,
                             "video_grid_thw": torch.tensor([list(grid)])})

    def batch_decode(self, ids, skip_special_tokens=True):
        return [fake_decode(r.tolist(), skip=(self.eos_token_id, self.pad_token_id) if skip_special_tokens else ()) for r in ids]


def run_case(name, use_grpo, beta, force_eos, seed, G=4, C=8, model="qwen2_vl", frames_shape=(4, 3, 56, 84), dtype=torch.float32, like=None):
    """dtype=torch.bfloat16 + like=<fp32 case name>: the SAME micro-step (weights, frames, completion ids forced to the fp32 capture's) run by
    the unmodified reference with the model in bf16 - every reference script trains in bf16 (timer1_trainer.py:244-246, :469).  The distance
    of this capture from the fp32 one is the reference's own bf16 noise: the yardstick for the HIP path's tolerances (tests/test_trainer_gpu.py)."""
    ref_main = load_ref_main()
    t1, _ = load_ref_trainers()
    from transformers import GenerationConfig
    cfg = tiny_test_25() if model == "qwen2_5_vl" else tiny_test()
    ops = RefOps()
    params = ModelParams(cfg, ops, seed=0)
    hf = hf_tiny(cfg)
    hf.load_state_dict({k: v.float() for k, v in params.export_hf_state_dict().items()}, strict=True)
    # reference policy != policy, so that KL and its gradient are exercised: ref = weights of ModelParams(seed=1) for the trainable part
    hf_ref = hf_tiny(cfg)
    sd_ref = ModelParams(cfg, ops, seed=0).export_hf_state_dict()
    g = torch.Generator().manual_seed(99)
    sd_ref = {k: (v.float() + 0.02 * torch.randn(v.shape, generator=g) * (0 if "visual.blocks" in k or "patch_embed" in k else 1)) for k, v in sd_ref.items()}
    hf_ref.load_state_dict(sd_ref, strict=True)

    frames = torch.randint(0, 256, frames_shape, generator=torch.Generator().manual_seed(7), dtype=torch.uint8).float()
    row = {"problem": "person sits down", "video_path": "x.mp4", "video_start": None, "video_end": None, "solution": (2.0, 12.0), "durations": 30.0}
    t1.process_vision_info_v3 = lambda conv, return_video_kwargs=True: (None, [frames], {"fps": [2.0]})   # test-time patch of a module attribute
    tr = object.__new__(t1.TimeR1_Trainer)
    acc = types.SimpleNamespace(device=torch.device("cpu"), gather_for_metrics=lambda x: x, unwrap_model=lambda m: m)
    proc = FakeProc(cfg)
    # rows 0, 1, 3 are overwritten with token ids whose decoded text exercises the reward branches (row 2 stays sampled):
    #   r0 "<think>step</think><answer>3 to 9</answer>"  r1 "<think>the</think><answer>1 to 2</answer>"  r3 "<answer>2 to 12</answer>.."
    def tok(*idx):
        return [26 * 2 + i for i in idx]
    force_rows = {0: tok(0, 22, 1, 2, 8, 4, 14, 3), 1: tok(0, 18, 1, 2, 6, 4, 7, 3), 3: tok(2, 7, 4, 6, 7, 3, 16, 16)}
    if like is not None:        # every row forced to the fp32 capture's completion (sampling from bf16 logits would pick other tokens)
        comp32 = torch.load(os.path.join(HERE, "grpo_step_%s.pt" % like), weights_only=False)["completion_ids"]
        force_rows, force_eos = {g_: comp32[g_].tolist() for g_ in range(G)}, None
    if dtype != torch.float32:
        hf, hf_ref = hf.to(dtype), hf_ref.to(dtype)
        frames = frames.to(dtype).float()      # (values are integers 0..255: exact in bf16)
    policy = Shim(hf, cfg.video_token_id, force_eos=force_eos, force_rows=force_rows)
    tr.__dict__.update(processing_class=proc, accelerator=acc, num_generations=G, beta=beta, use_grpo=use_grpo, epsilon_low=0.2, epsilon_high=0.2,
                       epsilon=0.2, reward_funcs=[ref_main.iou_timestamp_reward_v2, ref_main.format_reward], reward_processing_classes=[None, None],
                       metric_funcs=list(ref_main.metric_funcs_registry.values()), _metrics=collections.defaultdict(list),
                       ref_model=Shim(hf_ref, cfg.video_token_id).eval() if beta != 0 else None, prompt_type="v1", is_deepspeed_enabled=False,
                       _past=None, args=types.SimpleNamespace(device=torch.device("cpu"), past_index=-1),
                       generation_config=GenerationConfig(max_new_tokens=C, do_sample=True, temperature=1.0, num_return_sequences=G, pad_token_id=0))
    cap = {"calls": []}
    orig = t1.TimeR1_Trainer._get_per_token_logps

    def spy(self, model, input_ids, attention_mask, pixel_values_videos, video_grid_thw):
        lp, en = orig(self, model, input_ids, attention_mask, pixel_values_videos, video_grid_thw)
        cap["calls"].append(dict(ids=input_ids.clone(), logp=lp.detach().clone(), ent=en.detach().clone()))
        return lp, en
    tr._get_per_token_logps = types.MethodType(spy, tr)
    import contextlib
    import io
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        loss = tr.compute_loss(policy, [row])
        loss.backward()
    ids = cap["calls"][0]["ids"]
    P = ids.shape[1] - C
    comp = ids[:, P:]
    fx = {
        "case": name, "use_grpo": use_grpo, "beta": beta, "G": G, "C": C, "seed": seed, "param_seed": 0, "dtype": str(dtype).replace("torch.", ""), "like": like,
        "frames_seed": 7, "frames_shape": tuple(frames_shape), "model": model, "row": {k: v for k, v in row.items()}, "prompt_ids": ids[0, :P].tolist(), "completion_ids": comp.clone(),
        "completions": proc.batch_decode(comp), "logp": cap["calls"][0]["logp"][:, P - 1:].float(), "entropy": cap["calls"][0]["ent"][:, P - 1:].float(),
        "ref_logp": cap["calls"][1]["logp"][:, P - 1:].float() if beta != 0 else None, "loss": loss.detach().float().clone(),
        "metrics": {k: list(v) for k, v in tr._metrics.items()},
        "ref_noise_seed": 99,
        "grads": {k: p.grad.detach().float().clone() for k, p in hf.named_parameters() if p.grad is not None and
                  any(s in k for s in ("lm_head", "layers.0.self_attn.q_proj", "layers.1.mlp.down_proj", "merger.mlp.2", "merger.ln_q", "norm.weight", "layers.0.self_attn.v_proj.bias"))},
        "grad_norms": {k: float(p.grad.norm()) for k, p in hf.named_parameters() if p.grad is not None},
        "embed_grad_rows": {int(i): hf.model.language_model.embed_tokens.weight.grad[int(i)].float().clone() for i in set(comp.reshape(-1).tolist()[:6] + [5, 10])},
    }
    out = os.path.join(HERE, "grpo_step_%s.pt" % name)
    torch.save(fx, out)
    print(name, "loss", float(loss), "metrics", {k: round(v[0], 5) for k, v in tr._metrics.items()}, os.path.getsize(out), "bytes")


def run_case_ft(name, force_eos, seed, G=4, C=8, frames_shape=(6, 3, 56, 140), prompt_type="v2"):
    """One micro-step of the UNMODIFIED `TimeR1_Trainer_ft.compute_loss` (/root/reference/src/time_r1/rl/timer1_trainer_ft.py:536-852) the way
    `finetune.py:693-713` configures it for config 4: Qwen2.5-VL, PPO-clip loss (`--use_grpo false`), beta = 0 (no reference model), the dataset
    row carrying pre-decoded frames (`video_inputs` / `video_kwargs`, finetune.py:594-623), `metric_funcs` = finetune.py's registry.  Besides what
    run_case stores, the fixture keeps the chat conversation `make_conversation_video` produced (prompt template v2/v3 exist only in `_ft`) and every
    `metrics/<fn>` / `clip_ratio/*` value.  (`use_grpo=True` cannot be captured from `_ft`: :821 reads `coef_1`, which only the clip branch defines.)"""
    from _ref_harness import load_ref_finetune
    ref_ft = load_ref_finetune()
    _, t2 = load_ref_trainers()
    from transformers import GenerationConfig
    cfg = tiny_test_25()
    ops = RefOps()
    params = ModelParams(cfg, ops, seed=0)
    hf = hf_tiny(cfg)
    hf.load_state_dict({k: v.float() for k, v in params.export_hf_state_dict().items()}, strict=True)
    frames = torch.randint(0, 256, frames_shape, generator=torch.Generator().manual_seed(7), dtype=torch.uint8).float()
    row = {"problem": "person opens the door", "video_path": "x.mp4", "video_start": None, "video_end": None, "solution": (2.0, 12.0), "durations": 30.0,
           "video_inputs": [frames], "video_kwargs": {"fps": [2.0]}}
    tr = object.__new__(t2.TimeR1_Trainer_ft)
    acc = types.SimpleNamespace(device=torch.device("cpu"), gather_for_metrics=lambda x: x, unwrap_model=lambda m: m)
    seen = {}

    class Proc(FakeProc):
        def apply_chat_template(self, conv, tokenize=False, add_generation_prompt=True):
            seen["conversation"] = copy.deepcopy(conv)
            return "PROMPT"

        def __call__(self, text=None, images=None, videos=None, fps=None, **kw):
            seen["fps"] = list(fps)
            return super().__call__(text=text, images=images, videos=videos, fps=fps, **kw)

    def tok(*idx):
        return [26 * 2 + i for i in idx]
    force_rows = {0: tok(0, 22, 1, 2, 8, 4, 14, 3), 1: tok(0, 18, 1, 2, 6, 4, 7, 3), 3: tok(2, 7, 4, 6, 7, 3, 16, 16)}
    policy = Shim(hf, cfg.video_token_id, force_eos=force_eos, force_rows=force_rows)
    reward_funcs = [ref_ft.reward_funcs_registry["iou_v2"], ref_ft.reward_funcs_registry["format"]]
    metric_funcs = list(ref_ft.metric_funcs_registry.values())
    tr.__dict__.update(processing_class=Proc(cfg), accelerator=acc, num_generations=G, beta=0.0, use_grpo=False, epsilon_low=0.2, epsilon_high=0.2,
                       epsilon=0.2, reward_funcs=reward_funcs, reward_processing_classes=[None, None], metric_funcs=metric_funcs,
                       _metrics=collections.defaultdict(list), ref_model=None, prompt_type=prompt_type, is_deepspeed_enabled=False,
                       _past=None, args=types.SimpleNamespace(device=torch.device("cpu"), past_index=-1),
                       generation_config=GenerationConfig(max_new_tokens=C, do_sample=True, temperature=1.0, num_return_sequences=G, pad_token_id=0))
    cap = {"calls": []}
    orig = t2.TimeR1_Trainer_ft._get_per_token_logps

    def spy(self, model, input_ids, attention_mask, pixel_values_videos, video_grid_thw):
        lp, en = orig(self, model, input_ids, attention_mask, pixel_values_videos, video_grid_thw)
        cap["calls"].append(dict(ids=input_ids.clone(), logp=lp.detach().clone(), ent=en.detach().clone()))
        return lp, en
    tr._get_per_token_logps = types.MethodType(spy, tr)
    import contextlib
    import io
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        loss = tr.compute_loss(policy, [row])
        loss.backward()
    ids = cap["calls"][0]["ids"]
    P = ids.shape[1] - C
    comp = ids[:, P:]
    fx = {
        "case": name, "trainer": "TimeR1_Trainer_ft", "use_grpo": False, "beta": 0.0, "G": G, "C": C, "seed": seed, "param_seed": 0, "dtype": "float32", "like": None,
        "frames_seed": 7, "frames_shape": tuple(frames_shape), "model": "qwen2_5_vl", "prompt_type": prompt_type,
        "row": {k: v for k, v in row.items() if k not in ("video_inputs",)}, "conversation": seen["conversation"], "fps_seen": seen["fps"],
        "prompt_ids": ids[0, :P].tolist(), "completion_ids": comp.clone(), "completions": tr.processing_class.batch_decode(comp),
        "logp": cap["calls"][0]["logp"][:, P - 1:].float(), "entropy": cap["calls"][0]["ent"][:, P - 1:].float(), "ref_logp": None,
        "n_logp_calls": len(cap["calls"]), "loss": loss.detach().float().clone(), "metrics": {k: list(v) for k, v in tr._metrics.items()},
        "reward_func_names": [f.__name__ for f in reward_funcs], "metric_func_names": [f.__name__ for f in metric_funcs], "ref_noise_seed": 99,
        "grads": {k: p.grad.detach().float().clone() for k, p in hf.named_parameters() if p.grad is not None and
                  any(s in k for s in ("lm_head", "layers.0.self_attn.q_proj", "layers.1.mlp.down_proj", "merger.mlp.2", "merger.ln_q", "norm.weight", "layers.0.self_attn.v_proj.bias"))},
        "grad_norms": {k: float(p.grad.norm()) for k, p in hf.named_parameters() if p.grad is not None},
        "embed_grad_rows": {int(i): hf.model.language_model.embed_tokens.weight.grad[int(i)].float().clone() for i in set(comp.reshape(-1).tolist()[:6] + [5, 10])},
    }
    out = os.path.join(HERE, "grpo_step_%s.pt" % name)
    torch.save(fx, out)
    print(name, "loss", float(loss), "metrics", {k: round(v[0], 5) for k, v in tr._metrics.items()}, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    which = sys.argv[1:] or ["qwen2_vl", "qwen2_5_vl"]
    if "qwen2_vl" in which:
        run_case("grpo_beta", True, 0.04, None, 123)
        run_case("clip_beta", False, 0.04, {0: 2, 2: 5}, 124)
        run_case("grpo_nobeta_ragged", True, 0.0, {1: 0, 3: 6}, 125)
        run_case("clip_nobeta", False, 0.0, None, 126)
        run_case("grpo_beta_bf16", True, 0.04, None, 123, dtype=torch.bfloat16, like="grpo_beta")
    if "qwen2_5_vl" in which:
        # Qwen2.5-VL (the family the reference hard-codes): windowed ViT with ragged windows (84x112 -> 3x4 merged tokens, 2x2 windows)
        run_case("q25_grpo_beta", True, 0.04, None, 223, model="qwen2_5_vl", frames_shape=(4, 3, 84, 112))
        run_case("q25_clip_beta_ragged", False, 0.04, {0: 3, 2: 5}, 224, model="qwen2_5_vl", frames_shape=(6, 3, 56, 140))
        run_case("q25_grpo_beta_bf16", True, 0.04, None, 223, model="qwen2_5_vl", frames_shape=(4, 3, 84, 112), dtype=torch.bfloat16, like="q25_grpo_beta")
    if "ft" in which or not sys.argv[1:]:
        # TimeR1_Trainer_ft (config 4's class): clip branch, beta 0, ragged EOS, template v2, pre-decoded frames in the row, metric funcs
        run_case_ft("ft_clip_nobeta_ragged_v2", {0: 3, 2: 5}, 324)
