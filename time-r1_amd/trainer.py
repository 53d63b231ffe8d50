"""Drop-in trainer boundary: `TimeR1_Trainer` / `TimeR1_Trainer_ft` with the reference's constructor, `train()`, `compute_loss()`,
`log()`, `save_model()` and callback protocol (reference src/time_r1/rl/timer1_trainer.py:184-438, :512-793; ft variant
timer1_trainer_ft.py:551-563, :670-691, :789-842; wiring main.py:573-625), driving the MI355X engine instead of
transformers.Trainer + trl + DeepSpeed (all absent offline, and replaced here by design).

What is kept identical for the caller: constructor signature, dataset-row schema, identity collation (one prompt per device per
micro-step, reference fact "only inputs[0]" :524/:548-551), processor calls (apply_chat_template / __call__ / batch_decode),
reward callback protocol (prompts=, completions=, **row columns x G), metric keys, `state.global_step` / `trainer_state.json` /
`checkpoint-N` layout used by main.py's resume arithmetic (:589-618), TrainerCallback hooks (on_epoch_end etc., main.py:520-539).
What differs by construction: compute_loss also runs the backward (there is no autograd graph to return) and returns the loss value.
"""
import dataclasses
import json
import math
import os
import time
import types
from collections import defaultdict
from typing import Any, Callable, List, Optional, Union

import numpy as np
import torch

from .config import ModelConfig, PRESETS
from .dist import DataParallel, shard_world_ok
from .grpo import GRPOCore, eos_mask, group_advantages
from .model import Engine
from .optim import AdamWFlat
from .params import ModelParams
from . import vision_process as VP

SYSTEM_PROMPT = "You are a video analysis expert."

# Prompt wording is data the policy was trained with; kept verbatim (reference timer1_trainer.py:63-67, timer1_trainer_ft.py:61-85).
QUESTION_TEMPLATE_TG_v1 = """To accurately pinpoint the event "[EVENT]" in the video, determine the precise time period of the event.

Output your thought process within the <think> </think> tags, including analysis with either specific time ranges (xx.xx to xx.xx) in <timestep> </timestep> tags.

Then, provide the start and end times (in seconds, precise to two decimal places) in the format "start time to end time" within the <answer> </answer> tags. For example: "12.54 to 17.83"."""

QUESTION_TEMPLATE_TG_v2 = """To accurately pinpoint the event "[EVENT]" in the video, determine the precise time period of the event.

Provide the start and end times (in seconds, precise to two decimal places) in the format "start time to end time" within the <answer> </answer> tags. For example: "12.54 to 17.83"."""

QUESTION_TEMPLATE_TG_v3 = """Carefully analyze the video content to determine the precise time period during which "[EVENT]" occurs.  Within the `<think>` tags, provide a detailed description of your thought process, following the format below:
```
<think>
Step-by-step Analysis:
<timestep>Time period 1 (start time to end time)</timestep>: Describe the video content within this time period and determine if it is related to "[EVENT]".
<timestep>Time period 2 (start time to end time)</timestep>: Describe the video content within this time period and determine if it is related to "[EVENT]".
Based on the above analysis, state the precise time period during which "[EVENT]" occurs.
</think>
```
Finally, in the `<answer>` tags, provide the start and end times of "[EVENT]" in the format "start time to end time" (in seconds, precise to two decimal places). For example: "12.54 to 17.83".
```
<answer>
start time to end time
</answer>
```"""

_TEMPLATES = {"v1": QUESTION_TEMPLATE_TG_v1, "v2": QUESTION_TEMPLATE_TG_v2, "v3": QUESTION_TEMPLATE_TG_v3}


@dataclasses.dataclass
class GRPOConfig:
    """The training arguments the reference reads (trl.GRPOConfig + main.py:44-70 MY_GRPOConfig), as a plain dataclass."""
    output_dir: str = "timer1-GRPO"
    # GRPO
    num_generations: int = 8
    max_prompt_length: int = 512            # accepted, never enforced (reference quirk)
    max_completion_length: int = 256
    temperature: float = 0.9
    top_k: Optional[int] = 50               # transformers 4.51 GenerationConfig default
    beta: float = 0.04
    use_grpo: bool = False
    prompt_type: str = "v1"
    fix_vit: bool = True
    stop_at_eos: bool = False               # the reference's GenerationConfig carries no eos_token_id (a6): always C tokens
    rollout_batching: bool = True           # decode the prompts of one accumulation window together (same weights, same results)
    grad_wire_dtype: str = "bf16"           # data-parallel gradient exchange wire format ("bf16" | "fp32")
    shard_optimizer: Optional[bool] = None  # ZeRO-style: master/m/v on 1/world of every arena segment, reduce-scatter grads, all-gather bf16 weights.
                                            # None = on for 2 / 4 / 8 ranks (what a zero2 / zero3 `deepspeed` json of the reference scripts selects;
                                            # a non-zero json or False selects the replicated optimizer); no effect on one GPU
    gpu_video_preprocess: Optional[bool] = None   # uint8 frames -> fused HIP resize/normalise/patchify instead of the host processor's pixel path.
                                            # None = automatic: on whenever the row carries pre-decoded uint8 frames; False forces the host processor
    rollout_weight_dtype: str = "bf16"      # "fp8": the SAMPLING policy reads e4m3 copies of the decoder matrices (row scales, re-quantised every window);
                                            # log-probs, KL and the update keep bf16 weights (BASELINE config "fp8 weights")
    disable_log_print: bool = False         # keep log() from printing on rank 0 (bench.py prints exactly one JSON line)
    log_rollout_drift: Optional[bool] = None   # metric rollout_logp_drift = mean |logp under the SAMPLING policy's logits - policy logp| over the
                                            # completion tokens; None = on whenever the rollout reads quantised weights (or an importance cap is set)
    rollout_fp8_keep_bf16: Optional[tuple] = None   # fp8 sampling policies: matrices that stay bf16 ("qkv", "o", "gu", "down", "lm_head").  None = AUTO: the fp8-MFMA
    #                                     policy keeps the attention projections ("qkv", "o": 12 % of the decoder's weight bytes) in bf16 - at 16 decode rows their fp8 kernels
    #                                     are no faster than the bf16 ones (latency-bound; the fp8 q|k|v path also pays a separate rope / KV-append launch), and the policy
    #                                     drifts less: 589.5 ms / 0.244 nat against 593.6 ms / 0.303 nat with every matrix in fp8; () = every matrix fp8
    lazy_grad_zero: bool = True      # the optimizer leaves the decoder layers' large gradient matrices un-zeroed (the next window's first weight gradients
    #                                  overwrite them: Engine.lazy_zero_plan); False = zero the whole gradient arena every step
    rollout_importance_cap: Optional[float] = None   # c: advantage term weighted by min(exp(policy logp - sampling logp), c) per token (truncated
                                            # importance sampling).  None = AUTO: off for a bf16 sampling policy (the reference algebra unchanged), c = 2 for the
                                            # fp8 sampling policies, whose tokens are drawn ~0.2-0.3 nat off the update policy at 7B (e4m3's 3-bit
                                            # mantissa bounds this from below, so the policy-gradient term is CORRECTED instead); 0 = off explicitly
    dataloader_prefetch: int = 2            # batches whose host preprocessing (decode / resize / tokenise) runs ahead on a worker thread; 0 = inline
    rope_index_mode: str = "hf4"            # position rule of the transformers version the reference pins
    # optimisation (HF TrainingArguments names)
    learning_rate: float = 1e-6
    weight_decay: float = 0.0
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    max_grad_norm: float = 1.0
    lr_scheduler_type: str = "linear"
    warmup_steps: int = 0
    num_train_epochs: float = 1.0
    max_steps: int = -1
    per_device_train_batch_size: int = 1
    gradient_accumulation_steps: int = 1
    seed: int = 42
    data_seed: Optional[int] = None
    bf16: bool = True
    # bookkeeping
    logging_steps: int = 1
    save_strategy: str = "steps"            # "steps" | "epoch" | "no"
    save_steps: int = 500
    save_only_model: bool = False
    report_to: Any = None
    resume_from_checkpoint: Optional[str] = None
    # accepted for CLI compatibility; no effect on the MI355X engine (no recomputation / no DeepSpeed / no sliding window)
    gradient_checkpointing: bool = False
    deepspeed: Optional[str] = None
    model_init_kwargs: Optional[dict] = None
    slide_window: bool = False
    max_window_layers: int = 2
    sliding_window_length: int = 4096
    attn_implementation: str = "flash_attention_2"


class TrainerState:
    def __init__(self):
        self.global_step = 0
        self.max_steps = 0
        self.epoch = 0.0
        self.num_train_epochs = 0
        self.log_history = []
        self.is_world_process_zero = True
        self.is_local_process_zero = True

    def to_json(self):
        return {k: v for k, v in self.__dict__.items()}


class TrainerControl:
    def __init__(self):
        self.should_training_stop = False
        self.should_save = False
        self.should_log = False
        self.should_epoch_stop = False


def _call(cb, name, *a, **k):
    fn = getattr(cb, name, None)
    if fn is not None:
        r = fn(*a, **k)
        return r


class _PhaseClock:
    """Phase boundaries of the micro-steps (preprocess | vision | rollout | logps | backward | optimizer) as HIP events on the stream the
    kernels run on - recording costs ~2 us and never waits; the durations are read when `log()` builds the throughput keys.
    On a CPU op backend (tests) it falls back to the host clock."""

    def __init__(self, ops):
        dev = getattr(ops, "device", None)
        self.cuda = dev is not None and torch.device(dev).type == "cuda"
        self.marks = []

    def mark(self, name):
        if self.cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
        else:
            e = time.perf_counter()
        self.marks.append((name, e))
        if len(self.marks) > 4096:           # nobody is reading (a caller driving micro-steps without log()): keep the tail only
            del self.marks[:2048]

    def drain(self):
        """-> {phase: milliseconds} summed over the marks recorded so far ("start" marks open an interval and carry no time)."""
        marks, self.marks = self.marks, []
        out = {}
        if self.cuda and marks:
            marks[-1][1].synchronize()
        for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
            if n1 == "start":
                continue
            dt = e0.elapsed_time(e1) if self.cuda else (e1 - e0) * 1e3
            out[n1] = out.get(n1, 0.0) + dt
        return out


FP8_MFMA_KEEP_BF16 = ("qkv", "o")      # default of GRPOConfig.rollout_fp8_keep_bf16 for the fp8-MFMA sampling policy
FP8_IMPORTANCE_CAP = 2.0      # default truncation c of the per-token importance weight for quantised sampling policies (GRPOConfig.rollout_importance_cap)


def clip_ratio_metrics(logp, old_logp, advantages, mask, eps_low=0.2, eps_high=0.2):
    """Fractions of completion tokens whose probability ratio exp(logp - old_logp) left the PPO clip range on the side that matters for
    the sign of the advantage: (low, high, either), each sum(flag * mask) / sum(mask) (reference timer1_trainer_ft.py:820-829)."""
    coef = torch.exp(logp - old_logp)
    adv = advantages.reshape(-1, 1)
    m = mask.to(coef.dtype)
    is_low = (coef < 1 - eps_low) & (adv < 0)
    is_high = (coef > 1 + eps_high) & (adv > 0)
    tot = m.sum()
    return (is_low * m).sum() / tot, (is_high * m).sum() / tot, ((is_low | is_high) * m).sum() / tot


def config_from_hf(hc, name="model"):
    """transformers config dict (config.json / PretrainedConfig.to_dict()) of a Qwen2-VL or Qwen2.5-VL checkpoint -> ModelConfig."""
    tc = hc.get("text_config") or hc
    vc = hc["vision_config"]
    from .config import TextConfig, VisionConfig
    rope = tc.get("rope_parameters") or tc.get("rope_scaling") or {}
    text = TextConfig(vocab_size=tc["vocab_size"], hidden=tc["hidden_size"], intermediate=tc["intermediate_size"], n_layers=tc["num_hidden_layers"],
                      n_heads=tc["num_attention_heads"], n_kv_heads=tc["num_key_value_heads"], head_dim=tc["hidden_size"] // tc["num_attention_heads"],
                      rms_eps=tc.get("rms_norm_eps", 1e-6), rope_theta=float(rope.get("rope_theta", tc.get("rope_theta", 1e6))),
                      mrope_section=tuple(rope.get("mrope_section", (16, 24, 24))), tie_word_embeddings=bool(hc.get("tie_word_embeddings", tc.get("tie_word_embeddings", False))))
    tps = 2.0
    if "Qwen2_5" in "".join(hc.get("architectures") or []) or hc.get("model_type") == "qwen2_5_vl":
        # Qwen2.5-VL vision config (configuration_qwen2_5_vl.py): hidden_size = ViT width, out_hidden_size = LLM width
        vision = VisionConfig(depth=vc["depth"], embed_dim=vc["hidden_size"], num_heads=vc["num_heads"], mlp_dim=vc["intermediate_size"],
                              out_hidden=vc["out_hidden_size"], patch_size=vc.get("patch_size", 14), temporal_patch_size=vc.get("temporal_patch_size", 2),
                              spatial_merge_size=vc.get("spatial_merge_size", 2), in_channels=vc.get("in_channels", vc.get("in_chans", 3)),
                              variant="qwen2_5_vl", window_size=vc.get("window_size", 112),
                              fullatt_block_indexes=tuple(vc.get("fullatt_block_indexes", (7, 15, 23, 31))))
        tps = float(vc.get("tokens_per_second", 2))
    else:
        vision = VisionConfig(depth=vc["depth"], embed_dim=vc["embed_dim"], num_heads=vc["num_heads"], mlp_dim=int(vc["embed_dim"] * vc.get("mlp_ratio", 4)),
                              out_hidden=vc["hidden_size"], patch_size=vc.get("patch_size", 14), temporal_patch_size=vc.get("temporal_patch_size", 2),
                              spatial_merge_size=vc.get("spatial_merge_size", 2), in_channels=vc.get("in_channels", vc.get("in_chans", 3)))
    cfg = ModelConfig(text=text, vision=vision, image_token_id=hc["image_token_id"], video_token_id=hc["video_token_id"],
                      vision_start_token_id=hc["vision_start_token_id"], vision_end_token_id=hc["vision_end_token_id"],
                      eos_token_id=tc.get("eos_token_id", hc.get("eos_token_id", 151645)) or 151645, pad_token_id=tc.get("pad_token_id", hc.get("pad_token_id", 151643)) or 151643,
                      tokens_per_second=tps, name=name)
    if isinstance(cfg.eos_token_id, list):
        cfg.eos_token_id = cfg.eos_token_id[0]
    return cfg


def load_model_dir(path, ops):
    """HF checkpoint directory (config.json + *.safetensors) -> (ModelConfig, ModelParams)."""
    from safetensors.torch import load_file
    cfg = config_from_hf(json.load(open(os.path.join(path, "config.json"))), name=os.path.basename(os.path.normpath(path)))
    sd = {}
    for f in sorted(os.listdir(path)):
        if f.endswith(".safetensors"):
            sd.update(load_file(os.path.join(path, f)))
    params = ModelParams(cfg, ops, init="none")
    params.load_hf_state_dict(sd)
    return cfg, params


def load_hf_module(model, ops):
    """A loaded transformers model (reference timer1_trainer.py:184-206, :244-262 accepts a `PreTrainedModel` instance as well as a path) ->
    (ModelConfig, ModelParams): its config and state dict are copied into the engine's arenas; the module itself is not kept."""
    hc = model.config.to_dict()
    cfg = config_from_hf(hc, name=str(getattr(model.config, "_name_or_path", "") or type(model).__name__).rstrip("/").split("/")[-1])
    params = ModelParams(cfg, ops, init="none")
    params.load_hf_state_dict({k: v.detach() for k, v in model.state_dict().items()})
    return cfg, params


class TimeR1_Trainer:
    """GRPO post-training trainer (video decoded inside compute_loss, reference timer1_trainer.py:518-534)."""
    _is_ft = False

    def __init__(self, model, reward_funcs, metric_funcs, args: GRPOConfig = None, train_dataset=None, eval_dataset=None,
                 processing_class=None, reward_processing_classes=None, callbacks=None, optimizers=(None, None), peft_config=None,
                 max_pixels: Optional[int] = 12845056, min_pixels: Optional[int] = 3136, attn_implementation: str = "flash_attention_2",
                 ops=None):
        if args is None:
            name = model if isinstance(model, str) else getattr(getattr(model, "cfg", None), "name", None) or getattr(getattr(model, "config", None), "_name_or_path", None) or "model"
            args = GRPOConfig(output_dir="%s-GRPO" % str(name).split("/")[-1])
        self.args = args
        if peft_config is not None:
            raise NotImplementedError("LoRA/peft is not part of the MI355X engine (the reference scripts train full parameters)")
        if not getattr(args, "fix_vit", True):
            raise NotImplementedError("fix_vit=False (training the ViT blocks) is not implemented; every reference script sets fix_vit true")
        mik = getattr(args, "model_init_kwargs", None) or {}
        td = mik.get("torch_dtype")
        if isinstance(td, str) and td not in ("auto", "bfloat16", "float16", "float32"):   # reference :221-235
            raise ValueError("Invalid `torch_dtype` passed to `GRPOConfig`: %s" % td)
        if ops is None:
            from .ops import HipOps   # fails loudly without the HIP library / a GPU: there is no CPU fallback in the product
            ops = HipOps("cuda:%d" % int(os.environ.get("LOCAL_RANK", "0")))
            ops.use_priority_stream()
        self.ops = ops
        # ---- model
        if isinstance(model, str):
            if model in PRESETS:
                self.cfg = PRESETS[model]()
                self.params = ModelParams(self.cfg, ops, seed=args.seed)
            else:
                self.cfg, self.params = load_model_dir(model, ops)
        elif isinstance(model, ModelParams):
            self.cfg, self.params = model.cfg, model
        elif isinstance(model, ModelConfig):
            self.cfg, self.params = model, ModelParams(model, ops, seed=args.seed)
        elif hasattr(model, "state_dict") and hasattr(model, "config"):
            self.cfg, self.params = load_hf_module(model, ops)      # a loaded transformers model, like the reference accepts
        else:
            raise TypeError("model must be a checkpoint path, a preset name, a ModelConfig, a ModelParams or a loaded transformers model")
        self.model = self.params
        self.engine = Engine(self.cfg, ops, self.params)
        self.beta = args.beta
        self.ref_model = self.params.train.clone_weights_only() if self.beta != 0.0 else None    # reference :295-307
        # ---- processor
        if processing_class is None:
            if not isinstance(model, str) or model in PRESETS:
                raise ValueError("processing_class is required when the model is not a checkpoint directory")
            from transformers import AutoProcessor
            processing_class = AutoProcessor.from_pretrained(model)
            if hasattr(processing_class, "image_processor"):
                processing_class.image_processor.max_pixels = max_pixels     # reference :317-319
                processing_class.image_processor.min_pixels = min_pixels
        self.processing_class = processing_class
        if hasattr(processing_class, "tokenizer"):                       # reference :321-323 sets both unconditionally
            processing_class.pad_token_id = processing_class.tokenizer.pad_token_id
            processing_class.eos_token_id = processing_class.tokenizer.eos_token_id
        # ---- rewards / metrics
        if not isinstance(reward_funcs, list):
            reward_funcs = [reward_funcs]
        for f in reward_funcs:
            if not callable(f):
                raise NotImplementedError("reward models given as ids/modules are not supported; pass Python callables (main.py:416-420)")
        self.reward_funcs = reward_funcs
        self.metric_funcs = metric_funcs if isinstance(metric_funcs, list) else ([metric_funcs] if metric_funcs else [])
        self.reward_processing_classes = reward_processing_classes or [None] * len(reward_funcs)
        # ---- GRPO settings (reference :362-395)
        self.max_prompt_length = args.max_prompt_length
        self.max_completion_length = args.max_completion_length
        self.num_generations = args.num_generations
        self.use_grpo = args.use_grpo
        self.prompt_type = args.prompt_type
        self.epsilon_low = self.epsilon_high = 0.2          # hard-coded in the reference (:388-393)
        self.train_dataset, self.eval_dataset = train_dataset, eval_dataset
        self.data_collator = lambda features: features       # identity (:360-361)
        self.callbacks = list(callbacks or [])
        self.dp = DataParallel()
        self.accelerator = types.SimpleNamespace(device=ops.device, gather_for_metrics=self.dp.gather, num_processes=self.dp.world,
                                                 is_main_process=self.dp.rank == 0, unwrap_model=lambda m: m)
        self.core = GRPOCore(self.engine, self.ref_model, self.num_generations, self.max_completion_length, beta=self.beta,
                             use_grpo=self.use_grpo, temperature=args.temperature, top_k=args.top_k, seed=args.seed + 1000 * self.dp.rank,
                             rope_index_mode=args.rope_index_mode, stop_at_eos=args.stop_at_eos)
        self.core.roll.weight_dtype = getattr(args, "rollout_weight_dtype", "bf16")
        keep = getattr(args, "rollout_fp8_keep_bf16", None)
        if keep is None:
            keep = FP8_MFMA_KEEP_BF16 if self.core.roll.weight_dtype == "fp8-mfma" else ()
        self.core.roll.fp8_keep_bf16 = tuple(keep)
        drift = getattr(args, "log_rollout_drift", None)
        cap = getattr(args, "rollout_importance_cap", None)
        if cap is None and self.core.roll.weight_dtype != "bf16":
            cap = FP8_IMPORTANCE_CAP          # an fp8 sampling policy is off-policy by construction: truncated importance weights by default
        self._is_cap = float(cap) if cap else None
        self.core.roll.track_logp = bool(drift) if drift is not None else (self.core.roll.weight_dtype != "bf16" or self._is_cap is not None)
        if optimizers[0] is not None:
            raise NotImplementedError("custom torch optimizers are not supported; the engine owns a fused AdamW over its flat arena")
        self.optimizer = AdamWFlat(self.params, ops, lr=args.learning_rate, betas=(args.adam_beta1, args.adam_beta2), eps=args.adam_epsilon,
                                   weight_decay=args.weight_decay, max_grad_norm=args.max_grad_norm, dp=self.dp,
                                   grad_wire_dtype=torch.bfloat16 if getattr(args, "grad_wire_dtype", "bf16") == "bf16" else torch.float32,
                                   shard_optimizer=self._wants_shard(args, self.dp))
        # the decoder layers' large gradient matrices are overwritten by the first micro-step of every window: the optimizer does not zero them
        self.optimizer.lazy_zero = self.engine.lazy_zero_plan() if getattr(args, "lazy_grad_zero", True) else None
        self.engine.lazy_zero_active = self.optimizer.lazy_zero is not None
        self.optimizer.lazy_zero_ok = lambda eng=self.engine: bool(eng.wgrad_overwrite_first)      # re-checked at every step (A/B attribute)
        self._metrics_store = defaultdict(list)
        self._pending = []                   # micro-steps whose device-side metric values have not been fetched yet (_flush_metrics)
        self._clock = _PhaseClock(ops)
        self._tok_since_log, self._micro_at_log, self._t_last_log = 0.0, 0, None
        self.generated_tokens, self.phase_ms_total = 0.0, defaultdict(float)
        self.state = TrainerState()
        self.state.is_world_process_zero = self.dp.rank == 0
        self.control = TrainerControl()
        self.is_deepspeed_enabled = False
        self._micro = 0
        self._loss_acc, self._tr_loss_last, self._steps_per_epoch = [], 0.0, 0

    @staticmethod
    def _wants_shard(args, dp=None):
        so = getattr(args, "shard_optimizer", None)
        ds = getattr(args, "deepspeed", None)       # reference scripts: --deepspeed scripts/zero3.json | zero3_offload.json | zero2.json
        # None: follow `deepspeed` when one is given (every reference multi-GPU script passes a zero json); without one, N > 1 ranks still
        # default to sharding - the exchange moves the same bytes as the all-reduce and each rank streams 1/N of the AdamW state
        if so is not None:
            want = bool(so)
        elif isinstance(ds, str):
            want = "zero" in os.path.basename(ds).lower()
        else:
            want = dp is not None and dp.enabled
        if want and dp is not None and dp.enabled and not shard_world_ok(dp.world):
            # arena segments split into 1/2/4/8 equal 128-byte-aligned chunks (params.SEG_ALIGN); other world sizes train with the replicated
            # optimizer (same results, more optimizer-state memory) instead of failing at construction where the reference script ran
            if dp.rank == 0:
                print("[time-r1_amd] optimizer sharding supports 2 / 4 / 8 ranks; world size %d falls back to the replicated optimizer" % dp.world, flush=True)
            return False
        return want

    # ------------------------------------------------------------------------------------------------------ prompt building
    def make_conversation_video(self, example):
        if self.prompt_type not in _TEMPLATES or (not self._is_ft and self.prompt_type != "v1"):
            raise ValueError("unsupported prompt_type %r" % self.prompt_type)
        text = _TEMPLATES[self.prompt_type].replace("[EVENT]", example["problem"])
        return [{"role": "user", "content": [{"type": "text", "text": text},
                                             {"type": "video", "video": example["video_path"], "video_start": example.get("video_start"),
                                              "video_end": example.get("video_end"), "total_pixels": 3584 * 28 * 28, "min_pixels": 16 * 28 * 28}]}]

    def _video_inputs(self, example):
        """-> ([frames float T x 3 x H x W], [fps]). Non-ft: decode the whole file (the reference ignores video_start/end here)."""
        ele = {"type": "video", "video": example["video_path"], "total_pixels": 3584 * 28 * 28, "min_pixels": 16 * 28 * 28}
        if "video_frames" in example:      # pre-decoded uint8/float frames supplied by the caller (no decoder offline)
            ele["video"] = example["video_frames"]
        _, vids, kw = VP.process_vision_info_v3([[{"role": "user", "content": [ele]}]], return_video_kwargs=True)
        return vids, kw["fps"]

    def _prepare_inputs(self, inputs):
        return inputs

    def _prompt_ids(self, text, n_video_tokens):
        """Token ids of the chat-templated prompt with the single <|video_pad|> placeholder expanded to n_video_tokens copies
        (what Qwen2VLProcessor.__call__ does before tokenising, processing_qwen2_vl.py)."""
        pc = self.processing_class
        if hasattr(pc, "prompt_ids"):
            return pc.prompt_ids(text, n_video_tokens)
        tok = getattr(pc, "tokenizer", pc)
        pad = "<|video_pad|>"
        return tok(text.replace(pad, pad * n_video_tokens, 1), add_special_tokens=False)["input_ids"]

    # ------------------------------------------------------------------------------------------------------ the micro-step
    def compute_loss(self, model, inputs, return_outputs=False, num_items_in_batch=None):
        if return_outputs:
            raise ValueError("The GRPOTrainer does not support returning outputs")
        ctx = self._step_prepare(inputs)
        if ctx["forced"] is None:
            self.core.rollout(ctx["st"])
        return self._step_finish(ctx)

    def _host_prepare(self, inputs):
        """Host half of a micro-step's preparation (no GPU work): chat template, video decode / frame sampling / resize (or only the size
        plan when the pixels are produced on the GPU), processor call.  Runs inline or on the prefetch thread (decode and
        resize leave the step's critical path)."""
        example = inputs[0]
        prompts = [self.make_conversation_video(ex) for ex in inputs]
        prompts_text = [self.processing_class.apply_chat_template(p, tokenize=False, add_generation_prompt=True) for p in prompts]
        frames = example.get("video_frames")
        gpu_pre = getattr(self.args, "gpu_video_preprocess", None)
        if (gpu_pre is None or gpu_pre) and torch.is_tensor(frames) and frames.dtype == torch.uint8:
            # pre-decoded uint8 frames: size plan on the host (integers), pixels on the GPU (fused resize + normalise + patchify kernel)
            ele = {"total_pixels": 3584 * 28 * 28, "min_pixels": 16 * 28 * 28}
            T, _, H, W = frames.shape
            th, tw = VP.video_target_size(ele, T, H, W)
            v = self.cfg.vision          # the grid follows from the size plan, so the prompt is tokenised here (prefetch thread) too
            n_tok = ((T + v.temporal_patch_size - 1) // v.temporal_patch_size) * (th // v.patch_size) * (tw // v.patch_size) // v.merge_unit
            ids = np.asarray(self._prompt_ids(prompts_text[0], n_tok)).reshape(-1)
            return dict(prompts=prompts, text=prompts_text, gpu_frames=frames.contiguous(), target=(th, tw), ids_gpu=ids, n_tok=n_tok)
        video_inputs, fps_inputs = self._video_inputs(example)
        prompt_inputs = self.processing_class(text=[prompts_text[0]], images=None, videos=[video_inputs[0]], fps=[fps_inputs[0]], padding=True,
                                              return_tensors="pt", padding_side="left", add_special_tokens=False)
        return dict(prompts=prompts, text=prompts_text, ids=np.asarray(prompt_inputs["input_ids"]).reshape(-1),
                    pixels=prompt_inputs["pixel_values_videos"], grid=np.asarray(prompt_inputs["video_grid_thw"]))

    def _step_prepare(self, inputs):
        """Host preprocessing + vision tower for one micro-step (one prompt: reference facts :524, :548-551)."""
        example = inputs[0]
        hp = example.pop("_host_prepared", None) if isinstance(example, dict) else None
        hp = hp.result() if hp is not None else self._host_prepare(inputs)
        prompts, prompts_text = hp["prompts"], hp["text"]
        if "gpu_frames" in hp:
            v = self.cfg.vision
            pixels, grid = self.ops.video_preprocess(hp["gpu_frames"].to(self.ops.device, non_blocking=True), hp["target"], v.patch_dim_padded, v.patch_size,
                                                     v.temporal_patch_size, v.spatial_merge_size)
            self._clock.mark("preprocess")
            n_tok = grid[0] * grid[1] * grid[2] // v.merge_unit
            ids = hp["ids_gpu"] if hp.get("n_tok") == n_tok else np.asarray(self._prompt_ids(prompts_text[0], n_tok)).reshape(-1)
            st = self.core.prepare(ids, pixels, np.asarray([grid]))
        else:
            st = self.core.prepare(hp["ids"], hp["pixels"], hp["grid"])
        self._clock.mark("vision")
        forced = example.get("_forced_completion_ids")       # test hook: teacher-forced completions instead of sampling
        if forced is not None:
            from .positions import PackedLayout
            st.layout = PackedLayout(st.P, self.num_generations, self.max_completion_length)
            st.completion_ids = self.ops.tensor(np.asarray(forced, dtype=np.int32), torch.int32)
        return dict(inputs=inputs, st=st, prompts=prompts, forced=forced)

    def _step_finish(self, ctx, last_in_window=False):
        """Log-probs, rewards, loss gradient and backward of one prompt.  Returns the micro-step loss as a DEVICE scalar (HF's compute_loss
        returns a device tensor too): nothing in here waits for the GPU after the backward is enqueued - the metric values that live on the
        device (loss, KL, entropy, clip ratios) are packed into one small vector per micro-step and fetched with ONE copy when `log()` (or
        anybody reading `self._metrics`) asks for them, so the host prepares the next micro-step while this one's backward runs."""
        inputs, st, prompts = ctx["inputs"], ctx["st"], ctx["prompts"]
        G = self.num_generations
        tokens = st.completion_ids
        # tokens first, THEN the log-prob forwards: a device-to-host copy waits for everything queued before it on the stream, so in this
        # order the host decodes / scores / normalises while the GPU is busy with the policy and reference forwards
        comp_host = getattr(st, "completion_ids_host", None)
        if comp_host is None:
            comp_host = tokens.cpu().numpy()
        self.core.forward_logps(st)
        mask_np = eos_mask(comp_host, self.processing_class.eos_token_id)
        completions = self.processing_class.batch_decode(torch.as_tensor(comp_host), skip_special_tokens=True)
        prompts_rep = [p for p in prompts for _ in range(G)]
        reward_kwargs = {k: [] for k in inputs[0].keys() if k not in ("prompt", "completion") and not k.startswith("_")}
        for k in reward_kwargs:
            for ex in inputs:
                reward_kwargs[k].extend([ex[k]] * G)
        rewards_per_func = torch.zeros(len(prompts_rep), len(self.reward_funcs))
        for i, fn in enumerate(self.reward_funcs):
            rewards_per_func[:, i] = torch.tensor(fn(prompts=prompts_rep, completions=completions, **reward_kwargs), dtype=torch.float32)
        rewards, advantages, std = group_advantages(rewards_per_func, G)
        metric_vals = None
        if self._is_ft and self.metric_funcs:
            metric_vals = torch.stack([torch.tensor(fn(prompts=prompts_rep, completions=completions, **reward_kwargs), dtype=torch.float32)
                                       for fn in self.metric_funcs], 1)
        self._clock.mark("logps")
        scale = 1.0 / max(1, self.args.gradient_accumulation_steps)     # HF divides the loss by GA (model_accepts_loss_kwargs=False, :421-424)
        sync = None
        if last_in_window and self.dp.enabled:
            sync = self.optimizer.sync
            sync.begin()                      # overlap the gradient exchange with this (last) micro-step's backward
        mask_dev = self.ops.tensor(mask_np, torch.int32)
        adv_dev = self.ops.tensor(advantages.numpy(), torch.float32)
        # device-side metric pieces are computed BEFORE the backward releases the step's tensors; tiny [G, C] work on the same stream
        maskf = mask_dev.to(torch.float32)
        ent_mean = ((st.entropy.float() * maskf).sum(1) / maskf.sum(1).clamp(min=1)).mean()
        clip3 = None
        if self._is_ft and not self.use_grpo:   # reference timer1_trainer_ft.py:820-842 (undefined there for use_grpo: coef_1 does not exist)
            lp = st.logp.float()
            clip3 = torch.stack(clip_ratio_metrics(lp, lp, adv_dev, maskf, self.epsilon_low, self.epsilon_high))   # old policy == policy (one update per rollout)
        drift_t, tokw = None, None
        slp = getattr(st, "sample_logp", None)
        if slp is not None and ctx["forced"] is None:
            d = (st.logp.float() - slp.to(st.logp.device).float())          # log importance ratio of every drawn token: update policy / sampling policy
            drift_t = (d.abs() * maskf).sum() / maskf.sum().clamp(min=1)
            if self._is_cap is not None:
                tokw = torch.exp(d).clamp(max=float(self._is_cap))
        if last_in_window:      # the final gradient passes through this backward's weight-gradient epilogues: they leave its squared norm (AdamWFlat.step)
            self.engine.norm_sink = self.optimizer.norm_sink_begin(self.engine)
        try:
            out3, row_len = self.core.loss_backward(st, mask_dev, adv_dev, scale, grad_sync=sync, **({"tok_weight": tokw} if tokw is not None else {}))
        finally:
            self.engine.norm_sink = None
        self._clock.mark("backward")
        out3f = out3.float()
        z = torch.zeros(3, dtype=torch.float32, device=out3f.device)
        dev = torch.cat([out3f[:2], ent_mean.reshape(1).float(), clip3.float() if clip3 is not None else z,
                         drift_t.reshape(1).float() if drift_t is not None else z[:1]])      # fixed layout: loss, kl, entropy, clip x 3, drift
        # ---- metrics (reference :739-777; ft adds metrics/<fn> and clip ratios :789-842): per-sample host values + the device vector,
        # turned into the reference's gathered means by _flush_metrics
        self._pending.append(dict(length=mask_np.sum(1).astype(np.float32), rpf=rewards_per_func.numpy().copy(), reward=rewards.numpy().copy(),
                                  std=std.numpy().copy(), mvals=None if metric_vals is None else metric_vals.numpy(), dev=dev,
                                  has_clip=clip3 is not None, has_drift=drift_t is not None))
        self.last_completions = completions
        self.last_rewards = rewards
        return out3f[0]

    # ------------------------------------------------------------------------------------------------------ deferred metrics
    @property
    def _metrics(self):
        """The reference's `self._metrics` (defaultdict of per-micro-step lists, timer1_trainer.py:739-777).  Reading it resolves the micro-steps
        whose device-side values are still pending with one packed device-to-host copy.  Under data parallelism resolving them is a COLLECTIVE
        (the reference gathers every metric over the ranks), so a plain attribute read never does it - a rank-0-only callback or debug print
        reading `trainer._metrics` would otherwise enter an all-gather alone and hang until the process-group timeout.  There the pending
        micro-steps are resolved by `log()` (every rank calls it at the same optimizer step) or an explicit `flush_metrics()` on ALL ranks."""
        if not self.dp.enabled:
            self._flush_metrics()
        return self._metrics_store

    def flush_metrics(self):
        """Resolve the pending micro-steps' metrics now.  Collective under data parallelism: every rank must call it."""
        self._flush_metrics()
        return self._metrics_store

    def _flush_metrics(self):
        pend, self._pending = self._pending, []
        if not pend:
            return
        G, nf = self.num_generations, len(self.reward_funcs)
        nm = pend[0]["mvals"].shape[1] if pend[0]["mvals"] is not None else 0
        dev = torch.stack([r["dev"] for r in pend])                     # [n, nd] on the device
        host = np.stack([np.concatenate([r["length"], r["rpf"].reshape(-1), r["reward"], r["std"]] +
                                        ([r["mvals"].reshape(-1)] if nm else [])).astype(np.float32) for r in pend])
        if self.dp.enabled:
            both = torch.cat([self.ops.tensor(host, torch.float32), dev], 1)
            allv = self.dp.gather(both[None]).cpu()                      # [world, n, L]
        else:
            allv = torch.cat([torch.as_tensor(host), dev.cpu()], 1)[None]
        M = self._metrics_store
        o_rpf, o_rew, o_std, o_mv = G, G + G * nf, 2 * G + G * nf, 3 * G + G * nf
        o_dev = o_mv + G * nm
        for i, r in enumerate(pend):
            v = allv[:, i]                                               # [world, L]: rank-major, like accelerator.gather's concatenation
            M["completion_length"].append(v[:, :G].reshape(-1).mean().item())
            rpf = v[:, o_rpf:o_rew].reshape(-1, nf).mean(0)
            for j, fn in enumerate(self.reward_funcs):
                M["rewards/%s" % fn.__name__].append(rpf[j].item())
            M["reward"].append(v[:, o_rew:o_std].reshape(-1).mean().item())
            M["reward_std"].append(v[:, o_std:o_mv].reshape(-1).mean().item())
            if self.beta != 0.0:
                M["kl"].append(v[:, o_dev + 1].mean().item())
            M["generation_entropy"].append(v[:, o_dev + 2].mean().item())
            if nm:
                mv = v[:, o_mv:o_dev].reshape(-1, nm)
                for j, fn in enumerate(self.metric_funcs):
                    M["metrics/%s" % fn.__name__].append(mv[:, j].mean().item())
            if r["has_clip"]:
                g_low, g_high, g_reg = v[:, o_dev + 3], v[:, o_dev + 4], v[:, o_dev + 5]
                M["clip_ratio/low_mean"].append(g_low.nanmean().item())
                M["clip_ratio/low_min"].append(g_low[~g_low.isnan()].min().item() if (~g_low.isnan()).any() else float("nan"))
                M["clip_ratio/high_mean"].append(g_high.nanmean().item())
                M["clip_ratio/high_max"].append(g_high[~g_high.isnan()].max().item() if (~g_high.isnan()).any() else float("nan"))
                M["clip_ratio/region_mean"].append(g_reg.nanmean().item())
            if r.get("has_drift"):
                M["rollout_logp_drift"].append(v[:, o_dev + 6].mean().item())
            self._tok_since_log += float(v[:, :G].sum())
            self.generated_tokens += float(v[:, :G].sum())        # cumulative, all ranks

    def accumulation_window(self, batches):
        """All micro-steps of one optimizer step. With rollout_batching the G x len(batches) completions are decoded together
        (weights do not change inside the window, so this equals the reference's sequential micro-steps). Returns the losses
        (device scalars; nothing here waits for the backward)."""
        clock = self._clock
        clock.mark("start")
        if not getattr(self.args, "rollout_batching", True):
            # one prompt at a time, rollout and update interleaved: a sequential rollout keeps its saved prefill (prompt activations, K/V)
            # in the single slot-0 buffers, which the NEXT prompt's prefill overwrites - so each prompt is finished before the next starts
            losses = []
            for i, b in enumerate(batches):
                c = self._step_prepare(b)
                if c["forced"] is None:
                    self.core.rollout(c["st"])
                clock.mark("rollout")
                losses.append(self._step_finish(c, last_in_window=(i == len(batches) - 1)))
            return losses
        ctxs = [self._step_prepare(b) for b in batches]
        todo = [c["st"] for c in ctxs if c["forced"] is None]
        if len(todo) > 1:
            self.core.rollout_many(todo)
        elif todo:
            self.core.rollout(todo[0])
        for st in todo:
            st.completion_ids_host = st.completion_ids.cpu().numpy()        # one wait for the decode loop, ahead of every update
        clock.mark("rollout")
        return [self._step_finish(c, last_in_window=(i == len(ctxs) - 1)) for i, c in enumerate(ctxs)]

    def optimizer_window(self, window, t_start=None):
        """One optimizer step = the unit `train()` repeats: the window's micro-steps, the (clipped, fused) AdamW step with the data-parallel
        gradient exchange, LR schedule, `on_step_end`, logging and step-based checkpoints (TF trainer.py:1892-1961 inner loop body).
        `bench.py` times exactly this method."""
        a = self.args
        if self._t_last_log is None:         # driven without train() (bench.py): the throughput keys count from the first window
            self._t_last_log, self._micro_at_log = time.perf_counter(), self._micro
        self._loss_acc.extend(self.accumulation_window(window))
        self._micro += len(window)
        gnorm = self.optimizer.step(lr=self._lr(self.state.global_step))
        self._clock.mark("optimizer")
        self.state.global_step += 1
        if self._steps_per_epoch:
            self.state.epoch = self.state.global_step / self._steps_per_epoch
        for cb in self.callbacks:
            _call(cb, "on_step_end", a, self.state, self.control)
        if a.logging_steps and self.state.global_step % a.logging_steps == 0:
            # HF logs the mean of the micro-step losses since the last log (training_step returns loss / GA, summed over GA micro-steps);
            # the losses and the gradient norm are device scalars until here - this is the one place per optimizer step the host waits
            losses, self._loss_acc = self._loss_acc, []
            mean_loss = float(torch.stack([torch.as_tensor(x).float().reshape(()) for x in losses]).mean()) if losses else 0.0
            self._tr_loss_last = mean_loss
            self.log({"loss": round(mean_loss, 6), "grad_norm": float(gnorm), "learning_rate": self._lr(self.state.global_step - 1)}, t_start)
        if a.save_strategy == "steps" and a.save_steps and self.state.global_step % a.save_steps == 0:
            self._save_checkpoint()
        return gnorm

    # ------------------------------------------------------------------------------------------------------ training loop
    def get_train_dataloader(self):
        """Batches of `per_device_train_batch_size` dataset rows (identity collation), sharded rank-wise: perm(seed)[rank::world], the
        permutation wrapped to a multiple of `world` first (torch DistributedSampler / accelerate even_batches semantics), so every rank
        sees the SAME number of batches - ranks with different step counts would dead-lock in the gradient exchange."""
        n = len(self.train_dataset)
        bs = self.args.per_device_train_batch_size
        seed = self.args.data_seed if self.args.data_seed is not None else self.args.seed
        trainer = self

        class _Loader:
            def __init__(self):
                self.gen = torch.Generator().manual_seed(seed)
                self.skip_next = 0            # batches the NEXT iteration leaves out at its start (mid-epoch resume), without loading their rows

            def per_rank(self):
                return (n + trainer.dp.world - 1) // trainer.dp.world

            def __len__(self):
                return self.per_rank() // bs

            def skip_epochs(self, k):
                """Advance the sampler past k fully consumed epochs (resume) without touching the dataset."""
                for _ in range(k):
                    torch.randperm(n, generator=self.gen)

            def __iter__(self):
                perm = torch.randperm(n, generator=self.gen).tolist()
                world = trainer.dp.world
                total = self.per_rank() * world
                while len(perm) < total:                  # wrap-around padding, like DistributedSampler(drop_last=False)
                    perm += perm[: total - len(perm)]
                mine = perm[trainer.dp.rank::world]
                first, self.skip_next = self.skip_next * bs, 0
                for i in range(first, len(mine) - bs + 1, bs):
                    yield [trainer.train_dataset[j] for j in mine[i:i + bs]]
        return _Loader()

    def _lr(self, step):
        a = self.args
        base = a.learning_rate
        if a.warmup_steps and step < a.warmup_steps:
            return base * float(step) / float(max(1, a.warmup_steps))
        if a.lr_scheduler_type == "constant":
            return base
        prog = float(step - a.warmup_steps) / float(max(1, self.state.max_steps - a.warmup_steps))
        if a.lr_scheduler_type == "cosine":
            return base * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))
        return base * max(0.0, 1.0 - prog)     # linear (HF default)

    def _prefetching(self, loader, skip=0):
        """Iterate `loader` while a worker thread runs _host_prepare for the next `dataloader_prefetch` batches (video decode / resize /
        tokenisation overlap the GPU step; results travel with the batch under the private key `_host_prepared`)."""
        import collections
        import itertools
        depth = int(getattr(self.args, "dataloader_prefetch", 0) or 0)
        if skip and hasattr(loader, "skip_next"):
            loader.skip_next, skip = skip, 0
        it = itertools.islice(iter(loader), skip, None)
        if depth <= 0:
            yield from it
            return
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="tr1-prefetch")
        q = collections.deque()

        def fill():
            while len(q) < depth:
                try:
                    b = next(it)
                except StopIteration:
                    return
                b = [dict(ex) for ex in b]                # private copy: the future is attached to the row
                b[0]["_host_prepared"] = pool.submit(self._host_prepare, b)
                q.append(b)
        try:
            fill()
            while q:
                b = q.popleft()
                fill()
                yield b
            pool.shutdown(wait=False)        # exhausted: batches already handed out may still be waiting for their worker result
        except BaseException:                # closed early (max_steps / callback stop) or failed: drop the work nobody will read
            pool.shutdown(wait=False, cancel_futures=True)
            raise

    def training_step(self, inputs):
        return self.compute_loss(self.params, inputs)

    def train(self, resume_from_checkpoint=None):
        """The loop that replaces transformers.Trainer.train for main.py:589-625: accumulation windows of `gradient_accumulation_steps`
        micro-steps, optimizer step + LR schedule, callbacks, checkpoints, and HF's resume arithmetic: a checkpoint at global step s
        restarts in epoch s // steps_per_epoch after skipping the (s % steps_per_epoch) * GA batches that epoch already consumed; the
        run ends at state.max_steps (main.py sets max_steps = global_step + epochs * steps_per_epoch before resuming, :600-618)."""
        a = self.args
        loader = self.get_train_dataloader()
        ga = max(1, a.gradient_accumulation_steps)
        steps_per_epoch = max(len(loader) // ga, 1)
        self._steps_per_epoch = steps_per_epoch
        if self.state.max_steps <= 0:
            self.state.max_steps = a.max_steps if a.max_steps > 0 else math.ceil(a.num_train_epochs * steps_per_epoch)
        n_epochs = math.ceil(self.state.max_steps / steps_per_epoch)
        self.state.num_train_epochs = n_epochs
        start_step = 0
        ckpt = resume_from_checkpoint if resume_from_checkpoint is not None else a.resume_from_checkpoint
        if ckpt:
            start_step = self._load_checkpoint(ckpt)
        epoch = start_step // steps_per_epoch                  # fully trained epochs: the sampler is advanced, nothing is replayed
        loader.skip_epochs(epoch)
        skip_batches = (start_step % steps_per_epoch) * ga      # consumed part of the current epoch (skipped before any prefetch work)
        self.state.epoch = self.state.global_step / steps_per_epoch
        for cb in self.callbacks:
            _call(cb, "on_train_begin", a, self.state, self.control)
        t_start = time.time()
        self._t_last_log, self._micro_at_log = time.perf_counter(), self._micro
        self._loss_acc, self._tr_loss_last = [], 0.0
        self.control.should_training_stop = False
        while not self.control.should_training_stop and self.state.global_step < self.state.max_steps and epoch < n_epochs:
            window = []
            for batch in self._prefetching(loader, skip_batches):
                window.append(batch)
                if len(window) < ga:
                    continue
                self.optimizer_window(window, t_start)
                window = []
                if self.state.global_step >= self.state.max_steps or self.control.should_training_stop:
                    break
            if window and len(loader) < ga and self.state.global_step < self.state.max_steps and not self.control.should_training_stop:
                # an epoch shorter than one accumulation window (small dataset / many ranks): HF steps on the last batch of a short epoch
                # (TF trainer.py do_sync_step at steps_in_epoch) - without this the run would end after zero optimizer steps
                self.optimizer_window(window, t_start)
            elif window and self.dp.rank == 0:
                print("[time-r1_amd] epoch %d: %d trailing batch(es) do not fill an accumulation window of %d and are dropped (steps_per_epoch = "
                      "len(loader) // GA, the arithmetic main.py's resume logic uses)" % (epoch, len(window), ga), flush=True)
            skip_batches = 0
            if self.state.global_step < (epoch + 1) * steps_per_epoch:
                break                      # stopped inside the epoch (max_steps / callback): no epoch-end event
            epoch += 1
            self.state.epoch = float(epoch)
            for cb in self.callbacks:
                _call(cb, "on_epoch_end", a, self.state, self.control)
            if a.save_strategy == "epoch":
                self._save_checkpoint()
        for cb in self.callbacks:
            _call(cb, "on_train_end", a, self.state, self.control)
        if self._loss_acc:                 # micro-steps since the last log (logging_steps > 1)
            self._tr_loss_last = float(torch.stack([torch.as_tensor(x).float().reshape(()) for x in self._loss_acc]).mean())
            self._loss_acc = []
        return types.SimpleNamespace(global_step=self.state.global_step, training_loss=self._tr_loss_last, metrics={"train_runtime": time.time() - t_start})

    # ------------------------------------------------------------------------------------------------------ logging / saving
    def log(self, logs, start_time=None):
        """The reference's keys (timer1_trainer.py:784-793: means of the per-micro-step metric lists, then cleared) plus the throughput keys
        SURVEY 5.5 asks this build to emit beside them: `samples_per_sec` / `rollout_tokens_per_sec` (whole job) and the two roofline
        fractions of the step's dominant kernel families (`perf/*`, from HIP-event phase times and the algorithmic work of the shapes run)."""
        metrics = {k: sum(v) / len(v) for k, v in self.flush_metrics().items()}     # reference :784-793 (all ranks are here: see _metrics)
        logs = {**logs, **metrics, **self._throughput_keys()}
        if self.state.epoch is not None:
            logs["epoch"] = round(self.state.epoch, 4)
        self.state.log_history.append({**logs, "step": self.state.global_step})
        for cb in self.callbacks:
            _call(cb, "on_log", self.args, self.state, self.control, logs=logs)
        if self.dp.rank == 0 and not getattr(self.args, "disable_log_print", False):
            print(logs, flush=True)
        self._metrics_store.clear()

    def _throughput_keys(self):
        """Whole-job rates since the previous log.  Wall time is this rank's (ranks step in lock-step through the gradient exchange); generated
        tokens are the gathered completion lengths; phase times are HIP events on the compute stream (read here, after the step was enqueued)."""
        now = time.perf_counter()
        ph = self._clock.drain()
        n_micro = self._micro - self._micro_at_log
        out = {}
        if self._t_last_log is not None and n_micro > 0 and now > self._t_last_log:
            out["samples_per_sec"] = n_micro * self.dp.world / (now - self._t_last_log)
        toks, self._tok_since_log = self._tok_since_log, 0.0
        roll_ms = ph.get("rollout", 0.0)
        if roll_ms > 0 and toks > 0:
            out["rollout_tokens_per_sec"] = toks / (roll_ms * 1e-3)
        if n_micro > 0:
            w = self.core.drain_work()
            dec_ms = w.get("decode_ms_events") or 0.0
            if w.get("decode_bytes") and roll_ms > 0:
                # decode steps stream every decoder + lm_head weight once per step (+ the KV cache): HBM-bound family, peak 8 TB/s
                out["perf/decode_hbm_frac"] = w["decode_bytes"] / ((dec_ms or roll_ms) * 1e-3) / 8.0e12
            mm_ms = ph.get("logps", 0.0) + ph.get("backward", 0.0)
            if w.get("train_flops") and mm_ms > 0:
                # log-prob forwards + backward: MFMA-bound family, peak 2.5 PFLOP/s dense bf16
                out["perf/train_mfma_frac"] = w["train_flops"] / (mm_ms * 1e-3) / 2.5e15
        for k, v in ph.items():
            out["perf/ms_%s" % k] = v / max(n_micro, 1)
            self.phase_ms_total[k] += v
        self._t_last_log, self._micro_at_log = now, self._micro
        return out

    def save_model(self, output_dir=None, _internal_call=False):
        """16-bit weights under transformers key names in one safetensors file (what the reference's ZeRO-3 save gathers, zero3.json:32)."""
        output_dir = output_dir or self.args.output_dir
        if self.dp.rank != 0:
            return
        os.makedirs(output_dir, exist_ok=True)
        from safetensors.torch import save_file
        save_file(self.params.export_hf_state_dict(), os.path.join(output_dir, "model.safetensors"), metadata={"format": "pt"})
        json.dump(dataclasses.asdict(self.cfg), open(os.path.join(output_dir, "timer1_model_config.json"), "w"), indent=1)
        from .config import to_hf_config
        json.dump(to_hf_config(self.cfg), open(os.path.join(output_dir, "config.json"), "w"), indent=1)   # reference: save_pretrained via Trainer
        # HF Trainer._save also writes the processor / tokenizer, so `--model_name_or_path <output_dir>/checkpoint-N` reloads (train_rl_SF.sh, eval)
        sp = getattr(self.processing_class, "save_pretrained", None)
        if sp is not None:
            sp(output_dir)
        json.dump({"eos_token_id": self.cfg.eos_token_id, "pad_token_id": self.cfg.pad_token_id, "do_sample": True,
                   "temperature": float(getattr(self.args, "temperature", 1.0))}, open(os.path.join(output_dir, "generation_config.json"), "w"), indent=1)

    def _save_checkpoint(self):
        d = os.path.join(self.args.output_dir, "checkpoint-%d" % self.state.global_step)
        self.save_model(d)
        if self.dp.rank == 0:
            json.dump({"global_step": self.state.global_step, "max_steps": self.state.max_steps, "epoch": self.state.epoch,
                       "num_train_epochs": self.state.num_train_epochs, "log_history": self.state.log_history,
                       "train_batch_size": self.args.per_device_train_batch_size}, open(os.path.join(d, "trainer_state.json"), "w"), indent=1)
        if not self.args.save_only_model:
            os.makedirs(d, exist_ok=True)
            sd = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in self.optimizer.state_dict().items()}
            sd["rollout_calls"] = self.core.roll.calls
            sd["micro"] = self._micro
            torch.save(sd, os.path.join(d, "optimizer_rank%d.pt" % self.dp.rank))
            if self.ref_model is not None and self.dp.rank == 0:
                # the frozen reference policy is identical on every rank: one copy per checkpoint (15 GB at 7B), not one per rank
                torch.save({"ref_w16": self.ref_model.w16.cpu()}, os.path.join(d, "reference_policy.pt"))
        for cb in self.callbacks:
            _call(cb, "on_save", self.args, self.state, self.control)
        self.dp.barrier()

    def _load_checkpoint(self, d):
        from safetensors.torch import load_file
        self.params.load_hf_state_dict(load_file(os.path.join(d, "model.safetensors")))
        st = json.load(open(os.path.join(d, "trainer_state.json")))
        self.state.global_step = int(st["global_step"])
        self.state.log_history = st.get("log_history", [])
        self.state.epoch = float(st.get("epoch") or 0.0)
        opt = os.path.join(d, "optimizer_rank%d.pt" % self.dp.rank)
        if os.path.exists(opt):
            sd = torch.load(opt, weights_only=False)
            self.params.train.version += 1
            self.optimizer.load_state_dict({k: (v.to(self.ops.device) if torch.is_tensor(v) else v) for k, v in sd.items() if k in ("step", "master", "m", "v", "shard")})
            self.core.roll.calls = sd.get("rollout_calls", 0)
            self._micro = sd.get("micro", 0)
            refp = os.path.join(d, "reference_policy.pt")
            if self.ref_model is not None and os.path.exists(refp):
                self.ref_model.w16.copy_(torch.load(refp, weights_only=False)["ref_w16"].to(self.ops.device))
            elif self.ref_model is not None and "ref_w16" in sd:      # checkpoints written before round 3 embedded it per rank
                self.ref_model.w16.copy_(sd["ref_w16"].to(self.ops.device))
        return self.state.global_step

    def push_to_hub(self, *a, **k):
        raise RuntimeError("no network in this deployment; copy %s instead" % self.args.output_dir)

    def create_model_card(self, *a, **k):
        return None


class TimeR1_Trainer_ft(TimeR1_Trainer):
    """Fine-tune variant: dataset rows carry pre-decoded frames (`video_inputs`, `video_kwargs`), three prompt templates, shaping metrics
    (reference timer1_trainer_ft.py:511-563, :670-691, :789-842; row format finetune.py:594-623)."""
    _is_ft = True

    def _video_inputs(self, example):
        if example.get("video_inputs") is None:
            return super()._video_inputs(example)
        vids = example["video_inputs"]
        while isinstance(vids, (list, tuple)) and vids and isinstance(vids[0], (list, tuple)):
            vids = vids[0]          # the reference's dataset wraps the loaded list once more (finetune.py:598-611)
        kw = example.get("video_kwargs") or {"fps": [VP.FPS]}
        if isinstance(kw, (list, tuple)):
            kw = kw[0]
        vids = [torch.as_tensor(v).float() for v in vids]
        return vids, kw["fps"]
