"""PyTorch-ROCm custom ops over the C ABI: `torch.ops.timer1.*` (SURVEY.md 8b "op-level" seam).

Every op is registered with the dispatcher through `torch.library.custom_op` (schema, CUDA/HIP kernel, fake/meta kernel, autograd
formula that itself calls the matching `*_bwd` op), so the kernels compose with autograd, `torch.library.opcheck` and tracing, and can be
dropped into the REFERENCE's own HF model (`Qwen2_5_VLForConditionalGeneration`, reference src/time_r1/rl/timer1_trainer.py:244-262)
without this repo's engine:

    import time_r1_amd.torch_ops as T
    T.patch_hf_model(model)                      # Qwen2RMSNorm.forward -> timer1::rmsnorm, SwiGLU -> timer1::swiglu (+ optional nn.Linear -> timer1::linear)
    model = Qwen2_5_VLForConditionalGeneration.from_pretrained(path, attn_implementation="timer1_hip")      # after T.register_hf_attention()

Device contract: the implementations are HIP only (`device_types="cuda"`); a CPU tensor raises NotImplementedError from the dispatcher -
there is no fallback.  Kernels are launched on torch's CURRENT stream of the tensor's device; nothing is retained past the call.
Activations are bf16, statistics / gradients of parameters fp32 (cast to the parameter dtype by the wrappers).
"""
from typing import Optional, Tuple

import torch
from torch import Tensor

from .ops import HipOps

BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32
_OPS = {}


def _ops(t: Tensor) -> HipOps:
    d = t.device
    if d.type != "cuda":
        raise NotImplementedError("timer1 ops are HIP-only (got a %s tensor); there is no CPU fallback" % d.type)
    o = _OPS.get(d.index)
    if o is None:
        o = _OPS[d.index] = HipOps("cuda:%d" % (d.index if d.index is not None else torch.cuda.current_device()))
    return o


def _rows(x: Tensor) -> Tensor:
    return x.reshape(-1, x.shape[-1]).contiguous()


_op = lambda name, **k: torch.library.custom_op("timer1::" + name, mutates_args=k.pop("mutates_args", ()), device_types="cuda", **k)  # noqa: E731


# ===================================================================================================================== RMSNorm
# Qwen2RMSNorm (transformers modeling_qwen2_vl.py:96-110): y = w * (x * rsqrt(mean(x^2) + eps)).to(dtype)
@_op("rmsnorm_fwd")
def rmsnorm_fwd(x: Tensor, w: Tensor, eps: float) -> Tuple[Tensor, Tensor]:
    y, rstd, _ = _ops(x).rmsnorm_fwd(x, w, eps)
    return y, rstd


@rmsnorm_fwd.register_fake
def _(x, w, eps):
    return torch.empty_like(x), x.new_empty(x.shape[0], dtype=F32)


@_op("rmsnorm_bwd")
def rmsnorm_bwd(dy: Tensor, x: Tensor, w: Tensor, rstd: Tensor) -> Tuple[Tensor, Tensor]:
    dw = torch.zeros(w.numel(), dtype=F32, device=x.device)
    dx = _ops(x).rmsnorm_bwd(dy, x, w, rstd, dw=dw)
    return dx, dw


@rmsnorm_bwd.register_fake
def _(dy, x, w, rstd):
    return torch.empty_like(x), w.new_empty(w.numel(), dtype=F32)


def _rms_setup(ctx, inputs, output):
    x, w, _ = inputs
    ctx.save_for_backward(x, w, output[1])


def _rms_backward(ctx, dy, _drstd):
    x, w, rstd = ctx.saved_tensors
    dx, dw = torch.ops.timer1.rmsnorm_bwd(dy.contiguous(), x, w, rstd)
    return dx, dw.to(w.dtype), None


rmsnorm_fwd.register_autograd(_rms_backward, setup_context=_rms_setup)


def rmsnorm(x: Tensor, w: Tensor, eps: float = 1e-6) -> Tensor:
    """Any leading shape; last dim = hidden.  Differentiable in x and w."""
    y, _ = torch.ops.timer1.rmsnorm_fwd(_rows(x), w.contiguous(), float(eps))
    return y.view(x.shape)


# ====================================================================================================================== SwiGLU
# Qwen2MLP (modeling_qwen2_vl.py:459-466): down(silu(gate(x)) * up(x)); here on the fused [rows, 2I] = [gate | up] projection output
@_op("swiglu_fwd")
def swiglu_fwd(gu: Tensor) -> Tensor:
    return _ops(gu).swiglu_fwd(gu)


@swiglu_fwd.register_fake
def _(gu):
    return gu.new_empty(gu.shape[0], gu.shape[1] // 2)


@_op("swiglu_bwd")
def swiglu_bwd(da: Tensor, gu: Tensor) -> Tensor:
    return _ops(gu).swiglu_bwd(da, gu)


@swiglu_bwd.register_fake
def _(da, gu):
    return torch.empty_like(gu)


swiglu_fwd.register_autograd(lambda ctx, da: torch.ops.timer1.swiglu_bwd(da.contiguous(), ctx.saved_tensors[0]),
                             setup_context=lambda ctx, inputs, output: ctx.save_for_backward(inputs[0]))


def swiglu(gate_up: Tensor) -> Tensor:
    a = torch.ops.timer1.swiglu_fwd(_rows(gate_up))
    return a.view(*gate_up.shape[:-1], gate_up.shape[-1] // 2)


# ====================================================================================================================== Linear
# bf16 MFMA GEMM (LDS-staged 128x128 / 256x256 tiles): y = x W^T (+ b); dgrad reads W as stored (K-major "NN" form), wgrad accumulates in fp32
@_op("linear_fwd")
def linear_fwd(x: Tensor, w: Tensor, bias: Optional[Tensor]) -> Tensor:
    return _ops(x).gemm_nt(x, w, bias=bias)


@linear_fwd.register_fake
def _(x, w, bias):
    return x.new_empty(x.shape[0], w.shape[0])


@_op("linear_bwd")
def linear_bwd(dy: Tensor, x: Tensor, w: Tensor, need_bias: bool) -> Tuple[Tensor, Tensor, Tensor]:
    o = _ops(x)
    dx = o.gemm_nn(dy, w)
    dw = o.gemm_nt(o.transpose(dy), o.transpose(x), out_f32=True)
    db = torch.zeros(w.shape[0] if need_bias else 0, dtype=F32, device=x.device)
    if need_bias:
        o.colsum_accum(dy, db)
    return dx, dw, db


@linear_bwd.register_fake
def _(dy, x, w, need_bias):
    return torch.empty_like(x), w.new_empty(w.shape, dtype=F32), w.new_empty(w.shape[0] if need_bias else 0, dtype=F32)


def _lin_setup(ctx, inputs, output):
    x, w, b = inputs
    ctx.save_for_backward(x, w)
    ctx.has_bias = b is not None
    ctx.b_dtype = b.dtype if b is not None else None


def _lin_backward(ctx, dy):
    x, w = ctx.saved_tensors
    dx, dw, db = torch.ops.timer1.linear_bwd(dy.contiguous(), x, w, ctx.has_bias)
    return dx, dw.to(w.dtype), (db.to(ctx.b_dtype) if ctx.has_bias else None)


linear_fwd.register_autograd(_lin_backward, setup_context=_lin_setup)


def linear(x: Tensor, w: Tensor, bias: Optional[Tensor] = None) -> Tensor:
    """x [..., K] @ w[N, K]^T; K must be a multiple of 64 (the kernels' K tile)."""
    y = torch.ops.timer1.linear_fwd(_rows(x), w.contiguous(), bias)
    return y.view(*x.shape[:-1], w.shape[0])


# ===================================================================================================================== rotary
# rotate-half RoPE with per-token cos/sin tables [T, hd/2] fp32 (M-RoPE tables come from mrope_table; vision tables from vision_rope_table)
@_op("rope_fwd")
def rope_fwd(x: Tensor, cos: Tensor, sin: Tensor, n_heads: int, head_dim: int, backward: bool) -> Tensor:
    return _ops(x).rope_apply(x, n_heads, head_dim, cos, sin, backward=backward)


@rope_fwd.register_fake
def _(x, cos, sin, n_heads, head_dim, backward):
    return x.new_empty(x.shape[0], n_heads * head_dim)


def _rope_setup(ctx, inputs, output):
    _, cos, sin, ctx.nh, ctx.hd, ctx.bwd = inputs
    ctx.save_for_backward(cos, sin)


def _rope_backward(ctx, dy):        # the rotation is orthogonal: its gradient is the inverse rotation
    cos, sin = ctx.saved_tensors
    return torch.ops.timer1.rope_fwd(dy.contiguous(), cos, sin, ctx.nh, ctx.hd, not ctx.bwd), None, None, None, None, None


rope_fwd.register_autograd(_rope_backward, setup_context=_rope_setup)


def rope(x: Tensor, cos: Tensor, sin: Tensor, n_heads: int, head_dim: int) -> Tensor:
    """x [T, n_heads*head_dim] bf16; cos/sin [T, head_dim/2] fp32."""
    return torch.ops.timer1.rope_fwd(x.contiguous(), cos.contiguous(), sin.contiguous(), n_heads, head_dim, False)


@_op("mrope_table")
def mrope_table(pos3: Tensor, head_dim: int, s0: int, s1: int, s2: int, theta: float) -> Tuple[Tensor, Tensor]:
    return _ops(pos3).mrope_table(pos3, head_dim, (s0, s1, s2), theta)


@mrope_table.register_fake
def _(pos3, head_dim, s0, s1, s2, theta):
    T = pos3.shape[1]
    return pos3.new_empty(T, head_dim // 2, dtype=F32), pos3.new_empty(T, head_dim // 2, dtype=F32)


# =================================================================================================================== attention
# Two-interval mask per query row t: keys kv < pre[t] or lo[t] <= kv <= hi[t].  Causal: pre = 0, lo = 0, hi = t.  Varlen (cu_seqlens):
# lo = segment start, hi = segment end - 1.  Shared-prefix GRPO packing: pre = P for completion rows.  GQA by n_heads / n_kv.
@_op("attn_fwd")
def attn_fwd(q: Tensor, k: Tensor, v: Tensor, pre: Tensor, lo: Tensor, hi: Tensor, n_heads: int, n_kv: int, head_dim: int, scale: float) -> Tuple[Tensor, Tensor]:
    o = _ops(q)
    vt = o.pack_transpose(v, n_kv, n_kv, head_dim)
    out, lse = o.attn_fwd(q, k, vt, pre, lo, hi, n_heads, n_kv, k.shape[0], head_dim, scale, need_lse=True)
    return out, lse


@attn_fwd.register_fake
def _(q, k, v, pre, lo, hi, n_heads, n_kv, head_dim, scale):
    return q.new_empty(q.shape[0], n_heads * head_dim), q.new_empty(n_heads, q.shape[0], dtype=F32)


@_op("attn_bwd")
def attn_bwd(do: Tensor, q: Tensor, k: Tensor, v: Tensor, out: Tensor, lse: Tensor, pre: Tensor, lo: Tensor, hi: Tensor, n_heads: int, n_kv: int,
             head_dim: int, scale: float) -> Tuple[Tensor, Tensor, Tensor]:
    return _ops(q).attn_bwd(q, k, v, out, do, lse, pre, lo, hi, n_heads, n_kv, k.shape[0], head_dim, scale)


@attn_bwd.register_fake
def _(do, q, k, v, out, lse, pre, lo, hi, n_heads, n_kv, head_dim, scale):
    return torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)


def _attn_setup(ctx, inputs, output):
    q, k, v, pre, lo, hi, nh, nkv, hd, scale = inputs
    ctx.save_for_backward(q, k, v, output[0], output[1], pre, lo, hi)
    ctx.cfg = (nh, nkv, hd, scale)


def _attn_backward(ctx, do, _dlse):
    q, k, v, out, lse, pre, lo, hi = ctx.saved_tensors
    dq, dk, dv = torch.ops.timer1.attn_bwd(do.contiguous(), q, k, v, out, lse, pre, lo, hi, *ctx.cfg)
    return dq, dk, dv, None, None, None, None, None, None, None


attn_fwd.register_autograd(_attn_backward, setup_context=_attn_setup)


def attention(q: Tensor, k: Tensor, v: Tensor, pre: Tensor, lo: Tensor, hi: Tensor, n_heads: int, n_kv: int, head_dim: int, scale: Optional[float] = None) -> Tensor:
    """q [T, n_heads*hd], k / v [S, n_kv*hd] (S == T for self-attention over one packed sequence), masks int32 [T]."""
    scale = head_dim ** -0.5 if scale is None else scale
    return torch.ops.timer1.attn_fwd(q.contiguous(), k.contiguous(), v.contiguous(), pre, lo, hi, n_heads, n_kv, head_dim, float(scale))[0]


def causal_masks(T: int, device) -> Tuple[Tensor, Tensor, Tensor]:
    z = torch.zeros(T, dtype=I32, device=device)
    return z, z, torch.arange(T, dtype=I32, device=device)


def varlen_masks(cu_seqlens: Tensor, causal: bool = False) -> Tuple[Tensor, Tensor, Tensor]:
    """flash-attn style cu_seqlens [n+1] -> (pre, lo, hi) for non-causal (ViT) or causal segments."""
    cu = cu_seqlens.to(torch.long)
    T = int(cu[-1])
    idx = torch.arange(T, device=cu.device)
    seg = torch.searchsorted(cu[1:], idx, right=True)
    lo = cu[seg]
    hi = idx if causal else cu[seg + 1] - 1
    return torch.zeros(T, dtype=I32, device=cu.device), lo.to(I32), hi.to(I32)


# ========================================================================================================== vocabulary / loss side
@_op("logp_entropy_fwd")
def logp_entropy_fwd(logits: Tensor, targets: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    return _ops(logits).logp_entropy_fwd(logits, targets)


@logp_entropy_fwd.register_fake
def _(logits, targets):
    R = logits.shape[0]
    return logits.new_empty(R, dtype=F32), logits.new_empty(R, dtype=F32), logits.new_empty(R, dtype=F32)


@_op("logp_bwd")
def logp_bwd(logits: Tensor, targets: Tensor, lse: Tensor, dlogp: Tensor) -> Tensor:
    return _ops(logits).logp_bwd(logits, targets, lse, dlogp, inplace=False)


@logp_bwd.register_fake
def _(logits, targets, lse, dlogp):
    return torch.empty_like(logits)


def _lp_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1], output[2])


def _lp_backward(ctx, dlogp, _dent, _dlse):
    logits, targets, lse = ctx.saved_tensors           # the entropy is a logged metric (reference :473-481 computes it under no_grad)
    return torch.ops.timer1.logp_bwd(logits, targets, lse, dlogp.contiguous().float()), None


logp_entropy_fwd.register_autograd(_lp_backward, setup_context=_lp_setup)


def logp_entropy(logits: Tensor, targets: Tensor) -> Tuple[Tensor, Tensor]:
    """Per-row log-softmax gathered at `targets` + entropy of the row distribution, one pass over [R, V] bf16 logits
    (reference timer1_trainer.py:458-481).  Differentiable in logits through logp."""
    lp, ent, _ = torch.ops.timer1.logp_entropy_fwd(logits.contiguous(), targets.to(I32).contiguous())
    return lp, ent


@_op("lmhead_logp_entropy")
def lmhead_logp_entropy(hn: Tensor, w: Tensor, targets: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """(logp[target], entropy, lse) of softmax(hn @ w^T) per row with the statistics reduced in the GEMM epilogue: no [R, V] logits in HBM
    (forward only - the no-grad reference-policy log-probs of timer1_trainer.py:613-632)."""
    r = _ops(hn).lmhead_lse(hn, w, targets)
    if r is None:
        raise RuntimeError("lmhead_logp_entropy: needs V % 64 == 0, K % 64 == 0, V >= 256 and row-major operands")
    return r


@lmhead_logp_entropy.register_fake
def _(hn, w, targets):
    R = hn.shape[0]
    return hn.new_empty(R, dtype=F32), hn.new_empty(R, dtype=F32), hn.new_empty(R, dtype=F32)


@_op("grpo_loss")
def grpo_loss_op(logp: Tensor, ref_logp: Optional[Tensor], mask: Tensor, adv: Tensor, beta: float, use_grpo: bool, grad_scale: float) -> Tuple[Tensor, Tensor, Tensor]:
    dlogp, out3, row_len, _ = _ops(logp).grpo_loss(logp, ref_logp, mask, adv, beta, use_grpo, grad_scale)
    return out3, dlogp, row_len


@grpo_loss_op.register_fake
def _(logp, ref_logp, mask, adv, beta, use_grpo, grad_scale):
    return logp.new_empty(3), torch.empty_like(logp), logp.new_empty(logp.shape[0])


grpo_loss_op.register_autograd(lambda ctx, dout3, _d1, _d2: (ctx.saved_tensors[0] * dout3[0], None, None, None, None, None, None),
                               setup_context=lambda ctx, inputs, output: ctx.save_for_backward(output[1]))


def grpo_loss(logp: Tensor, ref_logp: Optional[Tensor], completion_mask: Tensor, advantages: Tensor, beta: float, use_grpo: bool) -> Tuple[Tensor, Tensor]:
    """-> (loss, mean k3-KL).  logp fp32 [G, C]; the loss algebra of reference timer1_trainer.py:635-737 (k3 KL, both loss branches) in one
    kernel that also produces dL/dlogp, which the autograd formula hands back."""
    out3, _, _ = torch.ops.timer1.grpo_loss(logp.contiguous(), ref_logp, completion_mask.to(I32).contiguous(), advantages.float().contiguous(),
                                            float(beta), bool(use_grpo), 1.0)
    return out3[0], out3[1]


# ============================================================================================================ sampler / optimizer / video
@_op("sample_tokens", mutates_args=("tokens", "finished"))
def sample_tokens(logits: Tensor, temperature: float, top_k: int, seed: int, step: Tensor, tokens: Tensor, finished: Tensor, eos_id: int, pad_id: int,
                  stop_at_eos: bool) -> None:
    _ops(logits).sample_tokens(logits, temperature, top_k, seed, step, tokens, finished, eos_id, pad_id, stop_at_eos)


@_op("adamw_step", mutates_args=("p32", "m", "v", "g", "p16"))
def adamw_step(p32: Tensor, m: Tensor, v: Tensor, g: Tensor, p16: Tensor, lr: float, beta1: float, beta2: float, eps: float, weight_decay: float, step: int,
               max_norm: float, grad_mult: float) -> Tensor:
    """Global-norm clip + AdamW on flat fp32 master / m / v / grad, bf16 working copy refreshed, grad zeroed.  Returns the pre-clip norm."""
    o = _ops(p32)
    ss = torch.zeros(1, dtype=F32, device=p32.device)
    o.sumsq_accum(g, ss)
    o.adamw_step(p32, m, v, g, p16, lr, beta1, beta2, eps, weight_decay, step, sumsq=ss, max_norm=max_norm, grad_mult=grad_mult, zero_grad=True)
    return ss.sqrt() * grad_mult


@adamw_step.register_fake
def _(p32, m, v, g, p16, lr, beta1, beta2, eps, weight_decay, step, max_norm, grad_mult):
    return p32.new_empty(1)


@_op("video_preprocess")
def video_preprocess(frames_u8: Tensor, out_h: int, out_w: int, k_pad: int) -> Tensor:
    """uint8 [T,3,H,W] -> bf16 [N_v, k_pad] normalised patches in the HF video processor's layout (resize + rescale + normalise + patchify)."""
    return _ops(frames_u8).video_preprocess(frames_u8, (out_h, out_w), k_pad)[0]


@video_preprocess.register_fake
def _(frames_u8, out_h, out_w, k_pad):
    T = (frames_u8.shape[0] + 1) // 2
    return frames_u8.new_empty(T * (out_h // 14) * (out_w // 14), k_pad, dtype=BF16)


OP_NAMES = ["rmsnorm_fwd", "rmsnorm_bwd", "swiglu_fwd", "swiglu_bwd", "linear_fwd", "linear_bwd", "rope_fwd", "mrope_table", "attn_fwd", "attn_bwd",
            "logp_entropy_fwd", "logp_bwd", "lmhead_logp_entropy", "grpo_loss", "sample_tokens", "adamw_step", "video_preprocess"]


# ================================================================================================ dropping the ops into an HF model
class TimeR1RMSNorm(torch.nn.Module):
    """Drop-in for Qwen2RMSNorm / Qwen2_5_VLRMSNorm (same parameter name `weight`, same `variance_epsilon`)."""

    def __init__(self, hidden_size, eps=1e-6, weight=None):
        super().__init__()
        self.weight = weight if weight is not None else torch.nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, hidden_states):
        return rmsnorm(hidden_states, self.weight, self.variance_epsilon)


def hf_attention_forward(module, query, key, value, attention_mask=None, dropout=0.0, scaling=None, is_causal=None, **kwargs):
    """transformers AttentionInterface signature: q [B, H, T, D], k / v [B, Hkv, S, D] -> ([B, T, H, D], None).
    Honours what transformers passes: `is_causal` (decoder: True; the Qwen2-VL / Qwen2.5-VL VISION blocks route through the same function
    with is_causal=False and flash-style `cu_seq_lens_q` -> bidirectional attention inside each frame / window), S > T (`generate` with a
    KV cache: the T new queries are the last T of the S keys, causal among themselves) and refuses what it cannot express (a padding /
    arbitrary additive mask) instead of silently ignoring it.  Reference call site: timer1_trainer.py:452-457 (logprob forward)."""
    B, Hq, T, D = query.shape
    Hkv, S = key.shape[1], key.shape[2]
    if S < T:
        raise NotImplementedError("timer1_hip attention: fewer keys (%d) than queries (%d)" % (S, T))
    causal = is_causal if is_causal is not None else bool(getattr(module, "is_causal", True))
    if attention_mask is not None:
        # transformers builds the plain causal triangle for sdpa-like backends, as an additive float mask (0 = keep) or as a bool mask
        # (True = keep); anything else (left padding, sliding windows, packed documents) cannot be expressed by one (pre, lo, hi) interval per
        # query here.  An explicit mask REPLACES is_causal in transformers' sdpa path, so an all-keep mask over T > 1 queries means bidirectional
        # attention even when the module is causal: it is honoured as such (it equals the triangle only for T == 1).
        keep = attention_mask if attention_mask.dtype == torch.bool else (attention_mask == 0)
        keep = keep.expand(B, -1, T, S).reshape(-1, T, S) if keep.dim() == 4 else keep.reshape(-1, T, S)
        tri = torch.ones(T, S, dtype=torch.bool, device=query.device).tril(S - T) if causal else torch.ones(T, S, dtype=torch.bool, device=query.device)
        full = bool((keep == tri).all())
        if not full and bool(keep.all()):
            full, causal = True, False
    else:
        full = True
    if not full:
        raise NotImplementedError("timer1_hip attention: a non-trivial attention_mask (padding / sliding window) was passed; use "
                                  "attn_implementation='sdpa' for padded batches")
    cu = kwargs.get("cu_seq_lens_q", kwargs.get("cu_seqlens"))
    if cu is not None and S == T:
        pre, lo, hi = varlen_masks(cu.to(query.device), causal=causal)
    elif causal:
        z = torch.zeros(T, dtype=I32, device=query.device)
        pre, lo, hi = z, z, torch.arange(S - T, S, dtype=I32, device=query.device)       # query t sees keys 0 .. (S - T) + t
    else:
        z = torch.zeros(T, dtype=I32, device=query.device)
        pre, lo, hi = z, z, torch.full((T,), S - 1, dtype=I32, device=query.device)
    outs = []
    for b in range(B):
        q2 = query[b].transpose(0, 1).reshape(T, Hq * D)
        k2 = key[b].transpose(0, 1).reshape(S, Hkv * D)
        v2 = value[b].transpose(0, 1).reshape(S, Hkv * D)
        outs.append(attention(q2, k2, v2, pre, lo, hi, Hq, Hkv, D, scaling).view(T, Hq, D))
    return torch.stack(outs, 0), None


def register_hf_attention(name="timer1_hip"):
    """After this, `attn_implementation="timer1_hip"` selects the HIP attention inside any transformers model."""
    from transformers import AttentionInterface
    AttentionInterface.register(name, hf_attention_forward)
    return name


def _is_silu(act):
    name = getattr(act, "__name__", type(act).__name__).lower()
    return "silu" in name and "quick" not in name


def patch_hf_model(model, rmsnorm_modules=True, swiglu_mlp=True, linears=False):
    """Swap the hot modules of an HF Qwen2-VL / Qwen2.5-VL model (bf16, on a HIP device) for the timer1 ops, in place.
    rmsnorm_modules: every *RMSNorm module's forward.  swiglu_mlp: MLPs with gate_proj / up_proj / down_proj compute act(gate)*up through
    timer1::swiglu.  linears: also route nn.Linear layers whose in_features % 64 == 0 through timer1::linear (MFMA GEMM + own dgrad / wgrad)."""
    import types
    n = dict(rmsnorm=0, mlp=0, linear=0)
    for mod in model.modules():
        cls = type(mod).__name__
        if rmsnorm_modules and cls.endswith("RMSNorm") and hasattr(mod, "weight"):
            eps = getattr(mod, "variance_epsilon", getattr(mod, "eps", 1e-6))
            mod.forward = types.MethodType(lambda self, x, _eps=eps: rmsnorm(x, self.weight, _eps), mod)
            n["rmsnorm"] += 1
        elif swiglu_mlp and all(hasattr(mod, a) for a in ("gate_proj", "up_proj", "down_proj")) and _is_silu(getattr(mod, "act_fn", None)):
            def mlp_forward(self, x):
                gu = torch.cat([self.gate_proj(x), self.up_proj(x)], dim=-1)
                return self.down_proj(swiglu(gu))
            mod.forward = types.MethodType(mlp_forward, mod)
            n["mlp"] += 1
        if linears and isinstance(mod, torch.nn.Linear) and mod.in_features % 64 == 0 and mod.out_features % 8 == 0:
            mod.forward = types.MethodType(lambda self, x: linear(x, self.weight, self.bias), mod)
            n["linear"] += 1
    return n
