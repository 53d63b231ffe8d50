"""One GRPO micro-step on device: vision tower once -> shared-prefix rollout -> policy / reference log-probs on the packed
sequence -> loss gradient -> hand-written backward.  Restates the tensor algebra of TimeR1_Trainer.compute_loss
(reference src/time_r1/rl/timer1_trainer.py:512-782) without the G-fold replication.

Host-side pieces that the reference also runs on the host (text decode, reward callbacks, group statistics on G floats)
stay in trainer.py; everything with a FLOP count lives here and goes through ops (HIP kernels).
"""

import numpy as np
import torch

from .model import Engine
from .positions import PackedLayout, rope_index
from .rollout import Rollout

I32 = torch.int32
F32 = torch.float32


class StepState:
    """Per-prompt device state handed between the phases of a micro-step."""
    pass


class GRPOCore:
    def __init__(self, engine: Engine, ref_arena=None, num_generations=8, max_completion_length=200, beta=0.04, use_grpo=False,
                 temperature=1.0, top_k=50, seed=1234, rope_index_mode="hf4", stop_at_eos=False, reuse_prefill=True):
        self.eng = engine
        self.ops = engine.ops
        self.cfg = engine.cfg
        self.ref_arena = ref_arena
        self.G, self.C = int(num_generations), int(max_completion_length)
        self.beta, self.use_grpo = float(beta), bool(use_grpo)
        self.rope_index_mode = rope_index_mode
        self.second_per_grid_t = 1.0
        # The rollout's prefill IS the policy forward over the prompt rows (same weights inside an accumulation window): keep its
        # activations and run the update's policy forward over the G*C completion rows only.
        self.reuse_prefill = bool(reuse_prefill)
        self.roll = Rollout(engine, self.G, self.C, temperature, top_k, seed, stop_at_eos)
        if self.beta != 0.0 and ref_arena is None:
            raise ValueError("beta != 0 needs a reference-policy arena (reference timer1_trainer.py:295-307)")
        # algorithmic work of what was run since the trainer last read it (TimeR1_Trainer.log -> perf/* keys): bytes the decode steps must
        # stream (HBM-bound family) and FLOPs of the log-prob forwards + backward (MFMA-bound family); formulas in DESIGN.md section 4
        self.work = dict(decode_bytes=0.0, train_flops=0.0, decode_ms_events=0.0)

    # ------------------------------------------------------------------------------------------------------- work accounting
    def _llm_flops(self, rows, pairs):
        t = self.cfg.text
        lin = 2.0 * rows * (t.hidden * t.qkv_dim + t.q_dim * t.hidden + 3 * t.hidden * t.intermediate)
        att = 4.0 * pairs * t.head_dim * t.n_heads
        return t.n_layers * (lin + att)

    def _count_decode(self, states):      # states: the prompt lengths P of one batched rollout
        t = self.cfg.text
        wb = 1.0 if self.roll.weight_dtype in ("fp8", "fp8-mfma") else 2.0
        w_bytes = wb * (t.n_layers * (t.hidden * t.qkv_dim + t.q_dim * t.hidden + 3 * t.hidden * t.intermediate) + t.vocab_size * t.hidden)
        G, C = self.G, self.C
        kv = sum(sum(P + G * s for s in range(1, C)) for P in states) * t.kv_dim * 2 * 2.0 * t.n_layers   # prefix once per prompt + every group's suffix
        self.work["decode_bytes"] += (C - 1) * w_bytes + kv
        for e0, e1 in getattr(self.roll, "decode_events", []):
            e1.synchronize()
            self.work["decode_ms_events"] += e0.elapsed_time(e1)
        self.roll.decode_events = []

    def _count_update(self, st, reused_prefill):
        t = self.cfg.text
        P, G, C = st.P, self.G, self.C
        M = P + G * C
        pairs_c = G * (C * P + C * (C + 1) / 2.0)
        pairs_all = P * (P + 1) / 2.0 + pairs_c
        head = 2.0 * G * C * t.vocab_size * t.hidden
        fl = (self._llm_flops(G * C, pairs_c) if reused_prefill else self._llm_flops(M, pairs_all)) + head        # policy forward
        if self.beta != 0.0:
            fl += self._llm_flops(M, pairs_all) + head                                                           # reference-policy forward
        lin_all = self._llm_flops(M, 0.0)
        fl += 2.0 * lin_all + 2.5 * (self._llm_flops(M, pairs_all) - lin_all) + 2.0 * head                       # backward (dgrad + wgrad; flash backward = 5 products)
        self.work["train_flops"] += fl

    # ------------------------------------------------------------------------------------------------------- phase 1
    def prepare(self, input_ids, pixel_values_videos, video_grid_thw):
        """input_ids: 1-D ints (prompt with <|video_pad|> expanded); pixel_values_videos: float [N_v, patch_dim] as produced by
        the HF video processor (reference timer1_trainer.py:547-565); video_grid_thw: [(t, h, w)]."""
        ops, cfg, eng = self.ops, self.cfg, self.eng
        st = StepState()
        ids = np.asarray(input_ids, dtype=np.int64).reshape(-1)
        grid = [tuple(int(x) for x in g) for g in np.asarray(video_grid_thw).reshape(-1, 3)]
        st.P = self.last_P = int(ids.shape[0])
        st.grid = grid
        st.prompt_ids_host = ids
        st.prompt_ids = ops.tensor(ids.astype(np.int32), I32)
        vid_rows = np.nonzero(ids == cfg.video_token_id)[0].astype(np.int32)
        st.vid_rows = ops.tensor(vid_rows, I32)
        # Qwen2.5-VL spaces temporal ids by tokens_per_second * second_per_grid_t (modeling_qwen2_5_vl.py:1043); the reference's logprob
        # forwards omit second_per_grid_ts (timer1_trainer.py:452-457) -> 1 s per grid step, which is also what its default FPS=2 gives
        # the rollout (temporal_patch_size / fps = 1.0), so one rule serves both phases here.
        interval = int(cfg.tokens_per_second * self.second_per_grid_t) if cfg.vision.variant == "qwen2_5_vl" else 1
        st.pos3_prompt, st.delta = rope_index(ids, grid, cfg.video_token_id, cfg.image_token_id, cfg.vision.spatial_merge_size,
                                              mode=self.rope_index_mode, time_interval=interval)
        v = cfg.vision
        pix = torch.as_tensor(pixel_values_videos)
        n_vid_tokens = sum(t * h * w for t, h, w in grid) // v.merge_unit
        assert n_vid_tokens == vid_rows.shape[0], "video pad tokens (%d) != merged patches (%d)" % (vid_rows.shape[0], n_vid_tokens)
        if pix.dim() == 2 and pix.shape[1] == v.patch_dim_padded and pix.dtype == ops.act_dtype and pix.device == ops.device:
            pp = pix        # already staged on the device in the kernels' layout (K padded to a multiple of 64)
        else:
            assert pix.dim() == 2 and pix.shape[1] == v.patch_dim, pix.shape
            pp = ops.zeros(pix.shape[0], v.patch_dim_padded)
            pp[:, : v.patch_dim] = pix.to(pp.device).to(pp.dtype)
        st.feats, st.vis_perm = eng.vit_features(pp, grid)        # frozen blocks: once per prompt (reference: 3 x G times)
        st.vid_embeds, st.merger_ctx = eng.merger_fwd(eng.params.train, st.feats, save=True, perm=st.vis_perm)
        return st

    # ------------------------------------------------------------------------------------------------------- phase 2
    def rollout(self, st):
        tokens, lay = self.roll.generate(self.eng.params.train, st.prompt_ids, st.vid_embeds, st.vid_rows, st.pos3_prompt, st.delta,
                                         save_prefill=self.reuse_prefill)
        st.layout = lay
        st.completion_ids = tokens        # int32 [G, C] on device
        st.prefill = self.roll.last_prefill[0] if self.reuse_prefill else None
        st.sample_logp = self.roll.last_sample_logp[0] if self.roll.last_sample_logp is not None else None
        self._pending_decode = (getattr(self, "_pending_decode", []) + [[st.P]])[-64:]
        return tokens

    def rollout_many(self, states):
        """Decode the prompts of one accumulation window together (weights are constant inside it): every weight byte streamed
        from HBM serves len(states)*G rows. Sampling streams stay per prompt, so tokens equal the one-by-one rollout's."""
        outs = self.roll.generate_many(self.eng.params.train, [(st.prompt_ids, st.vid_embeds, st.vid_rows, st.pos3_prompt, st.delta) for st in states],
                                       save_prefill=self.reuse_prefill)
        for b, (st, (tokens, lay)) in enumerate(zip(states, outs)):
            st.layout, st.completion_ids = lay, tokens
            st.prefill = self.roll.last_prefill[b] if self.reuse_prefill else None
            st.sample_logp = self.roll.last_sample_logp[b] if self.roll.last_sample_logp is not None else None
        self._pending_decode = (getattr(self, "_pending_decode", []) + [[st.P for st in states]])[-64:]
        return [st.completion_ids for st in states]

    def drain_work(self):
        """-> the work counters since the last call (decode events are resolved here, i.e. when the trainer logs - never inside a step)."""
        for states in getattr(self, "_pending_decode", []):
            self._count_decode(states)
        self._pending_decode = []
        w, self.work = self.work, dict.fromkeys(self.work, 0.0)
        return w

    # ------------------------------------------------------------------------------------------------------- phase 3
    def _packed_inputs(self, st):
        ops, lay = self.ops, st.layout
        comp = st.completion_ids
        st.ids_packed = torch.cat([st.prompt_ids, comp.reshape(-1)])
        st.pos3 = ops.tensor(lay.positions(st.pos3_prompt, st.delta), I32)
        t = self.cfg.text
        st.cos, st.sin = ops.mrope_table(st.pos3, t.head_dim, t.mrope_section, t.rope_theta)
        st.masks = [ops.tensor(a, I32) for a in lay.masks()]
        st.pred_rows = ops.tensor(lay.pred_rows(), I32)
        G, C = lay.G, lay.C
        st.targets = torch.cat([comp[:, 0], comp[:, 1:].reshape(-1)]).contiguous()
        # permutation between [G, C] order and pred_rows order
        perm = np.concatenate([np.arange(G) * C, (np.arange(G)[:, None] * C + 1 + np.arange(C - 1)[None, :]).reshape(-1)])
        st.perm = torch.as_tensor(perm, dtype=torch.long, device=comp.device)          # pred order -> flat (g, s) index
        st.inv_perm = torch.empty_like(st.perm)
        st.inv_perm[st.perm] = torch.arange(G * C, device=comp.device)

    def _to_gc(self, st, x_pred_order):
        return x_pred_order[st.inv_perm].view(st.layout.G, st.layout.C)

    def forward_logps(self, st):
        """Policy log-probs / entropy (activations saved for backward) and reference log-probs (no grad)."""
        eng, ops = self.eng, self.ops
        self._packed_inputs(st)
        tr = eng.params.train
        pf = getattr(st, "prefill", None)
        self._count_update(st, pf is not None and pf[0] is not None)
        if pf is not None and pf[0] is not None:
            # continuation: only the G*C completion rows; the prompt rows' activations and K/V come from the rollout's prefill
            P, M = st.P, st.layout.M
            pctx, kv = pf
            if pctx.get("stash"):            # large sequences: this prompt's prefill rows were parked; the one full buffer set is free now
                pctx = eng.unstash_ctx(pctx, P, M)
            hc = ops.gather_rows(tr.w("embed"), st.ids_packed[P:].contiguous())
            cmask = [m[P:].contiguous() for m in st.masks]
            hLc, cctx = eng.llm_fwd(tr, hc, st.cos[P:].contiguous(), st.sin[P:].contiguous(), cmask, save=True, kv_cache=kv, row0=P,
                                    bufs=pctx.get("bufs"))
            st.llm_ctx = eng.merge_ctx(pctx, cctx, kv, st.masks, st.cos, st.sin, M)
            # the head only needs the last prompt row (it predicts every group's first completion token) and the completion rows
            hL = ops.zeros(M, hLc.shape[1])
            hL[P:] = hLc
            hL[P - 1:P] = pctx["h_last"][P - 1:P]
            st.prefill = None
        else:
            h0 = eng.embed(tr, st.ids_packed, st.vid_embeds, st.vid_rows)
            hL, st.llm_ctx = eng.llm_fwd(tr, h0, st.cos, st.sin, st.masks, save=True, tail_from=eng.tail_rows_from(st.P, st.layout.M))    # the head reads rows >= P - 1 only (pred_rows)
        logp, ent, st.head_ctx = eng.head_fwd(tr, hL, st.pred_rows, st.targets, save=True)
        st.logp = self._to_gc(st, logp).contiguous()
        st.entropy = self._to_gc(st, ent).contiguous()
        st.ref_logp = None
        if self.beta != 0.0:
            ra = self.ref_arena
            ref_vid, _ = eng.merger_fwd(ra, st.feats, save=False, perm=st.vis_perm)
            h0r = eng.embed(ra, st.ids_packed, ref_vid, st.vid_rows)
            hLr, _ = eng.llm_fwd(ra, h0r, st.cos, st.sin, st.masks, save=False, tail_from=eng.tail_rows_from(st.P, st.layout.M))
            rlogp, _, _ = eng.head_fwd(ra, hLr, st.pred_rows, st.targets, save=False)
            st.ref_logp = self._to_gc(st, rlogp).contiguous()

    # ------------------------------------------------------------------------------------------------------- phase 4
    def loss_backward(self, st, completion_mask, advantages, grad_scale=1.0, grad_sync=None, tok_weight=None):
        """completion_mask int32 [G, C], advantages fp32 [G] (device). Accumulates grads into the trainable arena.
        grad_sync: a dist.GradSync in its begin() state when this is the LAST micro-step of the accumulation window - parameter ranges
        are handed to the all-reduce as soon as their gradients are final, overlapping the exchange with the rest of the backward.
        Returns (out3 = [loss, mean kl, sum mask], row_len [G]) as device tensors."""
        eng, ops = self.eng, self.ops
        tr = eng.params.train
        hook = None
        if grad_sync is not None and grad_sync.active:
            hook = lambda i: grad_sync.ready(*tr.range_of("l%d." % i))
        dlogp, out3, row_len, _ = ops.grpo_loss(st.logp, st.ref_logp, completion_mask, advantages, self.beta, self.use_grpo, grad_scale)
        if tok_weight is not None:
            # optional truncated importance weight rho[g, t] (a constant) on the advantage term, for completions drawn from a quantised sampling
            # policy: l = -rho * A + beta * kl = l_plain + (1 - rho) * A, same normalisation as the kernel.  None (the default) leaves the
            # reference algebra untouched - this branch is then not executed at all.
            m = completion_mask.to(torch.float32)
            w = (m / row_len.reshape(-1, 1).clamp(min=1.0) / float(st.layout.G)) if self.use_grpo else (m / out3[2].clamp(min=1.0))
            corr = (1.0 - tok_weight.to(torch.float32)) * advantages.reshape(-1, 1).to(torch.float32) * w
            dlogp = dlogp + (corr * float(grad_scale)).to(dlogp.dtype)
            out3 = out3.clone()
            out3[0] = out3[0] + corr.sum().to(out3.dtype)
        dl_pred = dlogp.reshape(-1)[st.perm].contiguous()
        dh = eng.head_bwd(st.head_ctx, dl_pred, st.layout.G)
        if hook is not None and not self.cfg.text.tie_word_embeddings:
            grad_sync.ready(*tr.range_of("norm"))          # final norm + untied lm_head gradients are complete after the head backward
            grad_sync.ready(*tr.range_of("lm_head"))
        dh0 = eng.llm_bwd(st.llm_ctx, dh, on_layer_done=hook)
        ids_g = st.ids_packed.clone()
        ids_g[st.vid_rows.long()] = -1
        dvid = eng.embed_bwd(dh0, ids_g, st.vid_rows)
        eng.merger_bwd(st.merger_ctx, dvid)
        st.llm_ctx = st.head_ctx = st.merger_ctx = None
        return out3, row_len


def eos_mask(completion_ids, eos_token_id):
    """completion_mask[g, t] = 1 for t <= first EOS (EOS kept), all ones when there is none (reference timer1_trainer.py:580-590).
    completion_ids: numpy int [G, C] -> numpy int32 [G, C]."""
    G, C = completion_ids.shape
    is_eos = completion_ids == eos_token_id
    eos_idx = np.full(G, C, dtype=np.int64)
    has = is_eos.any(1)
    eos_idx[has] = is_eos.argmax(1)[has]
    return (np.arange(C)[None, :] <= eos_idx[:, None]).astype(np.int32)


def group_advantages(rewards_per_func, num_generations):
    """rewards_per_func: torch fp32 [B*G, n_funcs] -> (rewards, advantages, std) exactly as reference timer1_trainer.py:700-712
    (sum over funcs, per-group mean, UNBIASED std, eps 1e-4)."""
    rewards = rewards_per_func.sum(dim=1)
    mean = rewards.view(-1, num_generations).mean(dim=1).repeat_interleave(num_generations, dim=0)
    std = rewards.view(-1, num_generations).std(dim=1).repeat_interleave(num_generations, dim=0)
    adv = (rewards - mean) / (std + 1e-4)
    return rewards, adv, std
