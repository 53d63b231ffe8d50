"""Rollout: G sampled completions per prompt, replacing `unwrapped_model.generate(..., num_return_sequences=G)`
(reference src/time_r1/rl/timer1_trainer.py:568-578, GenerationConfig at :371-377).

MI355X-first structure: the vision tower and the ~3.4k-token prompt are processed ONCE (the reference replicates
them G times), the prompt's K/V live once in the cache and are shared by all G rows through the two-interval attention mask, and
each decode step is a batch of single-token rows whose GEMMs take the HBM-streaming skinny kernel.  Decode is weight-bandwidth
bound, so the prompts of one gradient-accumulation window (weights are constant inside it) are decoded TOGETHER: with GA = 2 and
G = 8 every weight byte read from HBM serves 16 rows instead of 8.  Attention stays per prompt (own prefix, own KV cache).
"""
import os

import numpy as np
import torch

from .positions import PackedLayout

I32 = torch.int32


class KVCache:
    """Per layer ONE K buffer [B*s_cap, kv_dim] and ONE V^T buffer [kv_dim, B*s_cap] for the B prompts decoded together; prompt b owns
    slots [b*s_cap, (b+1)*s_cap) and sees them through views, so a single fused kernel appends the new K/V of all B*G rows."""

    def __init__(self, ops, n_layers, kv_dim, s_cap, B):
        self.s_cap, self.B = s_cap, B
        self.k = [ops.zeros(B * s_cap, kv_dim) for _ in range(n_layers)]
        self.vt = [ops.zeros(kv_dim, B * s_cap) for _ in range(n_layers)]

    def views(self, b):
        a, e = b * self.s_cap, (b + 1) * self.s_cap
        return [(k[a:e], vt[:, a:e]) for k, vt in zip(self.k, self.vt)]


class Rollout:
    def __init__(self, engine, num_generations, max_completion_length, temperature=1.0, top_k=50, seed=1234, stop_at_eos=False):
        self.eng = engine
        self.G, self.C = int(num_generations), int(max_completion_length)
        self.temperature, self.top_k, self.seed, self.stop_at_eos = float(temperature), int(top_k or 0), int(seed), bool(stop_at_eos)
        self._cache = None
        self.calls = 0
        self.force_nsplit = 0         # > 0: fixed split count of the decode attention (A/B runs)
        self.native_decode = True     # one native call per decode step (HipOps); False = op-by-op from the host (tests compare the two)
        self.weight_dtype = "bf16"    # "fp8" / "fp8-mfma": decode GEMMs read an e4m3 copy of the decoder matrices + lm_head (row scales), re-quantised
        #                               per call; "fp8" converts the codes to bf16 in registers (W8A16), "fp8-mfma" feeds them to the fp8 matrix
        #                               instruction with block-quantised e4m3 activations (W8A8, BASELINE config "CDNA4 fp8 MFMA")
        self._w8 = None
        self.fp8_keep_bf16 = ()       # matrices of the fp8 sampling policy that stay bf16: any of "qkv", "o", "gu", "down", "lm_head" (drift study)
        self.track_logp = False       # also record, per drawn token, its full-softmax log-prob under the logits it was SAMPLED from (one more pass
        #                               over the step's L2-resident logits): the drift of a quantised sampling policy against the bf16 policy of the
        #                               update is then a logged number (trainer metric rollout_logp_drift), and an importance weight is available
        self.last_sample_logp = None

    def _kv(self, B, s_cap):
        t = self.eng.cfg.text
        c = self._cache
        if c is None or c.s_cap != s_cap or c.B != B:
            self._cache = None      # release before re-allocating
            c = self._cache = KVCache(self.eng.ops, t.n_layers, t.kv_dim, s_cap, B)
        return c

    MATS = ("qkv.w", "o.w", "gu.w", "down.w")
    QBITS = {"qkv": 1, "o": 2, "gu": 4, "down": 8, "lm_head": 16}

    @staticmethod
    def cap_nsplit(nsplit, n_prompts, G, n_heads, n_kv, n_cus):
        """Split count of the decode attention such that (prompts x kv heads x 64-row query tiles) x splits stays within ONE round of blocks (a split-KV block
        takes a whole CU); never below 2 (the split-KV launch is what makes a decode step), unchanged when the launch is not split."""
        if nsplit <= 1:
            return nsplit
        groups = n_prompts * n_kv * ((G * (n_heads // n_kv) + 63) // 64)
        return max(2, min(nsplit, n_cus // max(1, groups)))

    def _n_cus(self):
        dev = getattr(self.eng.ops, "device", None)
        if dev is not None and torch.device(dev).type == "cuda":
            return int(torch.cuda.get_device_properties(dev).multi_processor_count)
        return 256

    def fp8_mask(self):
        keep = set(self.fp8_keep_bf16 or ())
        assert keep <= set(self.QBITS), "fp8_keep_bf16: unknown matrix in %r" % (keep,)
        return sum(b for n, b in self.QBITS.items() if n not in keep)

    def _quantize(self, arena, w_lm):
        """fp8 copies of the decode weights for THIS rollout (the weights change every optimizer step).  Only the sampling policy sees
        them: prefill, log-probs and the update keep bf16.  Buffers are allocated once and reused."""
        ops, t = self.eng.ops, self.eng.cfg.text
        if self._w8 is None:
            shapes = {"qkv.w": (t.qkv_dim, t.hidden), "o.w": (t.hidden, t.q_dim), "gu.w": (2 * t.intermediate, t.hidden), "down.w": (t.hidden, t.intermediate)}
            self._w8 = dict(layers=[{n: (torch.empty(*shapes[n], dtype=torch.uint8, device=ops.device), ops.empty(shapes[n][0], dtype=torch.float32))
                                     for n in self.MATS} for _ in range(t.n_layers)],
                            lm=(torch.empty(*w_lm.shape, dtype=torch.uint8, device=ops.device), ops.empty(w_lm.shape[0], dtype=torch.float32)))
        qm = self.fp8_mask()
        for i, L in enumerate(self._w8["layers"]):
            for n in self.MATS:
                if qm & self.QBITS[n[:-2]]:
                    ops.quantize_fp8_rows(arena.w("l%d.%s" % (i, n)), q=L[n][0], scale=L[n][1])
        if qm & 16:
            ops.quantize_fp8_rows(w_lm, q=self._w8["lm"][0], scale=self._w8["lm"][1])
        return self._w8

    def generate(self, arena, prompt_ids, vid_embeds, vid_rows, prompt_pos3, delta, save_prefill=False):
        """One prompt. prompt_ids: int32 device tensor [P]; prompt_pos3: numpy [3, P]; returns (tokens int32 [G, C] on device, layout)."""
        return self.generate_many(arena, [(prompt_ids, vid_embeds, vid_rows, prompt_pos3, delta)], save_prefill)[0]

    def generate_many(self, arena, items, save_prefill=False):
        """items: list of (prompt_ids, vid_embeds, vid_rows, prompt_pos3, delta), decoded together. Returns [(tokens [G, C], layout)].
        save_prefill: keep the prompt rows' activations so the policy forward of the update can skip the prompt (same weights).
        Each prompt keeps its own sampling stream (seed advances per prompt), so results do not depend on how prompts are batched."""
        eng, ops, cfg = self.eng, self.eng.ops, self.eng.cfg
        t = cfg.text
        G, C, B = self.G, self.C, len(items)
        qd, kvd, hd, half = t.q_dim, t.kv_dim, t.head_dim, t.head_dim // 2
        scale = hd ** -0.5
        w_lm = eng.params.lm_head_w(arena)
        steps = ops.tensor(np.arange(C, dtype=np.int32), I32)
        tokens_all = ops.zeros(B * G, C, dtype=I32)
        finished_all = ops.zeros(B * G, dtype=I32)
        slogp_all = ops.zeros(B * G, C, dtype=torch.float32) if self.track_logp else None
        per = []
        cos_rows, sin_rows = [], []
        lays = [PackedLayout(int(it[0].shape[0]), G, C) for it in items]
        cache = self._kv(B, max(l.S_cap for l in lays))
        for b, (prompt_ids, vid_embeds, vid_rows, prompt_pos3, delta) in enumerate(items):
            P = int(prompt_ids.shape[0])
            lay = lays[b]
            kv_views = cache.views(b)
            seed = self.seed + 7919 * self.calls
            self.calls += 1
            # ---- prefill (prompt once, K/V written straight into the cache)
            pos_p = ops.tensor(np.ascontiguousarray(prompt_pos3.astype(np.int32)), I32)
            cos, sin = ops.mrope_table(pos_p, t.head_dim, t.mrope_section, t.rope_theta)
            masks = [ops.tensor(a, I32) for a in lay.prompt_masks()]
            h = eng.embed(arena, prompt_ids, vid_embeds, vid_rows)
            # save_prefill: the prompt rows' activations go straight into [P + G*C, .] buffers that the update's continuation forward completes
            bufs, is_stash = eng.alloc_ctx_bufs(lay.M, slot=b, prefill_rows=P) if save_prefill else (None, False)
            # (only the last prompt row's output is read: first-token logits here, the first prediction row of the update later)
            hL, pctx = eng.llm_fwd(arena, h, cos, sin, masks, save=save_prefill, kv_cache=kv_views, bufs=bufs, tail_from=eng.tail_rows_from(P, lay.M))
            if is_stash:
                pctx["stash"] = True
            hn, _, _ = ops.rmsnorm_fwd(hL[P - 1:P], arena.w("norm"), t.rms_eps, need_rstd=False)
            logits = ops.gemm_nt(hn, w_lm)  # [1, V]
            tokens = tokens_all[b * G:(b + 1) * G]
            finished = finished_all[b * G:(b + 1) * G]
            ops.sample_tokens(logits.expand(G, logits.shape[1]), self.temperature, self.top_k, seed, steps[0:1], tokens, finished,
                              cfg.eos_token_id, cfg.pad_token_id, self.stop_at_eos)
            if slogp_all is not None:      # (the first token is drawn from the bf16 prefill's logits in every mode)
                slogp_all[b * G:(b + 1) * G, 0] = ops.logp_entropy_fwd(logits.expand(G, logits.shape[1]).contiguous(), tokens[:, 0].contiguous())[0]
            # ---- per-step tables (positions, slots, masks) built once
            comp_pos = (P + delta + np.arange(C, dtype=np.int64))
            pos_c = ops.tensor(np.repeat(comp_pos[None, :], 3, 0).astype(np.int32), I32)            # [3, C]
            cos_c, sin_c = ops.mrope_table(pos_c, t.head_dim, t.mrope_section, t.rope_theta)          # [C, hd/2]
            cos_rows.append(cos_c.view(C, 1, half).expand(C, G, half))
            sin_rows.append(sin_c.view(C, 1, half).expand(C, G, half))
            slots_all = ops.tensor(np.stack([lay.completion_slots(s) for s in range(C)]), I32)       # [C, G]
            pre_d, lo_d, _ = [ops.tensor(a, I32) for a in lay.decode_masks(0)]
            nsplit = max(1, min(28, ((P + 63) // 64 + 3) // 2))
            if self.force_nsplit:                            # tuning hook (A/B runs)
                nsplit = int(self.force_nsplit)
            per.append(dict(lay=lay, kv=kv_views, prefill_ctx=pctx, seed=seed, tokens=tokens, finished=finished, slots=slots_all, pre=pre_d, lo=lo_d, nsplit=nsplit))
        cos_all = torch.cat(cos_rows, 1).contiguous()      # [C, B*G, half]
        sin_all = torch.cat(sin_rows, 1).contiguous()
        pre_all = torch.cat([st["pre"] for st in per]).contiguous()
        lo_all = torch.cat([st["lo"] for st in per]).contiguous()
        hi_all = torch.cat([st["slots"] for st in per], 1).contiguous()        # [C, B*G]: hi of step s = the slot just appended (cache-local)
        nsplit = max(st["nsplit"] for st in per)
        # ONE round of blocks: a split-KV block takes a whole CU (145 KB of LDS), so (prompts x kv heads x 64-row query tiles) x splits above the CU count runs
        # as two rounds - 32 decode rows at 7B = 16 groups x 27 splits = 432 blocks took 25.1 us per layer at step 1 against 19.8 with 14 splits (round 5)
        if not self.force_nsplit:
            nsplit = self.cap_nsplit(nsplit, B, G, t.n_heads, t.n_kv_heads, self._n_cus())
        abs_slots = torch.cat([st["slots"] + b * cache.s_cap for b, st in enumerate(per)], 1).contiguous()   # [C, B*G] into the unified cache
        R = B * G
        fused = R <= 64        # the fused decode kernels hold all rows of a step in one MFMA column block set
        native = fused and self.native_decode and hasattr(ops, "decode_step")
        w8 = None
        qmask = self.fp8_mask()
        a8 = self.weight_dtype == "fp8-mfma"
        if self.weight_dtype in ("fp8", "fp8-mfma"):
            assert fused and t.hidden % 128 == 0 and t.intermediate % 128 == 0 and t.q_dim % 128 == 0, "fp8 decode: <= 64 rows, K % 128 == 0"
            w8 = self._quantize(arena, w_lm)
        if native:
            # whole decode step enqueued by ONE native call (csrc/decode.hip): the host stays ahead of the GPU, no idle gaps between kernels
            def layer_tensors(i):
                if w8 is None:
                    return [arena.w("l%d.%s" % (i, n)) for n in ("ln1", "qkv.w", "qkv.b", "o.w", "ln2", "gu.w", "down.w")] + [cache.k[i], cache.vt[i]]
                Q = w8["layers"][i]

                def mat(n):      # fp8 codes, or the bf16 weight itself where the mask keeps this matrix in bf16
                    return Q[n][0] if qmask & self.QBITS[n[:-2]] else arena.w("l%d.%s" % (i, n))
                return [arena.w("l%d.ln1" % i), mat("qkv.w"), arena.w("l%d.qkv.b" % i), mat("o.w"), arena.w("l%d.ln2" % i), mat("gu.w"), mat("down.w"),
                        cache.k[i], cache.vt[i], Q["qkv.w"][1], Q["o.w"][1], Q["gu.w"][1], Q["down.w"][1]]
            plan = ops.decode_plan([layer_tensors(i) for i in range(t.n_layers)], t.hidden, t.n_heads, t.n_kv_heads, hd, t.intermediate, t.vocab_size,
                                   R, B, cache.s_cap, nsplit, a8=a8, qmask=qmask)
            embed_p, norm_p = arena.w("embed").data_ptr(), arena.w("norm").data_ptr()
            lm_p = w_lm.data_ptr() if w8 is None else ((w8["lm"][0] if qmask & 16 else w_lm).data_ptr(), w8["lm"][1].data_ptr())
            cos_p, sin_p, slot_p, hi_p = cos_all.data_ptr(), sin_all.data_ptr(), abs_slots.data_ptr(), hi_all.data_ptr()
            pre_p, lo_p = pre_all.data_ptr(), lo_all.data_ptr()
            ids_buf = ops.zeros(R, dtype=I32)
            ids_p = ids_buf.data_ptr()

        timed = torch.device(ops.device).type == "cuda" if getattr(ops, "device", None) is not None else False
        if timed:          # HIP events around the decode loop (read by the trainer's log(), never waited for here)
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record()
        host_plan = ops.attn_plan(G, t.n_heads, t.n_kv_heads, B) if (not native and fused and hasattr(ops, "attn_plan")) else None
        for s in range(C - 1):
            if native:
                if s == 0:
                    ids_buf.copy_(tokens_all[:, 0])          # later steps: the sampler wrote the drawn tokens into ids_buf itself (next_ids)
                logits = ops.decode_step(plan, embed_p, norm_p, lm_p, ids_p, cos_p + s * R * half * 4, sin_p + s * R * half * 4, slot_p + s * R * 4,
                                         pre_p, lo_p, hi_p + s * R * 4, t.rms_eps, scale)
                # all prompts of the window in ONE sampler launch set; every prompt keeps its own Philox stream (seed_b = seed_0 + 7919 b)
                ops.sample_tokens(logits, self.temperature, self.top_k, per[0]["seed"], steps[s + 1:s + 2], tokens_all, finished_all, cfg.eos_token_id,
                                  cfg.pad_token_id, self.stop_at_eos, group_rows=G, seed_stride=7919, next_ids=ids_buf)
                if slogp_all is not None:
                    slogp_all[:, s + 1] = ops.logp_entropy_fwd(logits, tokens_all[:, s + 1].contiguous())[0]
                continue
            ids_s = tokens_all[:, s].contiguous()
            cs, sn = cos_all[s], sin_all[s]
            h = ops.gather_rows(arena.w("embed"), ids_s)
            for i in range(t.n_layers):
                p = "l%d." % i
                Q = w8["layers"][i] if w8 is not None else None
                q8 = (lambda n: Q is not None and bool(qmask & self.QBITS[n]))        # this matrix of the sampling policy is fp8
                if q8("qkv"):
                    qkv = ops.gemm_w8(h, Q["qkv.w"][0], Q["qkv.w"][1], lnw=arena.w(p + "ln1"), eps=t.rms_eps, bias=arena.w(p + "qkv.b"), a8=a8)
                elif fused and hd % 32 == 0:      # rmsnorm + q/k/v projection + M-RoPE + KV append: one launch (same choice as csrc/decode.hip)
                    qkv = None
                    q = ops.norm_gemm_qkv(h, arena.w(p + "ln1"), t.rms_eps, arena.w(p + "qkv.w"), arena.w(p + "qkv.b"), cs, sn, cache.k[i], cache.vt[i],
                                          abs_slots[s], t.n_heads, t.n_kv_heads, hd)
                elif fused:      # rmsnorm folded into the projection's operand load (one launch instead of two)
                    qkv = ops.norm_gemm(h, arena.w(p + "ln1"), t.rms_eps, arena.w(p + "qkv.w"), bias=arena.w(p + "qkv.b"))
                else:
                    xn, _, _ = ops.rmsnorm_fwd(h, arena.w(p + "ln1"), t.rms_eps, need_rstd=False)
                    qkv = ops.gemm_nt(xn, arena.w(p + "qkv.w"), bias=arena.w(p + "qkv.b"))
                if qkv is not None:
                    q = ops.decode_qkv_post(qkv, cs, sn, cache.k[i], cache.vt[i], abs_slots[s], t.n_heads, t.n_kv_heads, hd)
                # one launch for all prompts of the window: problem b = rows [b*G,(b+1)*G) over cache slots [b*s_cap, (b+1)*s_cap)
                # the per-step tile plan, exactly as the native step drives it (layer 0 publishes, the others read): both paths then run the
                # same kernels and sample the same tokens
                pk = {}
                if host_plan is not None and nsplit > 1:
                    pk = dict(plan=host_plan, plan_mode=1 if i == 0 else 2)
                o, _ = ops.attn_fwd(q, cache.k[i], cache.vt[i], pre_all, lo_all, hi_all[s], t.n_heads, t.n_kv_heads, cache.s_cap, hd, scale,
                                    nsplit=nsplit, need_lse=False, n_batch=B, kv_batch_slots=cache.s_cap, **pk)
                h2 = ops.gemm_w8(o, Q["o.w"][0], Q["o.w"][1], residual=h, a8=a8) if q8("o") else ops.gemm_nt(o, arena.w(p + "o.w"), residual=h)
                if q8("gu"):
                    a = ops.gemm_w8(h2, Q["gu.w"][0], Q["gu.w"][1], lnw=arena.w(p + "ln2"), eps=t.rms_eps, glu=True, a8=a8)
                elif fused:      # rmsnorm -> gate/up projection -> SwiGLU in one launch; the [R, 2I] intermediate never reaches HBM
                    a = ops.norm_gemm(h2, arena.w(p + "ln2"), t.rms_eps, arena.w(p + "gu.w"), glu=True)
                else:
                    xn2, _, _ = ops.rmsnorm_fwd(h2, arena.w(p + "ln2"), t.rms_eps, need_rstd=False)
                    a = ops.swiglu_fwd(ops.gemm_nt(xn2, arena.w(p + "gu.w")))
                if q8("down"):
                    h = ops.gemm_w8(a, Q["down.w"][0], Q["down.w"][1], residual=h2, a8=a8)
                elif fused and R >= 16 and t.intermediate >= 8192:      # same kernel choice as csrc/decode.hip (bitwise-equal paths)
                    h = ops.gemm_skinny_fixup(a, arena.w(p + "down.w"), residual=h2)
                else:
                    h = ops.gemm_nt(a, arena.w(p + "down.w"), residual=h2)
            if w8 is not None and qmask & 16:
                logits = ops.gemm_w8(h, w8["lm"][0], w8["lm"][1], lnw=arena.w("norm"), eps=t.rms_eps, a8=a8)
            elif fused:
                logits = ops.norm_gemm(h, arena.w("norm"), t.rms_eps, w_lm)
            else:
                hn, _, _ = ops.rmsnorm_fwd(h, arena.w("norm"), t.rms_eps, need_rstd=False)
                logits = ops.gemm_nt(hn, w_lm)
            for b, st in enumerate(per):
                ops.sample_tokens(logits[b * G:(b + 1) * G], self.temperature, self.top_k, st["seed"], steps[s + 1:s + 2], st["tokens"],
                                  st["finished"], cfg.eos_token_id, cfg.pad_token_id, self.stop_at_eos)
            if slogp_all is not None:
                slogp_all[:, s + 1] = ops.logp_entropy_fwd(logits.contiguous(), tokens_all[:, s + 1].contiguous())[0]
        if timed:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            self.decode_events = (getattr(self, "decode_events", []) + [(ev0, ev1)])[-64:]
        self.last_sample_logp = [slogp_all[b * G:(b + 1) * G] for b in range(B)] if slogp_all is not None else None
        self.last_prefill = [(st["prefill_ctx"], st["kv"]) for st in per]    # (saved prompt activations, cache views) per prompt
        return [(st["tokens"], st["lay"]) for st in per]
