"""Qwen2-VL forward and hand-written backward, expressed on the op interface of ops.py (HipOps in production).

No autograd: the GRPO update needs exactly one backward through a fixed architecture, so the engine saves what the HIP backward
kernels need (288 GB of HBM holds every activation of a 7B step - no recomputation, unlike the reference's gradient
checkpointing, scripts/posttrain/train_rl.sh:32) and walks the layers in reverse.

Reference call sites restated here (transformers/models/qwen2_vl/modeling_qwen2_vl.py, v5.15.0):
  vision tower   :251-274 PatchEmbed, :425-449 VisionBlock, :277-290 PatchMerger, :1021-1033 get_video_features
  decoder layer  :559-624; attention :501-556; MLP :459-466; final norm :839; lm_head :1323; video scatter :1170-1176
and the logprob stage of src/time_r1/rl/timer1_trainer.py:449-481.
"""
import os

import numpy as np
import torch

from .config import ModelConfig
from .params import ModelParams, Arena
from .positions import vision_hw_ids, vision_segments, vision_window_index, segments_from_cu

I32 = torch.int32
F32 = torch.float32


class Engine:
    def __init__(self, cfg: ModelConfig, ops, params: ModelParams):
        self.cfg, self.ops, self.params = cfg, ops, params
        self.overlap_wgrad = True      # weight gradients on a second HIP stream (attribute for A/B runs; measured equal in rounds 4-5: the backward is a serial sum)
        # weight gradients that stay on the MAIN stream (d = down, g = gate/up, o, q = qkv); the rest runs on the side stream beside the dgrad
        # chain.  With the dgrad reading the weights as stored (NN form) the main stream has slack: keeping the down projection's weight
        # gradient there balanced the two streams best on MI355X (backward 175 ms against 180 with everything on the side stream; with the main
        # chain on a high-priority stream: "d" 161, "" 164, "dq" 161, "o" / "dg" 174 ms).  TR1_WGRAD_MAIN overrides for A/B runs.
        self.wgrad_on_main = os.environ.get("TR1_WGRAD_MAIN", "d")
        self.wgrad_overwrite_first = True    # see _wgrad: relies on the optimizer zeroing the gradient arena and bumping arena.version (AdamWFlat.step)
        self._gw_ver = {}
        self.lazy_zero_active = False        # set by the owner of the optimizer when AdamWFlat.lazy_zero is in force (see _wgrad)
        self.fused_head = True    # lm_head -> logp / entropy in the GEMM epilogue where the logits are not kept
        self.wgrad_nn = True      # weight gradients read the saved activation as stored (attribute for A/B runs)
        self._side = None
        # set (by the owner of the optimizer) for the backward of a window's LAST micro-step: the weight-gradient epilogues of the decoder layers' large
        # matrices then also leave the squared norm of the final gradient (AdamWFlat.norm_sink_begin / step)
        self.norm_sink = None
        self.bwd_count = 0                   # decoder backward passes run so far: a sink is only valid if its backward was the LAST one before the step
        assert cfg.vision.variant in ("qwen2_vl", "qwen2_5_vl"), cfg.vision.variant

    # ================================================================================================= gradient helpers
    def _wgrad(self, dy, x, gw, key=None, bias_g=None, dyt=None):
        """gw[N,K] (fp32) += dy[M,N]^T @ x[M,K].  key (decoder-layer weights): the FIRST weight gradient after an optimizer step (the arena's
        version changed) overwrites gw instead of accumulating - AdamW left it at zero, so the result is identical and the GEMM epilogue skips
        reading 4 bytes per parameter (33 GB per accumulation window at 7B)."""
        ops = self.ops
        acc = True
        if key is not None:
            ver = getattr(self.params.train, "version", None)
            if ver is not None and self._gw_ver.get(key) != ver:
                self._gw_ver[key] = ver
                if self.wgrad_overwrite_first:
                    acc = False
                elif self.lazy_zero_active and key.split(".", 1)[-1] in self.OVERWRITTEN:
                    gw.zero_()      # overwrite was switched off after the optimizer left this matrix un-zeroed (AdamWFlat.lazy_zero): never accumulate onto stale values
        # bias_g: the Linear's bias gradient (column sums of dy) rides on the pass that builds dy^T
        if dyt is None:                                  # (the fused down dgrad hands over dgu^T from its epilogue)
            dyt = ops.transpose(dy, colsum=bias_g) if bias_g is not None else ops.transpose(dy)      # [N, Mp], zero-padded columns
        sink = self.norm_sink
        if sink is not None and key is not None and key.split(".", 1)[-1] in self.OVERWRITTEN and hasattr(ops, "wgrad_sumsq"):
            # last micro-step of the window: same GEMM, and its epilogue also leaves the sum of squares of the FINAL gradient values it stores
            kmaj = self.wgrad_nn and x.shape[1] >= 2 * dy.shape[1]
            sync = sink.get("sync")
            off = self.params.train.offsets[key][0]
            wire = sync.wire_view(off, gw.shape) if sync is not None else None      # data-parallel: the exchange's bf16 copy comes out of this epilogue too
            n = ops.wgrad_sumsq(dyt, x if kmaj else ops.transpose(x), gw, acc, sink["partials"], sink["n"], b_kmajor=kmaj, b_rows=x.shape[0], wire=wire)
            if n >= 0:
                sink["n"] += n
                sink["covered"].add(key)
                if wire is not None:
                    sync.mark_wire(off, off + gw.numel())
                return
        # K-major form: x is read as stored (no x^T copy; the padded token columns of dy^T are zero, so the rows re-read past M drop out).
        # Its transposing LDS reads cost 8-17 % of the GEMM rate (tools/bench_wgrad.py: 1060 against 1244 TFLOP/s at the down-projection
        # shape), so it only pays where the saved transpose is the larger piece: x at least twice as wide as dy (the down projection,
        # x = the 18944-column SwiGLU output: 735 -> 650 us per layer; gate/up, o, qkv and the lm_head stay on the transposed copy).
        nn = getattr(ops, "wgrad_nn", None)
        if nn is not None and self.wgrad_nn and x.shape[1] >= 2 * dy.shape[1] and nn(dyt, x, gw, acc):
            return
        ops.gemm_nt(dyt, ops.transpose(x), out_f32=True, out=gw, accumulate=acc)

    OVERWRITTEN = ("qkv.w", "o.w", "gu.w", "down.w")      # per-layer matrices whose first weight gradient of a window overwrites (see _wgrad)

    def lazy_zero_plan(self):
        """What the optimizer may leave un-zeroed: with wgrad_overwrite_first the four large matrices of every decoder layer are OVERWRITTEN by the first
        micro-step of the next window, so zeroing them (4 of AdamW's 34 bytes per parameter, 30 GB per step at 7B) is wasted work.  Returns
        dict(base, stride, count, keep=[(a, b) relative to a layer's start]) for AdamWFlat.lazy_zero, or None when the layout / settings do not allow it."""
        t, a = self.cfg.text, self.params.train
        if not self.wgrad_overwrite_first or a.grad is None or t.n_layers < 1:
            return None
        base = a.offsets["l0.ln1"][0]
        stride = (a.offsets["l1.ln1"][0] - base) if t.n_layers > 1 else (a.range_of("l0.")[1] - base)
        keep = []
        for i in range(t.n_layers):
            rel = []
            for nm in self.OVERWRITTEN:
                off, shape = a.offsets["l%d.%s" % (i, nm)]
                rel.append((off - base - i * stride, off - base - i * stride + int(np.prod(shape))))
            if i == 0:
                keep = rel
            elif rel != keep:
                return None          # layers are not laid out periodically
        if a.range_of("l%d." % (t.n_layers - 1))[1] > base + stride * t.n_layers:
            return None
        return dict(base=int(base), stride=int(stride), count=int(t.n_layers), keep=sorted(keep))

    # Weight gradients of the decoder layers on a second HIP stream: wgrad (dy^T x) and dgrad (dy W) of a Linear only share their input,
    # so the two GEMM chains run concurrently and each fills the CUs the other leaves idle in its last, partially filled round of tiles
    # (M = 5074 rows against 128/256-row tiles: 1.5-2.2 rounds per GEMM).
    def _side_stream(self):
        if not getattr(self.ops, "device", None) or torch.device(self.ops.device).type != "cuda" or not self.overlap_wgrad:
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.ops.device)
        return self._side

    def _wgrad_async(self, dy, x, gw, side, key=None, bias_g=None, dyt=None):
        if side is None:
            return self._wgrad(dy, x, gw, key, bias_g, dyt)
        main = torch.cuda.current_stream(self.ops.device)
        side.wait_stream(main)                         # dy (and every earlier write of gw) is ordered before the side work
        with torch.cuda.stream(side):
            self._wgrad(dy, x, gw, key, bias_g, dyt)
        dy.record_stream(side)                         # keep the caching allocator from recycling them under the side stream
        x.record_stream(side)
        if dyt is not None:
            dyt.record_stream(side)

    def _dgrad(self, dy, w, key=None):
        """dx[M,K] = dy[M,N] @ w[N,K].  The weight is the K-major operand of an "NN" GEMM: large problems read it as stored through the
        transposing LDS reads of the phased kernel (ops.gemm_nn), so no W^T copy is built (a cached W^T was measured SLOWER than a fresh one -
        the fresh copy is served from the Infinity Cache - and the NN form needs neither)."""
        assert dy.shape[1] % 64 == 0, "dgrad: N must be a multiple of 64"
        return self.ops.gemm_nn(dy, w)

    # ============================================================================================================ ViT
    def vit_features(self, pixels, grid_thw):
        """Frozen vision blocks. pixels: [N_v, patch_dim_padded] act dtype; grid_thw: list of (t,h,w).
        Returns (feats [N_v, embed], perm): perm is None for Qwen2-VL; for Qwen2.5-VL the rows of feats are in WINDOW order and perm is
        the int32 device tensor (window_index) merger_fwd/bwd use to restore the natural order of the merged tokens."""
        if self.cfg.vision.variant == "qwen2_5_vl":
            return self._vit_features_25(pixels, grid_thw)
        ops, v, fz = self.ops, self.cfg.vision, self.params.frozen
        E, H, hd = v.embed_dim, v.num_heads, v.head_dim
        hw = ops.tensor(vision_hw_ids(grid_thw, v.spatial_merge_size), I32)
        pre, lo, hi = [ops.tensor(a, I32) for a in vision_segments(grid_thw)]
        N = pixels.shape[0]
        assert hw.shape[0] == N, (hw.shape, N)
        cos, sin = ops.vision_rope_table(hw, hd)
        x = ops.gemm_nt(pixels, fz.w("patch.w"))
        scale = hd ** -0.5
        pad = self._vit_pad128(N)
        for i in range(v.depth):
            p = "v%d." % i
            y, _, _ = ops.layernorm_fwd(x, fz.w(p + "n1.w"), fz.w(p + "n1.b"), v.ln_eps, need_stats=False)
            x = self._vit_attention(i, y, x, cos, sin, pre, lo, hi, N, scale, pad)
            y, _, _ = ops.layernorm_fwd(x, fz.w(p + "n2.w"), fz.w(p + "n2.b"), v.ln_eps, need_stats=False)
            z = ops.gemm_quickgelu(y, fz.w(p + "fc1.w"), fz.w(p + "fc1.b"))          # fc1 + bias + QuickGELU in the GEMM epilogue
            x = ops.gemm_nt(z, fz.w(p + "fc2.w"), bias=fz.w(p + "fc2.b"), residual=x)
        return x, None

    # ---- vision attention sub-block (both towers): x + proj(attention(rope(q), rope(k), v))
    def _vit_pad128(self, N):
        """Head dim 80 on 128-wide zero-padded heads: feature d < 40 at column d, d + 40 at 48 + d (round 6; 64 + d before: the live features now end at 96 and
        the attention launch skips the last quarter of its MFMAs, ops.attn_fwd(live96=True)), so that (a) the q|k|v GEMM's epilogue adds the bias and
        applies the 2-D rotary embedding and (b) the head-dim-128 attention kernel (32x32x16 MFMA,
        K and V row-major: no V^T copy) runs the tower - 236 -> ~140 us per block at config 3 against the 96-wide 16x16 kernel, and the two rope launches
        and the V^T pack disappear.  Returns None when the shapes / backend do not allow it.  The padded output-projection weights (zero columns at the pad
        positions) are derived ONCE per version of the frozen arena."""
        ops, v, fz = self.ops, self.cfg.vision, self.params.frozen
        ok = getattr(ops, "vit_pad128_ok", None)
        if ok is None or not ok(v.num_heads, v.head_dim):
            return None
        H, hd, E, half = v.num_heads, v.head_dim, v.embed_dim, v.head_dim // 2
        hoff = 48 if half <= 48 else 64              # where the second rotary half starts inside a 128-wide head (csrc/gemm.hip EPI 7)
        cache = self.__dict__.setdefault("_vit_pad_cache", {})
        ver = (fz.w16.data_ptr(), fz.w16._version)         # torch's in-place version counter: any write to the frozen arena (loaders, tests) rebuilds
        if cache.get("ver") != ver:
            proj = []
            for i in range(v.depth):
                w = fz.w("v%d.proj.w" % i).view(E, H, hd)
                wp = ops.zeros(E, H, 128)
                wp[:, :, :half] = w[:, :, :half]
                wp[:, :, hoff:hoff + half] = w[:, :, half:]
                proj.append(wp.view(E, H * 128))
            cache.update(ver=ver, proj=proj, bufs=None)
        if cache.get("bufs") is None or cache["bufs"][0].shape[0] != N:
            cache["bufs"] = [ops.zeros(N, H * 128) for _ in range(4)]      # q, k, v, o: pad columns stay zero (the live-96 attention launch never writes o's columns 96..127)
        return dict(proj=cache["proj"], bufs=cache["bufs"], live96=hoff == 48)

    def _vit_attention(self, i, y, x, cos, sin, pre, lo, hi, N, scale, pad):
        ops, v, fz = self.ops, self.cfg.vision, self.params.frozen
        E, H, hd = v.embed_dim, v.num_heads, v.head_dim
        p = "v%d." % i
        if pad is not None:
            q, k, vv, o = pad["bufs"]
            ops.gemm_qkv_rope_vit(y, fz.w(p + "qkv.w"), fz.w(p + "qkv.b"), cos, sin, H, hd // 2, q, k, vv)
            ops.attn_fwd(q, k, None, pre, lo, hi, H, H, N, 128, scale, need_lse=False, v_rows=vv, out=o, live96=pad["live96"])
            return ops.gemm_nt(o, pad["proj"][i], bias=fz.w(p + "proj.b"), residual=x)
        qkv = ops.gemm_nt(y, fz.w(p + "qkv.w"), bias=fz.w(p + "qkv.b"))
        q = ops.rope_apply(qkv[:, :E], H, hd, cos, sin)
        k = ops.rope_apply(qkv[:, E:2 * E], H, hd, cos, sin)
        vt = ops.pack_transpose(qkv[:, 2 * E:], H, H, hd)
        o, _ = ops.attn_fwd(q, k, vt, pre, lo, hi, H, H, N, hd, scale, need_lse=False)
        return ops.gemm_nt(o, fz.w(p + "proj.w"), bias=fz.w(p + "proj.b"), residual=x)

    def _vit_features_25(self, pixels, grid_thw):
        """Qwen2.5-VL tower (transformers/models/qwen2_5_vl/modeling_qwen2_5_vl.py:408-470 forward, :294-325 block, :85-96 MLP):
        merged tokens are permuted once into window-major order, so BOTH attention flavours are contiguous segments of the same
        two-interval mask kernel - windows for most blocks, whole temporal patches for `fullatt_block_indexes` (a frame's windows
        stay contiguous after the permutation).  RMSNorm + biased SwiGLU MLP (width zero-padded to a multiple of 64)."""
        ops, v, fz = self.ops, self.cfg.vision, self.params.frozen
        E, H, hd, U = v.embed_dim, v.num_heads, v.head_dim, v.merge_unit
        N = pixels.shape[0]
        widx, cu_win = vision_window_index(grid_thw, v.spatial_merge_size, v.window_size, v.patch_size)
        hw = vision_hw_ids(grid_thw, v.spatial_merge_size)
        assert hw.shape[0] == N and widx.shape[0] * U == N, (hw.shape, widx.shape, N)
        hw = hw.reshape(N // U, U, 2)[widx].reshape(N, 2)
        cos, sin = ops.vision_rope_table(ops.tensor(np.ascontiguousarray(hw), I32), hd)
        seg_win = [ops.tensor(a, I32) for a in segments_from_cu(cu_win)]
        seg_full = [ops.tensor(a, I32) for a in vision_segments(grid_thw)]
        perm = ops.tensor(widx.astype(np.int32), I32)
        x = ops.gemm_nt(pixels, fz.w("patch.w"))
        x = ops.gather_rows(x.view(N // U, U * E), perm).view(N, E)
        scale = hd ** -0.5
        pad = self._vit_pad128(N)
        for i in range(v.depth):
            p = "v%d." % i
            pre, lo, hi = seg_full if i in v.fullatt_block_indexes else seg_win
            y, _, _ = ops.rmsnorm_fwd(x, fz.w(p + "n1.w"), v.ln_eps, need_rstd=False)
            x = self._vit_attention(i, y, x, cos, sin, pre, lo, hi, N, scale, pad)
            y, _, _ = ops.rmsnorm_fwd(x, fz.w(p + "n2.w"), v.ln_eps, need_rstd=False)
            a, _ = ops.gemm_glu(y, fz.w(p + "gu.w"), save_gu=False, bias=fz.w(p + "gu.b"))      # gate/up + bias + SwiGLU in the GEMM epilogue
            x = ops.gemm_nt(a, fz.w(p + "down.w"), bias=fz.w(p + "down.b"), residual=x)
        return x, perm

    def merger_fwd(self, arena: Arena, feats, save, perm=None):
        """PatchMerger (trainable even with fix_vit, reference timer1_trainer.py:277-280). feats [N_v, E] -> [N_v/4, out_hidden].
        Qwen2-VL: LayerNorm; Qwen2.5-VL: RMSNorm, and the output rows go back from window order to natural order
        (modeling_qwen2_5_vl.py:462-464: merged[argsort(window_index)])."""
        ops, v = self.ops, self.cfg.vision
        N, E = feats.shape
        if v.variant == "qwen2_5_vl":
            xn, rstd, _ = ops.rmsnorm_fwd(feats, arena.w("merger.ln.w"), v.ln_eps, need_rstd=save)
            mean = None
        else:
            xn, mean, rstd = ops.layernorm_fwd(feats, arena.w("merger.ln.w"), arena.w("merger.ln.b"), v.ln_eps, need_stats=save)
        xv = xn.view(N // v.merge_unit, E * v.merge_unit)
        z1 = ops.gemm_nt(xv, arena.w("merger.fc1.w"), bias=arena.w("merger.fc1.b"))
        g1 = ops.gelu_fwd(z1)
        out = ops.gemm_nt(g1, arena.w("merger.fc2.w"), bias=arena.w("merger.fc2.b"))
        if perm is not None:           # out_nat[perm[j]] = out_win[j]
            nat = ops.empty(*out.shape)
            ops.scatter_rows(out, perm, nat)
            out = nat
        ctx = dict(feats=feats, mean=mean, rstd=rstd, xv=xv, z1=z1, g1=g1, perm=perm) if save else None
        return out, ctx

    def merger_bwd(self, ctx, dout):
        ops, tr = self.ops, self.params.train
        if ctx["perm"] is not None:
            dout = ops.gather_rows(dout, ctx["perm"])
        self._wgrad(dout, ctx["g1"], tr.g("merger.fc2.w"))
        ops.colsum_accum(dout, tr.g("merger.fc2.b"))
        dg1 = self._dgrad(dout, tr.w("merger.fc2.w"))
        dz1 = ops.gelu_bwd(ctx["z1"], dg1)
        self._wgrad(dz1, ctx["xv"], tr.g("merger.fc1.w"))
        ops.colsum_accum(dz1, tr.g("merger.fc1.b"))
        dxv = self._dgrad(dz1, tr.w("merger.fc1.w"))
        dxn = dxv.view(ctx["feats"].shape)
        if self.cfg.vision.variant == "qwen2_5_vl":
            ops.rmsnorm_bwd(dxn, ctx["feats"], tr.w("merger.ln.w"), ctx["rstd"], dw=tr.g("merger.ln.w"))
        else:
            ops.layernorm_bwd(dxn, ctx["feats"], tr.w("merger.ln.w"), ctx["mean"], ctx["rstd"], tr.g("merger.ln.w"), tr.g("merger.ln.b"),
                              need_dx=False)  # the blocks below are frozen: no dx

    # ============================================================================================================ LLM
    def embed(self, arena: Arena, ids, vid_embeds=None, vid_rows=None):
        h = self.ops.gather_rows(arena.w("embed"), ids)
        if vid_embeds is not None:
            self.ops.scatter_rows(vid_embeds, vid_rows, h)
        return h

    SAVED = ("h", "xn", "v", "q", "o", "h2", "xn2", "gu", "a")

    # A full set of saved-activation buffers above this size is kept ONCE: further prompts of the accumulation window stash only their
    # prompt rows (written by the rollout prefill) and move them into the one full set when their update starts (unstash_ctx).
    CTX_STASH_GB = 40.0
    SHARE_A = True      # one shared SwiGLU-output buffer in the large-sequence regime (rebuilt in the backward); class attribute for A/B runs

    def ctx_bytes(self, rows):
        t = self.cfg.text
        return rows * t.n_layers * (2 * (4 * t.hidden + t.kv_dim + 2 * t.q_dim + 3 * t.intermediate) + 8)

    def alloc_ctx_bufs(self, total_rows, slot=0, prefill_rows=None):
        """Saved-activation buffers for a packed sequence of `total_rows` rows that is run in two pieces (prompt rows during the rollout
        prefill, completion rows in the update's continuation forward): both pieces write their rows in place, so the backward gets
        [M, .] tensors without a concatenation pass (that pass cost ~25 GB of copies per 7B micro-step).
        The buffers are owned by the engine and recycled: `slot` = index of the prompt inside the accumulation window; a slot is free
        again once that prompt's backward has been enqueued (same stream), and it only grows (no allocator churn of 24 GB blocks).
        Large sequences (config 4: 19 650 rows = 92 GB per set): slots >= 1 get a STASH of `prefill_rows` rows only - the window's rollouts
        are still decoded together (all prefills run first), but only one full set exists; returns (bufs, is_stash)."""
        ops, t = self.ops, self.cfg.text
        stash = slot > 0 and prefill_rows is not None and self.ctx_bytes(total_rows) > self.CTX_STASH_GB * 1e9
        rows = prefill_rows if stash else total_rows
        pool = self.__dict__.setdefault("_ctx_pool", {})
        key = ("stash", slot) if stash else slot
        ent = pool.get(key)
        if ent is None or ent[0] < rows:
            pool[key] = None          # release the smaller set before allocating the larger one
            cap = (rows + 255) // 256 * 256
            cols = dict(h=t.hidden, xn=t.hidden, v=t.kv_dim, q=t.q_dim, o=t.q_dim, h2=t.hidden, xn2=t.hidden, gu=2 * t.intermediate, a=t.intermediate)
            # Large sequences (the stashed-prefill regime): ONE SwiGLU-output buffer for all layers instead of one per layer - a = silu(g) u is consumed by its own
            # layer's down projection in the forward, and the backward rebuilds it from the saved gate/up tensor right before the down projection's weight gradient
            # (one elementwise pass, bit-identical: the fused epilogue computes a from the bf16-rounded g / u).  Config 4: 27 x 0.74 GB = 20 GB less on a
            # 250 GB step, which is what lets the tail-row and dgu^T fast paths stay on there without allocator retries.
            a_shared = ops.empty(cap, t.intermediate) if (self.SHARE_A and self.ctx_bytes(total_rows) > self.CTX_STASH_GB * 1e9 and t.n_layers > 1) else None
            full = []
            for _ in range(t.n_layers):
                L = {k: (a_shared if (k == "a" and a_shared is not None) else ops.empty(cap, c)) for k, c in cols.items()}
                L["rstd1"] = ops.empty(cap, dtype=F32)
                L["rstd2"] = ops.empty(cap, dtype=F32)
                full.append(L)
            ent = pool[key] = (cap, full)
        bufs = [{k: v[:rows] for k, v in L.items()} for L in ent[1]]
        return (bufs, stash) if prefill_rows is not None else bufs

    def unstash_ctx(self, pctx, prefill_rows, total_rows):
        """Move a stashed prefill (prompt rows of a later prompt of the window) into the one full buffer set; the previous prompt's backward has
        been enqueued on this stream, so the set is free.  ~1 read + 1 write of the prompt rows (config 4: 15 GB, 6 ms on a 3.4 s micro-step)."""
        full = self.alloc_ctx_bufs(total_rows, slot=0)
        P = prefill_rows
        shared_a = len(full) > 1 and full[0]["a"].data_ptr() == full[1]["a"].data_ptr()
        for dst, src in zip(full, pctx["bufs"]):
            for k, v in src.items():
                if k == "a" and shared_a:
                    continue              # rebuilt from gu in the backward (alloc_ctx_bufs)
                dst[k][:P].copy_(v[:P])
        pctx["bufs"] = full
        pctx["stash"] = False
        return pctx

    TAIL_SKIP = True     # class attribute (tests / A/B runs clear it): False = every row runs the whole last layer

    def tail_rows_from(self, P, M):
        """The `tail_from` a caller should pass for a packed sequence of P prompt rows in M rows, or None: the saving is the prompt's share of ONE layer, so
        only where the prompt dominates (config 3: 3 474 of 5 074 rows; config 4's prompt is a sixth of the rows).  Prefill, update forward and backward of one
        sequence must all use this one answer.  (Round 4 also switched it off in the large-sequence regime for allocator reasons; round 5 found the cause -
        blocks held across the two backward streams, see llm_bwd - and the memory gate is gone.)"""
        if not self.TAIL_SKIP or P < 2 or (P - 1) < 0.4 * M:
            return None
        return P - 1

    def llm_fwd(self, arena: Arena, h, cos, sin, masks, save, kv_cache=None, row0=0, bufs=None, tail_from=None):
        """Decoder stack over a packed sequence of M rows. masks = (pre, lo, hi) int32 [M] over slots == rows.
        kv_cache: optional list of (K [S_cap, kv_dim], VT [kv_dim, S_cap]) to be filled (rollout prefill).
        row0 > 0 ("continuation"): h holds only rows [row0, row0+M) of the packed sequence; their K/V are written to cache slots
        [row0, row0+M) and attention runs over slots [0, row0+M) - the prefix K/V of rows [0,row0) must already be in kv_cache.
        bufs (alloc_ctx_bufs): saved activations are written into rows [row0, row0+M) of these buffers instead of fresh tensors.
        tail_from (packed row index): the caller reads the stack's output only at rows >= tail_from (log-probs: the last prompt row and the completion
        rows; rollout prefill: the last prompt row).  The LAST layer's o projection and MLP then run on those rows only - the other rows' outputs feed
        nothing (their K / V, which later rows and the decode do read, come from the layer's input) - and rows < tail_from of the returned tensor and of the
        last layer's saved h2 / xn2 / gu / a / rstd2 are undefined; llm_bwd skips them the same way (their gradient is exactly zero).
        Returns (h_out, ctx) where ctx holds the saved activations when save=True."""
        ops, t = self.ops, self.cfg.text
        M = h.shape[0]
        S = row0 + M
        t0 = 0 if (tail_from is None or not self.TAIL_SKIP) else max(0, min(int(tail_from) - row0, M - 1))
        pre, lo, hi = masks
        qd, kvd, hd = t.q_dim, t.kv_dim, t.head_dim
        scale = hd ** -0.5
        layers = []
        inplace = save and bufs is not None

        def dst(i, key):
            return bufs[i][key][row0:S] if inplace else None
        if inplace:
            bufs[0]["h"][row0:S].copy_(h)
            h = bufs[0]["h"][row0:S]
        for i in range(t.n_layers):
            p = "l%d." % i
            xn, rstd1, _ = ops.rmsnorm_fwd(h, arena.w(p + "ln1"), t.rms_eps, need_rstd=save, out=dst(i, "xn"), rstd_out=dst(i, "rstd1"))
            # q|k|v projection + bias + M-RoPE in one launch (the GEMM epilogue rotates q and k; k lands in the KV cache rows when there is one).
            # The un-rotated q / k are never stored: nothing reads them again (the backward needs q, k rotated and v).
            rows_ok = getattr(ops, "attn_fwd_rows_ok", lambda *a, **kw: False)(hd)
            kc = vtc = None
            if kv_cache is not None:
                kc, vtc = kv_cache[i]
            q, k, v = ops.gemm_qkv_rope(xn, arena.w(p + "qkv.w"), arena.w(p + "qkv.b"), cos, sin, t.n_heads, t.n_kv_heads, hd, q_out=dst(i, "q"),
                                        k_out=kc[row0:S] if kc is not None else None, v_out=dst(i, "v"))
            # head dim 128: the forward kernel reads V row-major (it transposes in its LDS reads); V^T is then only built where a later decode
            # needs it in the cache (rollout prefill / continuation), not for the reference-policy and plain training forwards
            v_rows = None
            if kv_cache is not None:
                if row0 == 0:
                    vt = ops.pack_transpose(v, t.n_kv_heads, t.n_kv_heads, hd, out=vtc)
                    v_rows = v if rows_ok else None
                else:
                    vt = ops.pack_transpose(v, t.n_kv_heads, t.n_kv_heads, hd, out=vtc[:, row0:], zero_pad=False)
                    vt = vtc
                    if rows_ok and inplace:      # the prefix rows' V sits in the shared activation buffers (written by the prefill)
                        v_rows = bufs[i]["v"][:S]
                k_all = kc
            else:
                v_rows = v if rows_ok else None
                vt = ops.pack_transpose(v, t.n_kv_heads, t.n_kv_heads, hd) if v_rows is None else None
                k_all = k
            o, lse = ops.attn_fwd(q, k_all, vt, pre, lo, hi, t.n_heads, t.n_kv_heads, S, hd, scale, need_lse=save, out=dst(i, "o"),
                                  **({"v_rows": v_rows} if v_rows is not None else {}))
            if t0 > 0 and i == t.n_layers - 1:       # last layer: only rows >= tail_from go on (see the docstring)
                tl = lambda x: None if x is None else x[t0:]
                h2t = ops.gemm_nt(o[t0:], arena.w(p + "o.w"), residual=h[t0:], out=tl(dst(i, "h2")))
                xn2t, rstd2t, _ = ops.rmsnorm_fwd(h2t, arena.w(p + "ln2"), t.rms_eps, need_rstd=save, out=tl(dst(i, "xn2")), rstd_out=tl(dst(i, "rstd2")))
                at, gut = ops.gemm_glu(xn2t, arena.w(p + "gu.w"), a_out=tl(dst(i, "a")), gu_out=tl(dst(i, "gu")), save_gu=save)
                h_out = ops.empty(M, t.hidden)
                ops.gemm_nt(at, arena.w(p + "down.w"), residual=h2t, out=h_out[t0:])
                if save:
                    def full(x_t, key, **kw):         # [M, .] views for the backward: the shared buffers when there are some, else a tensor whose head is never read
                        if x_t is None:
                            return None
                        if inplace:
                            return bufs[i][key][row0:S]
                        f = ops.empty(M, *x_t.shape[1:], **kw)
                        f[t0:] = x_t
                        return f
                    h2, xn2, a, gu = full(h2t, "h2"), full(xn2t, "xn2"), full(at, "a"), full(gut, "gu")
                    rstd2 = full(rstd2t, "rstd2", dtype=F32)
                    layers.append(dict(h=h, rstd1=rstd1, xn=xn, v=v, q=q, k=k, o=o, lse=lse, h2=h2, rstd2=rstd2, xn2=xn2, gu=gu, a=a))
                h = h_out
                continue
            h2 = ops.gemm_nt(o, arena.w(p + "o.w"), residual=h, out=dst(i, "h2"))
            xn2, rstd2, _ = ops.rmsnorm_fwd(h2, arena.w(p + "ln2"), t.rms_eps, need_rstd=save, out=dst(i, "xn2"), rstd_out=dst(i, "rstd2"))
            a, gu = ops.gemm_glu(xn2, arena.w(p + "gu.w"), a_out=dst(i, "a"), gu_out=dst(i, "gu"), save_gu=save)     # SwiGLU in the GEMM epilogue
            h_out = ops.gemm_nt(a, arena.w(p + "down.w"), residual=h2, out=dst(i + 1, "h") if inplace and i + 1 < t.n_layers else None)
            if save:
                layers.append(dict(h=h, rstd1=rstd1, xn=xn, v=v, q=q, k=k, o=o, lse=lse, h2=h2, rstd2=rstd2, xn2=xn2, gu=gu, a=a))
            h = h_out
        ctx = dict(layers=layers, masks=masks, cos=cos, sin=sin, h_last=h, bufs=bufs if inplace else None, tail_from=row0 + t0) if save else None
        return h, ctx

    @staticmethod
    def merge_ctx(ctx_a, ctx_b, kv_cache, masks, cos, sin, M):
        """Stitch the saved activations of a prefix forward (rows [0,P)) and its continuation (rows [P,M)) into the [M, .] form
        llm_bwd expects. K comes from the cache (rows [0,M) in packed order); lse is [n_heads, rows] so it is joined along dim 1.
        When both pieces wrote into shared buffers (alloc_ctx_bufs) nothing but lse is copied."""
        layers = []
        shared = ctx_a.get("bufs") is not None and ctx_a.get("bufs") is ctx_b.get("bufs")
        for i, (a, b) in enumerate(zip(ctx_a["layers"], ctx_b["layers"])):
            L = {}
            for key in ("h", "xn", "v", "q", "o", "h2", "xn2", "gu", "a", "rstd1", "rstd2"):
                L[key] = ctx_a["bufs"][i][key][:M] if shared else torch.cat([a[key], b[key]], 0)
            L["lse"] = torch.cat([a["lse"], b["lse"]], 1).contiguous()
            L["k"] = kv_cache[i][0][:M]
            layers.append(L)
            ctx_a["layers"][i] = ctx_b["layers"][i] = None
        # (rows >= ctx_a's tail_from carry the last layer's MLP activations: the continuation ran all of its rows)
        return dict(layers=layers, masks=masks, cos=cos, sin=sin, tail_from=min(int(ctx_a.get("tail_from", 0)), int(ctx_b.get("tail_from", 0))))

    def llm_bwd(self, ctx, dh, on_layer_done=None):
        """dh: gradient wrt the decoder stack output [M, d]. Accumulates parameter grads; returns the gradient wrt the input embeddings.
        on_layer_done(i): called right after layer i's gradient kernels are enqueued (data-parallel overlap hook)."""
        ops, t, tr = self.ops, self.cfg.text, self.params.train
        self.bwd_count += 1
        pre, lo, hi = ctx["masks"]
        cos, sin = ctx["cos"], ctx["sin"]
        qd, kvd, hd = t.q_dim, t.kv_dim, t.head_dim
        scale = hd ** -0.5
        # Large sequences (the stashed-prefill regime, config 4): weight gradients stay on the MAIN stream.  A tensor produced on one stream and consumed on the
        # other (dgu^T, the transposes) keeps its caching-allocator block until the consumer's event has passed; with 1.5 GB tensors per layer and the side
        # stream a few layers behind, the allocator grew to 300 GB reserved for 250 GB allocated and went into retries (backward 0.65 -> 2.2 s) as soon as anything
        # else was added.  On one stream reserved == allocated (231 GB at config 4), and the second stream buys nothing there anyway (the backward is a
        # serial sum of its kernels).
        side = None if self.ctx_bytes(dh.shape[0]) > self.CTX_STASH_GB * 1e9 else self._side_stream()
        pending = None
        a_shared = t.n_layers > 1 and ctx["layers"][0]["a"].data_ptr() == ctx["layers"][1]["a"].data_ptr()
        t0 = int(ctx.get("tail_from", 0) or 0)      # rows < t0 never went through the last layer's o projection / MLP (llm_fwd tail_from): dh is zero there
        for i in reversed(range(t.n_layers)):
            p = "l%d." % i
            L = ctx["layers"][i]
            M = dh.shape[0]
            tail = t0 > 0 and i == t.n_layers - 1
            if tail:
                dh_full, Lf = dh, L
                dh = dh[t0:]
                L = dict(L)
                for key in ("a", "gu", "xn2", "h2", "rstd2", "o"):
                    L[key] = Lf[key][t0:]
            # h_out = a @ Wd^T + h2
            _sync = self.wgrad_on_main
            if a_shared:                  # one SwiGLU-output buffer for all layers (alloc_ctx_bufs): rebuild this layer's rows from its gate/up tensor, main stream
                ops.swiglu_fwd(L["gu"], out=L["a"])
            self._wgrad_async(dh, L["a"], tr.g(p + "down.w"), None if ("d" in _sync or a_shared) else side, key=p + "down.w")
            # down-projection dgrad with the SwiGLU backward in its epilogue, which also leaves dgu^T (the gate/up weight gradient's operand) from its LDS staging
            dgu, dgut = ops.dgrad_glu_bwd(dh, tr.w(p + "down.w"), L["gu"], want_t=True)
            self._wgrad_async(dgu, L["xn2"], tr.g(p + "gu.w"), None if "g" in _sync else side, key=p + "gu.w", dyt=dgut)
            dxn2 = self._dgrad(dgu, tr.w(p + "gu.w"), key=p + "gu.w")
            dh2 = ops.rmsnorm_bwd(dxn2, L["h2"], tr.w(p + "ln2"), L["rstd2"], dres=dh, dw=tr.g(p + "ln2"))
            # h2 = o @ Wo^T + h
            self._wgrad_async(dh2, L["o"], tr.g(p + "o.w"), None if "o" in _sync else side, key=p + "o.w")
            do = self._dgrad(dh2, tr.w(p + "o.w"), key=p + "o.w")
            if tail:                                  # back to all rows for the attention backward (every row's K / V took part): zero gradient above the tail
                L = Lf
                z = ops.zeros(M, qd); z[t0:] = do; do = z
                z = ops.zeros(M, t.hidden); z[t0:] = dh2; dh2 = z
            dqkv = ops.empty(M, t.qkv_dim)
            # attention backward writes dq | dk | dv straight into the columns of dqkv, dq and dk already rotated back (M-RoPE backward in the
            # kernels' epilogues); the bias gradient is summed by the transpose that feeds the weight gradient
            ops.attn_bwd(L["q"], L["k"], L["v"], L["o"], do, L["lse"], pre, lo, hi, t.n_heads, t.n_kv_heads, M, hd, scale,
                         dq_out=dqkv[:, :qd], dk_out=dqkv[:, qd:qd + kvd], dv_out=dqkv[:, qd + kvd:], rope=(cos, sin))
            self._wgrad_async(dqkv, L["xn"], tr.g(p + "qkv.w"), None if "q" in _sync else side, key=p + "qkv.w", bias_g=tr.g(p + "qkv.b"))
            dxn = self._dgrad(dqkv, tr.w(p + "qkv.w"), key=p + "qkv.w")
            dh = ops.rmsnorm_bwd(dxn, L["h"], tr.w(p + "ln1"), L["rstd1"], dres=dh2, dw=tr.g(p + "ln1"))
            ctx["layers"][i] = None  # release this layer's activations
            if on_layer_done is not None:
                if side is not None:
                    # this layer's side-stream weight gradients are final once `ev` has fired.  The announcement is deferred by one layer, so
                    # the main stream waits on an event that is (almost always) already past instead of draining the side stream per layer
                    ev = torch.cuda.Event()
                    ev.record(side)
                    if pending is not None:
                        torch.cuda.current_stream(self.ops.device).wait_event(pending[1])
                        on_layer_done(pending[0])
                    pending = (i, ev)
                else:
                    on_layer_done(i)
        if side is not None:
            torch.cuda.current_stream(self.ops.device).wait_stream(side)
        if pending is not None:
            on_layer_done(pending[0])
        return dh

    def embed_bwd(self, dh0, ids_for_grad, vid_rows=None):
        """ids_for_grad: token ids with -1 on rows whose embedding was replaced by a video feature."""
        self.ops.embed_bwd(dh0, ids_for_grad, self.params.train.g("embed"))
        if vid_rows is not None:
            return self.ops.gather_rows(dh0, vid_rows)
        return None

    # ====================================================================================================== logprob head
    # Logits exist at most for HEAD_CHUNK_ROWS prediction rows at a time.  Up to that many rows (config 3: G*C = 1600 rows, 0.49 GB of bf16
    # logits) the policy path keeps them for the backward; beyond it (config 4: 16 x 1024 rows = 5 GB) both passes walk the rows in chunks
    # and the backward recomputes a chunk's logits with one more lm_head GEMM (~1 ms per 1024 rows at 7B) instead of holding [G*C, V].
    HEAD_CHUNK_ROWS = 4096

    def head_fwd(self, arena: Arena, h_last, pred_rows, targets, save):
        """Final norm + lm_head on the rows that predict completion tokens only, then per-token log-prob and entropy
        (reference _get_per_token_logps materialises logits for all (G, L, V) - SURVEY 0.6)."""
        ops, t = self.ops, self.cfg.text
        hp = ops.gather_rows(h_last, pred_rows)
        hn, rstd, _ = ops.rmsnorm_fwd(hp, arena.w("norm"), t.rms_eps, need_rstd=save)
        w = self.params.lm_head_w(arena)
        R, ch = hp.shape[0], self.HEAD_CHUNK_ROWS
        fused = None
        if (not save or R > ch) and self.fused_head and hasattr(ops, "lmhead_lse"):
            # nobody reads these logits again (reference-policy forward; the large-R policy forward recomputes them chunk by chunk in the
            # backward): lm_head with the log-softmax statistics reduced in the GEMM epilogue - no [R, V] tensor in HBM at all
            fused = ops.lmhead_lse(hn, w, targets)
        if fused is not None:
            logits, (logp, ent, lse) = None, fused
        elif R <= ch:
            logits = ops.gemm_nt(hn, w)
            logp, ent, lse = ops.logp_entropy_fwd(logits, targets)
        else:
            logits, parts = None, []
            for a in range(0, R, ch):
                lg = ops.gemm_nt(hn[a:a + ch], w)
                parts.append(ops.logp_entropy_fwd(lg, targets[a:a + ch].contiguous()))
                del lg
            logp, ent, lse = [torch.cat([p[i] for p in parts]) for i in range(3)]
        ctx = dict(hp=hp, hn=hn, rstd=rstd, logits=logits, lse=lse, targets=targets, pred_rows=pred_rows, M=h_last.shape[0]) if save else None
        return logp, ent, ctx

    def head_bwd(self, ctx, dlogp, n_dup):
        """dlogp: [R] fp32 in pred_rows order. The first n_dup pred rows all alias one hidden row (the last prompt token)."""
        ops, t, tr = self.ops, self.cfg.text, self.params.train
        w, gw = self.params.lm_head_w(), self.params.lm_head_g()
        if ctx["logits"] is not None:
            dlogits = ops.logp_bwd(ctx["logits"], ctx["targets"], ctx["lse"], dlogp, inplace=True)
            self._wgrad(dlogits, ctx["hn"], gw, key="lm_head.w")        # first micro-step of a window: overwrite (the optimizer left it zero: same result, no 2.2 GB read)
            dhn = self._dgrad(dlogits, w, key="lm_head")
            ctx["logits"] = None
        else:           # chunked: recompute a chunk's logits, turn them into dlogits in place, feed both gradient GEMMs, drop them
            hn, R, ch = ctx["hn"], ctx["hn"].shape[0], self.HEAD_CHUNK_ROWS
            dhn = ops.empty(R, hn.shape[1])
            for a in range(0, R, ch):
                b = min(R, a + ch)
                lg = ops.gemm_nt(hn[a:b], w)
                dl = ops.logp_bwd(lg, ctx["targets"][a:b].contiguous(), ctx["lse"][a:b].contiguous(), dlogp[a:b].contiguous(), inplace=True)
                self._wgrad(dl, hn[a:b], gw, key="lm_head.w")           # (only the window's first chunk overwrites; later chunks and micro-steps accumulate)
                dhn[a:b] = self._dgrad(dl, w, key="lm_head")
                del lg, dl
        dhp = ops.rmsnorm_bwd(dhn, ctx["hp"], tr.w("norm"), ctx["rstd"], dw=tr.g("norm"))
        d = dhp.shape[1]
        dh = ops.zeros(ctx["M"], d)
        acc = ops.zeros(d, dtype=F32)
        ops.colsum_accum(dhp[:n_dup], acc)
        first = ops.cast_to_act(acc).view(1, d)
        ops.scatter_rows(first, ctx["pred_rows"][:1], dh)
        if dhp.shape[0] > n_dup:
            ops.scatter_rows(dhp[n_dup:], ctx["pred_rows"][n_dup:], dh)
        return dh
