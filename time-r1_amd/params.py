"""Parameter arenas: every weight is a view into ONE flat buffer per role, so the optimizer, the gradient all-reduce and the
reference-model snapshot are single large operations (MI355X: few, large HBM/xGMI transfers instead of thousands of small ones).

  trainable arena : bf16 working weights + fp32 master / m / v / grad   (LLM + patch-merger; reference timer1_trainer.py:272-280)
  frozen arena    : bf16 only (ViT patch-embed + blocks when fix_vit=True)
"""
import math
import re

import torch

from .config import ModelConfig

ALIGN = 64  # elements; keeps every view 128-byte aligned in bf16
SEG_ALIGN = 8 * ALIGN  # a SEGMENT (embed | one decoder layer | norm | lm_head | merger) starts on a multiple of this, so for world sizes 1/2/4/8
#                        it splits into `world` equal, 128-byte aligned chunks: the unit of the sharded optimizer's reduce-scatter / all-gather


def _segment_key(name):
    return name.split(".")[0]


def _specs_llm(cfg: ModelConfig):
    t = cfg.text
    s = [("embed", (t.vocab_size, t.hidden))]
    for i in range(t.n_layers):
        p = "l%d." % i
        s += [(p + "ln1", (t.hidden,)), (p + "qkv.w", (t.qkv_dim, t.hidden)), (p + "qkv.b", (t.qkv_dim,)), (p + "o.w", (t.hidden, t.q_dim)),
              (p + "ln2", (t.hidden,)), (p + "gu.w", (2 * t.intermediate, t.hidden)), (p + "down.w", (t.hidden, t.intermediate))]
    s += [("norm", (t.hidden,))]
    if not t.tie_word_embeddings:
        s += [("lm_head", (t.vocab_size, t.hidden))]
    return s


def _specs_merger(cfg: ModelConfig):
    v = cfg.vision
    m = v.embed_dim * v.merge_unit
    s = [("merger.ln.w", (v.embed_dim,))]
    if v.variant == "qwen2_vl":
        s += [("merger.ln.b", (v.embed_dim,))]
    s += [("merger.fc1.w", (m, m)), ("merger.fc1.b", (m,)), ("merger.fc2.w", (v.out_hidden, m)), ("merger.fc2.b", (v.out_hidden,))]
    return s


def _specs_vit(cfg: ModelConfig):
    v = cfg.vision
    s = [("patch.w", (v.embed_dim, v.patch_dim_padded))]
    if v.variant == "qwen2_5_vl":       # RMSNorm (no bias) + biased SwiGLU MLP, width padded with zeros to a multiple of 64
        ip = v.mlp_dim_padded
        for i in range(v.depth):
            p = "v%d." % i
            s += [(p + "n1.w", (v.embed_dim,)), (p + "qkv.w", (3 * v.embed_dim, v.embed_dim)), (p + "qkv.b", (3 * v.embed_dim,)),
                  (p + "proj.w", (v.embed_dim, v.embed_dim)), (p + "proj.b", (v.embed_dim,)), (p + "n2.w", (v.embed_dim,)),
                  (p + "gu.w", (2 * ip, v.embed_dim)), (p + "gu.b", (2 * ip,)), (p + "down.w", (v.embed_dim, ip)), (p + "down.b", (v.embed_dim,))]
        return s
    for i in range(v.depth):
        p = "v%d." % i
        s += [(p + "n1.w", (v.embed_dim,)), (p + "n1.b", (v.embed_dim,)), (p + "qkv.w", (3 * v.embed_dim, v.embed_dim)),
              (p + "qkv.b", (3 * v.embed_dim,)), (p + "proj.w", (v.embed_dim, v.embed_dim)), (p + "proj.b", (v.embed_dim,)),
              (p + "n2.w", (v.embed_dim,)), (p + "n2.b", (v.embed_dim,)), (p + "fc1.w", (v.mlp_dim, v.embed_dim)), (p + "fc1.b", (v.mlp_dim,)),
              (p + "fc2.w", (v.embed_dim, v.mlp_dim)), (p + "fc2.b", (v.embed_dim,))]
    return s


class Arena:
    def __init__(self, ops, specs, with_optimizer_state, with_grad=None):
        self.ops = ops
        self.specs = specs
        self.offsets = {}
        self.segments = []               # [(key, start, end)] contiguous, covering [0, numel); every boundary is a multiple of SEG_ALIGN
        off = 0
        for name, shape in specs:
            key = _segment_key(name)
            if not self.segments or self.segments[-1][0] != key:
                off = (off + SEG_ALIGN - 1) // SEG_ALIGN * SEG_ALIGN
                if self.segments:
                    self.segments[-1][2] = off
                self.segments.append([key, off, off])
            self.offsets[name] = (off, shape)
            off += (int(math.prod(shape)) + ALIGN - 1) // ALIGN * ALIGN
        off = (off + SEG_ALIGN - 1) // SEG_ALIGN * SEG_ALIGN
        if self.segments:
            self.segments[-1][2] = off
        self.segments = [tuple(s) for s in self.segments]
        self.numel = off
        self.shard = None                # (rank, world) once the optimizer state is sharded (set_shard)
        self.version = 0                 # bumped whenever the weights change (optimizer step, loaders): invalidates derived copies (Engine W^T cache)
        self.w16 = ops.zeros(off, dtype=ops.act_dtype)
        self.grad = self.master = self.m = self.v = None
        if with_grad is None:
            with_grad = with_optimizer_state
        if with_grad:
            self.grad = ops.zeros(off, dtype=torch.float32)
        if with_optimizer_state:
            self.master = ops.zeros(off, dtype=torch.float32)
            self.m = ops.zeros(off, dtype=torch.float32)
            self.v = ops.zeros(off, dtype=torch.float32)

    def view(self, flat, name):
        off, shape = self.offsets[name]
        return flat[off: off + int(math.prod(shape))].view(*shape)

    def w(self, name):
        return self.view(self.w16, name)

    def g(self, name):
        return self.view(self.grad, name)

    def names(self):
        return [n for n, _ in self.specs]

    def range_of(self, prefix):
        """[start, end) element range of the (contiguous) parameters whose name starts with `prefix`, padding included."""
        offs = [(self.offsets[n][0], n) for n, _ in self.specs if n.startswith(prefix)]
        if not offs:
            return 0, 0
        start = min(o for o, _ in offs)
        names = [n for n, _ in self.specs]
        last = max(names.index(n) for _, n in offs)
        end = self.offsets[names[last + 1]][0] if last + 1 < len(names) else self.numel
        return start, end

    def sync_master_from_w16(self):
        if self.master is not None:
            self.master.copy_(self.local_of(self.w16) if self.shard else self.w16)

    # ---- ZeRO-style optimizer-state sharding (reference scripts/zero3.json:22-33 shards optimizer state, gradients and parameters over the
    # data-parallel ranks; here 288 GB of HBM keep the bf16 weights and the fp32 gradient accumulator whole and only master / m / v - 12 of
    # the 18 bytes per parameter - are split): rank r owns the r-th of `world` equal chunks of EVERY segment, stored back to back.
    def chunks(self, rank=None, world=None):
        """[(global_start, global_end, local_start)] of the chunks `rank` owns (default: this arena's shard)."""
        r, w = self.shard if rank is None else (rank, world)
        out = []
        for _, a, b in self.segments:
            c = (b - a) // w
            out.append((a + r * c, a + (r + 1) * c, a // w))
        return out

    def local_of(self, flat):
        """The local shard (1/world of the arena) of a full flat tensor, as one contiguous tensor."""
        return torch.cat([flat[a:b] for a, b, _ in self.chunks()])

    def set_shard(self, rank, world):
        """Keep only this rank's 1/world of master / m / v.  Existing state is sliced; a weights-only arena gets master = fp32(w16), m = v = 0."""
        assert self.shard is None and SEG_ALIGN % (world * ALIGN) == 0, (self.shard, world)
        self.shard = (rank, world)
        n = self.numel // world
        for name in ("master", "m", "v"):
            full = getattr(self, name)
            if full is not None:
                setattr(self, name, self.local_of(full).clone())
            elif name == "master":
                self.master = self.ops.zeros(n, dtype=torch.float32)
                self.master.copy_(self.local_of(self.w16))
            else:
                setattr(self, name, self.ops.zeros(n, dtype=torch.float32))
            del full

    def clone_weights_only(self):
        """bf16-only snapshot (the frozen reference policy, reference timer1_trainer.py:295-307)."""
        a = Arena.__new__(Arena)
        a.ops, a.specs, a.offsets, a.numel, a.segments, a.shard = self.ops, self.specs, self.offsets, self.numel, self.segments, None
        a.w16 = self.w16.clone()
        a.grad = a.master = a.m = a.v = None
        return a


class ModelParams:
    """Qwen2-VL parameters: `train` arena (LLM + merger) and `frozen` arena (ViT)."""

    def __init__(self, cfg: ModelConfig, ops, seed=0, init="random", optimizer_state=True):
        self.cfg = cfg
        self.ops = ops
        self.train = Arena(ops, _specs_llm(cfg) + _specs_merger(cfg), with_optimizer_state=optimizer_state, with_grad=True)
        self.frozen = Arena(ops, _specs_vit(cfg), with_optimizer_state=False)
        if init == "random":
            self.init_random(seed)

    # lm_head is tied to the embedding for the 2B model (shared storage -> shared gradient view)
    def lm_head_w(self, arena=None):
        a = arena or self.train
        return a.w("embed") if self.cfg.text.tie_word_embeddings else a.w("lm_head")

    def lm_head_g(self):
        return self.train.g("embed") if self.cfg.text.tie_word_embeddings else self.train.g("lm_head")

    def init_random(self, seed=0, std=0.02):
        """SURVEY 8d synthetic weights: normal(0, 0.02), norm weights 1, biases small-random. Generated on host per tensor."""
        g = torch.Generator().manual_seed(seed)
        for arena in (self.train, self.frozen):
            for name, shape in arena.specs:
                if name.endswith("ln1") or name.endswith("ln2") or name == "norm" or name.endswith("ln.w") or name.endswith("n1.w") or name.endswith("n2.w"):
                    t = torch.ones(shape)
                elif name.endswith(".b"):
                    t = torch.randn(shape, generator=g) * std
                else:
                    t = torch.randn(shape, generator=g) * std
                    if name == "patch.w":
                        t[:, self.cfg.vision.patch_dim:] = 0
                arena.w(name).copy_(t.to(arena.w16.dtype))
        self._zero_vision_mlp_padding()
        self.train.sync_master_from_w16()

    def init_random_device(self, seed=0, std=0.02):
        """Same distribution as init_random but generated on the device (7B-scale benchmarks: no 30 GB host staging)."""
        dev = self.train.w16.device
        g = torch.Generator(device=dev).manual_seed(seed) if dev.type == "cuda" else torch.Generator().manual_seed(seed)
        for arena in (self.train, self.frozen):
            chunk = 1 << 28
            for a in range(0, arena.numel, chunk):
                b = min(arena.numel, a + chunk)
                arena.w16[a:b].copy_(torch.empty(b - a, dtype=torch.float32, device=dev).normal_(0, std, generator=g))
            for name, shape in arena.specs:
                if name.endswith("ln1") or name.endswith("ln2") or name == "norm" or name.endswith("ln.w") or name.endswith("n1.w") or name.endswith("n2.w"):
                    arena.w(name).fill_(1.0)
        self.frozen.w("patch.w")[:, self.cfg.vision.patch_dim:].zero_()
        self._zero_vision_mlp_padding()
        self.train.sync_master_from_w16()

    def _zero_vision_mlp_padding(self):
        v = self.cfg.vision
        if v.variant != "qwen2_5_vl" or v.mlp_dim_padded == v.mlp_dim:
            return
        i0, ip = v.mlp_dim, v.mlp_dim_padded
        for i in range(v.depth):
            p = "v%d." % i
            gu, gb, dw = self.frozen.w(p + "gu.w"), self.frozen.w(p + "gu.b"), self.frozen.w(p + "down.w")
            gu[i0:ip].zero_(); gu[ip + i0:].zero_(); gb[i0:ip].zero_(); gb[ip + i0:].zero_(); dw[:, i0:].zero_()

    # ---- HF checkpoint <-> arena ----------------------------------------------------------------------------------
    def load_hf_state_dict(self, sd):
        """Accepts transformers 4.51 (`model.layers.*`, `visual.*`) and 5.x (`model.language_model.*`, `model.visual.*`) key layouts."""
        self.train.version = getattr(self.train, "version", 0) + 1
        cfg = self.cfg

        def get(*cands):
            for c in cands:
                for k in sd:
                    if k.endswith(c):
                        return sd[k]
            raise KeyError(cands)

        tr, fz = self.train, self.frozen

        def put(arena, name, t):
            dst = arena.w(name)
            assert tuple(dst.shape) == tuple(t.shape), (name, dst.shape, t.shape)
            dst.copy_(t.to(dst.dtype))

        put(tr, "embed", get("embed_tokens.weight"))
        for i in range(cfg.text.n_layers):
            p, h = "l%d." % i, "layers.%d." % i
            put(tr, p + "ln1", get(h + "input_layernorm.weight"))
            put(tr, p + "qkv.w", torch.cat([get(h + "self_attn.q_proj.weight"), get(h + "self_attn.k_proj.weight"), get(h + "self_attn.v_proj.weight")], 0))
            put(tr, p + "qkv.b", torch.cat([get(h + "self_attn.q_proj.bias"), get(h + "self_attn.k_proj.bias"), get(h + "self_attn.v_proj.bias")], 0))
            put(tr, p + "o.w", get(h + "self_attn.o_proj.weight"))
            put(tr, p + "ln2", get(h + "post_attention_layernorm.weight"))
            put(tr, p + "gu.w", torch.cat([get(h + "mlp.gate_proj.weight"), get(h + "mlp.up_proj.weight")], 0))
            put(tr, p + "down.w", get(h + "mlp.down_proj.weight"))
        put(tr, "norm", get("language_model.norm.weight", "model.norm.weight"))
        if not cfg.text.tie_word_embeddings:
            put(tr, "lm_head", get("lm_head.weight"))
        put(tr, "merger.ln.w", get("merger.ln_q.weight"))
        if cfg.vision.variant == "qwen2_vl":
            put(tr, "merger.ln.b", get("merger.ln_q.bias"))
        put(tr, "merger.fc1.w", get("merger.mlp.0.weight"))
        put(tr, "merger.fc1.b", get("merger.mlp.0.bias"))
        put(tr, "merger.fc2.w", get("merger.mlp.2.weight"))
        put(tr, "merger.fc2.b", get("merger.mlp.2.bias"))
        v = cfg.vision
        pw = get("patch_embed.proj.weight").reshape(v.embed_dim, v.patch_dim)
        pwp = torch.zeros(v.embed_dim, v.patch_dim_padded, dtype=pw.dtype)
        pwp[:, : v.patch_dim] = pw
        put(fz, "patch.w", pwp)
        for i in range(v.depth):
            p, h = "v%d." % i, "blocks.%d." % i
            if v.variant == "qwen2_5_vl":
                i0, ip = v.mlp_dim, v.mlp_dim_padded
                put(fz, p + "n1.w", get(h + "norm1.weight"))
                put(fz, p + "n2.w", get(h + "norm2.weight"))
                for a, b in (("qkv", "attn.qkv"), ("proj", "attn.proj")):
                    put(fz, p + a + ".w", get(h + b + ".weight"))
                    put(fz, p + a + ".b", get(h + b + ".bias"))
                gw, uw, dw = get(h + "mlp.gate_proj.weight"), get(h + "mlp.up_proj.weight"), get(h + "mlp.down_proj.weight")
                guw = torch.zeros(2 * ip, v.embed_dim, dtype=gw.dtype)
                guw[:i0], guw[ip:ip + i0] = gw, uw
                gub = torch.zeros(2 * ip, dtype=gw.dtype)
                gub[:i0], gub[ip:ip + i0] = get(h + "mlp.gate_proj.bias"), get(h + "mlp.up_proj.bias")
                dwp = torch.zeros(v.embed_dim, ip, dtype=dw.dtype)
                dwp[:, :i0] = dw
                put(fz, p + "gu.w", guw)
                put(fz, p + "gu.b", gub)
                put(fz, p + "down.w", dwp)
                put(fz, p + "down.b", get(h + "mlp.down_proj.bias"))
                continue
            for a, b in (("n1", "norm1"), ("n2", "norm2"), ("qkv", "attn.qkv"), ("proj", "attn.proj"), ("fc1", "mlp.fc1"), ("fc2", "mlp.fc2")):
                put(fz, p + a + ".w", get(h + b + ".weight"))
                put(fz, p + a + ".b", get(h + b + ".bias"))
        self.train.sync_master_from_w16()

    def export_hf_state_dict(self):
        """Trainable + frozen weights under transformers-5.x key names (16-bit, like `stage3_gather_16bit_weights_on_model_save`)."""
        cfg, tr, fz = self.cfg, self.train, self.frozen
        t = cfg.text
        sd = {"model.language_model.embed_tokens.weight": tr.w("embed")}
        for i in range(t.n_layers):
            p, h = "l%d." % i, "model.language_model.layers.%d." % i
            qkvw, qkvb = tr.w(p + "qkv.w"), tr.w(p + "qkv.b")
            gu = tr.w(p + "gu.w")
            sd[h + "input_layernorm.weight"] = tr.w(p + "ln1")
            for nm, a, b in (("q_proj", 0, t.q_dim), ("k_proj", t.q_dim, t.q_dim + t.kv_dim), ("v_proj", t.q_dim + t.kv_dim, t.qkv_dim)):
                sd[h + "self_attn.%s.weight" % nm] = qkvw[a:b]
                sd[h + "self_attn.%s.bias" % nm] = qkvb[a:b]
            sd[h + "self_attn.o_proj.weight"] = tr.w(p + "o.w")
            sd[h + "post_attention_layernorm.weight"] = tr.w(p + "ln2")
            sd[h + "mlp.gate_proj.weight"] = gu[: t.intermediate]
            sd[h + "mlp.up_proj.weight"] = gu[t.intermediate:]
            sd[h + "mlp.down_proj.weight"] = tr.w(p + "down.w")
        sd["model.language_model.norm.weight"] = tr.w("norm")
        sd["lm_head.weight"] = self.lm_head_w()
        sd["model.visual.merger.ln_q.weight"] = tr.w("merger.ln.w")
        if cfg.vision.variant == "qwen2_vl":
            sd["model.visual.merger.ln_q.bias"] = tr.w("merger.ln.b")
        sd["model.visual.merger.mlp.0.weight"] = tr.w("merger.fc1.w")
        sd["model.visual.merger.mlp.0.bias"] = tr.w("merger.fc1.b")
        sd["model.visual.merger.mlp.2.weight"] = tr.w("merger.fc2.w")
        sd["model.visual.merger.mlp.2.bias"] = tr.w("merger.fc2.b")
        v = cfg.vision
        sd["model.visual.patch_embed.proj.weight"] = fz.w("patch.w")[:, : v.patch_dim].reshape(v.embed_dim, v.in_channels, v.temporal_patch_size, v.patch_size, v.patch_size)
        for i in range(v.depth):
            p, h = "v%d." % i, "model.visual.blocks.%d." % i
            if v.variant == "qwen2_5_vl":
                i0, ip = v.mlp_dim, v.mlp_dim_padded
                sd[h + "norm1.weight"], sd[h + "norm2.weight"] = fz.w(p + "n1.w"), fz.w(p + "n2.w")
                for a, b in (("qkv", "attn.qkv"), ("proj", "attn.proj")):
                    sd[h + b + ".weight"], sd[h + b + ".bias"] = fz.w(p + a + ".w"), fz.w(p + a + ".b")
                gu, gb = fz.w(p + "gu.w"), fz.w(p + "gu.b")
                sd[h + "mlp.gate_proj.weight"], sd[h + "mlp.up_proj.weight"] = gu[:i0], gu[ip:ip + i0]
                sd[h + "mlp.gate_proj.bias"], sd[h + "mlp.up_proj.bias"] = gb[:i0], gb[ip:ip + i0]
                sd[h + "mlp.down_proj.weight"], sd[h + "mlp.down_proj.bias"] = fz.w(p + "down.w")[:, :i0], fz.w(p + "down.b")
                continue
            for a, b in (("n1", "norm1"), ("n2", "norm2"), ("qkv", "attn.qkv"), ("proj", "attn.proj"), ("fc1", "mlp.fc1"), ("fc2", "mlp.fc2")):
                sd[h + b + ".weight"] = fz.w(p + a + ".w")
                sd[h + b + ".bias"] = fz.w(p + a + ".b")
        return {k: v_.detach().clone().contiguous().cpu() for k, v_ in sd.items()}
