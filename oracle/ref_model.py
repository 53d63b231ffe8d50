"""ORACLE - TEST INFRASTRUCTURE ONLY.  Plain-PyTorch (autograd) restatement of the Qwen2-VL forward exactly as the reference runs it:
the prompt is REPLICATED G times and every row is a full causal sequence (reference src/time_r1/rl/timer1_trainer.py:592-607 calls
model(input_ids[G, P+C], attention_mask, pixel_values_videos.repeat(G,1), video_grid_thw x G)).  Nothing is shared with the product
engine (time-r1_amd/model.py) except the parameter naming of params.py, so engine-vs-oracle agreement checks the packed / shared-prefix
formulation, the hand-written backward and the HIP kernels all at once.

Restated from transformers/models/qwen2_vl/modeling_qwen2_vl.py (v5.15.0): PatchEmbed :251-274, VisionBlock :425-449 (LayerNorm,
fused qkv, 2-D rope :225-248, per-frame attention via cu_seqlens vision_utils.py:42-65, quick_gelu MLP :293-301), PatchMerger :277-290,
Qwen2RMSNorm :96-110, M-RoPE :117-222, attention :317-339/:501-556 (GQA by repeat_kv, softmax fp32), MLP :459-466, decoder layer
:559-624, get_rope_index :914-1016.   Pinned against transformers + the reference by tests/test_oracle_vs_golden.py.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], -1)


def vision_tower_25(W, cfg, pixels, grid_thw):
    """Qwen2.5-VL tower restated WITHOUT the window permutation: attention is permutation-equivariant, so a block's output in
    natural patch order equals HF's (modeling_qwen2_5_vl.py:408-470) after its reorder -> blocks -> merger -> argsort round trip.
    Window membership follows vision_utils.py:130-188: merged token (t, i, j) of a video lies in window (t, i // mw, j // mw),
    mw = window_size // merge // patch.  RMSNorm :64-79, biased SwiGLU MLP :85-96 (only the first mlp_dim rows of the padded
    gate/up halves are read), merger with RMSNorm :135-149."""
    v = cfg.vision
    E, H, hd, m = v.embed_dim, v.num_heads, v.head_dim, v.spatial_merge_size
    i0, ip = v.mlp_dim, v.mlp_dim_padded
    x = pixels @ W["patch.w"][:, : v.patch_dim].t()
    mw = v.window_size // m // v.patch_size
    ids, frame, win = [], [], []
    nf = nwin = 0
    for t, h, w in grid_thw:
        hh, ww = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        hh = hh.reshape(h // m, m, w // m, m).permute(0, 2, 1, 3).reshape(-1)
        ww = ww.reshape(h // m, m, w // m, m).permute(0, 2, 1, 3).reshape(-1)
        ids.append(torch.stack([hh, ww], -1).repeat(t, 1))
        wid = (hh // m // mw) * 10000 + (ww // m // mw)          # window of each patch inside one temporal patch
        for ti in range(t):
            frame.append(torch.full((h * w,), nf + ti))
            win.append(wid + (nwin + ti) * 100000000)
        nf += t
        nwin += t
    ids, frame, win = torch.cat(ids, 0), torch.cat(frame), torch.cat(win)
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, hd // 2, 2, dtype=torch.float32) / (hd // 2)))
    freqs = (ids[:, :, None].float() * inv_freq[None, None, :]).flatten(1)
    emb = torch.cat([freqs, freqs], -1)
    cos, sin = emb.cos()[:, None, :], emb.sin()[:, None, :]
    mask_full = frame[:, None] == frame[None, :]
    mask_win = win[:, None] == win[None, :]
    N = x.shape[0]

    def rms(h, w):
        return w * (h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + v.ln_eps))
    for i in range(v.depth):
        p = "v%d." % i
        mask = mask_full if i in v.fullatt_block_indexes else mask_win
        qkv = (rms(x, W[p + "n1.w"]) @ W[p + "qkv.w"].t() + W[p + "qkv.b"]).reshape(N, 3, H, hd)
        q, k, val = qkv[:, 0], qkv[:, 1], qkv[:, 2]
        q = q * cos + _rot_half(q) * sin
        k = k * cos + _rot_half(k) * sin
        sc = torch.einsum("qhd,khd->hqk", q, k) * hd ** -0.5
        sc = sc.masked_fill(~mask[None], float("-inf"))
        o = torch.einsum("hqk,khd->qhd", sc.softmax(-1), val).reshape(N, E)
        x = x + o @ W[p + "proj.w"].t() + W[p + "proj.b"]
        y = rms(x, W[p + "n2.w"])
        gate = y @ W[p + "gu.w"][:i0].t() + W[p + "gu.b"][:i0]
        up = y @ W[p + "gu.w"][ip:ip + i0].t() + W[p + "gu.b"][ip:ip + i0]
        x = x + (F.silu(gate) * up) @ W[p + "down.w"][:, :i0].t() + W[p + "down.b"]
    y = rms(x, W["merger.ln.w"]).reshape(N // v.merge_unit, E * v.merge_unit)
    y = F.gelu(y @ W["merger.fc1.w"].t() + W["merger.fc1.b"])
    return y @ W["merger.fc2.w"].t() + W["merger.fc2.b"]


def vision_tower(W, cfg, pixels, grid_thw):
    """W: dict name -> tensor (params.py names). pixels [N_v, patch_dim]. Returns merged video embeddings [N_v/4, out_hidden]."""
    v = cfg.vision
    if v.variant == "qwen2_5_vl":
        return vision_tower_25(W, cfg, pixels, grid_thw)
    E, H, hd = v.embed_dim, v.num_heads, v.head_dim
    x = pixels @ W["patch.w"][:, : v.patch_dim].t()
    # 2-D rotary ids in merge-block order (vision_utils.py:81-127)
    ids = []
    for t, h, w in grid_thw:
        hh, ww = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        m = v.spatial_merge_size
        hh = hh.reshape(h // m, m, w // m, m).permute(0, 2, 1, 3).reshape(-1)
        ww = ww.reshape(h // m, m, w // m, m).permute(0, 2, 1, 3).reshape(-1)
        ids.append(torch.stack([hh, ww], -1).repeat(t, 1))
    ids = torch.cat(ids, 0)
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, hd // 2, 2, dtype=torch.float32) / (hd // 2)))
    freqs = (ids[:, :, None].float() * inv_freq[None, None, :]).flatten(1)        # [N, hd/2] = [h freqs | w freqs]
    emb = torch.cat([freqs, freqs], -1)
    cos, sin = emb.cos()[:, None, :], emb.sin()[:, None, :]
    seg = torch.repeat_interleave(torch.arange(sum(t for t, _, _ in grid_thw)), torch.tensor([h * w for t, h, w in grid_thw for _ in range(t)]))
    mask = seg[:, None] == seg[None, :]
    N = x.shape[0]
    for i in range(v.depth):
        p = "v%d." % i
        y = F.layer_norm(x, (E,), W[p + "n1.w"], W[p + "n1.b"], v.ln_eps)
        qkv = (y @ W[p + "qkv.w"].t() + W[p + "qkv.b"]).reshape(N, 3, H, hd)
        q, k, val = qkv[:, 0], qkv[:, 1], qkv[:, 2]
        q = q * cos + _rot_half(q) * sin
        k = k * cos + _rot_half(k) * sin
        s = torch.einsum("qhd,khd->hqk", q, k) * hd ** -0.5
        s = s.masked_fill(~mask[None], float("-inf"))
        o = torch.einsum("hqk,khd->qhd", s.softmax(-1), val).reshape(N, E)
        x = x + o @ W[p + "proj.w"].t() + W[p + "proj.b"]
        y = F.layer_norm(x, (E,), W[p + "n2.w"], W[p + "n2.b"], v.ln_eps)
        z = y @ W[p + "fc1.w"].t() + W[p + "fc1.b"]
        x = x + (z * torch.sigmoid(1.702 * z)) @ W[p + "fc2.w"].t() + W[p + "fc2.b"]
    y = F.layer_norm(x, (E,), W["merger.ln.w"], W["merger.ln.b"], v.ln_eps).reshape(N // v.merge_unit, E * v.merge_unit)
    y = F.gelu(y @ W["merger.fc1.w"].t() + W["merger.fc1.b"])
    return y @ W["merger.fc2.w"].t() + W["merger.fc2.b"]


def rope_index_ref(ids, grid_thw, video_token_id, merge, mode, interval=1):
    """Per-sequence 3-D positions (one row). Same rule as positions.rope_index but written independently (loop form)."""
    pos = [[], [], []]
    cur, i, gi, L = 0, 0, 0, len(ids)
    while i < L:
        if ids[i] != video_token_id:
            for a in range(3):
                pos[a].append(cur)
            cur += 1
            i += 1
        else:
            t, h, w = grid_thw[gi]
            gi += 1
            gh, gw = h // merge, w // merge
            mx = cur
            for tt in range(t):
                for hh in range(gh):
                    for ww in range(gw):
                        pos[0].append(cur + tt * interval); pos[1].append(cur + hh); pos[2].append(cur + ww)
                        mx = max(mx, cur + tt * interval, cur + hh, cur + ww)
            i += t * gh * gw
            cur = cur + max(h, w) // merge if mode == "hf5" else mx + 1
    return torch.tensor(pos)


def llm_logits(W, cfg, input_ids, vid_embeds, grid_thw, rope_mode="hf5"):
    """input_ids [B, L] (every row holds the same prompt incl. video pads), vid_embeds [T_vid, d] -> logits [B, L, V]."""
    t = cfg.text
    B, L = input_ids.shape
    x = W["embed"][input_ids]
    vid_mask = input_ids == cfg.video_token_id
    x = x.clone()
    x[vid_mask] = vid_embeds.repeat(B, 1).to(x.dtype)
    interval = int(cfg.tokens_per_second) if cfg.vision.variant == "qwen2_5_vl" else 1     # modeling_qwen2_5_vl.py:1043, 1 s per grid step
    pos = torch.stack([rope_index_ref(input_ids[b].tolist(), grid_thw, cfg.video_token_id, cfg.vision.spatial_merge_size, rope_mode, interval)
                       for b in range(B)], 1)  # [3,B,L]
    inv_freq = 1.0 / (t.rope_theta ** (torch.arange(0, t.head_dim, 2, dtype=torch.float32) / t.head_dim))
    freqs = pos[..., None].float() * inv_freq            # [3, B, L, hd/2]
    sec = list(t.mrope_section)
    chunks = torch.split(freqs, sec, dim=-1)
    f = torch.cat([chunks[i][i % 3] for i in range(3)], -1)   # [B, L, hd/2]
    emb = torch.cat([f, f], -1)
    cos, sin = emb.cos()[:, None], emb.sin()[:, None]          # [B,1,L,hd]
    causal = torch.tril(torch.ones(L, L, dtype=torch.bool))
    g = t.n_heads // t.n_kv_heads

    def rms(h, w):
        return w * (h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + t.rms_eps))
    for i in range(t.n_layers):
        p = "l%d." % i
        h = rms(x, W[p + "ln1"])
        qkv = h @ W[p + "qkv.w"].t() + W[p + "qkv.b"]
        q = qkv[..., : t.q_dim].reshape(B, L, t.n_heads, t.head_dim).transpose(1, 2)
        k = qkv[..., t.q_dim: t.q_dim + t.kv_dim].reshape(B, L, t.n_kv_heads, t.head_dim).transpose(1, 2)
        v = qkv[..., t.q_dim + t.kv_dim:].reshape(B, L, t.n_kv_heads, t.head_dim).transpose(1, 2)
        q = q * cos + _rot_half(q) * sin
        k = k * cos + _rot_half(k) * sin
        k, v = k.repeat_interleave(g, 1), v.repeat_interleave(g, 1)
        s = (q @ k.transpose(-1, -2)) * t.head_dim ** -0.5
        s = s.masked_fill(~causal, float("-inf"))
        o = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, L, t.q_dim)
        x = x + o @ W[p + "o.w"].t()
        h = rms(x, W[p + "ln2"])
        gu = h @ W[p + "gu.w"].t()
        x = x + (F.silu(gu[..., : t.intermediate]) * gu[..., t.intermediate:]) @ W[p + "down.w"].t()
    x = rms(x, W["norm"])
    head = W["embed"] if t.tie_word_embeddings else W["lm_head"]
    return x @ head.t()


def per_token_logps(logits, input_ids):
    """reference _get_per_token_logps (timer1_trainer.py:458-481): shift, log_softmax, gather, entropy."""
    lg = logits[:, :-1]
    tg = input_ids[:, 1:]
    lp = lg.log_softmax(-1)
    return lp.gather(-1, tg[..., None])[..., 0], -(lp.exp() * lp).sum(-1)


def weights_from_params(params, requires_grad=False, dtype=torch.float32):
    """dict name -> fp32 CPU tensor (leaf) from a ModelParams (any backend)."""
    W = {}
    for arena in (params.train, params.frozen):
        for name in arena.names():
            W[name] = arena.w(name).detach().to("cpu").to(dtype).clone()
    if requires_grad:
        for n in params.train.names():
            W[n].requires_grad_(True)
    return W
