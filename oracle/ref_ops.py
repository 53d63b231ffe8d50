"""ORACLE - TEST INFRASTRUCTURE ONLY.  CPU restatement (plain PyTorch, fp32 math) of every op behind the C ABI in
include/timer1_hip.h, with the same Python interface as time-r1_amd/ops.py:HipOps.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product path
(time-r1_amd/) never does and fails loudly when libtimer1_hip.so is missing.

Each function cites the reference line it restates ("ref:" = /root/reference, "TF:" =
transformers/models/qwen2_vl/modeling_qwen2_vl.py of transformers 5.15.0, the version importable in the build container;
the reference pins 4.51.1 - see SURVEY.md section 0).  Pinning: tests/golden/ holds outputs captured from the imported
reference + transformers; tests/test_oracle_vs_golden.py checks this file against them.
"""
import math

import numpy as np
import torch

F32 = torch.float32
I32 = torch.int32


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al. 2011), same constants as csrc/loss.hip."""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    mask = 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        h0, l0 = (p0 >> 32) & mask, p0 & mask
        h1, l1 = (p1 >> 32) & mask, p1 & mask
        c0, c1, c2, c3 = (h1 ^ c1 ^ k0) & mask, l1, (h0 ^ c3 ^ k1) & mask, l0
        k0 = (k0 + W0) & mask
        k1 = (k1 + W1) & mask
    return c0, c1, c2, c3


def philox_uniform(seed, row, step):
    c = philox4x32_10(row & 0xFFFFFFFF, step & 0xFFFFFFFF, 0, 0, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return (float(c[0] >> 8) + 0.5) * (1.0 / 16777216.0)


def visible_mask(pre, lo, hi, n_slots):
    """[T, n_slots] bool: key kv visible to token t iff kv < pre[t] or lo[t] <= kv <= hi[t] (csrc/attn_common.h)."""
    kv = torch.arange(n_slots)[None, :]
    pre, lo, hi = pre.long()[:, None], lo.long()[:, None], hi.long()[:, None]
    return (kv < pre) | ((kv >= lo) & (kv <= hi))


class RefOps:
    name = "ref"

    def __init__(self, act_dtype=torch.float32):
        self.act_dtype = act_dtype
        self.device = torch.device("cpu")

    # ---- memory helpers
    def empty(self, *shape, dtype=None):
        return torch.zeros(*shape, dtype=dtype or self.act_dtype)

    def zeros(self, *shape, dtype=None):
        return torch.zeros(*shape, dtype=dtype or self.act_dtype)

    def tensor(self, data, dtype):
        return torch.as_tensor(data, dtype=dtype)

    def _a(self, x):
        return x.to(self.act_dtype)

    # ---- GEMM (ref: nn.Linear, TF:501-504 / :459-466 / :251-274 / :277-290 / :1323)
    def gemm_nn(self, a, b):
        """C = a @ b with b K-major (dgrad of a Linear: dX = dY @ W)"""
        return a.float() @ b.float()

    def gemm_nt(self, a, b, bias=None, residual=None, out_f32=False, out=None, accumulate=False):
        c = a.float() @ b.float().t()
        if bias is not None:
            c = c + bias.float()
        if residual is not None:
            c = c + residual.float()
        if out is not None:
            if accumulate:
                out += c.to(out.dtype)
            else:
                out.copy_(c.to(out.dtype))
            return out
        return c if out_f32 else self._a(c)

    def transpose(self, x, pad_to=64, out=None, colsum=None):
        R, C = x.shape
        if colsum is not None:
            colsum += x.float().sum(0)
        Rp = (R + pad_to - 1) // pad_to * pad_to
        if out is None:
            out = torch.zeros(C, Rp, dtype=x.dtype)
        else:
            out.zero_()
        out[:, :R] = x.t()
        return out

    # ---- norms (ref: Qwen2RMSNorm TF:96-110; nn.LayerNorm)
    def rmsnorm_fwd(self, x, w, eps, residual=None, need_rstd=True, out=None, rstd_out=None):
        if out is not None or rstd_out is not None:
            y, rstd, xsum = self.rmsnorm_fwd(x, w, eps, residual=residual, need_rstd=need_rstd)
            if out is not None:
                out.copy_(y); y = out
            if rstd_out is not None and rstd is not None:
                rstd_out.copy_(rstd); rstd = rstd_out
            return y, rstd, xsum
        xsum = None
        if residual is not None:
            xsum = self._a(x.float() + residual.float())
            x = xsum
        x32 = x.float()
        rstd = torch.rsqrt(x32.pow(2).mean(-1) + eps)
        y = w.float() * self._a(x32 * rstd[:, None]).float()
        return self._a(y), rstd, xsum

    def rmsnorm_bwd(self, dy, x, w, rstd, dres=None, dw=None):
        xh = x.float() * rstd[:, None]
        gw = dy.float() * w.float()
        dot = (gw * xh).mean(-1, keepdim=True)
        dx = rstd[:, None] * (gw - xh * dot)
        if dres is not None:
            dx = dx + dres.float()
        if dw is not None:
            dw += (dy.float() * xh).sum(0)
        return self._a(dx)

    def layernorm_fwd(self, x, w, b, eps, need_stats=True):
        x32 = x.float()
        mean = x32.mean(-1)
        rstd = torch.rsqrt(x32.var(-1, unbiased=False) + eps)
        y = (x32 - mean[:, None]) * rstd[:, None] * w.float() + b.float()
        return self._a(y), mean, rstd

    def layernorm_bwd(self, dy, x, w, mean, rstd, dw, db, need_dx=False):
        xh = (x.float() - mean[:, None]) * rstd[:, None]
        g = dy.float()
        dw += (g * xh).sum(0)
        db += g.sum(0)
        if not need_dx:
            return None
        gw = g * w.float()
        dx = rstd[:, None] * (gw - gw.mean(-1, keepdim=True) - xh * (gw * xh).mean(-1, keepdim=True))
        return self._a(dx)

    # ---- activations (ref: TF:459-466 SwiGLU; TF:277-290 GELU; TF:293-301 quick_gelu)
    def norm_gemm_qkv(self, x, lnw, eps, wqkv, bias, cos, sin, kcache, vtcache, slots, n_heads, n_kv, head_dim):
        qkv = self.norm_gemm(x, lnw, eps, wqkv, bias=bias)
        return self.decode_qkv_post(qkv, cos, sin, kcache, vtcache, slots, n_heads, n_kv, head_dim)

    def quantize_fp8_rows(self, w, q=None, scale=None):
        """Per-row symmetric OCP e4m3 quantisation, same arithmetic as csrc/gemm_w8.hip (fp32: inv = 448 / amax, q = rne(w * inv))."""
        wf = w.float()
        amax = wf.abs().amax(1)
        ok = amax > 0
        c448 = torch.full_like(amax, 448.0)
        inv = torch.where(ok, c448 / amax, torch.ones_like(amax))      # tensor / tensor: correctly rounded (scalar / tensor is rcp * scalar in torch)
        sc = torch.where(ok, amax / c448, torch.ones_like(amax))
        qq = (wf * inv[:, None]).to(torch.float8_e4m3fn).view(torch.uint8)
        if q is not None:
            q.copy_(qq); qq = q
        if scale is not None:
            scale.copy_(sc); sc = scale
        return qq, sc

    @staticmethod
    def mx_quant_e4m3(x, block=32):
        """OCP-microscaling style activation quantisation of the W8A8 decode GEMM: per row and per `block` consecutive k one power-of-two
        scale 2^E, E = floor(log2(amax)) - 7 (values land in [128, 256) of e4m3's 448 range), elements rounded to e4m3 (nearest even);
        blocks with amax < 2^-119 are zero.  Returns the dequantised fp32 tensor."""
        M, K = x.shape
        xb = x.float().reshape(M, K // block, block)
        amax = xb.abs().amax(-1, keepdim=True)
        be = (amax.view(torch.int32) >> 23) & 0xff
        tiny = be < 8
        e = (be - 127 - 7).float()
        q = (xb * torch.exp2(-e)).to(torch.float8_e4m3fn).float() * torch.exp2(e)
        return torch.where(tiny, torch.zeros_like(q), q).reshape(M, K)

    def gemm_w8(self, x, q, scale, lnw=None, eps=1e-6, bias=None, residual=None, glu=False, a8=False):
        wd = q.view(torch.float8_e4m3fn).float()            # exactly representable in bf16: the kernel's register dequantisation is lossless
        if lnw is not None and a8:        # the fp8-MFMA kernel quantises x * lnw and applies the row's rstd to the accumulator
            xf = x.float()
            rstd = torch.rsqrt((xf * xf).mean(-1, keepdim=True) + eps)
            y = (self.mx_quant_e4m3(xf * lnw.float()[None, :]) @ wd.t()) * rstd * scale.float()[None, :]
        elif lnw is not None:
            x, _, _ = self.rmsnorm_fwd(x, lnw, eps, need_rstd=False)
            y = (x.float() @ wd.t()) * scale.float()[None, :]
        else:
            y = ((self.mx_quant_e4m3(x) if a8 else x.float()) @ wd.t()) * scale.float()[None, :]
        if bias is not None:
            y = y + bias.float()
        if residual is not None:
            y = y + residual.float()
        y = self._a(y)
        return self.swiglu_fwd(y) if glu else y

    def norm_gemm(self, x, lnw, eps, w, bias=None, glu=False):
        xn, _, _ = self.rmsnorm_fwd(x, lnw, eps, need_rstd=False)
        y = self.gemm_nt(xn, w, bias=bias)
        return self.swiglu_fwd(y) if glu else y

    # ---- fused-epilogue training GEMMs: by definition the compositions they replace (csrc/gemm.hip EPI 2 / 3 / 4)
    def gemm_quickgelu(self, x, w, bias=None):
        return self.quickgelu_fwd(self.gemm_nt(x, w, bias=bias))

    def gemm_glu(self, x, w_gu, a_out=None, gu_out=None, save_gu=True, bias=None):
        gu = self.gemm_nt(x, w_gu, bias=bias, out=gu_out)
        return self.swiglu_fwd(gu, out=a_out), (gu if save_gu else None)

    def gemm_qkv_rope(self, x, w_qkv, bias, cos, sin, n_heads, n_kv, head_dim, q_out=None, k_out=None, v_out=None):
        qd, kvd = n_heads * head_dim, n_kv * head_dim
        qkv = self.gemm_nt(x, w_qkv, bias=bias)
        q = self.rope_apply(qkv[:, :qd], n_heads, head_dim, cos, sin, out=q_out)
        k = self.rope_apply(qkv[:, qd:qd + kvd], n_kv, head_dim, cos, sin, out=k_out)
        v = qkv[:, qd + kvd:]
        if v_out is not None:
            v_out.copy_(v)
            v = v_out
        return q, k, v

    def dgrad_glu_bwd(self, dh, w_down, gu, want_t=False):
        dgu = self.swiglu_bwd(self.gemm_nn(dh, w_down), gu)
        return (dgu, None) if want_t else dgu

    def swiglu_fwd(self, gu, out=None):
        i = gu.shape[1] // 2
        g, u = gu[:, :i].float(), gu[:, i:].float()
        r = self._a(self._a(torch.nn.functional.silu(g)).float() * u)
        if out is not None:
            out.copy_(r)
            return out
        return r

    def swiglu_bwd(self, dout, gu):
        i = gu.shape[1] // 2
        g, u, d = gu[:, :i].float(), gu[:, i:].float(), dout.float()
        sg = torch.sigmoid(g)
        dg = d * u * (sg * (1 + g * (1 - sg)))
        du = d * (g * sg)
        return self._a(torch.cat([dg, du], 1))

    def gelu_fwd(self, x):
        return self._a(torch.nn.functional.gelu(x.float()))

    def gelu_bwd(self, x, dy):
        v = x.float()
        cdf = 0.5 * (1 + torch.erf(v / math.sqrt(2.0)))
        pdf = torch.exp(-0.5 * v * v) / math.sqrt(2 * math.pi)
        return self._a(dy.float() * (cdf + v * pdf))

    def quickgelu_fwd(self, x):
        v = x.float()
        return self._a(v * torch.sigmoid(1.702 * v))

    def add(self, a, b):
        return self._a(a.float() + b.float())

    def colsum_accum(self, dy, dbias):
        dbias += dy.float().sum(0)

    def cast_to_act(self, x_f32):
        return self._a(x_f32)

    def cast_to_f32(self, x):
        return x.float()

    # ---- rotary (ref: TF:117-222 M-RoPE, TF:225-248 vision rope)
    def mrope_table(self, pos3, head_dim, sections, theta):
        half = head_dim // 2
        inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))  # [half]
        axis = torch.cat([torch.full((s,), i, dtype=torch.long) for i, s in enumerate(sections)])  # [half]
        pos = pos3.float()[axis, :].t()  # [T, half]
        ang = pos * inv_freq[None, :]
        cos, sin = ang.cos(), ang.sin()
        # the reference casts cos/sin to the activation dtype (bf16 on GPU)
        return cos.to(torch.bfloat16).float().contiguous(), sin.to(torch.bfloat16).float().contiguous()

    def vision_rope_table(self, hw, head_dim, theta=10000.0):
        half = head_dim // 2
        q = half // 2
        inv_freq = 1.0 / (theta ** (torch.arange(0, half, 2, dtype=torch.float32) / half))  # [q]
        ang = torch.cat([hw[:, 0:1].float() * inv_freq[None, :], hw[:, 1:2].float() * inv_freq[None, :]], 1)  # [N, half]
        assert ang.shape[1] == half and q * 2 == half
        return ang.cos().contiguous(), ang.sin().contiguous()

    def rope_apply(self, x, n_heads, head_dim, cos, sin, backward=False, out=None):
        T = x.shape[0]
        half = head_dim // 2
        v = x[:, : n_heads * head_dim].float().reshape(T, n_heads, head_dim)
        a, b = v[..., :half], v[..., half:]
        c, s = cos[:, None, :], sin[:, None, :] * (-1.0 if backward else 1.0)
        r = torch.cat([a * c - b * s, b * c + a * s], -1).reshape(T, n_heads * head_dim)
        r = self._a(r)
        if out is not None:
            out[:, : n_heads * head_dim] = r
            return out
        return r

    # ---- gathers (ref: embed_tokens TF:1160, masked_scatter TF:1170-1176)
    def gather_rows(self, table, ids):
        return table[ids.long()].clone()

    def scatter_rows(self, src, idx, dst):
        dst[idx.long()] = src

    def embed_bwd(self, dout, ids, dtable):
        keep = ids >= 0
        dtable.index_add_(0, ids[keep].long(), dout[keep].float())

    # ---- attention (ref: TF:317-339 eager attention, softmax in fp32; varlen/causal structure via the two-interval mask)
    def pack_transpose(self, x, n_heads, n_kv, head_dim, ld_out=None, slots=None, out=None, zero_pad=True):
        T = x.shape[0]
        group = n_heads // n_kv
        if out is None:
            if ld_out is None:
                ld_out = (T * group + 63) // 64 * 64
            out = torch.zeros(n_kv * head_dim, ld_out, dtype=x.dtype)
        v = x[:, : n_heads * head_dim].reshape(T, n_kv, group, head_dim).permute(1, 3, 0, 2).reshape(n_kv * head_dim, T * group)
        if slots is not None:
            out[:, slots.long()] = v
        else:
            out[:, : T * group] = v
            if zero_pad:
                out[:, T * group:] = 0
        return out

    def scatter_slots(self, src, dst, slots):
        dst[slots.long(), : src.shape[1]] = src

    def decode_qkv_post(self, qkv, cos, sin, kcache, vtcache, slots, n_heads, n_kv, head_dim):
        qd, kvd = n_heads * head_dim, n_kv * head_dim
        q = self.rope_apply(qkv[:, :qd], n_heads, head_dim, cos, sin)
        k = self.rope_apply(qkv[:, qd:qd + kvd], n_kv, head_dim, cos, sin)
        kcache[slots.long()] = k
        vtcache[:, slots.long()] = qkv[:, qd + kvd:].t()
        return q

    def _dense_attn(self, q, k, v, pre, lo, hi, n_heads, n_kv, n_slots, head_dim, scale, chunk=1024):
        """Masked softmax attention, evaluated per block of query rows against only the key range those rows can see
        (a ViT frame segment, or the causal prefix) - same numbers as the dense [T, S] form, without materialising it."""
        T = q.shape[0]
        group = n_heads // n_kv
        kh_all = k[:n_slots].reshape(n_slots, n_kv, head_dim).transpose(0, 1)   # [n_kv, S, d]
        vh_all = v[:n_slots].reshape(n_slots, n_kv, head_dim).transpose(0, 1)
        outs, lses = [], []
        pre_l, lo_l, hi_l = pre.long(), lo.long(), hi.long()
        for a0 in range(0, T, chunk):
            a1 = min(T, a0 + chunk)
            pmax = int(pre_l[a0:a1].max())
            nonempty = hi_l[a0:a1] >= lo_l[a0:a1]
            if bool(nonempty.any()):
                lmin, hmax = int(lo_l[a0:a1][nonempty].min()), int(hi_l[a0:a1][nonempty].max())
            else:
                lmin, hmax = 0, -1
            k0 = 0 if pmax > 0 else max(0, min(lmin, n_slots))
            k1 = min(n_slots, max(pmax, hmax + 1))
            k1 = max(k1, k0 + 1)
            qh = q[a0:a1].reshape(a1 - a0, n_heads, head_dim).transpose(0, 1)                      # [H, t, d]
            kh = kh_all[:, k0:k1].repeat_interleave(group, 0)
            vh = vh_all[:, k0:k1].repeat_interleave(group, 0)
            s = (qh @ kh.transpose(1, 2)) * scale
            kv = torch.arange(k0, k1)[None, :]
            vis = (kv < pre_l[a0:a1, None]) | ((kv >= lo_l[a0:a1, None]) & (kv <= hi_l[a0:a1, None]))
            s = s.masked_fill(~vis[None], float("-inf"))
            lse = torch.logsumexp(s, -1)
            p = torch.nan_to_num(torch.exp(s - lse[..., None]), nan=0.0)
            outs.append((p @ vh).transpose(0, 1).reshape(a1 - a0, n_heads * head_dim))
            lses.append(lse)
        return torch.cat(outs, 0), torch.cat(lses, 1)

    def attn_fwd(self, q, k, vt, pre, lo, hi, n_heads, n_kv, n_slots, head_dim, scale, nsplit=1, need_lse=True, out=None, n_batch=1,
                 kv_batch_slots=0):
        if n_batch > 1:   # independent problems: rows [b*T,(b+1)*T) against cache slots [b*kv_batch_slots, ...)
            T = q.shape[0] // n_batch
            res = out if out is not None else torch.zeros(q.shape[0], n_heads * head_dim, dtype=self.act_dtype)
            for b in range(n_batch):
                a, e = b * T, (b + 1) * T
                s0 = b * kv_batch_slots
                self.attn_fwd(q[a:e], k[s0:], vt[:, s0:], pre[a:e], lo[a:e], hi[a:e], n_heads, n_kv, n_slots, head_dim, scale, need_lse=False,
                              out=res[a:e])
            return res, None
        v = vt[:, :n_slots].float().t().contiguous()  # [S, n_kv*hd]
        o, lse = self._dense_attn(q.float(), k.float(), v, pre, lo, hi, n_heads, n_kv, n_slots, head_dim, scale)
        if out is not None:
            out.copy_(self._a(o))
            return out, (lse if need_lse else None)
        return self._a(o), (lse if need_lse else None)

    def attn_bwd(self, q, k, v, o, do, lse, pre, lo, hi, n_heads, n_kv, n_slots, head_dim, scale, dv_out=None, dq_out=None, dk_out=None, rope=None):
        with torch.enable_grad():
            qf = q.float().detach().requires_grad_(True)
            kf = k[:n_slots].float().detach().requires_grad_(True)
            vf = v[:n_slots].float().detach().requires_grad_(True)
            of, _ = self._dense_attn(qf, kf, vf, pre, lo, hi, n_heads, n_kv, n_slots, head_dim, scale)
            dq, dk, dv = torch.autograd.grad(of, (qf, kf, vf), do.float())
        dq, dk, dv = self._a(dq), self._a(dk), self._a(dv)
        if rope is not None:        # gradients of the un-rotated projections: the transposed rotation, as the separate rope_apply(backward=True) calls
            dq = self.rope_apply(dq, n_heads, head_dim, rope[0], rope[1], backward=True)
            dk = self.rope_apply(dk, n_kv, head_dim, rope[0], rope[1], backward=True)
        outs = []
        for val, dst in ((dq, dq_out), (dk, dk_out), (dv, dv_out)):
            if dst is not None:
                dst.copy_(val)
                val = dst
            outs.append(val)
        return tuple(outs)

    # ---- video preprocessing (ref: vision_process.py:467-472 + HF video processor patchify)
    def video_preprocess(self, frames_u8, out_hw, k_pad, patch=14, temporal=2, merge=2, mean=None, std=None):
        from time_r1_amd import vision_process as VP
        x = VP.resize_frames(frames_u8, out_hw)
        pv, grid = VP.patchify(x, patch, temporal, merge)
        out = torch.zeros(pv.shape[0], k_pad, dtype=self.act_dtype)
        out[:, : pv.shape[1]] = pv.to(self.act_dtype)
        return out, grid

    # ---- vocabulary side (ref: timer1_trainer.py:458-481, :635-639, :713-737)
    def lmhead_lse(self, hn, w, targets):
        logits = self.gemm_nt(hn, w)              # rounded to the activation dtype like the materialised path
        return self.logp_entropy_fwd(logits, targets)

    def logp_entropy_fwd(self, logits, targets):
        lp = torch.log_softmax(logits.float(), -1)
        logp = lp.gather(1, targets.long()[:, None])[:, 0]
        ent = -(lp.exp() * lp).sum(-1)
        return logp, ent, torch.logsumexp(logits.float(), -1)

    def logp_bwd(self, logits, targets, lse, dlogp, inplace=True):
        p = torch.exp(logits.float() - lse[:, None])
        onehot = torch.zeros_like(p)
        onehot.scatter_(1, targets.long()[:, None], 1.0)
        d = self._a(dlogp[:, None] * (onehot - p))
        if inplace:
            logits.copy_(d)
            return logits
        return d

    def grpo_loss(self, logp, ref_logp, mask, adv, beta, use_grpo, grad_scale=1.0):
        G, C = logp.shape
        m = mask.float()
        kl = torch.zeros_like(logp)
        dkl = torch.zeros_like(logp)
        if ref_logp is not None:
            d = ref_logp - logp
            kl = torch.exp(d) - d - 1
            dkl = 1 - torch.exp(d)
        l = -adv[:, None] + beta * kl
        lens = m.sum(1)
        if use_grpo:
            loss = ((l * m).sum(1) / lens).mean()
            w = m / lens[:, None] / G
        else:
            loss = (l * m).sum() / m.sum()
            w = m / m.sum()
        dlogp = (-adv[:, None] + beta * dkl) * w * grad_scale
        klm = ((kl * m).sum(1) / lens).mean()
        out3 = torch.stack([loss, klm, m.sum()])
        return dlogp, out3, lens, (kl * m).sum(1)

    def sample_tokens(self, logits, temperature, top_k, seed, step_dev, tokens, finished, eos_id, pad_id, stop_at_eos, u_out=None, group_rows=0,
                      seed_stride=0):
        rows, V = logits.shape
        step = int(step_dev.item()) if step_dev is not None else 0
        x = logits.float().numpy().astype(np.float32) * np.float32(1.0 / temperature)
        for r in range(rows):
            if finished is not None and finished[r] and stop_at_eos:
                tokens[r, step] = pad_id
                continue
            xr = x[r]
            thr = -np.inf
            if top_k and 0 < top_k < V:
                thr = np.partition(xr, V - top_k)[V - top_k]
            keep = xr >= thr
            e = np.where(keep, np.exp((xr - xr.max()).astype(np.float64)), 0.0)
            u = philox_uniform(int(seed) + (r // group_rows) * int(seed_stride), r % group_rows, step) if group_rows else philox_uniform(int(seed), r, step)
            if u_out is not None:
                u_out[r] = u
            cdf = np.cumsum(e)
            tok = int(np.searchsorted(cdf, u * cdf[-1], side="left"))
            tok = min(tok, V - 1)
            while not keep[tok]:
                tok -= 1
            tokens[r, step] = tok
            if finished is not None and tok == eos_id:
                finished[r] = 1

    # ---- optimizer (ref: torch.optim.AdamW semantics = DeepSpeed FusedAdam adam_w_mode; clip = clip_grad_norm_)
    def sumsq_ranges_periodic(self, g, base, stride, count, rel_ranges, out_scalar):
        for l in range(count):
            for a, b in rel_ranges:
                out_scalar += (g[base + l * stride + a: base + l * stride + b].double() ** 2).sum().float()

    def zero_ranges_periodic(self, g, base, stride, count, rel_ranges):
        for l in range(count):
            for a, b in rel_ranges:
                g[base + l * stride + a: base + l * stride + b] = 0

    def sumsq_accum(self, g, out_scalar):
        out_scalar += (g.double() ** 2).sum().float()

    def adamw_step(self, p32, m, v, g, p16, lr, beta1, beta2, eps, weight_decay, step, sumsq=None, max_norm=0.0, grad_mult=1.0,
                   zero_grad=True, g16=None):
        g_acc = g
        if g16 is not None:
            g = g16.float()
        coef = grad_mult
        if sumsq is not None and max_norm > 0:
            norm = float(sumsq.sqrt()) * grad_mult
            coef = grad_mult * min(1.0, max_norm / (norm + 1e-6))
        gg = g * coef
        p32.mul_(1 - lr * weight_decay)
        m.mul_(beta1).add_(gg, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(gg, gg, value=1 - beta2)
        bc1 = 1 - beta1 ** step
        bc2 = 1 - beta2 ** step
        denom = v.sqrt() / math.sqrt(bc2) + eps
        p32.addcdiv_(m, denom, value=-lr / bc1)
        p16.copy_(p32.to(p16.dtype))
        if zero_grad:
            g_acc.zero_()
