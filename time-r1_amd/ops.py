"""Tensor-level wrappers over the C ABI (include/timer1_hip.h). PyTorch is used for device memory and streams only.

`HipOps` is the product backend. The engine (model.py / rollout.py / grpo.py) is written against this small interface so that
tests can drive the same host logic with `oracle.ref_ops.RefOps` on CPU - the product never imports oracle/.
"""
import os

import torch

from . import hip

BF16 = torch.bfloat16
F32 = torch.float32
I32 = torch.int32


def _p(t):
    return 0 if t is None else t.data_ptr()


def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1, "expected a row-major 2-D tensor/view, got strides %s" % (t.stride(),)
    return t.stride(0)


class HipOps:
    """All compute goes through libtimer1_hip.so on the current torch HIP stream. No fallbacks."""

    name = "hip"
    act_dtype = BF16

    def __init__(self, device="cuda:0"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise hip.HipError("HipOps needs a HIP device (got %s); there is no CPU fallback" % device)
        self.L = hip.lib()
        self._ws = {}

    def use_priority_stream(self):
        """Make a HIGH-priority HIP stream the current stream of this process (call once, before any device work).  The backward pass
        runs its weight gradients on a normal-priority side stream (Engine._side_stream): with both at the same priority every small
        kernel of the main chain queues behind the pending workgroups of a 1-block-per-CU weight-gradient GEMM (~150 us per launch);
        with the main chain ahead in the dispatcher's arbitration the 7B backward drops from 171 to 164 ms.  TR1_MAIN_PRIO=0 keeps the
        default stream (A/B measurements).
        With a torch.distributed process group in the process the default is the DEFAULT stream: the group's streams take HIP past its four hardware
        queues, and with the main chain on a priority stream every small kernel of the forward / backward then starts late (measured on one MI355X with a
        single-rank RCCL group, tools/ab_dp_single_rank.sh: log-probs 87 -> 101 ms, backward 152 -> 161 ms per micro-step; the default stream, or
        GPU_MAX_HW_QUEUES=2, restores 87 / 152).  TR1_MAIN_PRIO=1 forces the priority stream there too."""
        want = os.environ.get("TR1_MAIN_PRIO")
        if want == "0":
            return None
        if want is None:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                return None
        if getattr(self, "_main_stream", None) is None:
            self._main_stream = torch.cuda.Stream(device=self.device, priority=-1)
            torch.cuda.set_stream(self._main_stream)
        return self._main_stream

    def probe_hbm_read(self, buf, sink):
        """Measurement helper: one streaming read of `buf` (bench.py times it)."""
        self.L.call("tr1_probe_hbm_read", _p(buf), buf.numel() * buf.element_size(), _p(sink), self._s())

    def decode_profile_begin(self):
        """Measurement helper: native decode steps record HIP events around their projection GEMMs until decode_profile_end()."""
        self.L.call("tr1_decode_profile_begin")

    def decode_profile_end(self):
        """-> {family: (sum_ms, min_ms, launches)} for qkv / o / gate_up / down / lm_head since decode_profile_begin()."""
        import ctypes
        ms, mn, n = (ctypes.c_double * 5)(), (ctypes.c_double * 5)(), (ctypes.c_int64 * 5)()
        self.L.call("tr1_decode_profile_end", ms, mn, n)
        return {k: (ms[i], mn[i], int(n[i])) for i, k in enumerate(("qkv", "o", "gate_up", "down", "lm_head"))}

    # ---- memory helpers -------------------------------------------------------------------------------------------
    def empty(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or self.act_dtype, device=self.device)

    def zeros(self, *shape, dtype=None):
        return torch.zeros(*shape, dtype=dtype or self.act_dtype, device=self.device)

    def tensor(self, data, dtype):
        """Host data -> device tensor through PINNED staging memory and an asynchronous copy.  A copy from pageable memory is stream-ordered
        AND blocks the host until it has run, i.e. until every kernel queued before it has finished - each small index / mask / position
        table then drains the GPU queue and the host-side table building behind it shows up as GPU idle time (~15 ms per micro-step)."""
        t = torch.as_tensor(data, dtype=dtype)
        if t.device.type != "cpu":
            return t.to(self.device)
        if not t.is_contiguous():
            t = t.contiguous()
        return t.pin_memory().to(self.device, non_blocking=True)

    def _s(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _workspace(self, key, numel, dtype):
        key = (key, self._s())            # one scratch buffer per (purpose, stream): launches on different streams may overlap
        t = self._ws.get(key)
        if t is None or t.numel() < numel or t.dtype != dtype:
            t = torch.empty(max(int(numel), 1), dtype=dtype, device=self.device)
            self._ws[key] = t
        return t

    def _chk(self, *ts, dtype=BF16):
        for t in ts:
            if t is not None:
                assert t.device.type == "cuda" and t.dtype == dtype, (t.device, t.dtype, dtype)

    # ---- GEMM -----------------------------------------------------------------------------------------------------
    def _splitk_ok(self, M, N, K):
        """Few 256 x 256 tiles, many K tiles: the split-K form pays (class attribute SPLITK: A/B runs clear it)."""
        return bool(self.SPLITK and M > 64 and K >= 2048 and K % 64 == 0 and N % 8 == 0 and ((M + 255) // 256) * ((N + 255) // 256) <= 128 and 8 * M * N * 4 <= (1 << 31))

    def gemm_nn(self, a, b):
        """C[M,N] = a[M,K] @ b[K,N] (b K-major, e.g. the weight itself in a dgrad).  Large problems run the K-major form of the phased GEMM
        (no transposed copy); small ones transpose b and use gemm_nt."""
        self._chk(a, b)
        M, K = a.shape
        N = b.shape[1]
        assert b.shape[0] == K and a.stride(1) == 1 and b.stride(1) == 1
        # the K-major form exists for the 8-wave tiles only: with fewer than ~3/4 of the CUs covered by 256 x 256 tiles (the lm_head dgrad:
        # 1608 x 3584 outputs over K = 152064) the NT dispatch's 128 x 128 tiles on a transposed copy are faster (3.7 ms -> 2.6 ms)
        if M >= 512 and N >= 256 and K % 64 == 0 and N % 8 == 0 and ((M + 255) // 256) * ((N + 255) // 256) >= 192:
            c = self.empty(M, N)
            self.L.call("tr1_gemm_nn_bf16", _p(a), _p(b), _p(c), M, N, K, _ld(a), _ld(b), _ld(c), self._s())
            return c
        if self._splitk_ok(M, N, K) and a.stride(1) == 1 and b.stride(1) == 1:      # e.g. the lm_head's data gradient: the weight as stored, K = vocabulary
            c = self.empty(M, N)
            ws = self._workspace("gemm_splitk", 8 * M * N, F32)
            self.L.call("tr1_gemm_splitk_bf16", _p(a), _p(b), _p(c), None, None, M, N, K, _ld(a), _ld(b), _ld(c), 0, 1, _p(ws), ws.numel(), self._s())
            return c
        return self.gemm_nt(a, self.transpose(b))

    def wgrad_nn(self, dyt, x, gw, accumulate):
        """gw[N,K] fp32 (+)= dyt[N, Mp] @ x[M, K]  (dyt = transpose(dy), zero-padded to Mp = M rounded up to 64; x as stored).  Returns False
        when the K-major form does not cover the shape (the caller then transposes x and uses gemm_nt)."""
        N, Mp = dyt.shape
        M, K = x.shape
        if not (N >= 512 and K >= 256 and K % 8 == 0 and Mp % 64 == 0 and Mp >= M and x.stride(1) == 1 and x.stride(0) % 8 == 0 and dyt.stride(1) == 1):
            return False
        self._chk(dyt, x)
        assert gw.dtype == F32 and gw.shape == (N, K)
        self.L.call("tr1_gemm_nn_acc_f32", _p(dyt), _p(x), _p(gw), N, K, Mp, _ld(dyt), _ld(x), _ld(gw), int(accumulate), M, self._s())
        return True

    def gemm_nt(self, a, b, bias=None, residual=None, out_f32=False, out=None, accumulate=False):
        """C[M,N] = a[M,K] @ b[N,K]^T (+bias) (+residual); bf16 in, fp32 accumulate. K must be a multiple of 64."""
        self._chk(a, b, bias, residual)
        M, K = a.shape
        N, K2 = b.shape
        assert K == K2, (a.shape, b.shape)
        if out is None:
            assert not accumulate
            out = self.empty(M, N, dtype=F32 if out_f32 else BF16)
        assert out.shape == (M, N) and out.dtype == (F32 if out_f32 else BF16)
        # thin output over a long reduction (continuation down / o projections: 98 tiles of 256 x 256 for 256 CUs): deterministic S-way split-K
        if not out_f32 and self._splitk_ok(M, N, K) and a.stride(1) == 1 and b.stride(1) == 1:
            ws = self._workspace("gemm_splitk", 8 * M * N, F32)
            self.L.call("tr1_gemm_splitk_bf16", _p(a), _p(b), _p(out), _p(bias), _p(residual), M, N, K, _ld(a), _ld(b), _ld(out),
                        _ld(residual) if residual is not None else 0, 0, _p(ws), ws.numel(), self._s())
            return out
        self.L.call("tr1_gemm_nt_bf16", _p(a), _p(b), _p(out), _p(bias), _p(residual), M, N, K, _ld(a), _ld(b), _ld(out),
                    _ld(residual) if residual is not None else 0, int(out_f32), int(accumulate), self._s())
        return out

    # ---- fused-epilogue training GEMMs (bit-identical to the compositions in their `else` branches) ---------------------
    SPLITK = True        # the S-way split-K form of thin long-K GEMMs (class attribute for A/B runs)
    FUSE_EPI = os.environ.get("TR1_FUSE_EPI", "1") != "0"          # A/B switch: 0 = GEMM + separate elementwise kernels (the round-3 path)

    def gemm_quickgelu(self, x, w, bias=None):
        """quick_gelu(x @ w^T + bias) in one launch (Qwen2-VL vision MLP)."""
        self._chk(x, w, bias)
        M, K = x.shape
        N = w.shape[0]
        if self.FUSE_EPI and M > 64 and K % 64 == 0 and N % 8 == 0 and x.stride(1) == 1 and w.stride(1) == 1:
            y = self.empty(M, N)
            self.L.call("tr1_gemm_bias_quickgelu_bf16", _p(x), _p(w), _p(bias), _p(y), M, N, K, _ld(x), _ld(w), _ld(y), self._s())
            return y
        return self.quickgelu_fwd(self.gemm_nt(x, w, bias=bias))

    def gemm_glu(self, x, w_gu, a_out=None, gu_out=None, save_gu=True, bias=None):
        """(a, gu): a[M, I] = silu(x Wg^T + bg) * (x Wu^T + bu); gu = the projection [M, 2I] (None unless save_gu)."""
        self._chk(x, w_gu, a_out, gu_out, bias)
        M, K = x.shape
        I = w_gu.shape[0] // 2
        if self.FUSE_EPI and M > 64 and K % 64 == 0 and I % 8 == 0 and x.stride(1) == 1 and w_gu.stride(1) == 1:
            a = a_out if a_out is not None else self.empty(M, I)
            gu = (gu_out if gu_out is not None else self.empty(M, 2 * I)) if save_gu else None
            self.L.call("tr1_gemm_glu_bf16", _p(x), _p(w_gu), _p(bias), _p(a), _p(gu), M, I, K, _ld(x), _ld(w_gu), _ld(a), _ld(gu) if gu is not None else 0, self._s())
            return a, gu
        gu = self.gemm_nt(x, w_gu, bias=bias, out=gu_out)
        return self.swiglu_fwd(gu, out=a_out), (gu if save_gu else None)

    def gemm_qkv_rope(self, x, w_qkv, bias, cos, sin, n_heads, n_kv, head_dim, q_out=None, k_out=None, v_out=None):
        """(q, k, v) of the fused projection with bias, q and k rotated (M-RoPE tables cos / sin fp32 [M, head_dim / 2]); outputs may be views."""
        self._chk(x, w_qkv, bias, q_out, k_out, v_out)
        M, K = x.shape
        qd, kvd = n_heads * head_dim, n_kv * head_dim
        if (self.FUSE_EPI and M > 64 and head_dim == 128 and n_heads % 2 == 0 and n_kv % 2 == 0 and K % 64 == 0 and x.stride(1) == 1 and w_qkv.stride(1) == 1
                and bias is not None):
            assert cos.dtype == F32 and sin.dtype == F32 and cos.is_contiguous() and sin.is_contiguous() and cos.shape == (M, 64)
            q = q_out if q_out is not None else self.empty(M, qd)
            k = k_out if k_out is not None else self.empty(M, kvd)
            v = v_out if v_out is not None else self.empty(M, kvd)
            self.L.call("tr1_gemm_qkv_rope_bf16", _p(x), _p(w_qkv), _p(bias), _p(cos), _p(sin), _p(q), _ld(q), _p(k), _ld(k), _p(v), _ld(v), M, n_heads, n_kv,
                        head_dim, K, _ld(x), _ld(w_qkv), self._s())
            return q, k, v
        qkv = self.gemm_nt(x, w_qkv, bias=bias)
        q = self.rope_apply(qkv[:, :qd], n_heads, head_dim, cos, sin, out=q_out)
        k = self.rope_apply(qkv[:, qd:qd + kvd], n_kv, head_dim, cos, sin, out=k_out)
        v = qkv[:, qd + kvd:]
        if v_out is not None:
            v_out.copy_(v)
            v = v_out
        return q, k, v

    def vit_pad128_ok(self, n_heads, head_dim):
        """The vision tower can run on 128-wide zero-padded heads (fused q|k|v + rotary epilogue -> head-dim-128 attention kernel)."""
        half = head_dim // 2
        return bool(self.FUSE_EPI and self.FWD32 and head_dim < 128 and head_dim % 16 == 0 and half <= 64 and (n_heads * half) % 128 == 0)

    def gemm_qkv_rope_vit(self, x, w_qkv, bias, cos, sin, n_heads, half, q128, k128, v128):
        """Vision q|k|v projection + bias + rotary into 128-wide padded heads (pad columns of q128 / k128 / v128 must already be zero)."""
        self._chk(x, w_qkv, bias, q128, k128, v128)
        M, K = x.shape
        assert cos.dtype == F32 and sin.dtype == F32 and cos.is_contiguous() and sin.is_contiguous() and cos.shape == (M, half)
        assert w_qkv.shape[0] == 6 * n_heads * half and q128.shape == (M, n_heads * 128) == k128.shape == v128.shape
        self.L.call("tr1_gemm_qkv_rope_vit_bf16", _p(x), _p(w_qkv), _p(bias), _p(cos), _p(sin), _p(q128), _ld(q128), _p(k128), _ld(k128), _p(v128), _ld(v128),
                    M, n_heads, half, K, _ld(x), _ld(w_qkv), self._s())

    DGU_T = True        # dgu^T from the fused dgrad's epilogue instead of a transpose pass (class attribute for A/B runs)

    def dgrad_glu_bwd(self, dh, w_down, gu, want_t=False):
        """dgu[M, 2I] = swiglu_bwd(dh @ w_down, gu) with w_down [H, I] as stored (K-major operand).  want_t: -> (dgu, dgu^T or None); the fused kernel can
        write dgu^T [2I, Mp] (Mp = M rounded up to 64, padding zero: what transpose(dgu) returns) from its epilogue staging."""
        self._chk(dh, w_down, gu)
        M, H = dh.shape
        I = w_down.shape[1]
        if (self.FUSE_EPI and M >= 512 and I >= 256 and H % 64 == 0 and I % 8 == 0 and dh.stride(1) == 1 and w_down.stride(1) == 1 and gu.stride(1) == 1
                and ((M + 255) // 256) * ((I + 255) // 256) >= 192):
            dgu = self.empty(M, 2 * I)
            dgt = self.empty(2 * I, (M + 63) // 64 * 64) if (want_t and self.DGU_T) else None
            self.L.call("tr1_gemm_nn_glubwd_bf16", _p(dh), _p(w_down), _p(gu), _p(dgu), M, I, H, _ld(dh), _ld(w_down), _ld(gu), _ld(dgu),
                        _p(dgt), _ld(dgt) if dgt is not None else 0, self._s())
            return (dgu, dgt) if want_t else dgu
        dgu = self.swiglu_bwd(self.gemm_nn(dh, w_down), gu)
        return (dgu, None) if want_t else dgu

    def norm_gemm(self, x, lnw, eps, w, bias=None, glu=False):
        """Decode rows: rmsnorm(x; lnw) @ w^T (+bias), or with glu=True silu(gate)*up of the [2I, K] weight - one launch.  glu=2 (M <= 16): the same
        values FRAGMENT-MAJOR ([I / 32][16 rows][32 columns], the operand layout of gemm_oproj_frag) as a flat [16 * I] tensor."""
        self._chk(x, lnw, w, bias)
        M, K = x.shape
        N = w.shape[0] // 2 if glu else w.shape[0]
        assert x.stride(1) == 1 and w.stride(1) == 1 and w.shape[1] == K and lnw.numel() == K
        if int(glu) == 2:
            assert self.L.raw("tr1_norm_gemm_glu_frag_ok")(M, N, K)
            out = self.zeros(16 * N)
            self.L.call("tr1_norm_gemm_skinny", _p(x), _p(lnw), _p(w), _p(bias), _p(out), M, N, K, x.stride(0), w.stride(0), N, float(eps), 2, self._s())
            return out
        out = self.empty(M, N)
        self.L.call("tr1_norm_gemm_skinny", _p(x), _p(lnw), _p(w), _p(bias), _p(out), M, N, K, x.stride(0), w.stride(0), N, float(eps), int(glu), self._s())
        return out

    def norm_gemm_qkv(self, x, lnw, eps, wqkv, bias, cos, sin, kcache, vtcache, slots, n_heads, n_kv, head_dim):
        """Decode rows: rmsnorm -> q/k/v projection -> M-RoPE -> KV-cache append in one launch. Returns roped q [R, n_heads*hd]."""
        self._chk(x, lnw, wqkv, bias, kcache, vtcache)
        R, K = x.shape
        assert slots.dtype == I32 and cos.is_contiguous() and sin.is_contiguous() and cos.shape == (R, head_dim // 2) and cos.dtype == F32
        assert wqkv.shape == ((n_heads + 2 * n_kv) * head_dim, K) and x.stride(1) == 1 and wqkv.stride(1) == 1
        q = self.empty(R, n_heads * head_dim)
        self.L.call("tr1_norm_gemm_qkv", _p(x), _p(lnw), _p(wqkv), _p(bias), _p(cos), _p(sin), _p(q), _ld(q), _p(kcache), _ld(kcache), _p(vtcache),
                    _ld(vtcache), _p(slots), R, n_heads, n_kv, head_dim, K, x.stride(0), wqkv.stride(0), float(eps), self._s())
        return q

    # ---- fp8 weight storage (rollout only) -------------------------------------------------------------------------------
    def quantize_fp8_rows(self, w, q=None, scale=None):
        """bf16 [N, K] -> (fp8 e4m3 bytes uint8 [N, K], fp32 row scales [N]); q/scale may be preallocated views."""
        self._chk(w)
        N, K = w.shape
        assert w.stride(1) == 1
        if q is None:
            q = torch.empty(N, K, dtype=torch.uint8, device=self.device)
        if scale is None:
            scale = self.empty(N, dtype=F32)
        assert q.dtype == torch.uint8 and q.shape == (N, K) and q.stride(1) == 1 and scale.dtype == F32 and scale.numel() == N
        self.L.call("tr1_quantize_fp8_rows", _p(w), w.stride(0), _p(q), q.stride(0), _p(scale), N, K, self._s())
        return q, scale

    def gemm_w8(self, x, q, scale, lnw=None, eps=1e-6, bias=None, residual=None, glu=False, a8=False):
        """Decode rows x fp8 weights: act(x) @ dequant(q)^T * scale (+bias)(+residual); lnw folds rmsnorm in; glu: silu(gate)*up.
        a8=False: W8A16 (codes converted to bf16 in registers, bf16 MFMA); a8=True: W8A8 - fp8 MFMA, activations block-quantised to e4m3."""
        self._chk(x, lnw, bias, residual)
        M, K = x.shape
        N = q.shape[0] // 2 if glu else q.shape[0]
        assert q.dtype == torch.uint8 and q.shape[1] == K and x.stride(1) == 1 and q.stride(1) == 1 and scale.dtype == F32
        out = self.empty(M, N)
        if a8 and lnw is None and not glu and M <= 16 and K >= 8192 and K % 512 == 0 and N % 64 == 0:     # same choice as csrc/decode.hip (bitwise-equal paths)
            n = int(self.L.raw("tr1_gemm_skinny_fixup_workspace_floats")(M, N, K))
            key = "skinny_fix8_%d_%d_%d" % (M, N, K)
            ws = self._ws.get(key)
            if ws is None:
                ws = self._ws[key] = torch.zeros(n, dtype=F32, device=self.device)
            self.L.call("tr1_gemm_skinny_fixup_w8a8", _p(x), _p(q), _p(scale), _p(out), _p(bias), _p(residual), M, N, K, x.stride(0), q.stride(0), N,
                        residual.stride(0) if residual is not None else 0, _p(ws), n, self._s())
            return out
        self.L.call("tr1_gemm_skinny_w8a8" if a8 else "tr1_gemm_skinny_w8", _p(x), _p(lnw), _p(q), _p(scale), _p(bias), _p(residual), _p(out), M, N, K, x.stride(0), q.stride(0), N,
                    residual.stride(0) if residual is not None else 0, float(eps), int(glu), self._s())
        return out

    # ---- native decode-step driver ------------------------------------------------------------------------------------
    def decode_plan(self, layers, hidden, n_heads, n_kv, head_dim, inter, vocab, rows, n_batch, s_cap, nsplit, a8=False, qmask=31):
        """layers: per decoder layer the 9 tensors (ln1, qkv.w, qkv.b, o.w, ln2, gu.w, down.w, K cache, V^T cache).  Builds the host
        pointer table + device scratch once per rollout; decode_step then costs one C call per generated token."""
        import ctypes
        flat = [t for L in layers for t in L]
        per = len(flat) // max(len(layers), 1)
        assert per in (9, 13) and len(flat) == per * len(layers)      # 13 = fp8 matrices + their row scales (decode_step_w8)
        ptrs = (ctypes.c_void_p * len(flat))(*[t.data_ptr() for t in flat])
        dims = (ctypes.c_int64 * 12)(len(layers), hidden, n_heads, n_kv, head_dim, inter, vocab, rows, n_batch, s_cap, nsplit, int(qmask))   # qmask: fp8 steps only
        nbytes = int(self.L.raw("tr1_decode_step_workspace_bytes")(dims))
        work = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)     # zero: the split-K ticket counters inside start disarmed
        logits = self.empty(rows, vocab)
        return dict(ptrs=ptrs, ptrs_p=ctypes.cast(ptrs, ctypes.c_void_p), dims=dims, work=work, work_p=work.data_ptr(), nbytes=nbytes,
                    logits=logits, logits_p=logits.data_ptr(), keep=flat, stream=self._s(), w8=per == 13, a8=bool(a8))

    def decode_step(self, plan, embed_p, norm_p, lm_head_p, ids_p, cos_p, sin_p, slots_p, pre_p, lo_p, hi_p, eps, scale):
        """All arguments after `plan` are raw device addresses (ints): the caller precomputes row pointers into its step tables.
        Returns plan["logits"] [rows, vocab] (overwritten every step)."""
        if plan["w8"]:
            lm_q, lm_s = lm_head_p                  # (fp8 codes address, row scales address)
            self.L.call("tr1_decode_step_w8a8" if plan.get("a8") else "tr1_decode_step_w8", plan["ptrs_p"], plan["dims"], embed_p, norm_p, lm_q, lm_s, ids_p, cos_p, sin_p, slots_p, pre_p, lo_p,
                        hi_p, plan["work_p"], plan["nbytes"], plan["logits_p"], float(eps), float(scale), plan["stream"])
        else:
            self.L.call("tr1_decode_step", plan["ptrs_p"], plan["dims"], embed_p, norm_p, lm_head_p, ids_p, cos_p, sin_p, slots_p, pre_p, lo_p, hi_p,
                        plan["work_p"], plan["nbytes"], plan["logits_p"], float(eps), float(scale), plan["stream"])
        return plan["logits"]

    def attn_fwd_frag(self, q, k, vt, pre, lo, hi, n_heads, n_kv, n_slots, head_dim, scale, nsplit, n_batch=1, kv_batch_slots=0, plan=None, plan_mode=0):
        """Split-KV decode attention whose merged rows leave FRAGMENT-MAJOR (tr1_attn_fwd_planned_frag): -> bf16 [ceil(rows / 16) * 16 * n_heads * hd] with element
        (row m, feature k) at ((m // 16) * (n_heads * 4) + k // 32) * 512 + (m % 16) * 32 + k % 32 - the operand of gemm_oproj_frag."""
        self._chk(q, k, vt)
        rows = q.shape[0]
        T = rows // n_batch
        assert nsplit > 1 and head_dim == 128 and rows % n_batch == 0
        nws = n_batch * self.L.raw("tr1_attn_fwd_workspace_floats")(T, n_heads, n_kv, head_dim, nsplit)
        ws = self._workspace("attn_split", nws, F32)
        of = self.zeros((rows + 15) // 16 * 16 * n_heads * head_dim)
        self.L.call("tr1_attn_fwd_planned_frag", _p(q), _ld(q), _p(k), _ld(k), _p(vt), _ld(vt), _p(of), _p(pre), _p(lo), _p(hi), T, n_heads, n_kv, n_slots,
                    head_dim, float(scale), nsplit, _p(ws), nws, n_batch, kv_batch_slots, _p(plan), int(plan_mode), self._s())
        return of

    def gemm_oproj_frag(self, xfrag, w, M, residual=None):
        """c[M, N] = x @ w[N, K]^T (+ residual) for M <= 32 decode rows, x fragment-major (attn_fwd_frag): csrc/oproj.hip."""
        self._chk(xfrag, w, residual)
        N, K = w.shape
        assert self.L.raw("tr1_gemm_oproj_frag_ok")(M, N, K) and xfrag.numel() >= (M + 15) // 16 * 16 * K and w.stride(1) == 1
        c = self.empty(M, N)
        self.L.call("tr1_gemm_oproj_frag", _p(xfrag), _p(w), _p(residual), _p(c), M, N, K, _ld(w), residual.stride(0) if residual is not None else 0, _ld(c), self._s())
        return c

    def gemm_skinny_fixup(self, a, b, bias=None, residual=None):
        """Decode rows x narrow projection (o_proj / down_proj): split-K with in-kernel fixup; bf16 [M, N]."""
        self._chk(a, b, bias, residual)
        M, K = a.shape
        N = b.shape[0]
        assert a.stride(1) == 1 and b.stride(1) == 1 and b.shape[1] == K
        n = int(self.L.raw("tr1_gemm_skinny_fixup_workspace_floats")(M, N, K))
        key = "skinny_fix_%d_%d_%d" % (M, N, K)
        ws = self._ws.get(key)
        if ws is None:
            ws = self._ws[key] = torch.zeros(n, dtype=F32, device=self.device)     # ticket counters start at zero
        out = self.empty(M, N)
        self.L.call("tr1_gemm_skinny_fixup", _p(a), _p(b), _p(out), _p(bias), _p(residual), M, N, K, a.stride(0), b.stride(0), N,
                    residual.stride(0) if residual is not None else 0, _p(ws), n, self._s())
        return out

    def transpose(self, x, pad_to=64, out=None, colsum=None):
        """x[R,C] -> [C, Rpad] with zero-filled padding columns (Rpad = R rounded up to pad_to).  colsum (fp32 [C]): += the column sums of x
        (the bias gradient of a Linear, taken from the pass that builds dY^T for its weight gradient)."""
        self._chk(x)
        R, C = x.shape
        Rp = (R + pad_to - 1) // pad_to * pad_to
        if out is None:
            out = self.empty(C, Rp)
        assert out.shape[0] == C and out.shape[1] >= R
        if colsum is not None:
            assert colsum.dtype == F32 and colsum.numel() == C and colsum.is_contiguous()
            self.L.call("tr1_transpose_colsum_bf16", _p(x), _ld(x), _p(out), _ld(out), R, C, _p(colsum), self._s())
            return out
        self.L.call("tr1_transpose_bf16", _p(x), _ld(x), _p(out), _ld(out), R, C, self._s())
        return out

    # ---- norms ----------------------------------------------------------------------------------------------------
    def rmsnorm_fwd(self, x, w, eps, residual=None, need_rstd=True, out=None, rstd_out=None):
        self._chk(x, w, residual)
        rows, cols = x.shape
        assert x.is_contiguous() and (residual is None or residual.is_contiguous())
        y = out if out is not None else self.empty(rows, cols)
        assert y.is_contiguous() and y.shape == (rows, cols)
        rstd = (rstd_out if rstd_out is not None else self.empty(rows, dtype=F32)) if need_rstd else None
        xsum = self.empty(rows, cols) if residual is not None else None
        self.L.call("tr1_rmsnorm_fwd", _p(x), _p(residual), _p(w), _p(y), _p(xsum), _p(rstd), rows, cols, float(eps), self._s())
        return y, rstd, xsum

    def rmsnorm_bwd(self, dy, x, w, rstd, dres=None, dw=None):
        self._chk(dy, x, w, dres)
        rows, cols = x.shape
        assert dy.is_contiguous() and x.is_contiguous() and (dres is None or dres.is_contiguous())
        dx = self.empty(rows, cols)
        if dw is not None:
            assert dw.dtype == F32 and dw.numel() == cols
        ws, nws = None, 0
        if dw is not None:
            nws = int(self.L.raw("tr1_rmsnorm_bwd_workspace_floats")(rows, cols))
            ws = self._workspace("rmsnorm_bwd_dw", nws, F32)
        self.L.call("tr1_rmsnorm_bwd", _p(dy), _p(x), _p(w), _p(rstd), _p(dres), _p(dx), _p(dw), _p(ws), nws, rows, cols, self._s())
        return dx

    def layernorm_fwd(self, x, w, b, eps, need_stats=True):
        self._chk(x, w, b)
        rows, cols = x.shape
        assert x.is_contiguous()
        y = self.empty(rows, cols)
        mean = self.empty(rows, dtype=F32) if need_stats else None
        rstd = self.empty(rows, dtype=F32) if need_stats else None
        self.L.call("tr1_layernorm_fwd", _p(x), _p(w), _p(b), _p(y), _p(mean), _p(rstd), rows, cols, float(eps), self._s())
        return y, mean, rstd

    def layernorm_bwd(self, dy, x, w, mean, rstd, dw, db, need_dx=False):
        self._chk(dy, x, w)
        rows, cols = x.shape
        assert dy.is_contiguous() and x.is_contiguous() and dw.dtype == F32 and db.dtype == F32
        dx = self.empty(rows, cols) if need_dx else None
        self.L.call("tr1_layernorm_bwd", _p(dy), _p(x), _p(w), _p(mean), _p(rstd), _p(dx), _p(dw), _p(db), rows, cols, self._s())
        return dx

    # ---- activations ----------------------------------------------------------------------------------------------
    def swiglu_fwd(self, gu, out=None):
        self._chk(gu)
        rows, two_i = gu.shape
        assert gu.is_contiguous()
        if out is None:
            out = self.empty(rows, two_i // 2)
        assert out.is_contiguous() and out.shape == (rows, two_i // 2)
        self.L.call("tr1_swiglu_fwd", _p(gu), _p(out), rows, two_i // 2, self._s())
        return out

    def swiglu_bwd(self, dout, gu):
        self._chk(dout, gu)
        rows, two_i = gu.shape
        assert gu.is_contiguous() and dout.is_contiguous()
        dgu = self.empty(rows, two_i)
        self.L.call("tr1_swiglu_bwd", _p(dout), _p(gu), _p(dgu), rows, two_i // 2, self._s())
        return dgu

    def gelu_fwd(self, x):
        self._chk(x)
        assert x.is_contiguous()
        y = torch.empty_like(x)
        self.L.call("tr1_gelu_fwd", _p(x), _p(y), x.numel(), self._s())
        return y

    def gelu_bwd(self, x, dy):
        self._chk(x, dy)
        assert x.is_contiguous() and dy.is_contiguous()
        dx = torch.empty_like(x)
        self.L.call("tr1_gelu_bwd", _p(x), _p(dy), _p(dx), x.numel(), self._s())
        return dx

    def quickgelu_fwd(self, x):
        self._chk(x)
        assert x.is_contiguous()
        y = torch.empty_like(x)
        self.L.call("tr1_quickgelu_fwd", _p(x), _p(y), x.numel(), self._s())
        return y

    def add(self, a, b):
        self._chk(a, b)
        assert a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
        y = torch.empty_like(a)
        self.L.call("tr1_add_bf16", _p(a), _p(b), _p(y), a.numel(), self._s())
        return y

    def colsum_accum(self, dy, dbias):
        self._chk(dy)
        assert dy.is_contiguous() and dbias.dtype == F32 and dbias.numel() == dy.shape[1]
        self.L.call("tr1_colsum_accum", _p(dy), _p(dbias), dy.shape[0], dy.shape[1], self._s())

    def cast_to_act(self, x_f32):
        assert x_f32.dtype == F32 and x_f32.is_contiguous()
        y = torch.empty(x_f32.shape, dtype=BF16, device=self.device)
        self.L.call("tr1_cast_f32_to_bf16", _p(x_f32), _p(y), x_f32.numel(), self._s())
        return y

    def cast_to_f32(self, x):
        self._chk(x)
        assert x.is_contiguous()
        y = torch.empty(x.shape, dtype=F32, device=self.device)
        self.L.call("tr1_cast_bf16_to_f32", _p(x), _p(y), x.numel(), self._s())
        return y

    # ---- rotary ---------------------------------------------------------------------------------------------------
    def mrope_table(self, pos3, head_dim, sections, theta):
        assert pos3.dtype == I32 and pos3.is_contiguous() and pos3.shape[0] == 3
        T = pos3.shape[1]
        cos = self.empty(T, head_dim // 2, dtype=F32)
        sin = self.empty(T, head_dim // 2, dtype=F32)
        self.L.call("tr1_mrope_table", _p(pos3), _p(cos), _p(sin), T, head_dim, sections[0], sections[1], sections[2], float(theta), 1,
                    self._s())
        return cos, sin

    def vision_rope_table(self, hw, head_dim, theta=10000.0):
        assert hw.dtype == I32 and hw.is_contiguous() and hw.shape[1] == 2
        N = hw.shape[0]
        cos = self.empty(N, head_dim // 2, dtype=F32)
        sin = self.empty(N, head_dim // 2, dtype=F32)
        self.L.call("tr1_vision_rope_table", _p(hw), _p(cos), _p(sin), N, head_dim, float(theta), self._s())
        return cos, sin

    def rope_apply(self, x, n_heads, head_dim, cos, sin, backward=False, out=None):
        """x: [T, >= n_heads*head_dim] row-major view; rotates the first n_heads heads of every row."""
        self._chk(x)
        T = x.shape[0]
        assert cos.dtype == F32 and sin.dtype == F32 and cos.is_contiguous() and sin.is_contiguous() and cos.shape == (T, head_dim // 2)
        if out is None:
            out = self.empty(T, n_heads * head_dim)
        self.L.call("tr1_rope_apply", _p(x), _ld(x), _p(out), _ld(out), _p(cos), _p(sin), T, n_heads, head_dim, int(backward), self._s())
        return out

    # ---- gathers --------------------------------------------------------------------------------------------------
    def gather_rows(self, table, ids):
        self._chk(table)
        assert ids.dtype == I32 and table.is_contiguous()
        out = self.empty(ids.numel(), table.shape[1])
        self.L.call("tr1_gather_rows", _p(table), _p(ids), _p(out), ids.numel(), table.shape[1], self._s())
        return out

    def scatter_rows(self, src, idx, dst):
        self._chk(src, dst)
        assert idx.dtype == I32 and src.is_contiguous() and dst.is_contiguous() and src.shape[1] == dst.shape[1]
        self.L.call("tr1_scatter_rows", _p(src), _p(idx), _p(dst), idx.numel(), src.shape[1], self._s())

    def embed_bwd(self, dout, ids, dtable):
        self._chk(dout)
        assert ids.dtype == I32 and dout.is_contiguous() and dtable.dtype == F32
        self.L.call("tr1_embed_bwd", _p(dout), _p(ids), _p(dtable), ids.numel(), dout.shape[1], self._s())

    # ---- attention ------------------------------------------------------------------------------------------------
    def pack_transpose(self, x, n_heads, n_kv, head_dim, ld_out=None, slots=None, out=None, zero_pad=True):
        self._chk(x)
        T = x.shape[0]
        group = n_heads // n_kv
        if out is None:
            if ld_out is None:
                ld_out = (T * group + 63) // 64 * 64
            out = self.empty(n_kv * head_dim, ld_out)
        self.L.call("tr1_pack_transpose", _p(x), _ld(x), _p(out), _ld(out), _p(slots), T, n_heads, n_kv, head_dim,
                    out.shape[1] if (zero_pad and slots is None) else 0, self._s())
        return out

    def scatter_slots(self, src, dst, slots):
        self._chk(src, dst)
        assert slots.dtype == I32
        self.L.call("tr1_scatter_slots", _p(src), _ld(src), _p(dst), _ld(dst), _p(slots), src.shape[0], src.shape[1], self._s())

    def decode_qkv_post(self, qkv, cos, sin, kcache, vtcache, slots, n_heads, n_kv, head_dim):
        """RoPE(q), RoPE(k) -> K cache rows, v -> V^T cache columns, for the R new decode tokens. Returns roped q [R, n_heads*hd]."""
        self._chk(qkv, kcache, vtcache)
        R = qkv.shape[0]
        assert slots.dtype == I32 and cos.is_contiguous() and sin.is_contiguous() and cos.shape == (R, head_dim // 2)
        q = self.empty(R, n_heads * head_dim)
        self.L.call("tr1_decode_qkv_post", _p(qkv), _ld(qkv), _p(cos), _p(sin), _p(q), _ld(q), _p(kcache), _ld(kcache), _p(vtcache), _ld(vtcache),
                    _p(slots), R, n_heads, n_kv, head_dim, self._s())
        return q

    def attn_plan(self, T, n_heads, n_kv, n_batch=1):
        """int32 buffer for attn_fwd(plan=..., plan_mode=1|2): the relevant-tile lists of one decode step, shared by its layers."""
        return self.zeros(self.L.raw("tr1_attn_plan_ints")(T, n_heads, n_kv, n_batch), dtype=I32)

    FWD32 = True      # head dim 128, nsplit 1: the 32x32x16-MFMA forward over row-major K / V (class attribute: tests / tools may clear it)

    def attn_fwd(self, q, k, vt, pre, lo, hi, n_heads, n_kv, n_slots, head_dim, scale, nsplit=1, need_lse=True, out=None, n_batch=1,
                 kv_batch_slots=0, plan=None, plan_mode=0, v_rows=None, live96=False):
        """n_batch > 1: q/out/masks hold n_batch problems of T = rows/n_batch tokens each; problem b reads cache slots from b*kv_batch_slots.
        plan / plan_mode: split-KV decode only - mode 1 publishes the tile lists of these masks in `plan`, mode 2 reuses them (same masks).
        v_rows: V row-major [n_slots, n_kv*head_dim] (a view is fine).  With it, head dim 128 single-pass launches take the 32x32x16-MFMA
        kernel, which reads K and V as stored; `vt` may then be None.  live96 (with v_rows, head dim 128): features 96..127 of every head are zero in q / k / v
        (the vision towers' padded heads) - the kernel skips them and leaves columns 96..127 of `out` untouched."""
        assert pre.dtype == I32 and lo.dtype == I32 and hi.dtype == I32
        rows = q.shape[0]
        assert rows % n_batch == 0
        T = rows // n_batch
        o = out if out is not None else self.empty(rows, n_heads * head_dim)
        lse = self.empty(n_batch * n_heads, T, dtype=F32) if need_lse else None
        if v_rows is not None and self.attn_fwd_rows_ok(head_dim, nsplit, n_batch, n_slots=n_slots, ld=max(_ld(k), _ld(v_rows))):
            self._chk(q, k, v_rows)
            assert k.shape[0] >= n_slots and v_rows.shape[0] >= n_slots
            self.L.call("tr1_attn_fwd_rows_live96" if live96 else "tr1_attn_fwd_rows", _p(q), _ld(q), _p(k), _ld(k), _p(v_rows), _ld(v_rows), _p(o), _ld(o), _p(lse),
                        _p(pre), _p(lo), _p(hi), T, n_heads, n_kv, n_slots, head_dim, float(scale), self._s())
            return o, lse
        assert not live96, "live96 is a property of the row-major head-dim-128 launch"
        if vt is None and v_rows is not None:      # the row-major kernel does not take this launch (head dim, split-KV, > 4 GiB operands): V^T copy
            vt = self.pack_transpose(v_rows, n_kv, n_kv, head_dim)
        assert vt is not None, "attention: this shape needs the V^T operand"
        self._chk(q, k, vt)
        ws, nws = None, 0
        if nsplit > 1:
            nws = n_batch * self.L.raw("tr1_attn_fwd_workspace_floats")(T, n_heads, n_kv, head_dim, nsplit)
            ws = self._workspace("attn_split", nws, F32)
        if plan_mode:
            assert plan is not None and plan.dtype == I32 and plan.numel() >= self.L.raw("tr1_attn_plan_ints")(T, n_heads, n_kv, n_batch)
            self.L.call("tr1_attn_fwd_planned", _p(q), _ld(q), _p(k), _ld(k), _p(vt), _ld(vt), _p(o), _ld(o), _p(lse), _p(pre), _p(lo), _p(hi), T,
                        n_heads, n_kv, n_slots, head_dim, float(scale), nsplit, _p(ws), nws, n_batch, kv_batch_slots, _p(plan), int(plan_mode), self._s())
            return o, lse
        self.L.call("tr1_attn_fwd", _p(q), _ld(q), _p(k), _ld(k), _p(vt), _ld(vt), _p(o), _ld(o), _p(lse), _p(pre), _p(lo), _p(hi), T,
                    n_heads, n_kv, n_slots, head_dim, float(scale), nsplit, _p(ws), nws, n_batch, kv_batch_slots, self._s())
        return o, lse

    def attn_fwd_rows_ok(self, head_dim, nsplit=1, n_batch=1, n_slots=0, ld=0):
        """True when attn_fwd(v_rows=...) would take the row-major K / V kernel (callers can then skip building V^T).  The kernel addresses
        K / V with 32-bit DMA offsets: operands of n_slots rows x ld elements must stay below 4 GiB (csrc/attn_fwd32.hip, tr1_attn_fwd_rows)."""
        return self.FWD32 and head_dim == 128 and nsplit == 1 and n_batch == 1 and n_slots * ld * 2 < 0xffffffff

    def attn_bwd(self, q, k, v, o, do, lse, pre, lo, hi, n_heads, n_kv, n_slots, head_dim, scale, dv_out=None, dq_out=None, dk_out=None, rope=None):
        """-> dq [T, n_heads*hd], dk, dv [n_slots, n_kv*hd]. Builds the transposed operand copies it needs.
        rope=(cos, sin) (fp32 [T, hd/2]): dq / dk come back already multiplied by the transposed rotary matrix (the M-RoPE backward folded into the
        kernels' epilogues), i.e. they are the gradients of the UN-rotated projections; dq_out / dk_out / dv_out may be column views of one buffer."""
        self._chk(q, k, v, o, do)
        T = q.shape[0]
        group = n_heads // n_kv
        qt = dot = None                                     # the dQ kernel reads its K^T fragments from the K rows (KT argument unused)
        if (head_dim + 31) // 32 * 32 not in (64, 128):     # the 8-wave dK/dV kernel reads Q^T / dO^T straight from the row-major tiles
            qt = self.pack_transpose(q, n_heads, n_kv, head_dim)
            dot = self.pack_transpose(do, n_heads, n_kv, head_dim)
        dq = dq_out if dq_out is not None else self.empty(T, n_heads * head_dim)
        dk = dk_out if dk_out is not None else self.empty(n_slots, n_kv * head_dim)
        dv = dv_out if dv_out is not None else self.empty(n_slots, n_kv * head_dim)
        delta = self.empty(2 * n_heads, T, dtype=F32)       # [delta | log2-scaled LSE], both written by the backward's first kernel
        qmeta = self._workspace("attn_qmeta", 8 * ((T * group + 63) // 64), I32)
        nws = self.L.raw("tr1_attn_bwd_workspace_floats")(T, n_heads, n_kv, n_slots, head_dim)
        ws = self._workspace("attn_bwd_part", nws, F32) if nws else None
        args = (_p(q), _ld(q), _p(k), _ld(k), _p(v), _ld(v), None, 0, _p(qt), _ld(qt) if qt is not None else 0, _p(dot), _ld(dot) if dot is not None else 0,
                _p(o), _ld(o), _p(do), _ld(do), _p(lse), _p(delta), _p(dq), _ld(dq), _p(dk), _ld(dk), _p(dv), _ld(dv), _p(pre), _p(lo),
                _p(hi), _p(qmeta), _p(ws), nws, T, n_heads, n_kv, n_slots, head_dim, float(scale))
        if rope is not None and self.FUSE_EPI:
            cos, sin = rope
            assert cos.dtype == F32 and sin.dtype == F32 and cos.is_contiguous() and sin.is_contiguous() and cos.shape == (T, head_dim // 2) and n_slots == T
            self.L.call("tr1_attn_bwd_rope", *args, _p(cos), _p(sin), self._s())
            return dq, dk, dv
        self.L.call("tr1_attn_bwd", *args, self._s())
        if rope is not None:
            self.rope_apply(dq, n_heads, head_dim, rope[0], rope[1], backward=True, out=dq)
            self.rope_apply(dk, n_kv, head_dim, rope[0], rope[1], backward=True, out=dk)
        return dq, dk, dv

    # ---- video preprocessing --------------------------------------------------------------------------------------
    def video_preprocess(self, frames_u8, out_hw, k_pad, patch=14, temporal=2, merge=2, mean=(0.48145466, 0.4578275, 0.40821073),
                         std=(0.26862954, 0.26130258, 0.27577711)):
        """uint8 [T,3,H,W] (device) -> (bf16 [N_v, k_pad] normalised patches ready for the patch-embed GEMM, (t, h, w) grid)."""
        from .vision_process import aa_filter
        assert frames_u8.dtype == torch.uint8 and frames_u8.device.type == "cuda" and frames_u8.is_contiguous()
        T, C, H, W = frames_u8.shape
        Ho, Wo = out_hw
        T_out = (T + temporal - 1) // temporal * temporal
        key = ("aa", H, W, Ho, Wo)
        tabs = self._ws.get(key)
        if tabs is None:
            ymin, wy = aa_filter(H, Ho)
            xmin, wx = aa_filter(W, Wo)
            tabs = tuple(torch.as_tensor(a).to(self.device).contiguous() for a in (ymin, wy, xmin, wx))
            self._ws[key] = tabs
        ymin, wy, xmin, wx = tabs
        gt, gh, gw = T_out // temporal, Ho // patch, Wo // patch
        out = self.zeros(gt * gh * gw, k_pad)
        self.L.call("tr1_video_preprocess", _p(frames_u8), _p(out), _ld(out), _p(ymin), _p(wy), wy.shape[1], _p(xmin), _p(wx), wx.shape[1], T, T_out,
                    H, W, Ho, Wo, *[float(x) for x in mean], *[float(x) for x in std], patch, temporal, merge, self._s())
        return out, (gt, gh, gw)

    # ---- vocabulary side ------------------------------------------------------------------------------------------
    def logp_entropy_fwd(self, logits, targets):
        self._chk(logits)
        assert targets.dtype == I32
        R, V = logits.shape
        logp = self.empty(R, dtype=F32)
        ent = self.empty(R, dtype=F32)
        lse = self.empty(R, dtype=F32)
        self.L.call("tr1_logp_entropy_fwd", _p(logits), _ld(logits), _p(targets), _p(logp), _p(ent), _p(lse), R, V, self._s())
        return logp, ent, lse

    def lmhead_lse(self, hn, w, targets):
        """(logp[target], entropy, lse) of softmax(hn @ w^T) per row WITHOUT the [R, V] logits: the GEMM epilogue reduces each 64-column slice
        to (max, sum e, sum x e), a second kernel merges the slices of a row.  Returns None when the shape is outside the fused kernel's
        range (the caller then materialises the logits)."""
        self._chk(hn, w)
        R, K = hn.shape
        V = w.shape[0]
        if R == 0 or V % 64 or K % 64 or hn.stride(1) != 1 or w.stride(1) != 1 or hn.stride(0) % 8 or w.stride(0) % 8 or V < 256:
            return None
        assert targets.dtype == I32 and targets.numel() == R
        n = int(self.L.raw("tr1_lmhead_lse_workspace_floats")(R, V))
        ws = self._workspace("lmhead_lse", n, F32)
        logp, ent, lse = self.empty(R, dtype=F32), self.empty(R, dtype=F32), self.empty(R, dtype=F32)
        self.L.call("tr1_lmhead_lse_fwd", _p(hn), _p(w), _p(targets), _p(ws), n, _p(logp), _p(ent), _p(lse), R, V, K, hn.stride(0), w.stride(0), self._s())
        return logp, ent, lse

    def logp_bwd(self, logits, targets, lse, dlogp, inplace=True):
        self._chk(logits)
        R, V = logits.shape
        out = logits if inplace else torch.empty_like(logits)
        self.L.call("tr1_logp_bwd", _p(logits), _ld(logits), _p(targets), _p(lse), _p(dlogp), _p(out), _ld(out), R, V, self._s())
        return out

    def grpo_loss(self, logp, ref_logp, mask, adv, beta, use_grpo, grad_scale=1.0):
        G, C = logp.shape
        assert logp.dtype == F32 and mask.dtype == I32 and adv.dtype == F32 and logp.is_contiguous() and mask.is_contiguous()
        dlogp = self.empty(G, C, dtype=F32)
        out3 = self.empty(3, dtype=F32)
        row_len = self.empty(G, dtype=F32)
        row_kl = self.empty(G, dtype=F32)
        self.L.call("tr1_grpo_loss", _p(logp), _p(ref_logp), _p(mask), _p(adv), _p(dlogp), _p(out3), _p(row_len), _p(row_kl), G, C,
                    float(beta), int(bool(use_grpo)), float(grad_scale), self._s())
        return dlogp, out3, row_len, row_kl

    def sample_tokens(self, logits, temperature, top_k, seed, step_dev, tokens, finished, eos_id, pad_id, stop_at_eos, u_out=None, group_rows=0,
                      seed_stride=0, next_ids=None):
        """next_ids (int32 [rows], optional): also receives the drawn tokens (the next decode step's embedding gather reads it: no copy kernel)."""
        self._chk(logits)
        assert tokens.dtype == I32 and (step_dev is None or step_dev.dtype == I32)
        rows, V = logits.shape
        nws = self.L.raw("tr1_sample_workspace_words")(rows)
        if next_ids is not None:
            assert next_ids.dtype == I32 and next_ids.numel() == rows and next_ids.is_contiguous()
            key = ("sampler_step", rows)
            ws = self._ws.get(key)
            if ws is None:
                ws = self._ws[key] = torch.zeros(nws, dtype=I32, device=self.device)      # zero ONCE: the pick kernel leaves it zero after every call
            self.L.call("tr1_sample_tokens_step", _p(logits), _ld(logits), rows, V, float(temperature), int(top_k or 0), int(seed) & (2**64 - 1),
                        int(group_rows), int(seed_stride) & (2**64 - 1), _p(step_dev), _p(tokens), tokens.stride(0), _p(finished), int(eos_id), int(pad_id),
                        int(bool(stop_at_eos)), _p(u_out), _p(ws), nws, _p(next_ids), 1, self._s())
            return
        ws = self._workspace("sampler", nws, I32)
        self.L.call("tr1_sample_tokens", _p(logits), _ld(logits), rows, V, float(temperature), int(top_k or 0), int(seed) & (2**64 - 1),
                    int(group_rows), int(seed_stride) & (2**64 - 1), _p(step_dev), _p(tokens), tokens.stride(0), _p(finished), int(eos_id), int(pad_id), int(bool(stop_at_eos)), _p(u_out),
                    _p(ws), nws, self._s())

    # ---- optimizer ------------------------------------------------------------------------------------------------
    def sumsq_accum(self, g, out_scalar):
        assert g.dtype in (F32, BF16) and out_scalar.dtype == F32
        self.L.call("tr1_sumsq_accum" if g.dtype == F32 else "tr1_sumsq_accum_bf16", _p(g), g.numel(), _p(out_scalar), self._s())

    def wgrad_sumsq(self, a, b, gw, accumulate, partials, offset, b_kmajor=False, b_rows=0, wire=None):
        """gw[N, K] fp32 (+)= a[N, Mp] @ b^T (b = X^T [K, Mp]) or a @ b (b_kmajor: b = X [>= b_rows, K] as stored), and partials[offset : offset + n] receives
        the per-wave sums of squares of the values stored (n returned; -1 when the shape is not covered and nothing was launched)."""
        import ctypes
        N, Mp = a.shape
        K = b.shape[1] if b_kmajor else b.shape[0]
        if not (self.FUSE_EPI and N >= 512 and K >= 256 and K % 8 == 0 and Mp % 64 == 0 and a.stride(1) == 1 and b.stride(1) == 1 and _ld(a) % 8 == 0 and _ld(b) % 8 == 0):
            return -1
        if self.wgrad_sumsq_partials(N, K) > partials.numel() - int(offset) - 256:
            return -1                     # partials buffer too small for this matrix: the caller runs the plain GEMM and step() takes the full-arena norm
        self._chk(a, b)
        assert gw.dtype == F32 and gw.shape == (N, K) and partials.dtype == F32 and partials.is_contiguous()
        assert wire is None or (wire.dtype == torch.bfloat16 and wire.shape == (N, K) and wire.stride(1) == 1)      # the gradient exchange's bf16 copy of gw
        n = ctypes.c_int64(0)
        self.L.call("tr1_wgrad_f32_sumsq", _p(a), _p(b), _p(gw), N, K, Mp, _ld(a), _ld(b), _ld(gw), int(accumulate), int(b_kmajor), int(b_rows),
                    partials.data_ptr() + 4 * int(offset), partials.numel() - int(offset) - 256, ctypes.byref(n), _p(wire), _ld(wire) if wire is not None else 0, self._s())
        return int(n.value)

    @staticmethod
    def wgrad_sumsq_partials(N, K):
        """Upper bound of the partial sums one tr1_wgrad_f32_sumsq launch on an [N, K] gradient leaves: 8 per tile block, tiles >= 224 rows x 256 columns."""
        return 8 * ((int(N) + 223) // 224) * ((int(K) + 255) // 256)

    def sumsq_partials_accum(self, partials, n, out_scalar):
        assert partials.dtype == F32 and out_scalar.dtype == F32 and partials.numel() >= int(n) + 256
        self.L.call("tr1_sumsq_partials_accum", _p(partials), int(n), _p(out_scalar), self._s())

    def sumsq_ranges_periodic(self, g, base, stride, count, rel_ranges, out_scalar):
        import ctypes
        assert g.dtype == F32 and g.is_contiguous() and base + stride * count <= g.numel() and len(rel_ranges) <= 8
        flat = [int(x) for ab in rel_ranges for x in ab]
        arr = (ctypes.c_int64 * max(1, len(flat)))(*flat)
        self.L.call("tr1_sumsq_ranges_periodic", _p(g), int(base), int(stride), int(count), arr, len(rel_ranges), _p(out_scalar), self._s())

    def zero_ranges_periodic(self, g, base, stride, count, rel_ranges):
        """g[base + l*stride + r] = 0 for l < count and r in the half-open `rel_ranges` [(a, b), ...] (<= 8) of one period."""
        import ctypes
        assert g.dtype == F32 and g.is_contiguous() and base >= 0 and base + stride * count <= g.numel() and len(rel_ranges) <= 8
        flat = [int(x) for ab in rel_ranges for x in ab]
        arr = (ctypes.c_int64 * max(1, len(flat)))(*flat)
        self.L.call("tr1_zero_ranges_periodic", _p(g), int(base), int(stride), int(count), arr, len(rel_ranges), self._s())

    def adamw_step(self, p32, m, v, g, p16, lr, beta1, beta2, eps, weight_decay, step, sumsq=None, max_norm=0.0, grad_mult=1.0,
                   zero_grad=True, g16=None):
        """g16: read the gradient from this bf16 array instead of `g` (the all-reduced wire buffer); `g` is then only zeroed."""
        assert p32.dtype == F32 and m.dtype == F32 and v.dtype == F32 and g.dtype == F32 and p16.dtype == BF16
        n = p32.numel()
        assert m.numel() == n and v.numel() == n and g.numel() == n and p16.numel() == n
        if g16 is not None:
            assert g16.dtype == BF16 and g16.numel() == n
            self.L.call("tr1_adamw_step_g16", _p(p32), _p(m), _p(v), _p(g), _p(g16), _p(p16), n, float(lr), float(beta1), float(beta2), float(eps),
                        float(weight_decay), int(step), _p(sumsq), float(max_norm), float(grad_mult), int(zero_grad), self._s())
            return
        self.L.call("tr1_adamw_step", _p(p32), _p(m), _p(v), _p(g), _p(p16), n, float(lr), float(beta1), float(beta2), float(eps),
                    float(weight_decay), int(step), _p(sumsq), float(max_norm), float(grad_mult), int(zero_grad), self._s())
