"""GPU parity: every HIP kernel behind the C ABI vs the CPU oracle (oracle/ref_ops.py) on the same seeded inputs.

Inputs are drawn in fp32, rounded to bf16 once, and handed to both sides; the oracle computes in fp32. Tolerances are bf16-scale
(the HIP path stores bf16 activations, fp32 accumulate) and written next to each check.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

BF16 = torch.bfloat16


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF16)


def close(hip_t, ref_t, atol, rtol=2e-2, what=""):
    a = hip_t.detach().float().cpu()
    b = ref_t.detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert torch.isfinite(a).all(), what + ": non-finite values from the HIP path"
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bad.any(), "%s: %d/%d elements off, max err %.4g (ref %.4g) at %s" % (
        what, int(bad.sum()), bad.numel(), float(err.max()), float(b.flatten()[err.argmax()]), str(divmod(int(err.argmax()), b.shape[-1])))


def both(hip_ops, ref_ops, fn, *tensors, **kw):
    """Run ops.<fn> on GPU copies and on fp32 CPU copies."""
    dev = [t.to("cuda:0") if isinstance(t, torch.Tensor) else t for t in tensors]
    cpu = [t.float() if isinstance(t, torch.Tensor) and t.dtype == BF16 else t for t in tensors]
    return getattr(hip_ops, fn)(*dev, **kw), getattr(ref_ops, fn)(*cpu, **kw)


# ------------------------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(1, 128, 256), (8, 512, 3584), (16, 72, 320), (130, 200, 64), (257, 384, 192), (300, 1152, 512),
                                   (128, 128, 64), (513, 136, 1216),
                                   # decode regime with several 16-row groups (G = 16 / prompts batched over the accumulation window)
                                   (17, 72, 320), (24, 512, 3584), (32, 4096, 3584), (33, 200, 8192), (48, 16400, 8192), (64, 1152, 512), (32, 100096, 256)])
def test_gemm_nt(hip_ops, ref_ops, M, N, K):
    a, b, bias, res = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3), rnd(M, N, seed=4)
    atol = 0.02 * math.sqrt(K) * 0.1 + 0.02
    h, r = both(hip_ops, ref_ops, "gemm_nt", a, b)
    close(h, r, atol, what="gemm plain")
    h = hip_ops.gemm_nt(a.cuda(), b.cuda(), bias=bias.cuda(), residual=res.cuda())
    r = ref_ops.gemm_nt(a.float(), b.float(), bias=bias.float(), residual=res.float())
    close(h, r, atol, what="gemm bias+residual")
    # fp32 output + accumulate (weight-gradient mode)
    if M > 16:
        out = torch.full((M, N), 0.5, device="cuda:0")
        hip_ops.gemm_nt(a.cuda(), b.cuda(), out_f32=True, out=out, accumulate=True)
        close(out, a.float() @ b.float().t() + 0.5, 1e-3 * math.sqrt(K), rtol=1e-3, what="gemm f32 accumulate")
    out = hip_ops.gemm_nt(a.cuda(), b.cuda(), out_f32=True)
    close(out, a.float() @ b.float().t(), 1e-3 * math.sqrt(K), rtol=1e-3, what="gemm f32")


def test_gemm_strided_views(hip_ops, ref_ops):
    big = rnd(200, 448, seed=5)
    a = big[:, 64:192]  # lda = 448
    b = rnd(96, 128, seed=6, scale=0.1)
    h = hip_ops.gemm_nt(a.cuda(), b.cuda())
    close(h, a.float() @ b.float().t(), 0.05, what="gemm strided A")


@pytest.mark.parametrize("R,C", [(5, 64), (100, 136), (257, 64), (64, 4608)])
def test_transpose(hip_ops, ref_ops, R, C):
    x = rnd(R, C, seed=7)
    h, r = both(hip_ops, ref_ops, "transpose", x)
    assert torch.equal(h.float().cpu(), r.float()), "transpose must be bit exact (incl. zero padding)"


# ----------------------------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("rows,cols", [(1, 64), (37, 1536), (130, 3584), (9, 4096), (300, 1280)])
def test_rmsnorm(hip_ops, ref_ops, rows, cols):
    x, w, res, dy = rnd(rows, cols, seed=1, scale=2.0), rnd(cols, seed=2), rnd(rows, cols, seed=3), rnd(rows, cols, seed=4)
    (y, rstd, _), (yr, rstdr, _) = both(hip_ops, ref_ops, "rmsnorm_fwd", x, w, 1e-6)
    close(y, yr, 0.02, what="rmsnorm y")
    close(rstd, rstdr, 1e-5, rtol=1e-4, what="rmsnorm rstd")
    (y, rstd, xs), (yr, rstdr, xsr) = hip_ops.rmsnorm_fwd(x.cuda(), w.cuda(), 1e-6, residual=res.cuda()), \
        RefBF().rmsnorm_fwd(x, w, 1e-6, residual=res)
    assert torch.equal(xs.float().cpu(), xsr.float()), "fused residual sum must equal bf16(x+res) exactly"
    close(y, yr, 0.02, what="rmsnorm(res) y")
    # backward
    dw_h = torch.zeros(cols, device="cuda:0")
    dw_r = torch.zeros(cols)
    dx_h = hip_ops.rmsnorm_bwd(dy.cuda(), x.cuda(), w.cuda(), rstdr.cuda(), dres=res.cuda(), dw=dw_h)
    dx_r = ref_ops.rmsnorm_bwd(dy.float(), x.float(), w.float(), rstdr, dres=res.float(), dw=dw_r)
    close(dx_h, dx_r, 0.03, what="rmsnorm dx")
    close(dw_h, dw_r, 0.02 * math.sqrt(rows) + 1e-3, rtol=1e-3, what="rmsnorm dw")


def RefBF():
    from oracle.ref_ops import RefOps
    return RefOps(act_dtype=BF16)


@pytest.mark.parametrize("rows,cols", [(3, 64), (130, 1280)])
def test_layernorm(hip_ops, ref_ops, rows, cols):
    x, w, b, dy = rnd(rows, cols, seed=1, scale=2.0), rnd(cols, seed=2), rnd(cols, seed=3), rnd(rows, cols, seed=4)
    (y, mean, rstd), (yr, meanr, rstdr) = both(hip_ops, ref_ops, "layernorm_fwd", x, w, b, 1e-6)
    close(y, yr, 0.02, what="layernorm y")
    close(mean, meanr, 1e-5, rtol=1e-4, what="ln mean")
    close(rstd, rstdr, 1e-5, rtol=1e-4, what="ln rstd")
    dw_h, db_h = torch.zeros(cols, device="cuda:0"), torch.zeros(cols, device="cuda:0")
    dw_r, db_r = torch.zeros(cols), torch.zeros(cols)
    dx_h = hip_ops.layernorm_bwd(dy.cuda(), x.cuda(), w.cuda(), meanr.cuda(), rstdr.cuda(), dw_h, db_h, need_dx=True)
    dx_r = ref_ops.layernorm_bwd(dy.float(), x.float(), w.float(), meanr, rstdr, dw_r, db_r, need_dx=True)
    close(dx_h, dx_r, 0.03, what="ln dx")
    close(dw_h, dw_r, 0.02 * math.sqrt(rows) + 1e-3, rtol=1e-3, what="ln dw")
    close(db_h, db_r, 0.02 * math.sqrt(rows) + 1e-3, rtol=1e-3, what="ln db")


# ----------------------------------------------------------------------------------------------------------- elementwise
def test_activations(hip_ops, ref_ops):
    gu, dout = rnd(77, 2 * 136, seed=1, scale=2.0), rnd(77, 136, seed=2)
    h, r = both(hip_ops, ref_ops, "swiglu_fwd", gu)
    close(h, r, 0.02, what="swiglu fwd")
    h, r = both(hip_ops, ref_ops, "swiglu_bwd", dout, gu)
    close(h, r, 0.03, what="swiglu bwd")
    x, dy = rnd(50, 128, seed=3, scale=2.0), rnd(50, 128, seed=4)
    for fn in ("gelu_fwd", "quickgelu_fwd"):
        h, r = both(hip_ops, ref_ops, fn, x)
        close(h, r, 0.02, what=fn)
    h, r = both(hip_ops, ref_ops, "gelu_bwd", x, dy)
    close(h, r, 0.02, what="gelu bwd")
    h, r = both(hip_ops, ref_ops, "add", x, dy)
    close(h, r, 0.02, what="add")
    acc_h, acc_r = torch.ones(128, device="cuda:0"), torch.ones(128)
    hip_ops.colsum_accum(rnd(300, 128, seed=5).cuda(), acc_h)
    ref_ops.colsum_accum(rnd(300, 128, seed=5).float(), acc_r)
    close(acc_h, acc_r, 1e-3, rtol=1e-4, what="colsum")


def test_rope(hip_ops, ref_ops):
    T, nh, hd = 70, 6, 128
    g = torch.Generator().manual_seed(0)
    pos3 = torch.randint(0, 4000, (3, T), generator=g, dtype=torch.int32)
    (c, s), (cr, sr) = hip_ops.mrope_table(pos3.cuda(), hd, (16, 24, 24), 1e6), ref_ops.mrope_table(pos3, hd, (16, 24, 24), 1e6)
    # angles reach thousands of radians in fp32: allow one bf16 ulp of cos/sin (the table is bf16-rounded like the reference)
    close(c, cr, 8e-3, rtol=0, what="mrope cos")
    close(s, sr, 8e-3, rtol=0, what="mrope sin")
    x = rnd(T, (nh + 2) * hd, seed=1)
    for bwd in (False, True):
        h = hip_ops.rope_apply(x.cuda(), nh, hd, cr.cuda(), sr.cuda(), backward=bwd)
        r = ref_ops.rope_apply(x.float(), nh, hd, cr, sr, backward=bwd)
        close(h, r, 0.02, what="mrope apply bwd=%s" % bwd)
    # vision: head_dim 80
    hw = torch.randint(0, 46, (90, 2), generator=g, dtype=torch.int32)
    (c, s), (cr, sr) = hip_ops.vision_rope_table(hw.cuda(), 80), ref_ops.vision_rope_table(hw, 80)
    close(c, cr, 1e-4, rtol=0, what="vision cos")
    close(s, sr, 1e-4, rtol=0, what="vision sin")
    x = rnd(90, 3 * 4 * 80, seed=2)
    h = hip_ops.rope_apply(x.cuda(), 8, 80, cr.cuda(), sr.cuda())
    r = ref_ops.rope_apply(x.float(), 8, 80, cr, sr)
    close(h, r, 0.02, what="vision rope apply")


def test_gather_scatter_embed(hip_ops, ref_ops):
    table = rnd(50, 64, seed=1)
    ids = torch.tensor([3, 49, 0, 3, 7, 7, 7], dtype=torch.int32)
    h, r = hip_ops.gather_rows(table.cuda(), ids.cuda()), ref_ops.gather_rows(table.float(), ids)
    assert torch.equal(h.float().cpu(), r)
    dst_h, dst_r = torch.zeros(50, 64, dtype=BF16, device="cuda:0"), torch.zeros(50, 64)
    src = rnd(3, 64, seed=2)
    idx = torch.tensor([5, 1, 40], dtype=torch.int32)
    hip_ops.scatter_rows(src.cuda(), idx.cuda(), dst_h)
    ref_ops.scatter_rows(src.float(), idx, dst_r)
    assert torch.equal(dst_h.float().cpu(), dst_r)
    ids2 = torch.tensor([3, 49, -1, 3, 7, 7, 7], dtype=torch.int32)
    dt_h, dt_r = torch.zeros(50, 64, device="cuda:0"), torch.zeros(50, 64)
    dout = rnd(7, 64, seed=3)
    hip_ops.embed_bwd(dout.cuda(), ids2.cuda(), dt_h)
    ref_ops.embed_bwd(dout.float(), ids2, dt_r)
    close(dt_h, dt_r, 1e-5, rtol=1e-5, what="embed bwd")


@pytest.mark.parametrize("M,N,K", [(300, 512, 8192), (1600, 3584, 18944), (70, 200, 8320), (1600, 3584, 3584), (200, 264, 2112)])
def test_gemm_nt_split_k(hip_ops, ref_ops, M, N, K):
    """Thin outputs over a long K (the continuation forward's down / o projections) run a deterministic S-way split-K: all shares in one launch into fp32
    planes, summed in a fixed order with bias and residual.  Same result as the single-pass kernel up to the fp32 summation order; repeatable bit for bit."""
    a, b, bias, res = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05), rnd(N, seed=3), rnd(M, N, seed=4)
    assert hip_ops.SPLITK
    y = hip_ops.gemm_nt(a.cuda(), b.cuda(), bias=bias.cuda(), residual=res.cuda())
    y2 = hip_ops.gemm_nt(a.cuda(), b.cuda(), bias=bias.cuda(), residual=res.cuda())
    assert torch.equal(y, y2), "split-K must be deterministic"
    hip_ops.SPLITK = False
    try:
        y0 = hip_ops.gemm_nt(a.cuda(), b.cuda(), bias=bias.cuda(), residual=res.cuda())
    finally:
        del hip_ops.SPLITK
    r = ref_ops.gemm_nt(a.float(), b.float(), bias=bias.float(), residual=res.float())
    atol = 0.01 * math.sqrt(K) * 0.05 + 0.03
    close(y, r, atol, what="split-K vs oracle")
    close(y, y0.float().cpu(), atol, what="split-K vs single pass")


@pytest.mark.parametrize("M,K,N", [(1600, 152064 // 8, 3584), (300, 4096, 512), (129, 2112, 264)])
def test_gemm_nn_split_k_weight_as_stored(hip_ops, ref_ops, M, K, N):
    """dX = dY W with the weight as stored [K, N] and few output tiles (the lm_head's data gradient: K = vocabulary): split-K over the K-major form,
    no transposed weight copy."""
    a, w = rnd(M, K, seed=1, scale=0.1), rnd(K, N, seed=2, scale=0.1)
    assert hip_ops._splitk_ok(M, N, K)
    y = hip_ops.gemm_nn(a.cuda(), w.cuda())
    assert torch.equal(y, hip_ops.gemm_nn(a.cuda(), w.cuda())), "split-K must be deterministic"
    close(y, a.float() @ w.float(), 0.01 * math.sqrt(K) * 0.01 + 0.03, rtol=3e-2, what="NN split-K vs fp32")


# ------------------------------------------------------------------------------------------- fused-epilogue training GEMMs
class _unfused:
    """The same HipOps with the fused epilogues switched off (GEMM + separate elementwise kernels: the round-3 path)."""

    def __init__(self, ops):
        self.ops = ops

    def __enter__(self):
        self.ops.FUSE_EPI = False
        return self.ops

    def __exit__(self, *a):
        del self.ops.FUSE_EPI            # back to the class default


@pytest.mark.parametrize("M,I,K,save", [(300, 640, 256, True), (1111, 1032, 512, True), (700, 200, 128, False), (5074, 2048, 256, True), (65, 8, 64, True)])
def test_gemm_glu_fused_epilogue(hip_ops, ref_ops, M, I, K, save):
    """gate/up GEMM with SwiGLU in the epilogue (csrc/gemm.hip EPI 2): bit-identical to GEMM + swiglu_fwd, close to the oracle; I % 128 != 0 and
    output views included."""
    x, w = rnd(M, K, seed=1), rnd(2 * I, K, seed=2, scale=0.1)
    abuf = torch.zeros(M + 3, I + 16, dtype=BF16, device="cuda:0")
    gbuf = torch.zeros(M, 2 * I, dtype=BF16, device="cuda:0")
    a, gu = hip_ops.gemm_glu(x.cuda(), w.cuda(), a_out=abuf[3:, :I], gu_out=gbuf if save else None, save_gu=save)
    with _unfused(hip_ops) as o:
        a0, gu0 = o.gemm_glu(x.cuda(), w.cuda(), save_gu=True)
    assert torch.equal(a, a0), "fused SwiGLU epilogue differs from GEMM + swiglu_fwd"
    assert float(abuf[:3].abs().max()) == 0.0 and float(abuf[:, I:].abs().max()) == 0.0
    if save:
        assert gu.data_ptr() == gbuf.data_ptr() and torch.equal(gu, gu0)
    else:
        assert gu is None
    ar, gur = ref_ops.gemm_glu(x.float(), w.float())
    close(a, ar, 0.02 * math.sqrt(K) * 0.1 + 0.02, rtol=3e-2, what="glu a")


@pytest.mark.parametrize("M,N,K", [(13376 // 8, 5120, 1280), (300, 200, 64), (1000, 264, 128)])
def test_gemm_quickgelu_and_biased_glu_fused_epilogue(hip_ops, ref_ops, M, N, K):
    """Vision-tower MLPs: fc1 + bias + QuickGELU (Qwen2-VL) and gate/up + bias + SwiGLU (Qwen2.5-VL) in the GEMM epilogue, bit-identical to the
    separate kernels."""
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3)
    y = hip_ops.gemm_quickgelu(x.cuda(), w.cuda(), b.cuda())
    with _unfused(hip_ops) as o:
        y0 = o.gemm_quickgelu(x.cuda(), w.cuda(), b.cuda())
    assert torch.equal(y, y0)
    close(y, ref_ops.gemm_quickgelu(x.float(), w.float(), b.float()), 0.02 * math.sqrt(K) * 0.1 + 0.02, rtol=3e-2, what="quickgelu")
    I = N // 16 * 8                      # (the SwiGLU kernels need I % 8 == 0)
    a, _ = hip_ops.gemm_glu(x.cuda(), w.cuda()[: 2 * I], save_gu=False, bias=b.cuda()[: 2 * I])
    with _unfused(hip_ops) as o:
        a0, _ = o.gemm_glu(x.cuda(), w.cuda()[: 2 * I], save_gu=False, bias=b.cuda()[: 2 * I])
    assert torch.equal(a, a0)


@pytest.mark.parametrize("M,nh,nkv,K,row0", [(300, 4, 2, 256, 0), (1000, 28, 4, 512, 0), (333, 12, 2, 192, 77), (5074, 2, 2, 64, 0)])
def test_gemm_qkv_rope_fused_epilogue(hip_ops, ref_ops, M, nh, nkv, K, row0):
    """q|k|v projection + bias + M-RoPE in the GEMM epilogue (EPI 4): q / k / v bit-identical to GEMM + rope_apply; k written into rows
    [row0, row0 + M) of a wider cache buffer, v into a column view."""
    hd = 128
    qd, kvd = nh * hd, nkv * hd
    x, w, b = rnd(M, K, seed=1), rnd(qd + 2 * kvd, K, seed=2, scale=0.1), rnd(qd + 2 * kvd, seed=3)
    g = torch.Generator().manual_seed(5)
    ang = torch.rand(M, hd // 2, generator=g) * 6.28
    cos, sin = torch.cos(ang).to(BF16).float(), torch.sin(ang).to(BF16).float()
    kc = torch.zeros(row0 + M + 5, kvd, dtype=BF16, device="cuda:0")
    vb = torch.zeros(M, kvd + 8, dtype=BF16, device="cuda:0")
    q, k, v = hip_ops.gemm_qkv_rope(x.cuda(), w.cuda(), b.cuda(), cos.cuda(), sin.cuda(), nh, nkv, hd, k_out=kc[row0:row0 + M], v_out=vb[:, :kvd])
    with _unfused(hip_ops) as o:
        q0, k0, v0 = o.gemm_qkv_rope(x.cuda(), w.cuda(), b.cuda(), cos.cuda(), sin.cuda(), nh, nkv, hd)
    assert torch.equal(q, q0) and torch.equal(k, k0) and torch.equal(v, v0)
    assert k.data_ptr() == kc[row0:].data_ptr() and float(kc[:row0].abs().max() if row0 else 0.0) == 0.0 and float(kc[row0 + M:].abs().max()) == 0.0
    assert float(vb[:, kvd:].abs().max()) == 0.0
    qr, kr, vr = ref_ops.gemm_qkv_rope(x.float(), w.float(), b.float(), cos, sin, nh, nkv, hd)
    atol = 0.02 * math.sqrt(K) * 0.1 + 0.03
    close(q, qr, atol, rtol=3e-2, what="q rope"); close(k, kr, atol, rtol=3e-2, what="k rope"); close(v, vr, atol, rtol=3e-2, what="v")


@pytest.mark.parametrize("M,H,K", [(1000, 16, 1280), (300, 16, 128), (5074, 32, 64)])
def test_gemm_qkv_rope_vit_padded_heads(hip_ops, ref_ops, M, H, K):
    """Vision q|k|v projection + bias + 2-D rotary embedding written as 128-wide zero-padded heads (csrc/gemm.hip EPI 7): d < 40 at column d, d + 40 at 48 + d (round 6: live features end at 96).
    Bit-identical to GEMM + rope_apply(head dim 80) scattered into that layout; pad columns untouched (zero)."""
    hd, half = 80, 40
    E = H * hd
    x, w, b = rnd(M, K, seed=1), rnd(3 * E, K, seed=2, scale=0.1), rnd(3 * E, seed=3)
    ang = torch.rand(M, half, generator=torch.Generator().manual_seed(5)) * 6.28
    cos, sin = torch.cos(ang).contiguous(), torch.sin(ang).contiguous()
    assert hip_ops.vit_pad128_ok(H, hd)
    q128, k128, v128 = [torch.zeros(M, H * 128, dtype=BF16, device="cuda:0") for _ in range(3)]
    hip_ops.gemm_qkv_rope_vit(x.cuda(), w.cuda(), b.cuda(), cos.cuda(), sin.cuda(), H, half, q128, k128, v128)
    with _unfused(hip_ops) as o:
        qkv = o.gemm_nt(x.cuda(), w.cuda(), bias=b.cuda())
        q = o.rope_apply(qkv[:, :E], H, hd, cos.cuda(), sin.cuda())
        k = o.rope_apply(qkv[:, E:2 * E], H, hd, cos.cuda(), sin.cuda())
        v = qkv[:, 2 * E:]
    for got, want, nm in ((q128, q, "q"), (k128, k, "k"), (v128, v, "v")):
        g3, w3 = got.view(M, H, 128), want.reshape(M, H, hd)
        assert torch.equal(g3[:, :, :half], w3[:, :, :half]) and torch.equal(g3[:, :, 48:48 + half], w3[:, :, half:]), nm
        assert float(g3[:, :, half:48].abs().max()) == 0.0 and float(g3[:, :, 48 + half:].abs().max()) == 0.0, nm + " pad columns"
    qr = ref_ops.rope_apply(ref_ops.gemm_nt(x.float(), w.float(), bias=b.float())[:, :E], H, hd, cos, sin)
    close(q128.view(M, H, 128)[:, :, :half], qr.view(M, H, hd)[:, :, :half], 0.02 * math.sqrt(K) * 0.1 + 0.03, rtol=3e-2, what="q vs oracle")


@pytest.mark.parametrize("M,I,H", [(2048, 6400, 256), (1600, 8192, 128), (5074, 4096, 64), (1537, 8200, 64), (1440, 8192, 64), (1568, 8192, 64), (1824, 8192, 128), (2016, 8192, 64)])
def test_dgrad_glu_bwd_fused_epilogue(hip_ops, ref_ops, M, I, H):
    """Down-projection dgrad (weight as stored) with the SwiGLU backward in its epilogue (EPI 3): bit-identical to gemm_nn + swiglu_bwd."""
    dh, w, gu = rnd(M, H, seed=1), rnd(H, I, seed=2, scale=0.1), rnd(M, 2 * I, seed=3)
    d = hip_ops.dgrad_glu_bwd(dh.cuda(), w.cuda(), gu.cuda())
    with _unfused(hip_ops) as o:
        d0 = o.dgrad_glu_bwd(dh.cuda(), w.cuda(), gu.cuda())
    assert torch.equal(d, d0), "fused SwiGLU-backward epilogue differs from gemm_nn + swiglu_bwd"
    d2, dt = hip_ops.dgrad_glu_bwd(dh.cuda(), w.cuda(), gu.cuda(), want_t=True)
    assert dt is not None and torch.equal(d2, d0)
    assert torch.equal(dt, hip_ops.transpose(d0)), "dgu^T from the epilogue staging must equal transpose(dgu), zero padding included"
    if M <= 2048:
        close(d, ref_ops.dgrad_glu_bwd(dh.float(), w.float(), gu.float()), 0.02 * math.sqrt(H) * 0.1 + 0.03, rtol=4e-2, what="dgu")


# ------------------------------------------------------------------------------------------------------------- attention
def masks_causal(T):
    t = torch.arange(T, dtype=torch.int32)
    return torch.zeros(T, dtype=torch.int32), torch.zeros(T, dtype=torch.int32), t


def masks_prefix_shared(P, G, C):
    """Packed sequence: P prompt tokens then G groups of C completion tokens; slot == token index."""
    pre = torch.cat([torch.zeros(P), torch.full((G * C,), P)]).int()
    lo = torch.cat([torch.zeros(P), (P + torch.arange(G).repeat_interleave(C) * C).float()]).int()
    hi = torch.arange(P + G * C).int()
    return pre, lo, hi


def masks_segments(lengths):
    lo, hi = [], []
    a = 0
    for n in lengths:
        lo += [a] * n
        hi += [a + n - 1] * n
        a += n
    return torch.zeros(a, dtype=torch.int32), torch.tensor(lo, dtype=torch.int32), torch.tensor(hi, dtype=torch.int32)


def masks_prefix_only(P, n):
    """P causal prompt rows, then n rows that see the prompt through `pre` only: every third has an EMPTY [lo, hi], the others a 1-key interval."""
    pre = torch.cat([torch.zeros(P), torch.full((n,), P)]).int()
    idx = P + torch.arange(n)
    lo = torch.cat([torch.zeros(P), idx.float()]).int()
    hi = torch.cat([torch.arange(P).float(), torch.where(torch.arange(n) % 3 == 0, idx - 1, idx).float()]).int()
    return pre, lo, hi


ATT_CASES = [
    ("causal", 4, 2, 128, masks_causal(200)),
    ("causal-small-d", 4, 4, 32, masks_causal(77)),
    ("prefix-shared", 6, 2, 64, masks_prefix_shared(70, 3, 37)),
    ("prefix-shared-128", 14, 2, 128, masks_prefix_shared(150, 4, 21)),
    ("vit-segments", 4, 4, 80, masks_segments([60, 60, 60, 45])),
    ("windows", 2, 2, 80, masks_segments([16] * 9 + [4, 4, 7])),
    # head dim 128 takes the 32x32x16-MFMA backward kernels (round 3): ragged sizes, group 1 / 7, segments, rows that see only the prefix
    ("segments-128", 4, 4, 128, masks_segments([70, 3, 130, 64])),
    ("prefix-shared-128-g7", 7, 1, 128, masks_prefix_shared(333, 3, 50)),
    ("prefix-only-rows-128", 4, 2, 128, masks_prefix_only(90, 45)),
    ("tiny-128", 2, 2, 128, masks_causal(9)),
]


@pytest.mark.parametrize("name,nh,nkv,hd,m", ATT_CASES, ids=[c[0] for c in ATT_CASES])
def test_attention_fwd_bwd(hip_ops, ref_ops, name, nh, nkv, hd, m):
    pre, lo, hi = m
    T = pre.numel()
    S = T
    scale = hd ** -0.5
    q, k, v, do = rnd(T, nh * hd, seed=1), rnd(S, nkv * hd, seed=2), rnd(S, nkv * hd, seed=3), rnd(T, nh * hd, seed=4)
    vt_h = hip_ops.pack_transpose(v.cuda(), nkv, nkv, hd)
    vt_r = ref_ops.pack_transpose(v.float(), nkv, nkv, hd)
    assert torch.equal(vt_h.float().cpu(), vt_r), "pack_transpose must be exact"
    o_h, lse_h = hip_ops.attn_fwd(q.cuda(), k.cuda(), vt_h, pre.cuda(), lo.cuda(), hi.cuda(), nh, nkv, S, hd, scale)
    o_r, lse_r = ref_ops.attn_fwd(q.float(), k.float(), vt_r, pre, lo, hi, nh, nkv, S, hd, scale)
    close(o_h, o_r, 0.02, what=name + " O")
    close(lse_h, lse_r, 2e-3, rtol=1e-3, what=name + " lse")
    # split-KV must agree with the single pass
    o_s, lse_s = hip_ops.attn_fwd(q.cuda(), k.cuda(), vt_h, pre.cuda(), lo.cuda(), hi.cuda(), nh, nkv, S, hd, scale, nsplit=3)
    close(o_s, o_r, 0.02, what=name + " O split")
    close(lse_s, lse_r, 2e-3, rtol=1e-3, what=name + " lse split")
    if hd == 128:
        # the DEFAULT training / prefill / reference-forward kernel for head dim 128 (attn_fwd32_kernel, K and V read row-major): same op with
        # v_rows instead of V^T, against the oracle and against the 16x16-MFMA kernel above
        assert hip_ops.attn_fwd_rows_ok(hd)
        o_w, lse_w = hip_ops.attn_fwd(q.cuda(), k.cuda(), None, pre.cuda(), lo.cuda(), hi.cuda(), nh, nkv, S, hd, scale, v_rows=v.cuda())
        close(o_w, o_r, 0.02, what=name + " O (row-major V kernel)")
        close(lse_w, lse_r, 2e-3, rtol=1e-3, what=name + " lse (row-major V kernel)")
        close(o_w, o_h.float().cpu(), 0.02, what=name + " O rows vs V^T kernel")
        close(lse_w, lse_h.float().cpu(), 2e-3, rtol=1e-3, what=name + " lse rows vs V^T kernel")
        # round 6: the live-96 launch (the vision towers' padded heads: features 96..127 of q / k / v are zero) computes the SAME bits in columns 0..95 of every
        # head with three quarters of the MFMAs (skipped products are exact zeros) and leaves columns 96..127 of the output alone
        z = lambda t, n: (t.view(-1, n, hd) * (torch.arange(hd) < 96).to(t.dtype)).reshape(t.shape[0], n * hd).cuda()
        q6, k6, v6 = z(q, nh), z(k, nkv), z(v, nkv)
        o_full, lse_full = hip_ops.attn_fwd(q6, k6, None, pre.cuda(), lo.cuda(), hi.cuda(), nh, nkv, S, hd, scale, v_rows=v6)
        o_96 = torch.full((T, nh * hd), 7.0, dtype=BF16, device="cuda")
        _, lse_96 = hip_ops.attn_fwd(q6, k6, None, pre.cuda(), lo.cuda(), hi.cuda(), nh, nkv, S, hd, scale, v_rows=v6, out=o_96, live96=True)
        assert torch.equal(o_96.view(T, nh, hd)[:, :, :96], o_full.view(T, nh, hd)[:, :, :96]) and torch.equal(lse_96, lse_full), name + " live-96 launch"
        assert bool((o_96.view(T, nh, hd)[:, :, 96:] == 7.0).all()), name + " live-96 launch wrote beyond column 96"
    # backward (uses the oracle's O / lse so that only the backward kernels are under test)
    dq_h, dk_h, dv_h = hip_ops.attn_bwd(q.cuda(), k.cuda(), v.cuda(), o_r.to(BF16).cuda(), do.cuda(), lse_r.cuda(), pre.cuda(), lo.cuda(),
                                        hi.cuda(), nh, nkv, S, hd, scale)
    dq_r, dk_r, dv_r = ref_ops.attn_bwd(q.float(), k.float(), v.float(), o_r, do.float(), lse_r, pre, lo, hi, nh, nkv, S, hd, scale)
    close(dq_h, dq_r, 0.03, rtol=3e-2, what=name + " dQ")
    close(dk_h, dk_r, 0.03 * math.sqrt(nh // nkv) + 0.02, rtol=3e-2, what=name + " dK")
    close(dv_h, dv_r, 0.03 * math.sqrt(nh // nkv) + 0.02, rtol=3e-2, what=name + " dV")


FWD64_CASES = [
    # name, n_heads, n_kv, P, G, C, continuation, q scale (large: the lazy running maximum rescales often - the kernel's cold path)
    ("g7-kv4", 28, 4, 700, 8, 50, False, 1.0),
    ("g7-kv4-continuation", 28, 4, 210, 4, 33, True, 1.0),
    ("g6-kv2-hot-scores", 12, 2, 333, 8, 25, False, 6.0),
    ("g2-kv3", 6, 3, 333, 3, 77, False, 1.0),
    ("g1-kv16", 16, 16, 300, 2, 100, False, 1.0),
    ("one-tile", 28, 4, 5, 2, 3, False, 1.0),
    ("two-tokens", 28, 4, 1, 1, 1, False, 1.0),
]


@pytest.mark.parametrize("name,nh,nkv,P,G,C,cont,qs", FWD64_CASES, ids=[c[0] for c in FWD64_CASES])
def test_attention_fwd64_bit_identical_to_fwd32(hip_ops, name, nh, nkv, P, G, C, cont, qs, monkeypatch):
    """Round 6: attn_fwd64_kernel (64 query rows per wave, one wave per SIMD, softmax of tile t in the MFMA gaps of tiles t-1 / t+1; csrc/attn_fwd64.hip) is the
    head-dim-128 forward from 3 072 key slots on (TR1_FWD64=1 forces it at any size).  It keeps attn_fwd32_kernel's 32-row softmax groups, lazy-maximum decisions and summation orders, so O and the LSE must
    agree with that kernel BIT FOR BIT (the 32-row kernel is held to the oracle by the tests above and below; TR1_FWD64 is read per call)."""
    hd = 128
    qd, kvd = nh * hd, nkv * hd
    S = P + G * C
    pre, lo, hi = masks_prefix_shared(P, G, C)
    qkv = rnd(S, qd + 2 * kvd, seed=21).cuda()
    qkv[:, :qd] *= qs
    q_all, k_view, v_view = qkv[:, :qd], qkv[:, qd:qd + kvd], qkv[:, qd + kvd:]
    r0 = P if cont else 0
    args = (q_all[r0:], k_view, None, pre[r0:].cuda(), lo[r0:].cuda(), hi[r0:].cuda(), nh, nkv, S, hd, hd ** -0.5)
    monkeypatch.setenv("TR1_FWD64", "0")
    o32, l32 = hip_ops.attn_fwd(*args, v_rows=v_view)
    monkeypatch.setenv("TR1_FWD64", "1")
    o64, l64 = hip_ops.attn_fwd(*args, v_rows=v_view)
    torch.cuda.synchronize()
    assert not bool(torch.isnan(o64.float()).any())
    assert torch.equal(o32.view(torch.int16), o64.view(torch.int16)), name + " O"
    assert torch.equal(l32.view(torch.int32), l64.view(torch.int32)), name + " lse"


FWD64_MASK_CASES = [
    ("segments", 4, 4, lambda: masks_segments([70, 3, 130, 64, 200])),          # group 1, per-segment intervals (what a forced launch on a vision-style mask sees)
    ("prefix-only-rows", 4, 2, lambda: masks_prefix_only(90, 45)),               # rows whose second interval is empty or one key wide
    ("causal-tiny", 2, 2, lambda: masks_causal(9)),
    ("causal-g7-ragged", 28, 4, lambda: masks_causal(333)),                      # 333 slots: the last tile is partial, 2 331 packed rows: the last block too
    ("default-dispatch-3100-slots", 28, 4, lambda: masks_prefix_shared(2900, 2, 100)),   # TR1_FWD64 unset: the launch picks the 64-row kernel by itself
]


@pytest.mark.parametrize("name,nh,nkv,mk", FWD64_MASK_CASES, ids=[c[0] for c in FWD64_MASK_CASES])
def test_attention_fwd64_bit_identical_other_masks(hip_ops, name, nh, nkv, mk, monkeypatch):
    """The 64-row forward on the other mask families of the path (segments, prefix-only rows, plain causal, ragged sizes) and through the launch's own
    key-range rule: bit-identical O and LSE against the 32-row kernel, which the tests above hold to the oracle."""
    hd = 128
    pre, lo, hi = [t.cuda() for t in mk()]
    S = pre.numel()
    q, k, v = rnd(S, nh * hd, seed=41).cuda(), rnd(S, nkv * hd, seed=42).cuda(), rnd(S, nkv * hd, seed=43).cuda()
    args = (q, k, None, pre, lo, hi, nh, nkv, S, hd, hd ** -0.5)
    monkeypatch.setenv("TR1_FWD64", "0")
    o32, l32 = hip_ops.attn_fwd(*args, v_rows=v)
    if name.startswith("default-dispatch"):
        monkeypatch.delenv("TR1_FWD64")
    else:
        monkeypatch.setenv("TR1_FWD64", "1")
    o64, l64 = hip_ops.attn_fwd(*args, v_rows=v)
    torch.cuda.synchronize()
    assert not bool(torch.isnan(o64.float()).any())
    assert torch.equal(o32.view(torch.int16), o64.view(torch.int16)), name + " O"
    assert torch.equal(l32.view(torch.int32), l64.view(torch.int32)), name + " lse"


ROWS_CASES = [
    # name, n_heads, n_kv, P, G, C, continuation (q = completion rows only, T < n_slots)
    ("g7-kv4-packed", 28, 4, 210, 4, 33, False),          # Qwen2-VL-7B head layout; 342 * 7 packed rows: not a multiple of 256
    ("g7-kv4-continuation", 28, 4, 210, 4, 33, True),
    ("g6-kv2-continuation", 12, 2, 333, 8, 25, True),     # Qwen2-VL-2B head layout
    ("g1-continuation-tiny", 2, 2, 5, 2, 3, True),
]


@pytest.mark.parametrize("name,nh,nkv,P,G,C,cont", ROWS_CASES, ids=[c[0] for c in ROWS_CASES])
def test_attention_fwd_rows_strided_views_and_continuation(hip_ops, ref_ops, name, nh, nkv, P, G, C, cont):
    """tr1_attn_fwd_rows as model.llm_fwd calls it - K and V are COLUMN VIEWS of the fused [S, q|k|v] projection buffer (row stride =
    qkv width), the output is a view of a wider buffer, and the continuation forward passes only the completion rows as queries (T < n_slots)
    while K / V span the prompt + completion slots."""
    hd = 128
    qd, kvd = nh * hd, nkv * hd
    S = P + G * C
    pre, lo, hi = masks_prefix_shared(P, G, C)
    qkv = rnd(S, qd + 2 * kvd, seed=11).cuda()
    q_all, k_view, v_view = qkv[:, :qd], qkv[:, qd:qd + kvd], qkv[:, qd + kvd:]
    assert not k_view.is_contiguous() and not v_view.is_contiguous()
    r0 = P if cont else 0
    q = q_all[r0:]
    T = S - r0
    obuf = torch.zeros(S, qd + 64, dtype=BF16, device="cuda:0")
    o_view = obuf[r0:, :qd]
    o_w, lse_w = hip_ops.attn_fwd(q, k_view, None, pre[r0:].cuda(), lo[r0:].cuda(), hi[r0:].cuda(), nh, nkv, S, hd, hd ** -0.5, v_rows=v_view, out=o_view)
    assert o_w.data_ptr() == o_view.data_ptr()
    k32, v32 = k_view.float().cpu(), v_view.float().cpu()
    o_r, lse_r = ref_ops.attn_fwd(q.float().cpu(), k32, ref_ops.pack_transpose(v32, nkv, nkv, hd), pre[r0:], lo[r0:], hi[r0:], nh, nkv, S, hd, hd ** -0.5)
    close(o_w, o_r, 0.02, what=name + " O")
    close(lse_w, lse_r, 2e-3, rtol=1e-3, what=name + " lse")
    assert float(obuf[:, qd:].abs().max()) == 0.0 and (r0 == 0 or float(obuf[:r0].abs().max()) == 0.0), "writes outside the output view"
    # and against the V^T kernel (the path taken without v_rows) on contiguous copies of the same operands
    vt = hip_ops.pack_transpose(v_view.contiguous(), nkv, nkv, hd)
    o_t, lse_t = hip_ops.attn_fwd(q.contiguous(), k_view.contiguous(), vt, pre[r0:].cuda(), lo[r0:].cuda(), hi[r0:].cuda(), nh, nkv, S, hd, hd ** -0.5)
    close(o_w, o_t.float().cpu(), 0.02, what=name + " O rows vs V^T kernel")
    close(lse_w, lse_t.float().cpu(), 2e-3, rtol=1e-3, what=name + " lse rows vs V^T kernel")


@pytest.mark.parametrize("name,nh,nkv,hd,m", [c for c in ATT_CASES if c[0] in ("causal", "prefix-shared-128", "prefix-shared-128-g7", "prefix-shared", "tiny-128")] +
                         [("prefix-shared-big", 28, 4, 128, masks_prefix_shared(700, 8, 60))], ids=lambda v: v if isinstance(v, str) else None)
def test_attention_bwd_with_rope_backward_folded_in(hip_ops, ref_ops, name, nh, nkv, hd, m):
    """tr1_attn_bwd_rope: dQ / dK rotated by the transposed M-RoPE matrix inside the dQ kernel's epilogue and the dK / dV partial-sum kernel (head dim 128),
    written into column views of one [T, q|k|v] buffer - bit-identical to tr1_attn_bwd followed by rope_apply(backward=True), and close to the oracle."""
    pre, lo, hi = m
    T = pre.numel()
    scale = hd ** -0.5
    q, k, v, do = rnd(T, nh * hd, seed=1), rnd(T, nkv * hd, seed=2), rnd(T, nkv * hd, seed=3), rnd(T, nh * hd, seed=4)
    ang = torch.rand(T, hd // 2, generator=torch.Generator().manual_seed(9)) * 6.28
    cos, sin = torch.cos(ang).to(BF16).float(), torch.sin(ang).to(BF16).float()
    o_r, lse_r = ref_ops.attn_fwd(q.float(), k.float(), ref_ops.pack_transpose(v.float(), nkv, nkv, hd), pre, lo, hi, nh, nkv, T, hd, scale)
    qd, kvd = nh * hd, nkv * hd
    buf = torch.zeros(T, qd + 2 * kvd, dtype=BF16, device="cuda:0")
    args = (q.cuda(), k.cuda(), v.cuda(), o_r.to(BF16).cuda(), do.cuda(), lse_r.cuda(), pre.cuda(), lo.cuda(), hi.cuda(), nh, nkv, T, hd, scale)
    hip_ops.attn_bwd(*args, dq_out=buf[:, :qd], dk_out=buf[:, qd:qd + kvd], dv_out=buf[:, qd + kvd:], rope=(cos.cuda(), sin.cuda()))
    with _unfused(hip_ops) as o:
        dq0, dk0, dv0 = o.attn_bwd(*args, rope=(cos.cuda(), sin.cuda()))
    assert torch.equal(buf[:, :qd], dq0) and torch.equal(buf[:, qd:qd + kvd], dk0) and torch.equal(buf[:, qd + kvd:], dv0)
    dq_r, dk_r, dv_r = ref_ops.attn_bwd(q.float(), k.float(), v.float(), o_r, do.float(), lse_r, pre, lo, hi, nh, nkv, T, hd, scale, rope=(cos, sin))
    close(buf[:, :qd], dq_r, 0.04, rtol=4e-2, what=name + " dQ (rope bwd)")
    close(buf[:, qd:qd + kvd], dk_r, 0.04 * math.sqrt(nh // nkv) + 0.03, rtol=4e-2, what=name + " dK (rope bwd)")


@pytest.mark.parametrize("R,C", [(5074, 4608), (100, 72), (64, 64), (333, 200)])
def test_transpose_with_column_sums(hip_ops, ref_ops, R, C):
    x = rnd(R, C, seed=1)
    cs = torch.full((C,), 0.25, device="cuda:0")
    t = hip_ops.transpose(x.cuda(), colsum=cs)
    assert torch.equal(t, hip_ops.transpose(x.cuda())), "the transpose itself must not change"
    close(cs, x.float().sum(0) + 0.25, 1e-3 * math.sqrt(R), rtol=1e-4, what="column sums")


def test_attention_decode_over_cache(hip_ops, ref_ops):
    """G rollout rows, one new token each, over a KV cache = shared prompt prefix + per-row suffix slots (split-KV path)."""
    P, G, C, step, nh, nkv, hd = 300, 8, 20, 11, 14, 2, 128
    S = P + G * C
    k, v, q = rnd(S, nkv * hd, seed=1), rnd(S, nkv * hd, seed=2), rnd(G, nh * hd, seed=3)
    pre = torch.full((G,), P, dtype=torch.int32)
    lo = (P + torch.arange(G) * C).int()
    hi = (lo + step).int()
    vt_h = hip_ops.pack_transpose(v.cuda(), nkv, nkv, hd)
    vt_r = ref_ops.pack_transpose(v.float(), nkv, nkv, hd)
    o_r, _ = ref_ops.attn_fwd(q.float(), k.float(), vt_r, pre, lo, hi, nh, nkv, S, hd, hd ** -0.5)
    for nsplit in (1, 4, 16, 57, 64):
        o_h, _ = hip_ops.attn_fwd(q.cuda(), k.cuda(), vt_h, pre.cuda(), lo.cuda(), hi.cuda(), nh, nkv, S, hd, hd ** -0.5, nsplit=nsplit,
                                  need_lse=False)
        close(o_h, o_r, 0.02, what="decode attention nsplit=%d" % nsplit)


def test_kv_cache_append(hip_ops, ref_ops):
    nkv, hd, S = 2, 64, 256
    newk = rnd(5, nkv * hd, seed=1)
    slots = torch.tensor([7, 100, 101, 255, 0], dtype=torch.int32)
    kc_h, kc_r = torch.zeros(S, nkv * hd, dtype=BF16, device="cuda:0"), torch.zeros(S, nkv * hd)
    hip_ops.scatter_slots(newk.cuda(), kc_h, slots.cuda())
    ref_ops.scatter_slots(newk.float(), kc_r, slots)
    assert torch.equal(kc_h.float().cpu(), kc_r)
    vt_h, vt_r = torch.zeros(nkv * hd, S, dtype=BF16, device="cuda:0"), torch.zeros(nkv * hd, S)
    hip_ops.pack_transpose(newk.cuda(), nkv, nkv, hd, slots=slots.cuda(), out=vt_h)
    ref_ops.pack_transpose(newk.float(), nkv, nkv, hd, slots=slots, out=vt_r)
    assert torch.equal(vt_h.float().cpu(), vt_r)


# ------------------------------------------------------------------------------------------------------- vocabulary side
def test_logp_entropy(hip_ops, ref_ops):
    R, V = 37, 5000 + 8
    logits = rnd(R, V, seed=1, scale=3.0)
    tg = torch.randint(0, V, (R,), generator=torch.Generator().manual_seed(2), dtype=torch.int32)
    (lp, en, ls), (lpr, enr, lsr) = hip_ops.logp_entropy_fwd(logits.cuda(), tg.cuda()), ref_ops.logp_entropy_fwd(logits.float(), tg)
    close(lp, lpr, 1e-3, rtol=1e-4, what="logp")
    close(en, enr, 1e-3, rtol=1e-4, what="entropy")
    close(ls, lsr, 1e-3, rtol=1e-4, what="lse")
    dlogp = torch.randn(R, generator=torch.Generator().manual_seed(3))
    dh = hip_ops.logp_bwd(logits.cuda(), tg.cuda(), lsr.cuda(), dlogp.cuda(), inplace=False)
    dr = ref_ops.logp_bwd(logits.float(), tg, lsr, dlogp, inplace=False)
    close(dh, dr, 1e-3, rtol=1e-2, what="dlogits")


@pytest.mark.parametrize("use_grpo,beta", [(True, 0.04), (False, 0.04), (True, 0.0), (False, 0.0)])
def test_grpo_loss(hip_ops, ref_ops, use_grpo, beta):
    G, C = 8, 50
    g = torch.Generator().manual_seed(0)
    logp = -torch.rand(G, C, generator=g) * 3
    ref = (logp + 0.2 * torch.randn(G, C, generator=g)) if beta else None
    lens = torch.randint(1, C + 1, (G,), generator=g)
    mask = (torch.arange(C)[None, :] < lens[:, None]).int()
    adv = torch.randn(G, generator=g)
    outs_h = hip_ops.grpo_loss(logp.cuda(), ref.cuda() if ref is not None else None, mask.cuda(), adv.cuda(), beta, use_grpo, 0.5)
    outs_r = ref_ops.grpo_loss(logp, ref, mask, adv, beta, use_grpo, 0.5)
    for h, r, w in zip(outs_h, outs_r, ("dlogp", "out3", "row_len", "row_kl")):
        close(h, r, 1e-5, rtol=1e-4, what="grpo " + w)


@pytest.mark.parametrize("rows,V", [(8, 3000), (16, 152064), (3, 1001), (5, 40), (2, 16392)])
def test_sampler(hip_ops, ref_ops, rows, V):
    """(16, 152064): the rollout's shape (fused slice-sum + pick launch, 19 iterations per wave); 1001: V % 8 != 0 takes the separate kernels;
    40 / 16392: fewer chunks than waves / a ragged last segment."""
    C = 4
    logits = rnd(rows, V, seed=1, scale=2.5)
    for top_k in (0, 50):
        tok_h = torch.zeros(rows, C, dtype=torch.int32, device="cuda:0")
        tok_r = torch.zeros(rows, C, dtype=torch.int32)
        u_h = torch.zeros(rows, device="cuda:0")
        u_r = torch.zeros(rows)
        step = torch.tensor([2], dtype=torch.int32)
        hip_ops.sample_tokens(logits.cuda(), 0.9, top_k, 1234, step.cuda(), tok_h, None, 1, 0, False, u_out=u_h)
        ref_ops.sample_tokens(logits.float(), 0.9, top_k, 1234, step, tok_r, None, 1, 0, False, u_out=u_r)
        close(u_h, u_r, 1e-7, rtol=0, what="philox uniforms")  # integer RNG: must agree exactly
        # the drawn token must be the inverse-CDF answer for that uniform (allowing fp slack at interval edges)
        x = logits.float() / 0.9
        for r in range(rows):
            xr = x[r].double()
            keep = xr >= (torch.topk(xr, top_k).values[-1] if top_k and top_k < V else -1e30)
            p = torch.where(keep, (xr - xr.max()).exp(), torch.zeros_like(xr))
            cdf = torch.cumsum(p, 0) / p.sum()
            t = int(tok_h[r, 2])
            assert keep[t], "sampled a filtered token"
            lo_c = float(cdf[t - 1]) if t > 0 else 0.0
            assert lo_c - 1e-4 <= float(u_r[r]) <= float(cdf[t]) + 1e-4, (r, t, lo_c, float(u_r[r]), float(cdf[t]))
        assert (tok_h[:, [0, 1, 3]] == 0).all(), "only column *step is written"


def test_adamw_and_clip(hip_ops, ref_ops):
    n = 10007
    g = torch.Generator().manual_seed(0)
    p = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) * 3 for _ in range(3)]
    st_h = [p.clone().cuda(), torch.zeros(n, device="cuda:0"), torch.zeros(n, device="cuda:0"), None, torch.zeros(n, dtype=BF16, device="cuda:0")]
    st_r = [p.clone(), torch.zeros(n), torch.zeros(n), None, torch.zeros(n, dtype=BF16)]
    for step, gr in enumerate(grads, 1):
        gh, grr = gr.clone().cuda(), gr.clone()
        ss_h, ss_r = torch.zeros(1, device="cuda:0"), torch.zeros(1)
        hip_ops.sumsq_accum(gh, ss_h)
        ref_ops.sumsq_accum(grr, ss_r)
        close(ss_h, ss_r, 0, rtol=1e-4, what="sumsq")
        hip_ops.adamw_step(st_h[0], st_h[1], st_h[2], gh, st_h[4], 1e-3, 0.9, 0.999, 1e-8, 0.01, step, sumsq=ss_h, max_norm=1.0, grad_mult=0.5)
        ref_ops.adamw_step(st_r[0], st_r[1], st_r[2], grr, st_r[4], 1e-3, 0.9, 0.999, 1e-8, 0.01, step, sumsq=ss_r, max_norm=1.0, grad_mult=0.5)
        assert float(gh.abs().max()) == 0.0, "zero_grad"
    close(st_h[0], st_r[0], 1e-6, rtol=1e-5, what="adamw master")
    close(st_h[1], st_r[1], 1e-7, rtol=1e-4, what="adamw m")
    close(st_h[2], st_r[2], 1e-9, rtol=1e-4, what="adamw v")
    assert torch.equal(st_h[4].float().cpu(), st_h[0].to(BF16).float().cpu()), "bf16 copy = round(master)"
    # and against torch.optim.AdamW itself (the DeepSpeed/HF default optimizer semantics)
    pt = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([pt], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    for gr in grads:
        pt.grad = gr.clone() * 0.5
        torch.nn.utils.clip_grad_norm_([pt], 1.0)
        opt.step()
    close(st_h[0], pt.detach(), 1e-6, rtol=1e-5, what="adamw vs torch.optim.AdamW")


@pytest.mark.parametrize("R,V,K", [(48, 512, 128), (300, 1024, 256), (7, 256, 64), (1600, 152064, 3584),
                                   (64, 320, 128), (33, 448, 64), (1600, 151936, 1536)])       # V % 256 != 0: the 2B vocabulary (151936 = 593.5 x 256)
def test_lmhead_lse_fused_epilogue(hip_ops, ref_ops, R, V, K):
    """lm_head with the log-softmax statistics reduced in the GEMM epilogue (no [R, V] logits in HBM) == the materialised path
    (GEMM -> bf16 logits -> one-pass logp / entropy kernel), and == the oracle at sizes it runs in seconds."""
    g = torch.Generator().manual_seed(3)
    hn = rnd(R, K, seed=1, scale=1.0)
    w = rnd(V, K, seed=2, scale=2.0 / math.sqrt(K))
    tg = torch.randint(0, V, (R,), generator=g, dtype=torch.int32)
    tg[0], tg[R - 1] = 0, V - 1                                   # first / last vocabulary column (first / last 64-column slice)
    fused = hip_ops.lmhead_lse(hn.cuda(), w.cuda(), tg.cuda())
    assert fused is not None
    logits = hip_ops.gemm_nt(hn.cuda(), w.cuda())
    mat = hip_ops.logp_entropy_fwd(logits, tg.cuda())
    for a, b, name in zip(fused, mat, ("logp", "entropy", "lse")):
        assert torch.isfinite(a).all(), name
        close(a, b, 2e-3, rtol=1e-4, what="fused vs materialised " + name)
    if V <= 4096:
        # oracle on the SAME bf16-rounded logits (the reference's logits are bf16 too): only summation order and exp implementation differ
        lg = (hn.float() @ w.float().t()).to(BF16).float()
        ref = ref_ops.logp_entropy_fwd(lg, tg)
        close(fused[0], ref[0], 5e-3, rtol=0, what="fused logp vs oracle")
        close(fused[1], ref[1], 5e-3, rtol=0, what="fused entropy vs oracle")


def test_adamw_reads_bf16_wire_gradient(hip_ops):
    """Data-parallel form: norm and update consume the all-reduced gradient from its bf16 wire buffer; bit-equal to copying it back into the
    fp32 accumulator first (bf16 -> fp32 is exact), and the accumulator is zeroed without being read."""
    n = 100003
    g = torch.Generator().manual_seed(1)
    p = torch.randn(n, generator=g)
    wire = (torch.randn(n, generator=g) * 2).to(BF16).cuda()
    outs = []
    for use16 in (False, True):
        st = [p.clone().cuda(), torch.zeros(n, device="cuda:0"), torch.zeros(n, device="cuda:0"), torch.zeros(n, dtype=BF16, device="cuda:0")]
        acc = (wire.float() if not use16 else torch.full((n,), 7.0, device="cuda:0"))      # use16: garbage in the accumulator must not matter
        ss = torch.zeros(1, device="cuda:0")
        hip_ops.sumsq_accum(wire if use16 else acc, ss)
        hip_ops.adamw_step(st[0], st[1], st[2], acc, st[3], 1e-3, 0.9, 0.999, 1e-8, 0.01, 1, sumsq=ss, max_norm=1.0, grad_mult=0.25, g16=wire if use16 else None)
        assert float(acc.abs().max()) == 0.0
        outs.append((ss.cpu(), [t.clone().cpu() for t in st]))
    assert abs(float(outs[0][0]) - float(outs[1][0])) <= 1e-6 * float(outs[0][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.allclose(a.float(), b.float(), atol=1e-7, rtol=1e-6)


def test_decode_qkv_post(hip_ops, ref_ops):
    """Fused decode post-projection == rope(q), rope(k) -> K cache rows, v -> V^T cache columns."""
    R, nh, nkv, hd, S = 16, 14, 2, 128, 512
    qkv = rnd(R, (nh + 2 * nkv) * hd, seed=1)
    g = torch.Generator().manual_seed(0)
    pos3 = torch.randint(0, 4000, (3, R), generator=g, dtype=torch.int32)
    cos, sin = ref_ops.mrope_table(pos3, hd, (16, 24, 24), 1e6)
    slots = torch.randperm(S, generator=g)[:R].int()
    kc_h, vt_h = torch.zeros(S, nkv * hd, dtype=BF16, device="cuda:0"), torch.zeros(nkv * hd, S, dtype=BF16, device="cuda:0")
    kc_r, vt_r = torch.zeros(S, nkv * hd), torch.zeros(nkv * hd, S)
    q_h = hip_ops.decode_qkv_post(qkv.cuda(), cos.cuda(), sin.cuda(), kc_h, vt_h, slots.cuda(), nh, nkv, hd)
    q_r = ref_ops.decode_qkv_post(qkv.float(), cos, sin, kc_r, vt_r, slots, nh, nkv, hd)
    close(q_h, q_r, 0.02, what="fused q rope")
    close(kc_h, kc_r, 0.02, what="fused k append")
    assert torch.equal(vt_h.float().cpu(), vt_r), "fused v append must be exact"


def test_attention_decode_batched_prompts(hip_ops, ref_ops):
    """One launch for the decode rows of several prompts, each over its own region of a unified KV cache."""
    B, G, C, step, nh, nkv, hd = 3, 8, 12, 5, 14, 2, 128
    Ps = [300, 170, 420]
    s_cap = 576
    k, v = rnd(B * s_cap, nkv * hd, seed=1), rnd(B * s_cap, nkv * hd, seed=2)
    q = rnd(B * G, nh * hd, seed=3)
    pre = torch.cat([torch.full((G,), P, dtype=torch.int32) for P in Ps])
    lo = torch.cat([(P + torch.arange(G) * C).int() for P in Ps])
    hi = (lo + step).int()
    vt_h = torch.zeros(nkv * hd, B * s_cap, dtype=BF16, device="cuda:0")
    vt_r = torch.zeros(nkv * hd, B * s_cap)
    for b in range(B):
        vt_h[:, b * s_cap:(b + 1) * s_cap] = hip_ops.pack_transpose(v[b * s_cap:(b + 1) * s_cap].cuda(), nkv, nkv, hd)
        vt_r[:, b * s_cap:(b + 1) * s_cap] = ref_ops.pack_transpose(v[b * s_cap:(b + 1) * s_cap].float(), nkv, nkv, hd)
    o_r, _ = ref_ops.attn_fwd(q.float(), k.float(), vt_r, pre, lo, hi, nh, nkv, s_cap, hd, hd ** -0.5, n_batch=B, kv_batch_slots=s_cap)
    for nsplit in (1, 5, 12, 28):
        o_h, _ = hip_ops.attn_fwd(q.cuda(), k.cuda(), vt_h, pre.cuda(), lo.cuda(), hi.cuda(), nh, nkv, s_cap, hd, hd ** -0.5, nsplit=nsplit,
                                  need_lse=False, n_batch=B, kv_batch_slots=s_cap)
        close(o_h, o_r, 0.02, what="batched decode attention nsplit=%d" % nsplit)
        if nsplit > 1:
            # the per-step tile plan (layer 0 publishes, the other layers reuse): all three modes give the same bits, also with garbage in the plan
            plan = hip_ops.attn_plan(G, nh, nkv, B)
            plan.fill_(0x7fffffff)
            o_w, _ = hip_ops.attn_fwd(q.cuda(), k.cuda(), vt_h, pre.cuda(), lo.cuda(), hi.cuda(), nh, nkv, s_cap, hd, hd ** -0.5, nsplit=nsplit,
                                      need_lse=False, n_batch=B, kv_batch_slots=s_cap, plan=plan, plan_mode=1)
            q2 = rnd(B * G, nh * hd, seed=4)       # "another layer": different q / cache contents, same masks
            o_0, _ = hip_ops.attn_fwd(q2.cuda(), k.cuda(), vt_h, pre.cuda(), lo.cuda(), hi.cuda(), nh, nkv, s_cap, hd, hd ** -0.5, nsplit=nsplit,
                                      need_lse=False, n_batch=B, kv_batch_slots=s_cap)
            o_0 = o_0.clone()
            o_2, _ = hip_ops.attn_fwd(q2.cuda(), k.cuda(), vt_h, pre.cuda(), lo.cuda(), hi.cuda(), nh, nkv, s_cap, hd, hd ** -0.5, nsplit=nsplit,
                                      need_lse=False, n_batch=B, kv_batch_slots=s_cap, plan=plan, plan_mode=2)
            # publishing launch (mode 1) == plain launch bit for bit (same kernel); the reading launches (mode 2) take the LDS-DMA kernel at head
            # dim 128 (round 3: attn_dec32_kernel) - same tiles, same masks, another accumulation order
            assert torch.equal(o_w, o_h), "publishing the plan must not change the result (nsplit=%d)" % nsplit
            close(o_2, o_0, 0.01, what="plan-reading decode attention nsplit=%d" % nsplit)
            close(o_2, ref_ops.attn_fwd(q2.float(), k.float(), vt_r, pre, lo, hi, nh, nkv, s_cap, hd, hd ** -0.5, n_batch=B, kv_batch_slots=s_cap)[0], 0.02,
                  what="plan-reading decode attention vs oracle nsplit=%d" % nsplit)
            cnt = plan.view(B, -1, 1025)[:, :, 1024]
            assert int(cnt.min()) > 0 and int(cnt.max()) <= 1024, cnt

@pytest.mark.parametrize("B,G,nh,nkv,nsplit,N", [(2, 8, 28, 4, 28, 3584), (2, 8, 12, 2, 21, 1536), (1, 16, 14, 2, 28, 1792), (2, 16, 28, 4, 16, 3584),
                                                 (2, 5, 12, 4, 7, 512), (1, 3, 4, 4, 2, 4096), (1, 1, 28, 4, 28, 3584), (1, 17, 28, 4, 8, 3584), (3, 8, 28, 4, 8, 3584),
                                                 (1, 8, 4, 4, 64, 512)])
def test_decode_o_projection_on_fragment_major_attention_rows(hip_ops, B, G, nh, nkv, nsplit, N):
    """Round 5 (csrc/oproj.hip): the split-KV merge writes its rows fragment-major ([k / 32][16 rows][32 k]: the 16 rows x 64 bytes of an MFMA operand fragment
    are one contiguous KiB) and the o projection keeps its whole weight slice in flight with no cross-block fixup.  The fragment-major rows hold exactly the
    bits of the row-major merge; the projection agrees with a float64 product; 16- and 32-row forms, odd row counts, column blocks of 14 / 6 / 7 / 16 columns."""
    hd, C, step = 128, 12, 5
    Ps = [300, 170, 420][:B]
    s_cap = 576
    k, v = rnd(B * s_cap, nkv * hd, seed=1).cuda(), rnd(B * s_cap, nkv * hd, seed=2)
    pre = torch.cat([torch.full((G,), P, dtype=torch.int32) for P in Ps]).cuda()
    lo = torch.cat([(P + torch.arange(G) * C).int() for P in Ps]).cuda()
    hi = (lo + step).int()
    vt = torch.zeros(nkv * hd, B * s_cap, dtype=BF16, device="cuda:0")
    for b in range(B):
        vt[:, b * s_cap:(b + 1) * s_cap] = hip_ops.pack_transpose(v[b * s_cap:(b + 1) * s_cap].cuda(), nkv, nkv, hd)
    M, K = B * G, nh * hd
    assert hip_ops.L.raw("tr1_gemm_oproj_frag_ok")(M, N, K)
    plan = hip_ops.attn_plan(G, nh, nkv, B)
    q0 = rnd(M, K, seed=3).cuda()
    hip_ops.attn_fwd(q0, k, vt, pre, lo, hi, nh, nkv, s_cap, hd, hd ** -0.5, nsplit=nsplit, need_lse=False, n_batch=B, kv_batch_slots=s_cap, plan=plan, plan_mode=1)
    w = rnd(N, K, seed=4, scale=1.0 / math.sqrt(K)).cuda()
    for rep in range(3):
        q = rnd(M, K, seed=10 + rep).cuda()
        res = rnd(M, N, seed=40 + rep).cuda() if rep % 2 == 0 else None
        o_ref, _ = hip_ops.attn_fwd(q, k, vt, pre, lo, hi, nh, nkv, s_cap, hd, hd ** -0.5, nsplit=nsplit, need_lse=False, n_batch=B, kv_batch_slots=s_cap,
                                    plan=plan, plan_mode=2)
        of = hip_ops.attn_fwd_frag(q, k, vt, pre, lo, hi, nh, nkv, s_cap, hd, hd ** -0.5, nsplit, n_batch=B, kv_batch_slots=s_cap, plan=plan, plan_mode=2)
        Mp = (M + 15) // 16 * 16
        back = of.view(Mp // 16, K // 32, 16, 32).permute(0, 2, 1, 3).reshape(Mp, K)[:M]
        assert torch.equal(back, o_ref), "fragment-major rows differ from the row-major merge (rep %d)" % rep
        c = hip_ops.gemm_oproj_frag(of, w, M, residual=res)
        ref = o_ref.double() @ w.double().t() + (res.double() if res is not None else 0.0)
        close(c, ref.float().cpu(), 0.02, rtol=0.02, what="o projection on fragment-major rows, rep %d" % rep)


@pytest.mark.parametrize("M,H,I", [(16, 1536, 8960), (9, 1536, 8960), (16, 2048, 5632), (1, 1536, 8960)])
def test_decode_down_projection_on_the_fragment_major_swiglu_output(hip_ops, M, H, I):
    """Round 6 (Qwen2-VL-2B shapes): tr1_norm_gemm_skinny(glu = 2) writes silu(gate) * up fragment-major ([I / 32][16 rows][32 columns]) and the down projection runs on
    csrc/oproj.hip's long-K form (<= 8 columns per block, the block's weight slice over the WHOLE K in flight, no split-K fixup).  The fragment-major values are the bits of the
    row-major launch; the projection agrees with a float64 product and with the split-K + fixup kernel it replaces (another summation order: bf16 rounding apart)."""
    x = rnd(M, H, seed=1).cuda()
    lnw = (1.0 + 0.1 * rnd(H, seed=2).float()).to(BF16).cuda()
    wgu = rnd(2 * I, H, seed=3, scale=1.0 / math.sqrt(H)).cuda()
    wd = rnd(H, I, seed=4, scale=1.0 / math.sqrt(I)).cuda()
    res = rnd(M, H, seed=5).cuda()
    assert hip_ops.L.raw("tr1_gemm_oproj_frag_ok")(M, H, I) and hip_ops.L.raw("tr1_norm_gemm_glu_frag_ok")(M, I, H)
    a_row = hip_ops.norm_gemm(x, lnw, 1e-6, wgu, glu=True)
    for rep in range(3):      # (the first launch of a fresh process once read LDS stages before their DMA had landed: repeat)
        a_frag = hip_ops.norm_gemm(x, lnw, 1e-6, wgu, glu=2)
        back = a_frag.view(I // 32, 16, 32).permute(1, 0, 2).reshape(16, I)[:M]
        assert torch.equal(back, a_row), "fragment-major SwiGLU output differs from the row-major launch (rep %d)" % rep
        c = hip_ops.gemm_oproj_frag(a_frag, wd, M, residual=res)
        ref = a_row.double() @ wd.double().t() + res.double()
        close(c, ref.float().cpu(), 0.02, rtol=0.02, what="down projection on the fragment-major SwiGLU output, rep %d" % rep)
    old = hip_ops.gemm_skinny_fixup(a_row, wd, residual=res)
    close(c, old.cpu(), 0.05, rtol=0.03, what="long-K form vs split-K + fixup form")


def test_attention_decode_plan_fallback_when_the_list_is_too_long_for_a_reader_block(hip_ops, ref_ops):
    """A reader block holds one plan entry per thread: with few splits and many relevant tiles (n_rel > 256 * nsplit) the publishing launch stores
    count -1 and the reading launches take the full path - the same result."""
    G, C, step, nh, nkv, hd, P, nsplit = 4, 64, 63, 4, 1, 128, 40000, 2       # 625 prefix tiles + 4 suffix tiles > 256 * 2
    S = P + G * C
    k = (torch.randn(S, nkv * hd, generator=torch.Generator().manual_seed(1)) * 0.3).to(BF16).cuda()
    vt = (torch.randn(nkv * hd, S, generator=torch.Generator().manual_seed(2)) * 0.3).to(BF16).cuda()
    q = rnd(G, nh * hd, seed=3).cuda()
    pre = torch.full((G,), P, dtype=torch.int32).cuda()
    lo = (P + torch.arange(G) * C).int().cuda()
    hi = (lo + step).int()
    plan = hip_ops.attn_plan(G, nh, nkv, 1)
    o0, _ = hip_ops.attn_fwd(q, k, vt, pre, lo, hi, nh, nkv, S, hd, hd ** -0.5, nsplit=nsplit, need_lse=False)
    o0 = o0.clone()
    o1, _ = hip_ops.attn_fwd(q, k, vt, pre, lo, hi, nh, nkv, S, hd, hd ** -0.5, nsplit=nsplit, need_lse=False, plan=plan, plan_mode=1)
    o1 = o1.clone()
    assert int(plan[1024]) == -1, int(plan[1024])
    o2, _ = hip_ops.attn_fwd(q, k, vt, pre, lo, hi, nh, nkv, S, hd, hd ** -0.5, nsplit=nsplit, need_lse=False, plan=plan, plan_mode=2)
    assert torch.equal(o0, o1)
    close(o2, o0, 0.01, what="plan fallback (count -1): the reading kernel walks the block's tile ranges itself")
    with pytest.raises(RuntimeError):       # a plan needs split-KV
        hip_ops.attn_fwd(q, k, vt, pre, lo, hi, nh, nkv, S, hd, hd ** -0.5, nsplit=1, need_lse=False, plan=plan, plan_mode=1)


@pytest.mark.parametrize("T,H,W,Ho,Wo", [(8, 360, 640, 364, 644), (6, 360, 640, 308, 532), (3, 240, 320, 112, 140), (4, 100, 90, 196, 168)])
def test_video_preprocess_fused(hip_ops, ref_ops, T, H, W, Ho, Wo):
    """Fused resize (bicubic AA) + normalise + patchify vs the CPU chain (F.interpolate antialias -> round/clamp -> patchify)."""
    frames = torch.randint(0, 256, (T, 3, H, W), generator=torch.Generator().manual_seed(T), dtype=torch.uint8)
    out_h, grid_h = hip_ops.video_preprocess(frames.cuda(), (Ho, Wo), 1216)
    out_r, grid_r = ref_ops.video_preprocess(frames, (Ho, Wo), 1216)
    assert tuple(grid_h) == tuple(grid_r) and out_h.shape == out_r.shape
    a, b = out_h.float().cpu(), out_r.float()
    assert float(a[:, 1176:].abs().max()) == 0.0, "K padding must stay zero"
    # one uint8 level = 1/(255*std) ~ 0.0146-0.0150 after normalisation; the direct 2-D sum vs ATen's two-pass filter may round a
    # value sitting on a .5 boundary the other way: allow <= 1 level on < 0.1 % of the pixels, bf16 rounding (<= 0.008) on the rest
    err = (a - b).abs()
    assert float(err.max()) <= 0.0151 + 0.008
    assert float((err > 0.009).float().mean()) < 1e-3


def test_video_preprocess_matches_hf_processor_golden(hip_ops):
    """The fused HIP preprocessing kernel (target size == source size: the resize taps collapse to identity) against pixel_values_videos
    captured from transformers' Qwen2VLVideoProcessor (tests/golden/patchify_hf.pt): layout, odd-frame padding, constants; bf16 output."""
    import os
    fx = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "patchify_hf.pt"), weights_only=False)
    for c in fx["cases"]:
        frames = torch.randint(0, 256, tuple(c["shape"]), generator=torch.Generator().manual_seed(c["seed"]), dtype=torch.uint8)
        out, grid = hip_ops.video_preprocess(frames.cuda(), tuple(c["shape"][2:]), 1216)
        assert list(grid) == list(c["video_grid_thw"])
        a, want = out.float().cpu(), c["pixel_values_videos"]
        assert a.shape == (want.shape[0], 1216) and float(a[:, 1176:].abs().max()) == 0.0
        assert float((a[:, :1176] - want).abs().max()) <= 0.008 + 1e-5, float((a[:, :1176] - want).abs().max())      # bf16 rounding of |x| < 2.3


@pytest.mark.parametrize("M,N,K,glu", [(8, 512, 256, False), (16, 4608, 3584, False), (16, 100032, 256, False), (32, 100096, 320, False), (5, 72, 320, False), (16, 1024, 3584, True), (13, 200, 512, True),
                                       (32, 4608, 3584, False), (24, 1024, 1536, True), (64, 512, 3584, False), (40, 136, 832, True),
                                       # LDS-streamed GLU kernel (M <= 16, K = 3584 / 1536): full gate/up width, fewer pairs than CUs, ragged row counts
                                       (5, 18944, 3584, True), (1, 48, 3584, True), (16, 8960, 1536, True), (9, 4112, 1536, True), (16, 11008, 2048, True),
                                       # 17 .. 32 rows: the two-fragment LDS-streamed form (config 4: two prompts x G = 16 rows per decode step)
                                       (32, 18944, 3584, True), (17, 4112, 3584, True), (20, 11008, 2048, True), (29, 48, 1536, True),
                                       # round 3: wide plain projection (the lm_head) through the LDS stream - column pairs (n, n + N/2), with bias, ragged rows, both vocabularies
                                       (16, 152064, 3584, False), (7, 151936, 1536, False), (1, 65536, 2048, False), (32, 152064, 3584, False), (19, 65536, 1536, False)])
def test_norm_gemm_fused(hip_ops, ref_ops, M, N, K, glu):
    """rmsnorm folded into the decode GEMM (and SwiGLU into its epilogue) vs the unfused oracle composition."""
    x, lnw = rnd(M, K, seed=1, scale=2.0), (1.0 + 0.1 * rnd(K, seed=2).float()).to(BF16)
    # GLU: products of two O(sqrt(K)*scale) values amplify the bf16 rounding of either factor - keep gate/up at O(1) like real activations
    w = rnd(2 * N if glu else N, K, seed=3, scale=1.5 / math.sqrt(K) if glu else 0.1)
    bias = None if glu else rnd(N, seed=4)
    h = hip_ops.norm_gemm(x.cuda(), lnw.cuda(), 1e-6, w.cuda(), bias=None if bias is None else bias.cuda(), glu=glu)
    r = ref_ops.norm_gemm(x.float(), lnw.float(), 1e-6, w.float(), bias=None if bias is None else bias.float(), glu=glu)
    close(h, r, 0.02 * math.sqrt(K) * 0.1 + 0.03, rtol=0.02, what="norm_gemm glu=%s" % glu)
    # and against the unfused HIP path (same rounding points except the norm scale order)
    xn, _, _ = hip_ops.rmsnorm_fwd(x.cuda(), lnw.cuda(), 1e-6, need_rstd=False)
    y = hip_ops.gemm_nt(xn, w.cuda(), bias=None if bias is None else bias.cuda())
    if glu:
        y = hip_ops.swiglu_fwd(y)
    close(h, y.float().cpu(), 0.02 * math.sqrt(K) * 0.1 + 0.03, rtol=0.02, what="norm_gemm vs unfused HIP")


@pytest.mark.parametrize("M,N,K", [(16, 3584, 18944), (16, 3584, 3584), (8, 1536, 8960), (5, 200, 2048), (32, 3584, 18944), (24, 1536, 1536), (64, 512, 4096),
                                   (16, 128, 256), (3, 128, 18944), (1, 64, 8192)])
def test_gemm_skinny_fixup(hip_ops, M, N, K):
    """Cross-block split-K with in-kernel fixup == the single-pass skinny GEMM (fp32 sums in a different order: bf16-ulp tolerance),
    launched repeatedly to exercise the self re-arming ticket counters."""
    a, b = rnd(M, K, seed=1).cuda(), rnd(N, K, seed=2, scale=1.0 / math.sqrt(K)).cuda()
    bias, res = rnd(N, seed=3).cuda(), rnd(M, N, seed=4).cuda()
    want = hip_ops.gemm_nt(a, b, bias=bias, residual=res).float()
    ref = a.float() @ b.float().t() + bias.float() + res.float()
    for rep in range(40):      # fresh activations every launch: a stale partial tile from the previous launch would show up as an error
        a = rnd(M, K, seed=100 + rep).cuda()
        want = hip_ops.gemm_nt(a, b, bias=bias, residual=res).float()
        ref = a.float() @ b.float().t() + bias.float() + res.float()
        got = hip_ops.gemm_skinny_fixup(a, b, bias=bias, residual=res).float()
        assert (got - ref).abs().max() <= 0.02 + 0.01 * ref.abs().max(), rep
        assert (got - want).abs().max() <= 2.0 ** -7 * max(1.0, float(want.abs().max())), rep
    plain = hip_ops.gemm_skinny_fixup(a, b).float()
    assert (plain - a.float() @ b.float().t()).abs().max() <= 0.02 + 0.01 * ref.abs().max()


@pytest.mark.parametrize("N,K", [(64, 128), (200, 3584), (37, 18944), (16, 256)])
def test_quantize_fp8_rows_bit_exact(hip_ops, ref_ops, N, K):
    """Row-wise e4m3 quantiser: codes and scales equal torch's float8_e4m3fn cast of the same fp32 products (integer/byte work: bit exact)."""
    w = rnd(N, K, seed=11, scale=0.05)
    w[3 % N] = 0                                    # an all-zero row: scale 1, codes 0
    w[5 % N, 7] = 3.0                               # an outlier that sets the row scale
    q, sc = hip_ops.quantize_fp8_rows(w.cuda())
    rq, rsc = ref_ops.quantize_fp8_rows(w.float())
    assert torch.equal(sc.cpu(), rsc)
    assert torch.equal(q.cpu(), rq)


@pytest.mark.parametrize("M,N,K,mode", [(16, 4608, 3584, "norm"), (8, 512, 256, "plain"), (16, 3584, 18944, "res"), (16, 1024, 3584, "glu"),
                                        (32, 4608, 3584, "norm"), (24, 512, 1536, "glu"), (64, 512, 3584, "res"), (5, 72, 384, "plain"), (40, 136, 1024, "glu"),
                                        # full-width 7B decode shapes: lm_head over V = 152064 and the gate/up projection
                                        (16, 152064, 3584, "norm"), (16, 18944, 3584, "glu"),
                                        # LDS-streamed fp8 gate/up (a8, M <= 16, hidden 3584 / 2048 / 1536): ragged rows, fewer pairs than CUs
                                        (5, 18944, 3584, "glu"), (7, 3584, 18944, "res"), (16, 8960, 1536, "glu"), (9, 4112, 1536, "glu"), (16, 11008, 2048, "glu"), (1, 48, 3584, "glu")])
@pytest.mark.parametrize("a8", [False, True])
def test_gemm_w8(hip_ops, ref_ops, M, N, K, mode, a8):
    """fp8-weight decode GEMM vs the oracle on the SAME quantised weights.  a8=False: register dequantisation + bf16 MFMA (W8A16);
    a8=True: fp8 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4) with the activations block-quantised to e4m3 in the operand load (W8A8) -
    the oracle applies the same per-row / per-32-k power-of-two scaling and e4m3 rounding, so only the accumulation order differs."""
    glu = mode == "glu"
    x = rnd(M, K, seed=1, scale=2.0 if mode in ("norm", "glu") else 1.0)
    w = rnd(2 * N if glu else N, K, seed=3, scale=1.5 / math.sqrt(K) if glu else 0.1)
    lnw = (1.0 + 0.1 * rnd(K, seed=2).float()).to(BF16) if mode in ("norm", "glu") else None
    bias = rnd(N, seed=4) if mode in ("norm", "plain") else None
    res = rnd(M, N, seed=5) if mode == "res" else None
    q, sc = hip_ops.quantize_fp8_rows(w.cuda())
    c = lambda t: None if t is None else t.cuda()
    f = lambda t: None if t is None else t.float()
    h = hip_ops.gemm_w8(x.cuda(), q, sc, lnw=c(lnw), eps=1e-6, bias=c(bias), residual=c(res), glu=glu, a8=a8)
    r = ref_ops.gemm_w8(x.float(), q.cpu(), sc.cpu(), lnw=f(lnw), eps=1e-6, bias=f(bias), residual=f(res), glu=glu, a8=a8)
    close(h, r, 0.02 * math.sqrt(K) * 0.1 + 0.03, rtol=0.02, what="gemm_w8 %s a8=%s" % (mode, a8))
    # the quantisation itself: within the e4m3 step of the bf16 GEMM (3 mantissa bits -> ~3 % rms per weight / activation, averaged over K)
    if mode == "plain":
        full = x.float() @ w.float().t() + bias.float()
        assert (h.float().cpu() - full).norm() / full.norm() < (0.07 if a8 else 0.05)


@pytest.mark.parametrize("R,nh,nkv,hd,K", [(16, 28, 4, 128, 3584), (8, 4, 2, 32, 128), (32, 12, 2, 128, 1536), (5, 4, 1, 64, 256), (64, 4, 2, 32, 128)])
def test_norm_gemm_qkv_fused(hip_ops, ref_ops, R, nh, nkv, hd, K):
    """rmsnorm -> qkv projection -> M-RoPE -> KV append in one launch == the two-kernel path (norm_gemm + decode_qkv_post), bit for bit,
    and == the oracle composition within bf16 tolerance."""
    N = (nh + 2 * nkv) * hd
    x, lnw = rnd(R, K, seed=1, scale=2.0), (1.0 + 0.1 * rnd(K, seed=2).float()).to(BF16)
    w, b = rnd(N, K, seed=3, scale=1.0 / math.sqrt(K)), rnd(N, seed=4)
    S = 96
    pos = torch.randint(0, 500, (3, R), generator=torch.Generator().manual_seed(5)).int()
    sec = {128: (16, 24, 24), 64: (8, 12, 12), 32: (4, 6, 6)}[hd]
    cos, sin = hip_ops.mrope_table(pos.cuda(), hd, sec, 1e6)
    slots = torch.randperm(S, generator=torch.Generator().manual_seed(6))[:R].int()
    kc0, vt0 = rnd(S, nkv * hd, seed=7), rnd(nkv * hd, S, seed=8)
    outs = []
    for fused in (True, False):
        kc, vt = kc0.clone().cuda(), vt0.clone().cuda()
        if fused:
            q = hip_ops.norm_gemm_qkv(x.cuda(), lnw.cuda(), 1e-6, w.cuda(), b.cuda(), cos, sin, kc, vt, slots.cuda(), nh, nkv, hd)
        else:
            qkv = hip_ops.norm_gemm(x.cuda(), lnw.cuda(), 1e-6, w.cuda(), bias=b.cuda())
            q = hip_ops.decode_qkv_post(qkv, cos, sin, kc, vt, slots.cuda(), nh, nkv, hd)
        outs.append((q.cpu(), kc.cpu(), vt.cpu()))
    for a, c in zip(outs[0], outs[1]):
        assert torch.equal(a, c)
    kc, vt = kc0.clone().float(), vt0.clone().float()
    rq = ref_ops.norm_gemm_qkv(x.float(), lnw.float(), 1e-6, w.float(), b.float(), cos.cpu(), sin.cpu(), kc, vt, slots, nh, nkv, hd)
    close(outs[0][0], rq, 0.05, what="fused qkv: q")
    close(outs[0][1], kc, 0.05, what="fused qkv: K cache")
    close(outs[0][2], vt, 0.05, what="fused qkv: V^T cache")


@pytest.mark.parametrize("M,N,K", [(512, 256, 64), (1000, 520, 192), (5074, 3584, 4608), (1600, 3584, 18944), (700, 264, 1024), (2049, 1288, 320)])
def test_gemm_nn(hip_ops, M, N, K):
    """K-major B operand (dgrad dX = dY @ W reads the weight as stored): transposing LDS reads vs a float64 product, and bit-equal to the
    NT kernel fed with the transposed copy (same tiles, same k order inside every MFMA, same accumulation order)."""
    a, b = rnd(M, K, seed=1).cuda(), rnd(K, N, seed=2, scale=1.0 / math.sqrt(K)).cuda()
    got = hip_ops.gemm_nn(a, b)
    ref = (a.double() @ b.double()).float()
    close(got, ref.cpu(), 0.03, rtol=0.02, what="gemm_nn")
    nt = hip_ops.gemm_nt(a, b.t().contiguous())
    assert torch.equal(got, nt), "NN and NT forms accumulate in the same order"


@pytest.mark.parametrize("shape", [(1024, 768, 640), (1000, 776, 704), (552, 264, 64)])
@pytest.mark.parametrize("kmajor", [False, True])
def test_weight_gradient_epilogue_also_writes_the_bf16_wire_copy(hip_ops, kmajor, shape):
    """Round 5 (data-parallel per-rank tax): tr1_wgrad_f32_sumsq can also leave bf16(final gradient) in the gradient exchange's staging arena, so
    GradSync / ShardSync skip their 6-byte-per-parameter staging pass for the large matrices.  The wire copy equals gw.to(bf16) bit for bit (overwrite and
    accumulate, NT and K-major operand forms), gw and the sums of squares are unchanged."""
    N, K, T = shape       # (the second and third shapes end in partial tiles in both directions: 224 / 256-row x 256-column blocks)
    dy = rnd(T, N, seed=1, scale=0.1).cuda()
    x = rnd(T, K, seed=2, scale=0.5).cuda()
    dyt = hip_ops.transpose(dy)
    b = x if kmajor else hip_ops.transpose(x)
    part = torch.zeros(1 << 16, dtype=torch.float32, device="cuda")
    for acc in (False, True):
        gw0 = (rnd(N, K, seed=3).float().cuda() if acc else torch.full((N, K), 7.0, device="cuda"))
        gw_a, gw_b = gw0.clone(), gw0.clone()
        wire = torch.full((N, K), 3.0, dtype=BF16, device="cuda")
        n1 = hip_ops.wgrad_sumsq(dyt, b, gw_a, acc, part, 0, b_kmajor=kmajor, b_rows=T)
        s1 = part[:n1].clone()
        n2 = hip_ops.wgrad_sumsq(dyt, b, gw_b, acc, part, 0, b_kmajor=kmajor, b_rows=T, wire=wire)
        assert n1 == n2 and n1 > 0 and torch.equal(gw_a, gw_b) and torch.equal(s1, part[:n2])
        assert torch.equal(wire, gw_b.to(BF16)), "the wire copy must be the bf16 rounding of the stored gradient"
        want = (gw0.double() if acc else 0.0) + dy.double().t() @ x.double()
        close(gw_b.cpu(), want.float().cpu(), 0.02, rtol=0.02, what="wgrad with wire copy")
