// FP8 (OCP e4m3) weight storage for the rollout's decode GEMMs (BASELINE config "fp8 weights"): the decode step is bound by streaming the
// weights from HBM, so halving their bytes is worth more than any arithmetic.  Weights are quantised per output row
// (scale[n] = amax_n / 448) once per optimizer step; the decode kernels read 16 fp8 per lane (one 16-byte load), convert them to bf16 in
// registers (v_cvt_pk_f32_fp8 + v_cvt_pk_bf16_f32, ~20 % of the VALU at the HBM-bound rate) and run the same bf16 MFMA tiles against bf16
// activations (W8A16): activations keep their precision, the row scale is applied to the fp32 accumulator in the epilogue.
// Only the SAMPLING policy is quantised - log-probs, KL and the update use the bf16 weights.
//
// Kernel structure = gemm_skinny_kernel / norm_gemm_skinny_kernel (gemm.hip) with a 128-element k-step:
//   lane (u, g) of a 16-row weight fragment loads W[row u][k0 + h*64 + g*16 .. +16] (h = 0, 1) and splits it into two bf16x8 MFMA operands
//   (bytes 0-7, bytes 8-15); the activation fragments are loaded from the same k positions, so any k permutation cancels.
#include "tr1_common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(2))) float f32x2_t;

TR1_DEV void fp8x16_to_bf16(u32x4_t q, bf16x8_t& lo, bf16x8_t& hi) {
    u32x4_t a, b;
#pragma unroll
    for (int w = 0; w < 2; ++w) {        // dwords 0,1 -> lo (k 0..7); dwords 2,3 -> hi (k 8..15)
        const f32x2_t p0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)q[w], false), p1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)q[w], true);
        a[2 * w] = pack2bf(p0[0], p0[1]); a[2 * w + 1] = pack2bf(p1[0], p1[1]);
        const f32x2_t r0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)q[2 + w], false), r1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)q[2 + w], true);
        b[2 * w] = pack2bf(r0[0], r0[1]); b[2 * w + 1] = pack2bf(r1[0], r1[1]);
    }
    lo = __builtin_bit_cast(bf16x8_t, a); hi = __builtin_bit_cast(bf16x8_t, b);
}

TR1_DEV float silu_w8(float x) { return x / (1.f + __expf(-x)); }
TR1_DEV bf16x8_t scale_frag_sumsq_w8(bf16x8_t x, bf16x8_t w, float& ss) {
    const u32x4_t xu = __builtin_bit_cast(u32x4_t, x), wu = __builtin_bit_cast(u32x4_t, w);
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a = bflo(xu[e]), b = bfhi(xu[e]);
        ss = fmaf(a, a, fmaf(b, b, ss));
        o[e] = pack2bf(a * bflo(wu[e]), b * bfhi(wu[e]));
    }
    return __builtin_bit_cast(bf16x8_t, o);
}

// out[M, N] = act(x)[M, K] * dequant(W)[N, K]^T (* wscale[n]) (+ bias) (+ residual);  NORM: act = rmsnorm(.; lnw) folded in;  GLU: W holds
// gate rows then up rows (up_off apart) and out = silu(gate) * up.
template <int WAVES, int UNROLL, int NCOL, int MG, bool NORM, bool GLU>
__global__ __launch_bounds__(WAVES * 64) void gemm_skinny_w8_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ lnw,
                                                                    const unsigned char* __restrict__ W, const float* __restrict__ wscale,
                                                                    bf16_t* __restrict__ C, const bf16_t* __restrict__ bias,
                                                                    const bf16_t* __restrict__ residual, int M, int64_t N, int64_t K, int64_t ldx,
                                                                    int64_t ldw, int64_t ldc, int64_t ldr, float eps, int64_t up_off) {
    static_assert(!GLU || NCOL % 2 == 0, "GLU: NCOL/2 gate column groups + the matching NCOL/2 up groups");
    constexpr int NOUT = GLU ? NCOL / 2 : NCOL;          // output column groups per block
    __shared__ __attribute__((aligned(16))) float red[WAVES][NCOL][MG][16][17];
    __shared__ float ssred[WAVES][MG][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u = lane & 15, g = lane >> 4;
    const int64_t n0 = (int64_t)blockIdx.x * 16 * NOUT;
    const unsigned char* wp[NCOL];
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
        int64_t wrow = n0 + (c % NOUT) * 16 + u;
        if (wrow >= N) wrow = N - 1;
        if (GLU && c >= NOUT) wrow += up_off;
        wp[c] = W + wrow * ldw + g * 16;
    }
    const bf16_t* xp[MG];      // rows >= M re-read row M-1; their outputs are never stored
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) xp[mg] = X + (int64_t)(mg * 16 + u < M ? mg * 16 + u : (M - 1)) * ldx + g * 16;
    const bf16_t* lp = NORM ? lnw + g * 16 : nullptr;
    const int64_t nsteps = K / 128;
    const int64_t s_per = (nsteps + WAVES - 1) / WAVES;
    const int64_t s0 = wave * s_per;
    int64_t s1 = s0 + s_per; if (s1 > nsteps) s1 = nsteps;
    f32x4_t acc[NCOL][MG];
    float ss[MG];
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) {
        ss[mg] = 0.f;
#pragma unroll
        for (int c = 0; c < NCOL; ++c) acc[c][mg] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    for (int64_t s = s0; s < s1; s += UNROLL) {
        u32x4_t wq[UNROLL][NCOL][2];
        bf16x8_t xa[UNROLL][MG][2][2], la[UNROLL][2][2];
#pragma unroll
        for (int q = 0; q < UNROLL; ++q) {
            const int64_t st = s + q < s1 ? s + q : s1 - 1;          // surplus buffers of the last trip re-read the final step (unused)
            const int64_t k = st * 128;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int c = 0; c < NCOL; ++c) wq[q][c][h] = *reinterpret_cast<const u32x4_t*>(wp[c] + k + h * 64);
#pragma unroll
                for (int mg = 0; mg < MG; ++mg) {
                    xa[q][mg][h][0] = *reinterpret_cast<const bf16x8_t*>(xp[mg] + k + h * 64);
                    xa[q][mg][h][1] = *reinterpret_cast<const bf16x8_t*>(xp[mg] + k + h * 64 + 8);
                }
                if (NORM) {
                    la[q][h][0] = *reinterpret_cast<const bf16x8_t*>(lp + k + h * 64);
                    la[q][h][1] = *reinterpret_cast<const bf16x8_t*>(lp + k + h * 64 + 8);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < UNROLL; ++q) {
            if (s + q < s1) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    bf16x8_t xf[MG][2];
#pragma unroll
                    for (int mg = 0; mg < MG; ++mg)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            xf[mg][j] = NORM ? scale_frag_sumsq_w8(xa[q][mg][h][j], la[q][h][j], ss[mg]) : xa[q][mg][h][j];
#pragma unroll
                    for (int c = 0; c < NCOL; ++c) {
                        bf16x8_t w0, w1;
                        fp8x16_to_bf16(wq[q][c][h], w0, w1);
#pragma unroll
                        for (int mg = 0; mg < MG; ++mg) {
                            acc[c][mg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, xf[mg][0], acc[c][mg], 0, 0, 0);
                            acc[c][mg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, xf[mg][1], acc[c][mg], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) {
        if (NORM) {
            float v = ss[mg];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (g == 0) ssred[wave][mg][u] = v;
        }
#pragma unroll
        for (int c = 0; c < NCOL; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][c][mg][u][g * 4 + r] = acc[c][mg][r];
    }
    __syncthreads();
    const float inv_k = 1.f / (float)K;
    for (int i = threadIdx.x; i < NOUT * MG * 256; i += WAVES * 64) {   // (output column group, row group, m, n)
        const int c = i / (MG * 256), mg = (i >> 8) % MG, mm = (i >> 4) & 15, nn = i & 15;
        const int m = mg * 16 + mm;
        const int64_t n = n0 + c * 16 + nn;
        if (m < M && n < N) {
            float sq = 0.f, v = 0.f, v2 = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                if (NORM) sq += ssred[w][mg][mm];
                v += red[w][c][mg][mm][nn];
                if (GLU) v2 += red[w][NOUT + c][mg][mm][nn];
            }
            const float rstd = NORM ? rsqrtf(sq * inv_k + eps) : 1.f;
            v *= rstd * wscale[n];
            if (GLU) {
                const float gt = bf2f(f2bf(v)), up = bf2f(f2bf(v2 * rstd * wscale[up_off + n]));
                C[(int64_t)m * ldc + n] = f2bf(bf2f(f2bf(silu_w8(gt))) * up);
            } else {
                if (bias) v += bf2f(bias[n]);
                if (residual) v += bf2f(residual[(int64_t)m * ldr + n]);
                C[(int64_t)m * ldc + n] = f2bf(v);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- W8A8: fp8 MFMA
// BASELINE config 5 ("CDNA4 fp8 MFMA"): the same weight stream, but the fp8 codes go into the matrix core AS fp8 - no conversion VALU on
// the weight path at all - through v_mfma_scale_f32_16x16x128_f8f6f4 (one instruction per 16 x 16 x 128 step; block-scaled "MX" form).
// The activations are quantised on the fly, per row and per 32 consecutive k, to e4m3 with a power-of-two (E8M0) block scale - the
// OCP microscaling recipe - so a block's range follows its own values and no row-wide amax pass over x is needed; the weight keeps its
// per-output-row fp32 scale (applied to the accumulator in the epilogue) and enters the MFMA with block scale 1.
// Operand layout of the instruction, probed on MI355X (tools/probe_mfma_f8.hip, tools/probe_mfma_f8_scale.hip): lane (u = lane % 16,
// g = lane / 16) supplies row/column u; byte t of its 32 operand bytes is logical k = (t / 16) * 64 + g * 16 + t % 16 - i.e. two 16-byte
// halves, exactly the two loads `k + h*64 + g*16` the W8A16 kernel already issues; the E8M0 scale of logical block b (k in [32b, 32b+32))
// is read from byte 0 of the scale VGPR of lane (u, b).  Block b = half b/2 of the lane pair g in {2(b%2), 2(b%2)+1}.
typedef __attribute__((ext_vector_type(8))) int i32x8_t;

// 16 floats -> 16 e4m3 codes (4 dwords), v_cvt_pk_fp8_f32 rounds to nearest even
TR1_DEV void quant16_fp8(const float (&v)[16], float mul, int (&o)[4]) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * d] * mul, v[4 * d + 1] * mul, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * d + 2] * mul, v[4 * d + 3] * mul, w, true);
        o[d] = w;
    }
}

// XLDS (round 3, MG = 1): the bf16 activation rows and the norm weight - 8 of the 12 vector loads of a k-step at NCOL = 2, all in the MFMA operand
// layout that costs 64 L1 tag look-ups per KiB (see gemm.hip) - reach the lanes through ONE DMA copy into LDS per block: x as [K/128 segments][16 rows]
// [256 bytes] with chunk c of row r at c ^ r, lnw as it is.  Only the fp8 weights stay on the vector-memory path.  Same values, same order.
template <int WAVES, int UNROLL, int NCOL, int MG, bool NORM, bool GLU, bool XLDS = false>
__global__ __launch_bounds__(WAVES * 64) void gemm_skinny_w8a8_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ lnw,
                                                                      const unsigned char* __restrict__ W, const float* __restrict__ wscale,
                                                                      bf16_t* __restrict__ C, const bf16_t* __restrict__ bias,
                                                                      const bf16_t* __restrict__ residual, int M, int64_t N, int64_t K, int64_t ldx,
                                                                      int64_t ldw, int64_t ldc, int64_t ldr, float eps, int64_t up_off) {
    static_assert(!GLU || NCOL % 2 == 0, "GLU: NCOL/2 gate column groups + the matching NCOL/2 up groups");
    constexpr int NOUT = GLU ? NCOL / 2 : NCOL;
    __shared__ __attribute__((aligned(16))) float red[WAVES][NCOL][MG][16][17];
    __shared__ float ssred[WAVES][MG][16];
    extern __shared__ __attribute__((aligned(1024))) char w8_xs[];             // XLDS: x image (K * 32 bytes), then lnw (K * 2 bytes)
    static_assert(!XLDS || MG == 1, "the LDS copy of x holds 16 rows");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u = lane & 15, g = lane >> 4;
    const int64_t n0 = (int64_t)blockIdx.x * 16 * NOUT;
    const unsigned char* wp[NCOL];
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
        int64_t wrow = n0 + (c % NOUT) * 16 + u;
        if (wrow >= N) wrow = N - 1;
        if (GLU && c >= NOUT) wrow += up_off;
        wp[c] = W + wrow * ldw + g * 16;
    }
    const bf16_t* xp[MG];
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) xp[mg] = X + (int64_t)(mg * 16 + u < M ? mg * 16 + u : (M - 1)) * ldx + g * 16;
    const bf16_t* lp = NORM ? lnw + g * 16 : nullptr;
    const int64_t nsteps = K / 128;
    const int64_t s_per = (nsteps + WAVES - 1) / WAVES;
    const int64_t s0 = wave * s_per;
    int64_t s1 = s0 + s_per; if (s1 > nsteps) s1 = nsteps;
    f32x4_t acc[NCOL][MG];
    float ss[MG];
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) {
        ss[mg] = 0.f;
#pragma unroll
        for (int c = 0; c < NCOL; ++c) acc[c][mg] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    // the lane that holds the other 16 values of my blocks, and the lane whose block exponents I have to present to the MFMA
    const int partner = lane ^ 16;
    const int src_lane = u + 16 * (2 * (g & 1));             // first lane of the pair that holds block g (its half g / 2)
    const unsigned xs_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)w8_xs;
    const unsigned ls_base = xs_base + (unsigned)(K * 32);
    if (XLDS) {
        const int n_x = (int)(K >> 5), n_l = NORM ? (int)((K * 2 + 1023) >> 10) : 0;      // 1 KiB per instruction: 4 rows x 256 bytes of x / 512 columns of lnw
        const int w0 = __builtin_amdgcn_readfirstlane(wave);
        for (int i = w0; i < n_x; i += WAVES) {
            const int r = (i & 3) * 4 + (lane >> 4);
            const unsigned off = (unsigned)((r < M ? r : M - 1) * (int)ldx + (i >> 2) * 128 + (((lane & 15) ^ r) << 3)) * 2u;
            const unsigned dst = xs_base + (unsigned)i * 1024u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(off), "s"(X) : "memory", "m0");
        }
        for (int i = w0; i < n_l; i += WAVES) {
            int64_t col = (int64_t)i * 512 + lane * 8;
            if (col + 8 > K) col = K - 8;
            const unsigned off = (unsigned)col * 2u, dst = ls_base + (unsigned)i * 1024u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(off), "s"(lnw) : "memory", "m0");
        }
    }
    // XLDS: the weight loads of a trip are requested one trip ahead (wn), the first ones BEFORE the wait for the x copy - only they may still be in
    // flight when the copy has landed (counted vmcnt), so the weight stream starts at kernel entry
    u32x4_t wn[UNROLL][NCOL][2];
#define W8_LOADW(dst, s_)                                                                                         \
    do { _Pragma("unroll") for (int q = 0; q < UNROLL; ++q) {                                                     \
        const int64_t k__ = ((s_) + q < s1 ? (s_) + q : s1 - 1) * 128;                                            \
        _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                             \
        _Pragma("unroll") for (int c = 0; c < NCOL; ++c) dst[q][c][h] = *reinterpret_cast<const u32x4_t*>(wp[c] + k__ + h * 64); } } while (0)
    if (XLDS) {
        static_assert(!XLDS || (UNROLL * NCOL * 2 == 4 || UNROLL * NCOL * 2 == 8), "counted wait below");
        if (s0 < s1) {
            W8_LOADW(wn, s0);
            if (UNROLL * NCOL * 2 == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    for (int64_t s = s0; s < s1; s += UNROLL) {
        u32x4_t wq[UNROLL][NCOL][2];
        u32x4_t xa[UNROLL][MG][2][2], la[UNROLL][2][2];
        if (XLDS) {
#pragma unroll
            for (int q = 0; q < UNROLL; ++q)
#pragma unroll
                for (int c = 0; c < NCOL; ++c) { wq[q][c][0] = wn[q][c][0]; wq[q][c][1] = wn[q][c][1]; }
            if (s + UNROLL < s1) W8_LOADW(wn, s + UNROLL);
        }
#pragma unroll
        for (int q = 0; q < UNROLL; ++q) {
            const int64_t st = s + q < s1 ? s + q : s1 - 1;
            const int64_t k = st * 128;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (!XLDS) {
#pragma unroll
                for (int c = 0; c < NCOL; ++c) wq[q][c][h] = *reinterpret_cast<const u32x4_t*>(wp[c] + k + h * 64);
                }
                if (XLDS) {
                    typedef const __attribute__((address_space(3))) u32x4_t* xs_ptr_t;
                    const unsigned xrow = xs_base + (unsigned)st * 4096u + (unsigned)u * 256u;
                    xa[q][0][h][0] = *(xs_ptr_t)(uintptr_t)(xrow + (unsigned)(((h * 8 + g * 2) ^ u) << 4));
                    xa[q][0][h][1] = *(xs_ptr_t)(uintptr_t)(xrow + (unsigned)(((h * 8 + g * 2 + 1) ^ u) << 4));
                    if (NORM) {
                        const unsigned lrow = ls_base + (unsigned)(k + h * 64 + g * 16) * 2u;
                        la[q][h][0] = *(xs_ptr_t)(uintptr_t)lrow;
                        la[q][h][1] = *(xs_ptr_t)(uintptr_t)(lrow + 16u);
                    }
                } else {
#pragma unroll
                for (int mg = 0; mg < MG; ++mg) {
                    xa[q][mg][h][0] = *reinterpret_cast<const u32x4_t*>(xp[mg] + k + h * 64);
                    xa[q][mg][h][1] = *reinterpret_cast<const u32x4_t*>(xp[mg] + k + h * 64 + 8);
                }
                if (NORM) {
                    la[q][h][0] = *reinterpret_cast<const u32x4_t*>(lp + k + h * 64);
                    la[q][h][1] = *reinterpret_cast<const u32x4_t*>(lp + k + h * 64 + 8);
                }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < UNROLL; ++q) {
            if (s + q < s1) {
                i32x8_t xq[MG];
                int xs[MG];
#pragma unroll
                for (int mg = 0; mg < MG; ++mg) {
                    float v[2][16];
                    float am[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        am[h] = 0.f;
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float a = bflo(xa[q][mg][h][j][e]), b = bfhi(xa[q][mg][h][j][e]);
                                if (NORM) {
                                    ss[mg] = fmaf(a, a, fmaf(b, b, ss[mg]));
                                    a *= bflo(la[q][h][j][e]); b *= bfhi(la[q][h][j][e]);
                                }
                                v[h][j * 8 + 2 * e] = a; v[h][j * 8 + 2 * e + 1] = b;
                                am[h] = fmaxf(am[h], fmaxf(fabsf(a), fabsf(b)));
                            }
                        am[h] = fmaxf(am[h], __shfl(am[h], partner, 64));             // the block = my 16 values + the partner lane's 16
                    }
                    // block exponent: values scaled into [0, 256) (e4m3 max 448): E8M0 byte = biased exponent of amax - 7; blocks below 2^-119 are zero
                    int eb[2]; float mul[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int be = (int)(__float_as_uint(am[h]) >> 23);
                        const bool tiny = be < 8;
                        eb[h] = tiny ? 0 : be - 7;
                        mul[h] = tiny ? 0.f : __uint_as_float((unsigned)(261 - be) << 23);
                    }
                    int o0[4], o1[4];
                    quant16_fp8(v[0], mul[0], o0);
                    quant16_fp8(v[1], mul[1], o1);
                    xq[mg] = (i32x8_t){o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o1[2], o1[3]};
                    const int both = __shfl(eb[0] | (eb[1] << 8), src_lane, 64);
                    xs[mg] = (g >> 1) ? (both >> 8) & 0xff : both & 0xff;
                }
#pragma unroll
                for (int c = 0; c < NCOL; ++c) {
                    const i32x8_t wf = {(int)wq[q][c][0][0], (int)wq[q][c][0][1], (int)wq[q][c][0][2], (int)wq[q][c][0][3],
                                        (int)wq[q][c][1][0], (int)wq[q][c][1][1], (int)wq[q][c][1][2], (int)wq[q][c][1][3]};
#pragma unroll
                    for (int mg = 0; mg < MG; ++mg)
                        acc[c][mg] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf, xq[mg], acc[c][mg], 0, 0, 0, 127, 0, xs[mg]);
                }
            }
        }
    }
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) {
        if (NORM) {
            float v = ss[mg];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (g == 0) ssred[wave][mg][u] = v;
        }
#pragma unroll
        for (int c = 0; c < NCOL; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][c][mg][u][g * 4 + r] = acc[c][mg][r];
    }
    __syncthreads();
    const float inv_k = 1.f / (float)K;
    for (int i = threadIdx.x; i < NOUT * MG * 256; i += WAVES * 64) {
        const int c = i / (MG * 256), mg = (i >> 8) % MG, mm = (i >> 4) & 15, nn = i & 15;
        const int m = mg * 16 + mm;
        const int64_t n = n0 + c * 16 + nn;
        if (m < M && n < N) {
            float sq = 0.f, v = 0.f, v2 = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                if (NORM) sq += ssred[w][mg][mm];
                v += red[w][c][mg][mm][nn];
                if (GLU) v2 += red[w][NOUT + c][mg][mm][nn];
            }
            const float rstd = NORM ? rsqrtf(sq * inv_k + eps) : 1.f;
            v *= rstd * wscale[n];
            if (GLU) {
                const float gt = bf2f(f2bf(v)), up = bf2f(f2bf(v2 * rstd * wscale[up_off + n]));
                C[(int64_t)m * ldc + n] = f2bf(bf2f(f2bf(silu_w8(gt))) * up);
            } else {
                if (bias) v += bf2f(bias[n]);
                if (residual) v += bf2f(residual[(int64_t)m * ldr + n]);
                C[(int64_t)m * ldc + n] = f2bf(v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ W8A8, LDS-streamed gate/up (M <= 16)
// The fp8 twin of norm_glu_lds_kernel (gemm.hip): weights travel HBM -> LDS in full 128-byte row runs (global_load_lds, 8 rows per wave
// instruction) - with 1-byte codes a run is 128 k = exactly one 16x16x128 MFMA step; each of the NW waves owns a K/NW = 512-wide slice
// (4 stages), its own ring of R stages and a counted vmcnt, no barrier in the stream.  A block is persistent over a range of 16-column
// pairs (16 gate rows + 16 up rows): the block-quantised activation fragments x' = e4m3(x * lnw / 2^E) of the wave's k-slice and their
// E8M0 scales are built ONCE and stay in registers (4 x (8 + 1) VGPRs), sum x^2 once per block.  Stage image, swizzle (keyA8 on the DMA's
// SOURCE address) and the two ds_read_b128 per operand are those of the bf16 kernel: chunk h*4 + g of row u = the lane's half h.
typedef const __attribute__((address_space(1))) void* w8_gptr_t;
typedef __attribute__((address_space(3))) void* w8_lptr_t;
TR1_DEV int keyA8(int row) { return (row >> 1) & 7; }
#define W8_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

template <int NW, int R>
__global__ __launch_bounds__(NW * 64) void norm_glu_lds_f8_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ lnw, const unsigned char* __restrict__ W,
                                                                  const float* __restrict__ wscale, bf16_t* __restrict__ C, int M, int64_t N, int64_t K,
                                                                  int64_t ldx, int64_t ldw, int64_t ldc, float eps, int64_t up_off) {
    constexpr int NST = 4, STAGE = 4096, REDW = 2 * 16 * 17;
    extern __shared__ __attribute__((aligned(16))) char glu8_lds[];       // red[2][NW][REDW] f32 | ssq[NW][16] | [NW waves][R stages][4 KiB]
    float* red = reinterpret_cast<float*>(glu8_lds);
    float* ssq = red + 2 * NW * REDW;
    char* rings = glu8_lds + (2 * NW * REDW + NW * 16) * sizeof(float);
    static_assert(((2 * NW * REDW + NW * 16) * sizeof(float)) % 16 == 0 && (2 * NW * REDW + NW * 16) * sizeof(float) >= 3 * 128, "ring base");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, u = lane & 15, g = lane >> 4;
    const int64_t NP = (N + 15) / 16;
    const int64_t p0 = NP * blockIdx.x / gridDim.x, p1 = NP * (blockIdx.x + 1) / gridDim.x;
    const int npair = (int)(p1 - p0);
    const int64_t kb = (int64_t)wave * 512;
    char* ring = rings + wave * R * STAGE;
    const int total = npair * NST;
    const unsigned char* pg[2]; const unsigned char* pu[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 8 * j + (lane >> 3);
        pg[j] = W + (p0 * 16 + r) * ldw + kb + (((lane & 7) ^ keyA8(r)) << 4);
        pu[j] = pg[j] + up_off * ldw;
    }
    const int64_t pair_step = 16 * ldw;
    int islot = 0;
#define G8_ISSUE(ST) do {                                                                                                \
        char* dst__ = ring + islot * STAGE - (ST) * 128;   /* the instruction offset is added to the LDS address as well */  \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                  \
            __builtin_amdgcn_global_load_lds((w8_gptr_t)pg[j], (w8_lptr_t)(dst__ + j * 1024), 16, (ST) * 128, 2);        \
            __builtin_amdgcn_global_load_lds((w8_gptr_t)pu[j], (w8_lptr_t)(dst__ + 2048 + j * 1024), 16, (ST) * 128, 2); \
        }                                                                                                                \
        islot = (islot + 1 == R) ? 0 : islot + 1;                                                                        \
    } while (0)
#define G8_ISSUE_ST(ST) do { switch (ST) { case 0: G8_ISSUE(0); break; case 1: G8_ISSUE(1); break; case 2: G8_ISSUE(2); break; default: G8_ISSUE(3); break; } } while (0)
#define G8_NEXT_PAIR() do { _Pragma("unroll") for (int j = 0; j < 2; ++j) { pg[j] += pair_step; pu[j] += pair_step; } } while (0)
    // ---- block-quantised activation fragments of this wave's k-slice (once per block).  Round 3 (same finding as norm_glu_lds_kernel in gemm.hip:
    // 32 vector loads per lane in the MFMA operand layout = 64 L1 tag look-ups per KiB, in dependent batches, while only R-1 weight stages were in
    // flight): the wave's x slice (16 rows x 512 columns = NST stages of 16 rows x 256 bytes) and its norm-weight slice are copied by DMA - x into the
    // wave's own still empty ring (+ one 4 KiB stage behind the rings when R < NST), lnw into a private KiB - and read back from LDS; the weight
    // stream starts right after.  Same values in the same order.
    static_assert(R + 1 >= NST, "x staging: ring + one extra stage");
    char* const x_extra = rings + NW * R * STAGE + wave * STAGE;
    char* const lnw_lds = rings + NW * R * STAGE + NW * STAGE + wave * 1024;
    {
        int64_t col = kb + lane * 8;
        if (col + 8 > K) col = K - 8;
        __builtin_amdgcn_global_load_lds((w8_gptr_t)(lnw + col), (w8_lptr_t)lnw_lds, 16, 0, 0);
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            char* dst = st < R ? ring + st * STAGE : x_extra;
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {                              // 4 rows x 256 bytes per instruction; row r keeps logical chunk c at c ^ r
                const int r = 4 * i4 + (lane >> 4);
                const bf16_t* src = X + (int64_t)(r < M ? r : M - 1) * ldx + kb + st * 128 + (((lane & 15) ^ r) << 3);
                __builtin_amdgcn_global_load_lds((w8_gptr_t)src, (w8_lptr_t)(dst + i4 * 1024), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    i32x8_t xq[NST]; int xs[NST];
    {
        float ss = 0.f;
        const int partner = lane ^ 16, src_lane = u + 16 * (2 * (g & 1));
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const char* xst = (st < R ? ring + st * STAGE : x_extra) + u * 256;
            float v[2][16], am[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                am[h] = 0.f;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const u32x4_t xv = *reinterpret_cast<const u32x4_t*>(xst + (((h * 8 + g * 2 + j) ^ u) << 4));
                    const u32x4_t lv = *reinterpret_cast<const u32x4_t*>(lnw_lds + (st * 128 + h * 64 + g * 16 + j * 8) * 2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a = bflo(xv[e]), b = bfhi(xv[e]);
                        ss = fmaf(a, a, fmaf(b, b, ss));
                        a *= bflo(lv[e]); b *= bfhi(lv[e]);
                        v[h][j * 8 + 2 * e] = a; v[h][j * 8 + 2 * e + 1] = b;
                        am[h] = fmaxf(am[h], fmaxf(fabsf(a), fabsf(b)));
                    }
                }
                am[h] = fmaxf(am[h], __shfl(am[h], partner, 64));
            }
            int eb[2]; float mul[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int be = (int)(__float_as_uint(am[h]) >> 23);
                const bool tiny = be < 8;
                eb[h] = tiny ? 0 : be - 7;
                mul[h] = tiny ? 0.f : __uint_as_float((unsigned)(261 - be) << 23);
            }
            int o0[4], o1[4];
            quant16_fp8(v[0], mul[0], o0);
            quant16_fp8(v[1], mul[1], o1);
            xq[st] = (i32x8_t){o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o1[2], o1[3]};
            const int both = __shfl(eb[0] | (eb[1] << 8), src_lane, 64);
            xs[st] = (g >> 1) ? (both >> 8) & 0xff : both & 0xff;
        }
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        if (g == 0) ssq[wave * 16 + u] = ss;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // the ring is free again: start the weight stream
#pragma unroll
    for (int i = 0; i < R - 1; ++i) {
        if (i < total) {
            if (i > 0 && i % NST == 0) G8_NEXT_PAIR();
            G8_ISSUE_ST(i % NST);
        }
    }
    W8_BARRIER();
    float rstd;
    {
        float sq = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sq += ssq[w * 16 + ((threadIdx.x >> 4) & 15)];
        rstd = rsqrtf(sq * (1.f / (float)K) + eps);
    }
    const int rd_off = u * 128;
    const int kA = keyA8(u);
    int cslot = 0;
    for (int pi = 0; pi < npair; ++pi) {
        f32x4_t ag = {0.f, 0.f, 0.f, 0.f}, au = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const int item = pi * NST + st;
            if (item + R - 1 < total) {
                if ((st + R - 1) % NST == 0) G8_NEXT_PAIR();
                G8_ISSUE_ST((st + R - 1) % NST);
            }
            const int rem = total - 1 - item;
            if (rem >= R - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (R - 1)) : "memory");
            else if (rem == 2 && R > 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (rem == 1 && R > 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const char* sb = ring + cslot * STAGE + rd_off;
            cslot = (cslot + 1 == R) ? 0 : cslot + 1;
            const u32x4_t g0 = *reinterpret_cast<const u32x4_t*>(sb + ((g ^ kA) << 4)), g1 = *reinterpret_cast<const u32x4_t*>(sb + (((4 + g) ^ kA) << 4));
            const u32x4_t u0 = *reinterpret_cast<const u32x4_t*>(sb + 2048 + ((g ^ kA) << 4)), u1 = *reinterpret_cast<const u32x4_t*>(sb + 2048 + (((4 + g) ^ kA) << 4));
            const i32x8_t wg = {(int)g0[0], (int)g0[1], (int)g0[2], (int)g0[3], (int)g1[0], (int)g1[1], (int)g1[2], (int)g1[3]};
            const i32x8_t wu = {(int)u0[0], (int)u0[1], (int)u0[2], (int)u0[3], (int)u1[0], (int)u1[1], (int)u1[2], (int)u1[3]};
            ag = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wg, xq[st], ag, 0, 0, 0, 127, 0, xs[st]);
            au = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wu, xq[st], au, 0, 0, 0, 127, 0, xs[st]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        float* rw = red + ((pi & 1) * NW + wave) * REDW;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rw[u * 17 + g * 4 + r] = ag[r];
            rw[16 * 17 + u * 17 + g * 4 + r] = au[r];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W8_BARRIER();
        for (int i = threadIdx.x; i < 256; i += NW * 64) {              // 256 outputs of the pair; blocks of 3 waves take two trips
            const int mm = i >> 4, nn = i & 15;
            const float* rb = red + (pi & 1) * NW * REDW;
            float v = 0.f, v2 = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { v += rb[w * REDW + mm * 17 + nn]; v2 += rb[w * REDW + 16 * 17 + mm * 17 + nn]; }
            const int64_t n = (p0 + pi) * 16 + nn;
            if (mm < M && n < N) {
                float rs = rstd;
                if (NW * 64 < 256) { float sq = 0.f; _Pragma("unroll") for (int w = 0; w < NW; ++w) sq += ssq[w * 16 + mm]; rs = rsqrtf(sq * (1.f / (float)K) + eps); }
                v *= rs * wscale[n];
                const float gt = bf2f(f2bf(v)), up = bf2f(f2bf(v2 * rs * wscale[up_off + n]));
                C[(int64_t)mm * ldc + n] = f2bf(bf2f(f2bf(silu_w8(gt))) * up);
            }
        }
    }
#undef G8_ISSUE
#undef G8_ISSUE_ST
#undef G8_NEXT_PAIR
}

// ------------------------------------------------------------------------------- W8A8, LDS-streamed split-K + fixup projection (M <= 16)
// The fp8 twin of gemm_skinny_lds_fix_kernel (gemm.hip; decode down projection: N = 3584 columns, K = 18944): a block owns 64 output columns
// and one of gridDim.y K-slabs; its waves take the slab's 128-wide stages round-robin, each with a two-slot ring: 64 weight rows x 128 bytes
// of fp8 (8 KiB) + the 16 activation rows x 128 k of bf16 (2 x 16 x 128 bytes, quantised per lane pair right before the MFMA).  The partial
// tiles go through the same workspace and ticket protocol (the last-arriving slab sums the tiles in slab order); the weight's row scale is
// applied once, to the summed tile.
template <int WAVES, int NWI = 8>      // NWI = weight DMA instructions (8 rows each) per stage: 8 = 64-column blocks, 7 = 56-column blocks (3584 columns: 64 groups x 4 slabs = every CU)
__global__ __launch_bounds__(WAVES * 64) void gemm_skinny_lds_fix_f8_kernel(const bf16_t* __restrict__ X, const unsigned char* __restrict__ W,
                                                                            const float* __restrict__ wscale, bf16_t* __restrict__ C,
                                                                            const bf16_t* __restrict__ bias, const bf16_t* __restrict__ residual, int M,
                                                                            int64_t N, int64_t K, int64_t ldx, int64_t ldw, int64_t ldc, int64_t ldr,
                                                                            float* __restrict__ fix_ws, int* __restrict__ fix_cnt) {
    constexpr int NC = 4, STAGE = NWI * 1024 + 4096, TILE = NC * 256, COLS = NWI * 8;
    extern __shared__ __attribute__((aligned(16))) char sk8_lds[];         // [WAVES][2][STAGE]; afterwards red[WAVES][NC][16][17] f32; ticket at the end
    int* s_ticket = reinterpret_cast<int*>(sk8_lds + WAVES * 2 * STAGE);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, u = lane & 15, g = lane >> 4;
    const int64_t n0 = (int64_t)blockIdx.x * COLS;
    const int64_t kslab = K / gridDim.y, k0 = (int64_t)blockIdx.y * kslab;
    const int nst = (int)(kslab / 128);
    const int n_my = wave < nst ? (nst - wave + WAVES - 1) / WAVES : 0;
    char* ring = sk8_lds + wave * 2 * STAGE;
    const unsigned char* pw[NWI]; const bf16_t* px[4];
#pragma unroll
    for (int j = 0; j < NWI; ++j) {
        const int r = 8 * j + (lane >> 3);
        int64_t row = n0 + r; if (row >= N) row = N - 1;
        pw[j] = W + row * ldw + k0 + (int64_t)wave * 128 + (((lane & 7) ^ keyA8(r)) << 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {            // activation stage = 32 "rows" of 128 bytes: row rho = h * 16 + m holds x[m][k + h*64 .. +64]
        const int rho = 8 * j + (lane >> 3), m = rho & 15, h = rho >> 4;
        px[j] = X + (int64_t)(m < M ? m : M - 1) * ldx + k0 + (int64_t)wave * 128 + h * 64 + (((lane & 7) ^ keyA8(rho)) << 3);
    }
#define SK8_ISSUE(SLOT) do {                                                                                              \
        char* dst__ = ring + (SLOT) * STAGE;                                                                              \
        _Pragma("unroll") for (int j = 0; j < NWI; ++j) { __builtin_amdgcn_global_load_lds((w8_gptr_t)pw[j], (w8_lptr_t)(dst__ + j * 1024), 16, 0, 2); pw[j] += WAVES * 128; } \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) { __builtin_amdgcn_global_load_lds((w8_gptr_t)px[j], (w8_lptr_t)(dst__ + NWI * 1024 + j * 1024), 16, 0, 0); px[j] += WAVES * 128; } \
    } while (0)
    f32x4_t acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int rd_w = u * 128, kA = keyA8(u);
    const int partner = lane ^ 16, src_lane = u + 16 * (2 * (g & 1));
    auto consume = [&](int slot) {
        const char* sb = ring + slot * STAGE;
        const char* xb = sb + NWI * 1024;      // (56-column stages: the MFMA's weight rows 56..63 read into this area - they feed output columns that are never stored)
        float v[2][16], am[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            am[h] = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const u32x4_t xv = *reinterpret_cast<const u32x4_t*>(xb + (h * 16 + u) * 128 + (((2 * g + j) ^ kA) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = bflo(xv[e]), b = bfhi(xv[e]);
                    v[h][j * 8 + 2 * e] = a; v[h][j * 8 + 2 * e + 1] = b;
                    am[h] = fmaxf(am[h], fmaxf(fabsf(a), fabsf(b)));
                }
            }
            am[h] = fmaxf(am[h], __shfl(am[h], partner, 64));
        }
        int eb[2]; float mul[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int be = (int)(__float_as_uint(am[h]) >> 23);
            const bool tiny = be < 8;
            eb[h] = tiny ? 0 : be - 7;
            mul[h] = tiny ? 0.f : __uint_as_float((unsigned)(261 - be) << 23);
        }
        int o0[4], o1[4];
        quant16_fp8(v[0], mul[0], o0);
        quant16_fp8(v[1], mul[1], o1);
        const i32x8_t xq = {o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o1[2], o1[3]};
        const int both = __shfl(eb[0] | (eb[1] << 8), src_lane, 64);
        const int xs = (g >> 1) ? (both >> 8) & 0xff : both & 0xff;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const u32x4_t w0 = *reinterpret_cast<const u32x4_t*>(sb + c * 2048 + rd_w + ((g ^ kA) << 4));
            const u32x4_t w1 = *reinterpret_cast<const u32x4_t*>(sb + c * 2048 + rd_w + (((4 + g) ^ kA) << 4));
            const i32x8_t wf = {(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3], (int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
            acc[c] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf, xq, acc[c], 0, 0, 0, 127, 0, xs);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    if (n_my > 0) SK8_ISSUE(0);
    for (int i = 0; i < n_my; i += 2) {
        if (i + 1 < n_my) { SK8_ISSUE(1); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWI + 4) : "memory"); } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        consume(0);
        if (i + 1 < n_my) {
            if (i + 2 < n_my) { SK8_ISSUE(0); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWI + 4) : "memory"); } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            consume(1);
        }
    }
#undef SK8_ISSUE
    W8_BARRIER();
    float* red = reinterpret_cast<float*>(sk8_lds);                         // [WAVES][NC][16][17]
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * NC + c) * 16 + u) * 17 + g * 4 + r] = acc[c][r];
    __syncthreads();
    float* mine = fix_ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * TILE;
    for (int i = threadIdx.x; i < TILE; i += WAVES * 64) {
        const int c = i >> 8, mm = (i >> 4) & 15, nn = i & 15;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) v += red[((w * NC + c) * 16 + mm) * 17 + nn];
        __hip_atomic_store(mine + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) *s_ticket = __hip_atomic_fetch_add(&fix_cnt[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*s_ticket != (int)gridDim.y - 1) return;
    for (int i = threadIdx.x; i < TILE; i += WAVES * 64) {
        const int c = i >> 8, mm = (i >> 4) & 15, nn = i & 15;
        const int64_t n = n0 + c * 16 + nn;
        float v = 0.f;
        if (gridDim.y == 4) {       // unrolled: the four device-scope loads in flight together; same sum in the same order (see gemm_skinny_lds_fix_kernel)
            float t[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                t[ks] = __hip_atomic_load(fix_ws + ((int64_t)ks * gridDim.x + blockIdx.x) * TILE + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = ((t[0] + t[1]) + t[2]) + t[3];
        } else {
            for (int ks = 0; ks < (int)gridDim.y; ++ks)
                v += __hip_atomic_load(fix_ws + ((int64_t)ks * gridDim.x + blockIdx.x) * TILE + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (mm < M && n < N && c * 16 + nn < COLS) {
            v *= wscale[n];
            if (bias) v += bf2f(bias[n]);
            if (residual) v += bf2f(residual[(int64_t)mm * ldr + n]);
            C[(int64_t)mm * ldc + n] = f2bf(v);
        }
    }
    if (threadIdx.x == 0) fix_cnt[blockIdx.x] = 0;
}

// Per-row symmetric quantisation: scale[n] = amax_n / 448 (1 for an all-zero row), q = fp8_e4m3(w * (448 / amax_n)), round to nearest even.
__global__ __launch_bounds__(256) void quant_fp8_rows_kernel(const bf16_t* __restrict__ w, unsigned char* __restrict__ q, float* __restrict__ scale,
                                                             int64_t K, int64_t ldw, int64_t ldq) {
    __shared__ float red[16];
    const int64_t row = blockIdx.x;
    const bf16_t* src = w + row * ldw;
    float amax = 0.f;
    for (int64_t k = threadIdx.x * 8; k < K; k += 256 * 8) {
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(src + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(bflo(v[e])), fabsf(bfhi(v[e]))));
    }
    amax = block_max(amax, red);
    // correctly rounded divisions (hipcc's default fp32 '/' is a 2.5-ulp reciprocal sequence): codes must be reproducible bit for bit
    const float inv = amax > 0.f ? __fdiv_rn(448.0f, amax) : 1.0f;
    if (threadIdx.x == 0) scale[row] = amax > 0.f ? __fdiv_rn(amax, 448.0f) : 1.0f;
    unsigned char* dst = q + row * ldq;
    for (int64_t k = threadIdx.x * 8; k < K; k += 256 * 8) {
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(src + k);
        int lo = 0, hi = 0;            // v_cvt_pk_fp8_f32: round to nearest even, two codes per instruction
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(bflo(v[0]) * inv, bfhi(v[0]) * inv, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(bflo(v[1]) * inv, bfhi(v[1]) * inv, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(bflo(v[2]) * inv, bfhi(v[2]) * inv, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(bflo(v[3]) * inv, bfhi(v[3]) * inv, hi, true);
        u32x2_t o = {(unsigned)lo, (unsigned)hi};
        *reinterpret_cast<u32x2_t*>(dst + k) = o;
    }
}

extern "C" int tr1_quantize_fp8_rows(const void* w_bf16, int64_t ldw, void* q_fp8, int64_t ldq, void* scale_f32, int64_t N, int64_t K, void* stream) {
    TR1_CHECK_ARG(K % 8 == 0 && ldw % 8 == 0 && ldq % 8 == 0, "quantize_fp8_rows: K and leading dimensions must be multiples of 8");
    if (N == 0) return 0;
    hipLaunchKernelGGL(quant_fp8_rows_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w_bf16, (unsigned char*)q_fp8,
                       (float*)scale_f32, K, ldw, ldq);
    TR1_LAUNCH_CHECK();
}

extern "C" int tr1_gemm_skinny_w8(const void* x, const void* lnw, const void* W_fp8, const void* wscale, const void* bias, const void* residual,
                                  void* out, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw, int64_t ldc, int64_t ldr, float eps,
                                  int glu, void* stream) {
    TR1_CHECK_ARG(K % 128 == 0 && K >= 128, "gemm_skinny_w8: K must be a positive multiple of 128");
    TR1_CHECK_ARG(M >= 1 && M <= 64, "gemm_skinny_w8: 1 <= M <= 64 (decode rows)");
    TR1_CHECK_ARG(N % 8 == 0 && ldx % 8 == 0 && ldw % 16 == 0 && ldc % 8 == 0 && (!residual || ldr % 8 == 0), "gemm_skinny_w8: N%8, ldx%8, ldw%16, ldc%8");
    TR1_CHECK_ARG(!glu || (lnw && !bias && !residual), "gemm_skinny_w8: the GLU form is norm + gate/up only");
    hipStream_t s = (hipStream_t)stream;
    const int mg = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
#define W8(WV, UN, NC, MGR, NRM, GL)                                                                                                       \
    hipLaunchKernelGGL((gemm_skinny_w8_kernel<WV, UN, NC, MGR, NRM, GL>), dim3((unsigned)((N + (GL ? 8 * NC : 16 * NC) - 1) / (GL ? 8 * NC : 16 * NC))), \
                       dim3(WV * 64), 0, s, (const bf16_t*)x, (const bf16_t*)lnw, (const unsigned char*)W_fp8, (const float*)wscale,       \
                       (bf16_t*)out, (const bf16_t*)bias, (const bf16_t*)residual, (int)M, N, K, ldx, ldw, ldc, ldr, eps, N)
#define W8_MG(WV, UN, NC, NRM, GL)                                                         \
    do { if (mg == 1) W8(WV, UN, NC, 1, NRM, GL); else if (mg == 2) W8(WV, UN, NC, 2, NRM, GL); else W8(WV, 1, NC, 4, NRM, GL); } while (0)
    {   // tuning hook for tools/microbench.py w8: TR1_W8_CFG=<waves><unroll><ncol> (M <= 16 only)
        static int cfg = -1;
        if (cfg < 0) { cfg = 0; }
        if (cfg && !glu && M <= 16) {
            const bool nrm = lnw != nullptr;
#define W8C(WV, UN, NC) do { if (nrm) W8(WV, UN, NC, 1, true, false); else W8(WV, UN, NC, 1, false, false); TR1_LAUNCH_CHECK(); } while (0)
            switch (cfg) {
                case 442: W8C(4, 4, 2);
                case 424: W8C(4, 2, 4);
                case 444: W8C(4, 4, 4);
                case 822: W8C(8, 2, 2);
                case 824: W8C(8, 2, 4);
                case 814: W8C(8, 1, 4);
                case 842: W8C(8, 4, 2);
                case 441: W8C(4, 4, 1);
                case 841: W8C(8, 4, 1);
                default: break;
            }
#undef W8C
        }
    }
    // column groups per block: the activations are re-read from L2 by every block, and with fp8 weights they are as many bytes as a
    // 2-group weight slab - 4 groups halve that traffic (measured, M = 16: lm_head 168 -> 140 us)
    if (glu) { if (mg == 1) W8(4, 2, 4, 1, true, true); else if (mg == 2) W8(4, 2, 4, 2, true, true); else W8(4, 1, 2, 4, true, true); }
    else if (lnw && N >= 100000) { if (mg == 1) W8(4, 2, 4, 1, true, false); else if (mg == 2) W8(4, 2, 4, 2, true, false); else W8(4, 1, 2, 4, true, false); }
    else if (lnw) W8_MG(4, 2, 2, true, false);
    else if (K >= 8192) W8_MG(8, 2, 1, false, false);
    else W8_MG(4, 2, 1, false, false);
#undef W8_MG
#undef W8
    TR1_LAUNCH_CHECK();
}

// W8A8: the same dispatch on the fp8-MFMA kernel (activations block-quantised to e4m3 in the operand load)
extern "C" int tr1_gemm_skinny_w8a8(const void* x, const void* lnw, const void* W_fp8, const void* wscale, const void* bias, const void* residual,
                                  void* out, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw, int64_t ldc, int64_t ldr, float eps,
                                  int glu, void* stream) {
    TR1_CHECK_ARG(K % 128 == 0 && K >= 128, "gemm_skinny_w8a8: K must be a positive multiple of 128");
    TR1_CHECK_ARG(M >= 1 && M <= 64, "gemm_skinny_w8a8: 1 <= M <= 64 (decode rows)");
    TR1_CHECK_ARG(N % 8 == 0 && ldx % 8 == 0 && ldw % 16 == 0 && ldc % 8 == 0 && (!residual || ldr % 8 == 0), "gemm_skinny_w8a8: N%8, ldx%8, ldw%16, ldc%8");
    TR1_CHECK_ARG(!glu || (lnw && !bias && !residual), "gemm_skinny_w8a8: the GLU form is norm + gate/up only");
    hipStream_t s = (hipStream_t)stream;
    const int mg = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
#define W8(WV, UN, NC, MGR, NRM, GL)                                                                                                       \
    hipLaunchKernelGGL((gemm_skinny_w8a8_kernel<WV, UN, NC, MGR, NRM, GL>), dim3((unsigned)((N + (GL ? 8 * NC : 16 * NC) - 1) / (GL ? 8 * NC : 16 * NC))), \
                       dim3(WV * 64), 0, s, (const bf16_t*)x, (const bf16_t*)lnw, (const unsigned char*)W_fp8, (const float*)wscale,       \
                       (bf16_t*)out, (const bf16_t*)bias, (const bf16_t*)residual, (int)M, N, K, ldx, ldw, ldc, ldr, eps, N)
#define W8_MG(WV, UN, NC, NRM, GL)                                                         \
    do { if (mg == 1) W8(WV, UN, NC, 1, NRM, GL); else if (mg == 2) W8(WV, UN, NC, 2, NRM, GL); else W8(WV, 1, NC, 4, NRM, GL); } while (0)
    {   // tuning hook for tools/microbench.py w8: TR1_W8A8_CFG=<waves><unroll><ncol> (M <= 16 only)
        static int cfg = -1;
        if (cfg < 0) { cfg = 0; }
        if (cfg && !glu && M <= 16) {
            const bool nrm = lnw != nullptr;
#define W8C(WV, UN, NC) do { if (nrm) W8(WV, UN, NC, 1, true, false); else W8(WV, UN, NC, 1, false, false); TR1_LAUNCH_CHECK(); } while (0)
            switch (cfg) {
                case 442: W8C(4, 4, 2);
                case 424: W8C(4, 2, 4);
                case 444: W8C(4, 4, 4);
                case 822: W8C(8, 2, 2);
                case 824: W8C(8, 2, 4);
                case 814: W8C(8, 1, 4);
                case 842: W8C(8, 4, 2);
                case 441: W8C(4, 4, 1);
                case 841: W8C(8, 4, 1);
                default: break;
            }
#undef W8C
        }
    }
    // column groups per block: the activations are re-read from L2 by every block, and with fp8 weights they are as many bytes as a
    // 2-group weight slab - 4 groups halve that traffic (measured, M = 16: lm_head 168 -> 140 us)
    {   // gate/up at <= 16 rows, hidden 3584 / 2048 / 1536: the LDS-streamed form (TR1_W8_GLU_LDS=0: register-fragment form, A/B runs)
        static int glu_lds = -1;
        if (glu_lds < 0) { glu_lds = 1; }
        const int64_t nw = K / 512;
        if (glu && M <= 16 && glu_lds && K % 512 == 0 && (nw == 7 || nw == 4 || nw == 3) && N % 16 == 0) {
            constexpr int RING = 3;
            static int n_cu = 0;
            const size_t dyn = (size_t)nw * RING * 4096 + (2 * nw * 2 * 16 * 17 + nw * 16) * sizeof(float) + (size_t)nw * (4096 + 1024);      // + one x stage and the norm-weight KiB per wave
            if (!n_cu) {
                hipDeviceProp_t prop; int dev = 0;
                hipGetDevice(&dev); hipGetDeviceProperties(&prop, dev);
                n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
                const int mx = 7 * RING * 4096 + (2 * 7 * 2 * 16 * 17 + 7 * 16) * (int)sizeof(float) + 7 * (4096 + 1024);
                hipFuncSetAttribute(reinterpret_cast<const void*>(&norm_glu_lds_f8_kernel<7, RING>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
                hipFuncSetAttribute(reinterpret_cast<const void*>(&norm_glu_lds_f8_kernel<4, RING>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
                hipFuncSetAttribute(reinterpret_cast<const void*>(&norm_glu_lds_f8_kernel<3, RING>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
            }
            const int64_t NP = N / 16;
            const unsigned grid = (unsigned)(NP < n_cu ? NP : n_cu);
#define G8L(NWV) hipLaunchKernelGGL((norm_glu_lds_f8_kernel<NWV, RING>), dim3(grid), dim3(NWV * 64), dyn, s, (const bf16_t*)x, (const bf16_t*)lnw, \
                                    (const unsigned char*)W_fp8, (const float*)wscale, (bf16_t*)out, (int)M, N, K, ldx, ldw, ldc, eps, N)
            if (nw == 7) G8L(7); else if (nw == 4) G8L(4); else G8L(3);
#undef G8L
            TR1_LAUNCH_CHECK();
        }
    }
    if (glu) { if (mg == 1) W8(4, 2, 4, 1, true, true); else if (mg == 2) W8(4, 2, 4, 2, true, true); else W8(4, 1, 2, 4, true, true); }
    else if (lnw && N >= 100000) { if (mg == 1) W8(4, 2, 4, 1, true, false); else if (mg == 2) W8(4, 2, 4, 2, true, false); else W8(4, 1, 2, 4, true, false); }
    else {
// x image (32 bytes per k: 16 rows of bf16) + the norm weight, which arrives in whole 1 KiB DMA instructions (the last one may run past K * 2 bytes)
#define W8X_LDS(K_) ((int64_t)(K_) * 32 + (((int64_t)(K_) * 2 + 1023) / 1024) * 1024)
        static int xlds = -1;                        // TR1_W8A8_XLDS=0: activation rows / norm weight through the vector-memory path (A/B measurements)
        if (xlds < 0) { xlds = 1; }
        const bool x_ok = xlds && mg == 1 && K < 8192 && K % 128 == 0 && W8X_LDS(K) <= 128 * 1024 && (int64_t)M * ldx * 2 < 0x7fffffffLL;
#define W8X(NC, NRM) do {                                                                                                                        \
            static bool attr_ = false;                                                                                                           \
            if (!attr_) { hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_w8a8_kernel<4, 2, NC, 1, NRM, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); attr_ = true; } \
            hipLaunchKernelGGL((gemm_skinny_w8a8_kernel<4, 2, NC, 1, NRM, false, true>), dim3((unsigned)((N + 16 * NC - 1) / (16 * NC))), dim3(256), (size_t)W8X_LDS(K), s, \
                               (const bf16_t*)x, (const bf16_t*)lnw, (const unsigned char*)W_fp8, (const float*)wscale, (bf16_t*)out, (const bf16_t*)bias,  \
                               (const bf16_t*)residual, (int)M, N, K, ldx, ldw, ldc, ldr, eps, N);                                              \
        } while (0)
        if (lnw) { if (x_ok) W8X(2, true); else W8_MG(4, 2, 2, true, false); }
        else if (K >= 8192) W8_MG(8, 2, 1, false, false);
        else { if (x_ok) W8X(1, false); else W8_MG(4, 2, 1, false, false); }
#undef W8X
    }
#undef W8_MG
#undef W8
    TR1_LAUNCH_CHECK();
}

// Split-K + fixup form of the W8A8 projection for M <= 16 rows and wide K (decode down projection).  Workspace = the one of
// tr1_gemm_skinny_fixup for the same (M, N, K) (fp32 tiles + self re-arming ticket counters, zero-filled once by the caller).
extern "C" int tr1_gemm_skinny_fixup_w8a8(const void* x, const void* W_fp8, const void* wscale, void* out, const void* bias, const void* residual,
                                          int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw, int64_t ldc, int64_t ldr, void* ws_f32,
                                          int64_t ws_floats, void* stream) {
    constexpr int KS = 4, WV = 6;
    TR1_CHECK_ARG(M >= 1 && M <= 16 && K % (KS * 128) == 0 && N % 64 == 0, "gemm_skinny_fixup_w8a8: 1 <= M <= 16, K % 512 == 0, N % 64 == 0");
    TR1_CHECK_ARG(ldx % 8 == 0 && ldw % 16 == 0 && ldc % 8 == 0 && (!residual || ldr % 8 == 0), "gemm_skinny_fixup_w8a8: ldx%8, ldw%16, ldc%8");
    // round 5: 56-column blocks where that fills the chip (N = 3584: 64 groups x 4 slabs = 256 blocks instead of 224) - the 11 KiB stages then leave room for
    // a seventh wave (7 x 2 x 11 KiB = 154 KiB); same K slabs, same ordered fixup (what the bf16 kernel got in round 3)
    if (N % 56 == 0 && (N / 56) * KS <= 256 && N / 56 > N / 64) {
        constexpr int WV7 = 7;
        const int64_t g56 = N / 56;
        TR1_CHECK_ARG(ws_f32 && ws_floats >= KS * g56 * 4 * 256 + g56, "gemm_skinny_fixup_w8a8: workspace too small");
        float* tiles56 = (float*)ws_f32;
        int* cnt56 = (int*)(tiles56 + KS * g56 * 4 * 256);
        const size_t dyn56 = WV7 * 2 * (7 * 1024 + 4096) + 16;
        static bool attr56 = false;
        if (!attr56) { hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_lds_fix_f8_kernel<WV7, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn56); attr56 = true; }
        hipLaunchKernelGGL((gemm_skinny_lds_fix_f8_kernel<WV7, 7>), dim3((unsigned)g56, KS), dim3(WV7 * 64), dyn56, (hipStream_t)stream, (const bf16_t*)x,
                           (const unsigned char*)W_fp8, (const float*)wscale, (bf16_t*)out, (const bf16_t*)bias, (const bf16_t*)residual, (int)M, N, K, ldx,
                           ldw, ldc, ldr, tiles56, cnt56);
        TR1_LAUNCH_CHECK();
    }
    const int64_t groups = N / 64;
    TR1_CHECK_ARG(ws_f32 && ws_floats >= KS * groups * 4 * 256 + groups, "gemm_skinny_fixup_w8a8: workspace too small");
    float* tiles = (float*)ws_f32;
    int* cnt = (int*)(tiles + KS * groups * 4 * 256);
    const size_t dyn = WV * 2 * 12288 + 16;
    static bool attr_set = false;
    if (!attr_set) { hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_lds_fix_f8_kernel<WV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); attr_set = true; }
    hipLaunchKernelGGL((gemm_skinny_lds_fix_f8_kernel<WV>), dim3((unsigned)groups, KS), dim3(WV * 64), dyn, (hipStream_t)stream, (const bf16_t*)x,
                       (const unsigned char*)W_fp8, (const float*)wscale, (bf16_t*)out, (const bf16_t*)bias, (const bf16_t*)residual, (int)M, N, K, ldx,
                       ldw, ldc, ldr, tiles, cnt);
    TR1_LAUNCH_CHECK();
}
