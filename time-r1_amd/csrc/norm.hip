// RMSNorm / LayerNorm forward + backward for gfx950.
// One wave64 per row, 16-byte (8 x bf16) accesses, the row lives in registers between the
// reduction and the scale pass (cols <= 4096), fp32 math, HBM-bound by construction.
//
// Semantics follow the reference model's norm layers:
//   RMSNorm  : transformers/models/qwen2_vl/modeling_qwen2_vl.py:96-110  (cast to bf16 BEFORE the weight multiply)
//   LayerNorm: torch.nn.LayerNorm(eps=1e-6) in Qwen2-VL VisionBlock / PatchMerger (:277-301, :425-449)
#include "tr1_common.h"

#define NORM_MAXC 8  // 8 chunks x 8 elems x 64 lanes = 4096 columns

template <bool HAS_RES, bool HOIST_W>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ res,
                                                          const bf16_t* __restrict__ w, bf16_t* __restrict__ y,
                                                          bf16_t* __restrict__ xsum, float* __restrict__ rstd_out,
                                                          int rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int nch = cols >> 3;  // 16-byte chunks per row
    for (int r = wave; r < rows; r += nwaves) {
        const u32x4_t* xr = reinterpret_cast<const u32x4_t*>(x + (size_t)r * cols);
        const u32x4_t* rr = HAS_RES ? reinterpret_cast<const u32x4_t*>(res + (size_t)r * cols) : nullptr;
        u32x4_t* sr = HAS_RES ? reinterpret_cast<u32x4_t*>(xsum + (size_t)r * cols) : nullptr;
        float v[NORM_MAXC][8];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NORM_MAXC; ++i) {
            const int c = lane + i * 64;
            if (c < nch) {
                u32x4_t p = xr[c];
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[i][2 * j] = bflo(p[j]); v[i][2 * j + 1] = bfhi(p[j]); }
                if (HAS_RES) {
                    u32x4_t q = rr[c];
                    u32x4_t o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // residual stream is bf16: round the sum first (x + res in bf16), then normalise that
                        unsigned s = pack2bf(v[i][2 * j] + bflo(q[j]), v[i][2 * j + 1] + bfhi(q[j]));
                        o[j] = s; v[i][2 * j] = bflo(s); v[i][2 * j + 1] = bfhi(s);
                    }
                    sr[c] = o;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) ss += v[i][j] * v[i][j];
            }
        }
        // The weight row does not depend on the reduction.  For the few-row decode case (latency bound) its loads are issued before
        // the cross-lane sum so the two latencies overlap; for many rows (bandwidth bound) they stay behind it to save registers.
        const u32x4_t* wr = reinterpret_cast<const u32x4_t*>(w);
        u32x4_t wv[NORM_MAXC];
        if (HOIST_W) {
#pragma unroll
            for (int i = 0; i < NORM_MAXC; ++i) {
                const int c = lane + i * 64;
                if (c < nch) wv[i] = wr[c];
            }
        }
        ss = wave_sum(ss);
        const float rstd = rsqrtf(ss / (float)cols + eps);
        if (lane == 0 && rstd_out) rstd_out[r] = rstd;
        u32x4_t* yr = reinterpret_cast<u32x4_t*>(y + (size_t)r * cols);
#pragma unroll
        for (int i = 0; i < NORM_MAXC; ++i) {
            const int c = lane + i * 64;
            if (c < nch) {
                u32x4_t o;
                if (!HOIST_W) wv[i] = wr[c];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // bf16(x*rstd) then * weight, rounded again (reference cast order)
                    float a = bf2f(f2bf(v[i][2 * j] * rstd)) * bflo(wv[i][j]);
                    float b = bf2f(f2bf(v[i][2 * j + 1] * rstd)) * bfhi(wv[i][j]);
                    o[j] = pack2bf(a, b);
                }
                yr[c] = o;
            }
        }
    }
}

// dx = rstd * (dy*w - xhat * mean(dy*w*xhat)) (+ dres) ; dw[c] += sum_r dy*xhat  (fp32 atomics, one per block per column)
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                          const bf16_t* __restrict__ w, const float* __restrict__ rstd_in,
                                                          const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx,
                                                          float* __restrict__ dw_part, int rows, int cols) {
    extern __shared__ __attribute__((aligned(16))) float lds_dw[];  // [4 waves][cols]
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int nch = cols >> 3;
    float dwa[NORM_MAXC][8];
#pragma unroll
    for (int i = 0; i < NORM_MAXC; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) dwa[i][j] = 0.f;
    const u32x4_t* wr = reinterpret_cast<const u32x4_t*>(w);
    for (int r = wave; r < rows; r += nwaves) {
        const u32x4_t* xr = reinterpret_cast<const u32x4_t*>(x + (size_t)r * cols);
        const u32x4_t* gr = reinterpret_cast<const u32x4_t*>(dy + (size_t)r * cols);
        const float rstd = rstd_in[r];
        float xh[NORM_MAXC][8], gw[NORM_MAXC][8];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < NORM_MAXC; ++i) {
            const int c = lane + i * 64;
            if (c < nch) {
                u32x4_t xp = xr[c], gp = gr[c], wp = wr[c];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x0 = bflo(xp[j]) * rstd, x1 = bfhi(xp[j]) * rstd;
                    float g0 = bflo(gp[j]), g1 = bfhi(gp[j]);
                    xh[i][2 * j] = x0; xh[i][2 * j + 1] = x1;
                    dwa[i][2 * j] += g0 * x0; dwa[i][2 * j + 1] += g1 * x1;
                    g0 *= bflo(wp[j]); g1 *= bfhi(wp[j]);
                    gw[i][2 * j] = g0; gw[i][2 * j + 1] = g1;
                    dot += g0 * x0 + g1 * x1;
                }
            }
        }
        dot = wave_sum(dot) / (float)cols;
        u32x4_t* dxr = reinterpret_cast<u32x4_t*>(dx + (size_t)r * cols);
        const u32x4_t* drr = dres ? reinterpret_cast<const u32x4_t*>(dres + (size_t)r * cols) : nullptr;
#pragma unroll
        for (int i = 0; i < NORM_MAXC; ++i) {
            const int c = lane + i * 64;
            if (c < nch) {
                u32x4_t o, d = {0, 0, 0, 0};
                if (drr) d = drr[c];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float a = rstd * (gw[i][2 * j] - xh[i][2 * j] * dot) + bflo(d[j]);
                    float b = rstd * (gw[i][2 * j + 1] - xh[i][2 * j + 1] * dot) + bfhi(d[j]);
                    o[j] = pack2bf(a, b);
                }
                dxr[c] = o;
            }
        }
    }
    if (dw_part) {      // per-block partial row (fixed summation order: no atomics, the weight gradient is reproducible)
        float* mine = lds_dw + (threadIdx.x >> 6) * cols;
#pragma unroll
        for (int i = 0; i < NORM_MAXC; ++i) {
            const int c = lane + i * 64;
            if (c < nch) {
                *reinterpret_cast<f32x4_t*>(mine + c * 8) = (f32x4_t){dwa[i][0], dwa[i][1], dwa[i][2], dwa[i][3]};
                *reinterpret_cast<f32x4_t*>(mine + c * 8 + 4) = (f32x4_t){dwa[i][4], dwa[i][5], dwa[i][6], dwa[i][7]};
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < cols; c += blockDim.x)
            dw_part[(size_t)blockIdx.x * cols + c] = (lds_dw[c] + lds_dw[cols + c]) + (lds_dw[2 * cols + c] + lds_dw[3 * cols + c]);
    }
}

// dw[c] += sum_b part[b][c], b in ascending order inside each of 16 row groups, groups combined in a fixed order.  64 columns per block.
__global__ __launch_bounds__(1024) void norm_dw_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nblk, int cols) {
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int per = (nblk + 15) / 16, b0 = rg * per, b1 = min(nblk, b0 + per);
    float s = 0.f;
    if (c < cols) {
#pragma unroll 8
        for (int b = b0; b < b1; ++b) s += part[(size_t)b * cols + c];
    }
    red[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][cl];
        dw[c] += t;
    }
}

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            const bf16_t* __restrict__ b, bf16_t* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int nch = cols >> 3;
    const u32x4_t* wr = reinterpret_cast<const u32x4_t*>(w);
    const u32x4_t* br = reinterpret_cast<const u32x4_t*>(b);
    for (int r = wave; r < rows; r += nwaves) {
        const u32x4_t* xr = reinterpret_cast<const u32x4_t*>(x + (size_t)r * cols);
        float v[NORM_MAXC][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NORM_MAXC; ++i) {
            const int c = lane + i * 64;
            if (c < nch) {
                u32x4_t p = xr[c];
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[i][2 * j] = bflo(p[j]); v[i][2 * j + 1] = bfhi(p[j]); s += v[i][2 * j] + v[i][2 * j + 1]; }
            }
        }
        const float mean = wave_sum(s) / (float)cols;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NORM_MAXC; ++i) {
            const int c = lane + i * 64;
            if (c < nch) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { float d = v[i][j] - mean; ss += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(ss) / (float)cols + eps);
        if (lane == 0) {
            if (mean_out) mean_out[r] = mean;
            if (rstd_out) rstd_out[r] = rstd;
        }
        u32x4_t* yr = reinterpret_cast<u32x4_t*>(y + (size_t)r * cols);
#pragma unroll
        for (int i = 0; i < NORM_MAXC; ++i) {
            const int c = lane + i * 64;
            if (c < nch) {
                u32x4_t wv = wr[c], bv = br[c], o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float a = (v[i][2 * j] - mean) * rstd * bflo(wv[j]) + bflo(bv[j]);
                    float c2 = (v[i][2 * j + 1] - mean) * rstd * bfhi(wv[j]) + bfhi(bv[j]);
                    o[j] = pack2bf(a, c2);
                }
                yr[c] = o;
            }
        }
    }
}

// Parameter gradients of LayerNorm (the patch-merger's ln_q is trainable while the ViT below it is frozen,
// reference timer1_trainer.py:272-280), so only dgamma / dbeta are needed; dx is optional.
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                            const bf16_t* __restrict__ w, const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in, bf16_t* __restrict__ dx,
                                                            float* __restrict__ dw, float* __restrict__ db, int rows, int cols) {
    extern __shared__ __attribute__((aligned(16))) float lds_acc[];  // [2*cols]
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int nch = cols >> 3;
    for (int c = threadIdx.x; c < 2 * cols; c += blockDim.x) lds_acc[c] = 0.f;
    __syncthreads();
    float dwa[NORM_MAXC][8], dba[NORM_MAXC][8];
#pragma unroll
    for (int i = 0; i < NORM_MAXC; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { dwa[i][j] = 0.f; dba[i][j] = 0.f; }
    const u32x4_t* wr = reinterpret_cast<const u32x4_t*>(w);
    for (int r = wave; r < rows; r += nwaves) {
        const u32x4_t* xr = reinterpret_cast<const u32x4_t*>(x + (size_t)r * cols);
        const u32x4_t* gr = reinterpret_cast<const u32x4_t*>(dy + (size_t)r * cols);
        const float mean = mean_in[r], rstd = rstd_in[r];
        float xh[NORM_MAXC][8], gw[NORM_MAXC][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NORM_MAXC; ++i) {
            const int c = lane + i * 64;
            if (c < nch) {
                u32x4_t xp = xr[c], gp = gr[c], wp = wr[c];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x0 = (bflo(xp[j]) - mean) * rstd, x1 = (bfhi(xp[j]) - mean) * rstd;
                    float g0 = bflo(gp[j]), g1 = bfhi(gp[j]);
                    xh[i][2 * j] = x0; xh[i][2 * j + 1] = x1;
                    dwa[i][2 * j] += g0 * x0; dwa[i][2 * j + 1] += g1 * x1;
                    dba[i][2 * j] += g0; dba[i][2 * j + 1] += g1;
                    g0 *= bflo(wp[j]); g1 *= bfhi(wp[j]);
                    gw[i][2 * j] = g0; gw[i][2 * j + 1] = g1;
                    s1 += g0 + g1; s2 += g0 * x0 + g1 * x1;
                }
            }
        }
        if (dx) {
            s1 = wave_sum(s1) / (float)cols; s2 = wave_sum(s2) / (float)cols;
            u32x4_t* dxr = reinterpret_cast<u32x4_t*>(dx + (size_t)r * cols);
#pragma unroll
            for (int i = 0; i < NORM_MAXC; ++i) {
                const int c = lane + i * 64;
                if (c < nch) {
                    u32x4_t o;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        o[j] = pack2bf(rstd * (gw[i][2 * j] - s1 - xh[i][2 * j] * s2),
                                       rstd * (gw[i][2 * j + 1] - s1 - xh[i][2 * j + 1] * s2));
                    dxr[c] = o;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NORM_MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { atomicAdd(&lds_acc[c * 8 + j], dwa[i][j]); atomicAdd(&lds_acc[cols + c * 8 + j], dba[i][j]); }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += blockDim.x) { atomicAdd(&dw[c], lds_acc[c]); atomicAdd(&db[c], lds_acc[cols + c]); }
}

static inline int norm_grid(int rows) { int g = (rows + 3) / 4; if (g > 2048) g = 2048; if (g < 1) g = 1; return g; }

extern "C" int tr1_rmsnorm_fwd(const void* x, const void* residual, const void* w, void* y, void* xsum, void* rstd,
                               int64_t rows, int64_t cols, float eps, void* stream) {
    TR1_CHECK_ARG(cols % 8 == 0 && cols <= 64 * 8 * NORM_MAXC, "rmsnorm: cols must be a multiple of 8 and <= 4096");
    TR1_CHECK_ARG(!residual || xsum, "rmsnorm: residual given without xsum output");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
#define RMS_LAUNCH(HR, HW)                                                                                                             \
    hipLaunchKernelGGL((rmsnorm_fwd_kernel<HR, HW>), dim3(norm_grid(rows)), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)residual, \
                       (const bf16_t*)w, (bf16_t*)y, (bf16_t*)xsum, (float*)rstd, (int)rows, (int)cols, eps)
    const bool few = rows <= 256;
    if (residual) { if (few) RMS_LAUNCH(true, true); else RMS_LAUNCH(true, false); }
    else { if (few) RMS_LAUNCH(false, true); else RMS_LAUNCH(false, false); }
#undef RMS_LAUNCH
    TR1_LAUNCH_CHECK();
}

static int norm_bwd_blocks(int64_t rows) { int g = (int)((rows + 3) / 4); return g > 512 ? 512 : g; }

extern "C" int64_t tr1_rmsnorm_bwd_workspace_floats(int64_t rows, int64_t cols) { return (int64_t)norm_bwd_blocks(rows) * cols; }

extern "C" int tr1_rmsnorm_bwd(const void* dy, const void* x, const void* w, const void* rstd, const void* dres, void* dx,
                               void* dw_f32, void* ws_f32, int64_t ws_floats, int64_t rows, int64_t cols, void* stream) {
    TR1_CHECK_ARG(cols % 8 == 0 && cols <= 64 * 8 * NORM_MAXC, "rmsnorm_bwd: cols must be a multiple of 8 and <= 4096");
    TR1_CHECK_ARG(dy && x && w && rstd && dx, "rmsnorm_bwd: dy, x, w, rstd and dx are required");
    if (rows == 0) return 0;
    const int g = norm_bwd_blocks(rows);
    TR1_CHECK_ARG(!dw_f32 || (ws_f32 && ws_floats >= (int64_t)g * cols), "rmsnorm_bwd: workspace too small (tr1_rmsnorm_bwd_workspace_floats)");
    hipLaunchKernelGGL(rmsnorm_bwd_kernel, dim3(g), dim3(256), dw_f32 ? 4 * cols * sizeof(float) : 0, (hipStream_t)stream, (const bf16_t*)dy,
                       (const bf16_t*)x, (const bf16_t*)w, (const float*)rstd, (const bf16_t*)dres, (bf16_t*)dx, dw_f32 ? (float*)ws_f32 : nullptr,
                       (int)rows, (int)cols);
    if (dw_f32)
        hipLaunchKernelGGL(norm_dw_reduce_kernel, dim3((unsigned)((cols + 63) / 64)), dim3(1024), 0, (hipStream_t)stream, (const float*)ws_f32,
                           (float*)dw_f32, g, (int)cols);
    TR1_LAUNCH_CHECK();
}

extern "C" int tr1_layernorm_fwd(const void* x, const void* w, const void* b, void* y, void* mean, void* rstd, int64_t rows,
                                 int64_t cols, float eps, void* stream) {
    TR1_CHECK_ARG(cols % 8 == 0 && cols <= 64 * 8 * NORM_MAXC, "layernorm: cols must be a multiple of 8 and <= 4096");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(norm_grid(rows)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, (float*)mean, (float*)rstd, (int)rows, (int)cols, eps);
    TR1_LAUNCH_CHECK();
}

extern "C" int tr1_layernorm_bwd(const void* dy, const void* x, const void* w, const void* mean, const void* rstd, void* dx,
                                 void* dw_f32, void* db_f32, int64_t rows, int64_t cols, void* stream) {
    TR1_CHECK_ARG(cols % 8 == 0 && cols <= 64 * 8 * NORM_MAXC, "layernorm_bwd: cols must be a multiple of 8 and <= 4096");
    TR1_CHECK_ARG(dw_f32 && db_f32, "layernorm_bwd: dw/db required");
    if (rows == 0) return 0;
    int g = (int)((rows + 3) / 4); if (g > 512) g = 512;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(g), dim3(256), 2 * cols * sizeof(float), (hipStream_t)stream, (const bf16_t*)dy,
                       (const bf16_t*)x, (const bf16_t*)w, (const float*)mean, (const float*)rstd, (bf16_t*)dx, (float*)dw_f32,
                       (float*)db_f32, (int)rows, (int)cols);
    TR1_LAUNCH_CHECK();
}
