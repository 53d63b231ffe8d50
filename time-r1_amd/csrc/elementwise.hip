// Memory-bound elementwise / gather / transpose kernels for gfx950. All bf16 traffic is 16 bytes per lane.
//
// Reference semantics:
//   SwiGLU     down(silu(gate(x)) * up(x))          transformers/models/qwen2_vl/modeling_qwen2_vl.py:459-466
//   QuickGELU  x * sigmoid(1.702 x) (ViT MLP)       :293-301
//   GELU(erf)  PatchMerger MLP                      :277-290
//   M-RoPE     :117-222 ; vision 2-D RoPE :225-248 (+ vision_utils.py:81-127)
#include "tr1_common.h"

// ---------------------------------------------------------------- activations
TR1_DEV float silu_f(float x) { return x / (1.f + __expf(-x)); }

// gu: [rows, 2*inter] (gate | up), out: [rows, inter]
__global__ void swiglu_fwd_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ out, int64_t rows, int inter) {
    const int nch = inter >> 3;
    const int64_t total = rows * nch;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / nch; const int c = (int)(i - r * nch);
        const u32x4_t g = *reinterpret_cast<const u32x4_t*>(gu + r * 2 * inter + c * 8);
        const u32x4_t u = *reinterpret_cast<const u32x4_t*>(gu + r * 2 * inter + inter + c * 8);
        u32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // silu output is rounded to bf16 before the product (two separate bf16 ops in the reference)
            float a = bf2f(f2bf(silu_f(bflo(g[j])))) * bflo(u[j]);
            float b = bf2f(f2bf(silu_f(bfhi(g[j])))) * bfhi(u[j]);
            o[j] = pack2bf(a, b);
        }
        *reinterpret_cast<u32x4_t*>(out + r * inter + c * 8) = o;
    }
}

// dgu = [dout*u*silu'(g) | dout*silu(g)]
__global__ void swiglu_bwd_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ gu, bf16_t* __restrict__ dgu,
                                  int64_t rows, int inter) {
    const int nch = inter >> 3;
    const int64_t total = rows * nch;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / nch; const int c = (int)(i - r * nch);
        const u32x4_t g = *reinterpret_cast<const u32x4_t*>(gu + r * 2 * inter + c * 8);
        const u32x4_t u = *reinterpret_cast<const u32x4_t*>(gu + r * 2 * inter + inter + c * 8);
        const u32x4_t d = *reinterpret_cast<const u32x4_t*>(dout + r * inter + c * 8);
        u32x4_t og, ou;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gg[2] = {bflo(g[j]), bfhi(g[j])}, uu[2] = {bflo(u[j]), bfhi(u[j])}, dd[2] = {bflo(d[j]), bfhi(d[j])};
            float rg[2], ru[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float sg = 1.f / (1.f + __expf(-gg[k]));
                float si = gg[k] * sg;
                rg[k] = dd[k] * uu[k] * (sg * (1.f + gg[k] * (1.f - sg)));
                ru[k] = dd[k] * si;
            }
            og[j] = pack2bf(rg[0], rg[1]); ou[j] = pack2bf(ru[0], ru[1]);
        }
        *reinterpret_cast<u32x4_t*>(dgu + r * 2 * inter + c * 8) = og;
        *reinterpret_cast<u32x4_t*>(dgu + r * 2 * inter + inter + c * 8) = ou;
    }
}

// mode 0: gelu(erf) fwd, 1: quick_gelu fwd, 2: gelu(erf) bwd (needs dy), 3: silu fwd? (unused)
template <int MODE>
__global__ void act_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, bf16_t* __restrict__ y, int64_t nchunks) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nchunks; i += (int64_t)gridDim.x * blockDim.x) {
        const u32x4_t p = reinterpret_cast<const u32x4_t*>(x)[i];
        u32x4_t d = {0, 0, 0, 0};
        if (MODE == 2) d = reinterpret_cast<const u32x4_t*>(dy)[i];
        u32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[2] = {bflo(p[j]), bfhi(p[j])}, g[2] = {bflo(d[j]), bfhi(d[j])}, r[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (MODE == 0) r[k] = 0.5f * v[k] * (1.f + erff(v[k] * 0.70710678118654752f));
                else if (MODE == 1) r[k] = v[k] / (1.f + __expf(-1.702f * v[k]));
                else {
                    float cdf = 0.5f * (1.f + erff(v[k] * 0.70710678118654752f));
                    float pdf = 0.3989422804014327f * __expf(-0.5f * v[k] * v[k]);
                    r[k] = g[k] * (cdf + v[k] * pdf);
                }
            }
            o[j] = pack2bf(r[0], r[1]);
        }
        reinterpret_cast<u32x4_t*>(y)[i] = o;
    }
}

__global__ void add_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ y, int64_t nchunks) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nchunks; i += (int64_t)gridDim.x * blockDim.x) {
        const u32x4_t p = reinterpret_cast<const u32x4_t*>(a)[i], q = reinterpret_cast<const u32x4_t*>(b)[i];
        u32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = pack2bf(bflo(p[j]) + bflo(q[j]), bfhi(p[j]) + bfhi(q[j]));
        reinterpret_cast<u32x4_t*>(y)[i] = o;
    }
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = f2bf(x[i]);
}
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = bf2f(x[i]);
}

// ---------------------------------------------------------------- rotary tables
// LLM M-RoPE table: pos3 [3, T] int32 (t,h,w); cos/sin [T, half] with half = head_dim/2.
// Channel i < half uses axis a(i) chosen by cumulative sections (sec_t | sec_h | sec_w), angle = pos[a(i)] * theta^(-2i/head_dim).
// The reference casts cos/sin to the activation dtype (bf16) before use; round_bf16 reproduces that.
__global__ void mrope_table_kernel(const int* __restrict__ pos3, float* __restrict__ cosb, float* __restrict__ sinb, int T, int half,
                                   int sec_t, int sec_h, float theta, int round_bf16) {
    const int64_t total = (int64_t)T * half;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(idx / half), i = (int)(idx - (int64_t)t * half);
        const int axis = (i < sec_t) ? 0 : ((i < sec_t + sec_h) ? 1 : 2);
        const float inv_freq = 1.0f / powf(theta, (float)(2 * i) / (float)(2 * half));
        const float ang = (float)pos3[(int64_t)axis * T + t] * inv_freq;
        float c = cosf(ang), s = sinf(ang);
        if (round_bf16) { c = bf2f(f2bf(c)); s = bf2f(f2bf(s)); }
        cosb[idx] = c; sinb[idx] = s;
    }
}

// Vision 2-D table: hw [N, 2] int32 (h, w); cos/sin [N, half], half = head_dim/2 (=40): first half/2 channels use h, next w;
// inv_freq_j = theta^(-2j/half) for j < half/2. fp32 (the reference applies vision rope in fp32).
__global__ void vision_rope_table_kernel(const int* __restrict__ hw, float* __restrict__ cosb, float* __restrict__ sinb, int N, int half,
                                         float theta) {
    const int64_t total = (int64_t)N * half;
    const int q = half >> 1;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(idx / half), i = (int)(idx - (int64_t)t * half);
        const int axis = (i < q) ? 0 : 1;
        const int j = (i < q) ? i : i - q;
        const float inv_freq = 1.0f / powf(theta, (float)(2 * j) / (float)half);
        const float ang = (float)hw[(int64_t)t * 2 + axis] * inv_freq;
        cosb[idx] = cosf(ang); sinb[idx] = sinf(ang);
    }
}

// Generic rotate-half RoPE on [T, n_heads, head_dim] slices living inside rows of `in` (row stride ld_in elements),
// written to `out` (row stride ld_out). out[j] = x[j]*c - x[j+half]*s*sgn ; out[j+half] = x[j+half]*c + x[j]*s*sgn.
// sgn=+1 forward, -1 backward (the adjoint rotation). half must be a multiple of 8 or handled scalar (vision half=40 -> 8 ok).
__global__ void rope_apply_kernel(const bf16_t* __restrict__ in, int64_t ld_in, bf16_t* __restrict__ out, int64_t ld_out,
                                  const float* __restrict__ cosb, const float* __restrict__ sinb, int T, int n_heads, int head_dim,
                                  float sgn) {
    const int half = head_dim >> 1;
    const int hc = half >> 3;  // 16-byte chunks per half
    const int64_t total = (int64_t)T * n_heads * hc;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % hc);
        const int64_t th = idx / hc;
        const int h = (int)(th % n_heads);
        const int64_t t = th / n_heads;
        const bf16_t* src = in + t * ld_in + (int64_t)h * head_dim + c * 8;
        bf16_t* dst = out + t * ld_out + (int64_t)h * head_dim + c * 8;
        const u32x4_t a = *reinterpret_cast<const u32x4_t*>(src);
        const u32x4_t b = *reinterpret_cast<const u32x4_t*>(src + half);
        const float* cp = cosb + t * half + c * 8;
        const float* sp = sinb + t * half + c * 8;
        u32x4_t oa, ob;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float c0 = cp[2 * j], c1 = cp[2 * j + 1], s0 = sp[2 * j] * sgn, s1 = sp[2 * j + 1] * sgn;
            const float a0 = bflo(a[j]), a1 = bfhi(a[j]), b0 = bflo(b[j]), b1 = bfhi(b[j]);
            oa[j] = pack2bf(a0 * c0 - b0 * s0, a1 * c1 - b1 * s1);
            ob[j] = pack2bf(b0 * c0 + a0 * s0, b1 * c1 + a1 * s1);
        }
        *reinterpret_cast<u32x4_t*>(dst) = oa;
        *reinterpret_cast<u32x4_t*>(dst + half) = ob;
    }
}

// ---------------------------------------------------------------- gathers / scatters
// out[t, :] = table[ids[t], :]
__global__ void gather_rows_kernel(const bf16_t* __restrict__ table, const int* __restrict__ ids, bf16_t* __restrict__ out, int64_t T,
                                   int cols) {
    const int nch = cols >> 3;
    const int64_t total = T * nch;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / nch; const int c = (int)(i - t * nch);
        reinterpret_cast<u32x4_t*>(out + t * cols)[c] = reinterpret_cast<const u32x4_t*>(table + (int64_t)ids[t] * cols)[c];
    }
}
// dst[idx[t], :] = src[t, :]
__global__ void scatter_rows_kernel(const bf16_t* __restrict__ src, const int* __restrict__ idx, bf16_t* __restrict__ dst, int64_t T,
                                    int cols) {
    const int nch = cols >> 3;
    const int64_t total = T * nch;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / nch; const int c = (int)(i - t * nch);
        reinterpret_cast<u32x4_t*>(dst + (int64_t)idx[t] * cols)[c] = reinterpret_cast<const u32x4_t*>(src + t * cols)[c];
    }
}
// dtable[ids[t], :] += dout[t, :]  (fp32 atomics; rows with ids[t] < 0 are skipped)
__global__ void embed_bwd_kernel(const bf16_t* __restrict__ dout, const int* __restrict__ ids, float* __restrict__ dtable, int64_t T,
                                 int cols) {
    const int64_t total = T * cols;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / cols; const int c = (int)(i - t * cols);
        const int id = ids[t];
        if (id >= 0) atomicAdd(&dtable[(int64_t)id * cols + c], bf2f(dout[i]));
    }
}

// dbias[c] += sum_r dy[r, c]
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ dy, float* __restrict__ dbias, int64_t rows, int cols,
                                                     int rows_per_block) {
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block; if (r1 > rows) r1 = rows;
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (c >= cols) return;
    float a0 = 0.f, a1 = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
        const unsigned p = *reinterpret_cast<const unsigned*>(dy + r * cols + c);
        a0 += bflo(p); a1 += bfhi(p);
    }
    atomicAdd(&dbias[c], a0); atomicAdd(&dbias[c + 1], a1);
}

// ---------------------------------------------------------------- transpose (with zero padding)
// out[c, r] = in[r, c] for r < R, c < C; out has leading dimension ld_out >= R and columns [R, ld_out) are zero-filled.
// 64 x 64 tile through LDS with 16-byte global accesses on both sides: rows are read 8 elements per lane, the transposed tile is written
// 8 elements per lane (the 2-byte-per-lane form this replaces ran at ~1 TB/s and was 4.6 % of the 7B step's kernel time).
// COLSUM: the tile's column sums are added to colsum[c] (fp32 atomics, one per column and block) - the bias gradient of a Linear taken from the pass that
// builds dY^T for its weight gradient instead of a second read of dY (tr1_colsum_accum).
template <bool COLSUM>
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ in, int64_t ld_in, bf16_t* __restrict__ out,
                                                        int64_t ld_out, int64_t R, int64_t C, float* __restrict__ colsum) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64][72];           // 144-byte rows: 16-byte aligned, conflict-free column reads
    __shared__ float csum[COLSUM ? 8 : 1][64];
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const bool vec_in = (ld_in % 8 == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = threadIdx.x + i * 256;                             // 64 rows x 8 chunks
        const int rr = idx >> 3, cc = (idx & 7) * 8;
        const int64_t r = r0 + rr, c = c0 + cc;
        u32x4_t v = {0, 0, 0, 0};
        if (r < R && c < C) {
            if (vec_in && c + 8 <= C) v = *reinterpret_cast<const u32x4_t*>(in + r * ld_in + c);
            else {
                bf16_t t[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = (c + e < C) ? in[r * ld_in + c + e] : (bf16_t)0;
                v = (u32x4_t){t[0] | ((unsigned)t[1] << 16), t[2] | ((unsigned)t[3] << 16), t[4] | ((unsigned)t[5] << 16), t[6] | ((unsigned)t[7] << 16)};
            }
        }
        *reinterpret_cast<u32x4_t*>(&tile[rr][(((cc >> 3) ^ ((rr >> 3) & 7)) << 3)]) = v;       // 16-byte chunk c of row r sits at c ^ (r / 8 % 8)
    }
    __syncthreads();
    const bool vec_out = (ld_out % 8 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        // 64 output rows (c) x 8 chunks of 8 source rows; 8 consecutive lanes write ONE output row's 128 bytes (whole lines per store instruction - the
        // round-1 map had the 64 lanes of an instruction write 16 bytes into 64 different rows).  With the chunk swizzle above the 2-byte column reads of a
        // wave (8 output rows x 8 chunks) land on 32 distinct banks.
        const int idx = threadIdx.x + i * 256;
        const int cc = idx >> 3, ch = idx & 7, rr = ch * 8;
        const int64_t c = c0 + cc, r = r0 + rr;
        if (COLSUM) csum[ch][cc] = 0.f;
        if (c >= C || r >= ld_out) continue;
        bf16_t t[8];
        const int col = (((cc >> 3) ^ ch) << 3) | (cc & 7);                // (rows rr .. rr + 7 share r / 8 % 8 == ch)
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = tile[rr + e][col];              // rows >= R were zero-filled on load: they are the padding
        if (COLSUM) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) a += bf2f(t[e]);
            csum[ch][cc] = a;
        }
        if (vec_out && r + 8 <= ld_out) {
            const u32x4_t v = {t[0] | ((unsigned)t[1] << 16), t[2] | ((unsigned)t[3] << 16), t[4] | ((unsigned)t[5] << 16), t[6] | ((unsigned)t[7] << 16)};
            *reinterpret_cast<u32x4_t*>(out + c * ld_out + r) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (r + e < ld_out) out[c * ld_out + r + e] = t[e];
        }
    }
    if (COLSUM) {
        __syncthreads();
        if (threadIdx.x < 64 && c0 + threadIdx.x < C && r0 < R) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) a += csum[j][threadIdx.x];
            atomicAdd(&colsum[c0 + threadIdx.x], a);
        }
    }
}

// ---------------------------------------------------------------- C ABI
#define EW_GRID(n) tr1_grid_1d((n), 256, 4096)

extern "C" int tr1_swiglu_fwd(const void* gu, void* out, int64_t rows, int64_t inter, void* stream) {
    TR1_CHECK_ARG(inter % 8 == 0, "swiglu: inter must be a multiple of 8");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(EW_GRID(rows * inter / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gu,
                       (bf16_t*)out, rows, (int)inter);
    TR1_LAUNCH_CHECK();
}
extern "C" int tr1_swiglu_bwd(const void* dout, const void* gu, void* dgu, int64_t rows, int64_t inter, void* stream) {
    TR1_CHECK_ARG(inter % 8 == 0, "swiglu_bwd: inter must be a multiple of 8");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(EW_GRID(rows * inter / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dout,
                       (const bf16_t*)gu, (bf16_t*)dgu, rows, (int)inter);
    TR1_LAUNCH_CHECK();
}
extern "C" int tr1_gelu_fwd(const void* x, void* y, int64_t n, void* stream) {
    TR1_CHECK_ARG(n % 8 == 0, "gelu: n must be a multiple of 8");
    if (n == 0) return 0;
    hipLaunchKernelGGL(act_kernel<0>, dim3(EW_GRID(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)nullptr,
                       (bf16_t*)y, n / 8);
    TR1_LAUNCH_CHECK();
}
extern "C" int tr1_quickgelu_fwd(const void* x, void* y, int64_t n, void* stream) {
    TR1_CHECK_ARG(n % 8 == 0, "quickgelu: n must be a multiple of 8");
    if (n == 0) return 0;
    hipLaunchKernelGGL(act_kernel<1>, dim3(EW_GRID(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)nullptr,
                       (bf16_t*)y, n / 8);
    TR1_LAUNCH_CHECK();
}
extern "C" int tr1_gelu_bwd(const void* x, const void* dy, void* dx, int64_t n, void* stream) {
    TR1_CHECK_ARG(n % 8 == 0, "gelu_bwd: n must be a multiple of 8");
    if (n == 0) return 0;
    hipLaunchKernelGGL(act_kernel<2>, dim3(EW_GRID(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)dy,
                       (bf16_t*)dx, n / 8);
    TR1_LAUNCH_CHECK();
}
extern "C" int tr1_add_bf16(const void* a, const void* b, void* y, int64_t n, void* stream) {
    TR1_CHECK_ARG(n % 8 == 0, "add: n must be a multiple of 8");
    if (n == 0) return 0;
    hipLaunchKernelGGL(add_kernel, dim3(EW_GRID(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y,
                       n / 8);
    TR1_LAUNCH_CHECK();
}
extern "C" int tr1_cast_f32_to_bf16(const void* x, void* y, int64_t n, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(EW_GRID(n)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (bf16_t*)y, n);
    TR1_LAUNCH_CHECK();
}
extern "C" int tr1_cast_bf16_to_f32(const void* x, void* y, int64_t n, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(EW_GRID(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (float*)y, n);
    TR1_LAUNCH_CHECK();
}
extern "C" int tr1_mrope_table(const void* pos3, void* cosb, void* sinb, int64_t T, int64_t head_dim, int64_t sec_t, int64_t sec_h,
                               int64_t sec_w, float theta, int round_bf16, void* stream) {
    TR1_CHECK_ARG(head_dim % 2 == 0 && sec_t + sec_h + sec_w == head_dim / 2, "mrope_table: sections must sum to head_dim/2");
    if (T == 0) return 0;
    hipLaunchKernelGGL(mrope_table_kernel, dim3(EW_GRID(T * head_dim / 2)), dim3(256), 0, (hipStream_t)stream, (const int*)pos3,
                       (float*)cosb, (float*)sinb, (int)T, (int)(head_dim / 2), (int)sec_t, (int)sec_h, theta, round_bf16);
    TR1_LAUNCH_CHECK();
}
extern "C" int tr1_vision_rope_table(const void* hw, void* cosb, void* sinb, int64_t N, int64_t head_dim, float theta, void* stream) {
    TR1_CHECK_ARG(head_dim % 4 == 0, "vision_rope_table: head_dim must be a multiple of 4");
    if (N == 0) return 0;
    hipLaunchKernelGGL(vision_rope_table_kernel, dim3(EW_GRID(N * head_dim / 2)), dim3(256), 0, (hipStream_t)stream, (const int*)hw,
                       (float*)cosb, (float*)sinb, (int)N, (int)(head_dim / 2), theta);
    TR1_LAUNCH_CHECK();
}
extern "C" int tr1_rope_apply(const void* in, int64_t ld_in, void* out, int64_t ld_out, const void* cosb, const void* sinb, int64_t T,
                              int64_t n_heads, int64_t head_dim, int backward, void* stream) {
    TR1_CHECK_ARG(head_dim % 16 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0, "rope_apply: head_dim%16, ld%8 required");
    if (T == 0) return 0;
    hipLaunchKernelGGL(rope_apply_kernel, dim3(EW_GRID(T * n_heads * head_dim / 16)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in,
                       ld_in, (bf16_t*)out, ld_out, (const float*)cosb, (const float*)sinb, (int)T, (int)n_heads, (int)head_dim,
                       backward ? -1.f : 1.f);
    TR1_LAUNCH_CHECK();
}
extern "C" int tr1_gather_rows(const void* table, const void* ids, void* out, int64_t T, int64_t cols, void* stream) {
    TR1_CHECK_ARG(cols % 8 == 0, "gather_rows: cols must be a multiple of 8");
    if (T == 0) return 0;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(EW_GRID(T * cols / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)table,
                       (const int*)ids, (bf16_t*)out, T, (int)cols);
    TR1_LAUNCH_CHECK();
}
extern "C" int tr1_scatter_rows(const void* src, const void* idx, void* dst, int64_t T, int64_t cols, void* stream) {
    TR1_CHECK_ARG(cols % 8 == 0, "scatter_rows: cols must be a multiple of 8");
    if (T == 0) return 0;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(EW_GRID(T * cols / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                       (const int*)idx, (bf16_t*)dst, T, (int)cols);
    TR1_LAUNCH_CHECK();
}
extern "C" int tr1_embed_bwd(const void* dout, const void* ids, void* dtable_f32, int64_t T, int64_t cols, void* stream) {
    if (T == 0) return 0;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(EW_GRID(T * cols)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dout, (const int*)ids,
                       (float*)dtable_f32, T, (int)cols);
    TR1_LAUNCH_CHECK();
}
extern "C" int tr1_colsum_accum(const void* dy, void* dbias_f32, int64_t rows, int64_t cols, void* stream) {
    TR1_CHECK_ARG(cols % 2 == 0, "colsum: cols must be even");
    if (rows == 0) return 0;
    const int rpb = 128;
    dim3 grid((unsigned)((cols / 2 + 255) / 256), (unsigned)((rows + rpb - 1) / rpb));
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (float*)dbias_f32, rows, (int)cols, rpb);
    TR1_LAUNCH_CHECK();
}
extern "C" int tr1_transpose_bf16(const void* in, int64_t ld_in, void* out, int64_t ld_out, int64_t R, int64_t C, void* stream) {
    TR1_CHECK_ARG(ld_out >= R && ld_in >= C, "transpose: bad leading dimensions");
    if (R == 0 || C == 0) return 0;
    dim3 grid((unsigned)((C + 63) / 64), (unsigned)((ld_out + 63) / 64));
    hipLaunchKernelGGL(transpose_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, ld_in, (bf16_t*)out, ld_out, R, C, (float*)nullptr);
    TR1_LAUNCH_CHECK();
}
// The same transpose, and colsum_f32[c] += sum_r in[r, c] (the bias gradient rides on the pass that builds dY^T)
extern "C" int tr1_transpose_colsum_bf16(const void* in, int64_t ld_in, void* out, int64_t ld_out, int64_t R, int64_t C, void* colsum_f32, void* stream) {
    TR1_CHECK_ARG(ld_out >= R && ld_in >= C && colsum_f32, "transpose_colsum: bad leading dimensions / missing colsum");
    if (R == 0 || C == 0) return 0;
    dim3 grid((unsigned)((C + 63) / 64), (unsigned)((ld_out + 63) / 64));
    hipLaunchKernelGGL(transpose_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, ld_in, (bf16_t*)out, ld_out, R, C, (float*)colsum_f32);
    TR1_LAUNCH_CHECK();
}

// ---------------------------------------------------------------- fused decode post-projection
// One launch per decode layer-step instead of five: M-RoPE on the new token's q and k heads, append k to the K cache row
// `slots[r]`, scatter v into the transposed V cache column `slots[r]` (reference: Qwen2VLAttention.forward TF:521-556 + DynamicCache.update).
// qkv: [R, (n_heads + 2*n_kv)*hd]; q_out: [R, n_heads*hd]; kcache: [slots, n_kv*hd]; vtcache: [n_kv*hd, vt_ld].
__global__ void decode_qkv_post_kernel(const bf16_t* __restrict__ qkv, int64_t ld, const float* __restrict__ cosb, const float* __restrict__ sinb,
                                       bf16_t* __restrict__ q_out, int64_t ld_q, bf16_t* __restrict__ kcache, int64_t k_ld,
                                       bf16_t* __restrict__ vtcache, int64_t vt_ld, const int* __restrict__ slots, int R, int n_heads, int n_kv,
                                       int hd) {
    const int half = hd >> 1, hc = half >> 3;
    const int heads_total = n_heads + 2 * n_kv;
    const int64_t total = (int64_t)R * heads_total * hc;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % hc);
        const int64_t rh = idx / hc;
        const int h = (int)(rh % heads_total);
        const int r = (int)(rh / heads_total);
        const bf16_t* src = qkv + (int64_t)r * ld + (int64_t)h * hd + c * 8;
        const u32x4_t a = *reinterpret_cast<const u32x4_t*>(src);
        const u32x4_t b = *reinterpret_cast<const u32x4_t*>(src + half);
        const int slot = slots[r];
        if (h < n_heads + n_kv) {
            const float* cp = cosb + (int64_t)r * half + c * 8;
            const float* sp = sinb + (int64_t)r * half + c * 8;
            u32x4_t oa, ob;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float c0 = cp[2 * j], c1 = cp[2 * j + 1], s0 = sp[2 * j], s1 = sp[2 * j + 1];
                const float a0 = bflo(a[j]), a1 = bfhi(a[j]), b0 = bflo(b[j]), b1 = bfhi(b[j]);
                float x0, y0, x1, y1;
                rope_pair(a0, b0, c0, s0, x0, y0);
                rope_pair(a1, b1, c1, s1, x1, y1);
                oa[j] = pack2bf(x0, x1);
                ob[j] = pack2bf(y0, y1);
            }
            bf16_t* dst = (h < n_heads) ? (q_out + (int64_t)r * ld_q + (int64_t)h * hd + c * 8)
                                        : (kcache + (int64_t)slot * k_ld + (int64_t)(h - n_heads) * hd + c * 8);
            *reinterpret_cast<u32x4_t*>(dst) = oa;
            *reinterpret_cast<u32x4_t*>(dst + half) = ob;
        } else {
            const int kvh = h - n_heads - n_kv;
            bf16_t* col = vtcache + ((int64_t)kvh * hd + c * 8) * vt_ld + slot;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                col[(int64_t)(2 * j) * vt_ld] = (bf16_t)(a[j] & 0xffffu);
                col[(int64_t)(2 * j + 1) * vt_ld] = (bf16_t)(a[j] >> 16);
                col[(int64_t)(half + 2 * j) * vt_ld] = (bf16_t)(b[j] & 0xffffu);
                col[(int64_t)(half + 2 * j + 1) * vt_ld] = (bf16_t)(b[j] >> 16);
            }
        }
    }
}

extern "C" int tr1_decode_qkv_post(const void* qkv, int64_t ld, const void* cosb, const void* sinb, void* q_out, int64_t ld_q, void* kcache,
                                   int64_t k_ld, void* vtcache, int64_t vt_ld, const void* slots, int64_t R, int64_t n_heads, int64_t n_kv,
                                   int64_t head_dim, void* stream) {
    TR1_CHECK_ARG(head_dim % 16 == 0 && ld % 8 == 0 && ld_q % 8 == 0 && k_ld % 8 == 0, "decode_qkv_post: head_dim%16 and ld%8 required");
    if (R == 0) return 0;
    const int64_t total = R * (n_heads + 2 * n_kv) * (head_dim / 16);
    hipLaunchKernelGGL(decode_qkv_post_kernel, dim3(tr1_grid_1d(total, 256, 1024)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, ld,
                       (const float*)cosb, (const float*)sinb, (bf16_t*)q_out, ld_q, (bf16_t*)kcache, k_ld, (bf16_t*)vtcache, vt_ld, (const int*)slots,
                       (int)R, (int)n_heads, (int)n_kv, (int)head_dim);
    TR1_LAUNCH_CHECK();
}
