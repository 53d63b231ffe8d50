// Fused AdamW over ONE flat parameter arena (fp32 master / m / v / grad, bf16 working copy) + global grad-norm clip.
// Replaces DeepSpeed's FusedAdam / DeepSpeedCPUAdam on the reference path (scripts/zero3.json:13-21, zero3_offload.json:24-31);
// update rule = torch.optim.AdamW (decoupled weight decay, bias correction), HF defaults lr/betas/eps, max_grad_norm 1.0.
// HBM-bound: 16 B read (g, p, m, v) + 14 B written (p, m, v, bf16 copy) [+4 B when the gradient is zeroed in place] per parameter.
#include "tr1_common.h"

// every array is streamed exactly once per step (258 GB for the 7B arena): nt loads/stores keep it from turning over L2 / the memory-side cache
#ifndef TR1_OPT_NT
#define TR1_OPT_NT 1
#endif
#if TR1_OPT_NT
#define OPT_LD(ptr) __builtin_nontemporal_load(ptr)
#define OPT_ST(val, ptr) __builtin_nontemporal_store(val, ptr)
#else
#define OPT_LD(ptr) (*(ptr))
#define OPT_ST(val, ptr) (*(ptr) = (val))
#endif

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4_t v = OPT_LD(reinterpret_cast<const f32x4_t*>(g) + i);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0) for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) s += g[i] * g[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(out, s);
}

// the same sum over a bf16 array (data-parallel runs: the all-reduced gradient stays in its bf16 wire buffer and is consumed from there)
__global__ __launch_bounds__(256) void sumsq_bf16_kernel(const bf16_t* __restrict__ g, int64_t n, float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    const int64_t n8 = n >> 3;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const u32x4_t v = OPT_LD(reinterpret_cast<const u32x4_t*>(g) + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float a = bflo(v[e]), b = bfhi(v[e]); s += a * a + b * b; }
    }
    if (blockIdx.x == 0) for (int64_t i = n8 * 8 + threadIdx.x; i < n; i += blockDim.x) { const float a = bf2f(g[i]); s += a * a; }
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(out, s);
}

// sumsq: device scalar with sum of squared grads (already multiplied by nothing); clip coefficient derived in-kernel so the
// step needs no host round trip:  coef = grad_mult * min(1, max_norm / (grad_mult*sqrt(sumsq) + 1e-6)).
// G16: the gradient is read from a bf16 array `g16` (the all-reduced wire buffer); `g` (fp32 accumulator) is then only zeroed, never read.
template <bool G16>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, float* __restrict__ g,
                                                    bf16_t* __restrict__ p16, int64_t n, float lr, float beta1, float beta2, float eps,
                                                    float wd, float bc1, float bc2_sqrt, const float* __restrict__ sumsq, float max_norm,
                                                    float grad_mult, int zero_grad, const bf16_t* __restrict__ g16) {
    float coef = grad_mult;
    if (sumsq && max_norm > 0.f) {
        const float norm = sqrtf(*sumsq) * grad_mult;
        coef = grad_mult * fminf(1.f, max_norm / (norm + 1e-6f));
    }
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4_t pv = OPT_LD(reinterpret_cast<f32x4_t*>(p) + i), mv = OPT_LD(reinterpret_cast<f32x4_t*>(m) + i), vv = OPT_LD(reinterpret_cast<f32x4_t*>(v) + i);
        f32x4_t gv;
        if (G16) { const u32x2_t w = OPT_LD(reinterpret_cast<const u32x2_t*>(g16) + i); gv = (f32x4_t){bflo(w[0]), bfhi(w[0]), bflo(w[1]), bfhi(w[1])}; }
        else gv = OPT_LD(reinterpret_cast<f32x4_t*>(g) + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gg = gv[j] * coef;
            pv[j] *= (1.f - lr * wd);
            mv[j] = beta1 * mv[j] + (1.f - beta1) * gg;
            vv[j] = beta2 * vv[j] + (1.f - beta2) * gg * gg;
            const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
            pv[j] -= (lr / bc1) * (mv[j] / denom);
        }
        OPT_ST(pv, reinterpret_cast<f32x4_t*>(p) + i); OPT_ST(mv, reinterpret_cast<f32x4_t*>(m) + i); OPT_ST(vv, reinterpret_cast<f32x4_t*>(v) + i);
        const u32x2_t w = {pack2bf(pv[0], pv[1]), pack2bf(pv[2], pv[3])};
        OPT_ST(w, reinterpret_cast<u32x2_t*>(p16) + i);
        if (zero_grad) OPT_ST(((f32x4_t){0.f, 0.f, 0.f, 0.f}), reinterpret_cast<f32x4_t*>(g) + i);
    }
    if (blockIdx.x == 0) {
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
            const float gg = (G16 ? bf2f(g16[i]) : g[i]) * coef;
            float pv = p[i] * (1.f - lr * wd);
            const float mv = beta1 * m[i] + (1.f - beta1) * gg, vv = beta2 * v[i] + (1.f - beta2) * gg * gg;
            pv -= (lr / bc1) * (mv / (sqrtf(vv) / bc2_sqrt + eps));
            p[i] = pv; m[i] = mv; v[i] = vv; p16[i] = f2bf(pv);
            if (zero_grad) g[i] = 0.f;
        }
    }
}

extern "C" int tr1_sumsq_accum(const void* g, int64_t n, void* out_scalar, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(sumsq_kernel, dim3(tr1_grid_1d(n / 4 + 1, 256, 2048)), dim3(256), 0, (hipStream_t)stream, (const float*)g, n, (float*)out_scalar);
    TR1_LAUNCH_CHECK();
}

extern "C" int tr1_adamw_step(void* p_f32, void* m_f32, void* v_f32, void* g_f32, void* p_bf16, int64_t n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int64_t step, const void* sumsq_scalar, float max_norm, float grad_mult,
                              int zero_grad, void* stream) {
    TR1_CHECK_ARG(step >= 1, "adamw: step counts from 1");
    if (n == 0) return 0;
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    hipLaunchKernelGGL(adamw_kernel<false>, dim3(tr1_grid_1d(n / 4 + 1, 256, 4096)), dim3(256), 0, (hipStream_t)stream, (float*)p_f32, (float*)m_f32,
                       (float*)v_f32, (float*)g_f32, (bf16_t*)p_bf16, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt,
                       (const float*)sumsq_scalar, max_norm, grad_mult, zero_grad, (const bf16_t*)nullptr);
    TR1_LAUNCH_CHECK();
}

// Data-parallel form: the gradient SUM over ranks is read from the bf16 wire buffer `g_bf16` (no bf16 -> fp32 copy-back pass); the fp32
// accumulator `g_f32` is only zeroed (zero_grad) for the next window.
extern "C" int tr1_adamw_step_g16(void* p_f32, void* m_f32, void* v_f32, void* g_f32, const void* g_bf16, void* p_bf16, int64_t n, float lr, float beta1,
                                  float beta2, float eps, float weight_decay, int64_t step, const void* sumsq_scalar, float max_norm, float grad_mult,
                                  int zero_grad, void* stream) {
    TR1_CHECK_ARG(step >= 1 && g_bf16, "adamw_g16: step counts from 1, bf16 gradient required");
    if (n == 0) return 0;
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    hipLaunchKernelGGL(adamw_kernel<true>, dim3(tr1_grid_1d(n / 4 + 1, 256, 4096)), dim3(256), 0, (hipStream_t)stream, (float*)p_f32, (float*)m_f32,
                       (float*)v_f32, (float*)g_f32, (bf16_t*)p_bf16, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt,
                       (const float*)sumsq_scalar, max_norm, grad_mult, zero_grad, (const bf16_t*)g_bf16);
    TR1_LAUNCH_CHECK();
}

extern "C" int tr1_sumsq_accum_bf16(const void* g_bf16, int64_t n, void* out_scalar, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(sumsq_bf16_kernel, dim3(tr1_grid_1d(n / 8 + 1, 256, 2048)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g_bf16, n, (float*)out_scalar);
    TR1_LAUNCH_CHECK();
}

// g[base + l * stride + r] = 0 for l < count and r in the (at most 8) half-open ranges rel[2i], rel[2i+1]: the SMALL per-layer tensors of the gradient arena
// (norm weights, biases, padding).  The weight-gradient GEMMs OVERWRITE the large matrices on the first micro-step of every accumulation window, so the
// optimizer no longer zeroes those (4 of its 34 bytes per parameter); what still accumulates from zero is cleared by this launch (a few KB per layer).
struct ZeroRanges { int64_t a[8], b[8]; int n; };
__global__ __launch_bounds__(256) void zero_periodic_kernel(float* __restrict__ g, int64_t base, int64_t stride, ZeroRanges zr) {
    float* gl = g + base + (int64_t)blockIdx.y * stride;
    for (int r = 0; r < zr.n; ++r)
        for (int64_t i = zr.a[r] + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < zr.b[r]; i += (int64_t)gridDim.x * blockDim.x) gl[i] = 0.f;
}
extern "C" int tr1_zero_ranges_periodic(void* g_f32, int64_t base, int64_t stride, int64_t count, const int64_t* rel_ranges, int64_t n_ranges, void* stream) {
    TR1_CHECK_ARG(n_ranges >= 0 && n_ranges <= 8 && (n_ranges == 0 || rel_ranges), "zero_ranges_periodic: at most 8 ranges");
    if (count <= 0 || n_ranges == 0) return 0;
    ZeroRanges zr; zr.n = (int)n_ranges;
    int64_t longest = 1;
    for (int i = 0; i < zr.n; ++i) {
        zr.a[i] = rel_ranges[2 * i]; zr.b[i] = rel_ranges[2 * i + 1];
        TR1_CHECK_ARG(zr.a[i] >= 0 && zr.b[i] >= zr.a[i] && zr.b[i] <= stride, "zero_ranges_periodic: range outside the period");
        if (zr.b[i] - zr.a[i] > longest) longest = zr.b[i] - zr.a[i];
    }
    const unsigned gx = (unsigned)((longest + 255) / 256 < 64 ? (longest + 255) / 256 : 64);
    hipLaunchKernelGGL(zero_periodic_kernel, dim3(gx, (unsigned)count), dim3(256), 0, (hipStream_t)stream, (float*)g_f32, base, stride, zr);
    TR1_LAUNCH_CHECK();
}

// out += sum of partials[0 .. n) in a FIXED order (one block: thread t adds partials[t], partials[t + 1024], ...; then the block tree) - the per-wave sums of
// squares the weight-gradient epilogues left (tr1_wgrad_f32_sumsq)
// two fixed-order levels (a single 1024-thread block took 0.33 ms for the 7B step's 720 k partials): level 1 - block b sums its contiguous share and leaves
// the result in scratch[b]; level 2 - one wave-sized block adds the <= 256 block sums to `out`
__global__ __launch_bounds__(256) void sumsq_partials_kernel(const float* __restrict__ part, int64_t n, float* __restrict__ scratch) {
    __shared__ float red[16];
    const int64_t per = (n + gridDim.x - 1) / gridDim.x, a = (int64_t)blockIdx.x * per, b = a + per < n ? a + per : n;
    float s = 0.f;
    for (int64_t i = a + threadIdx.x; i < b; i += 256) s += part[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) scratch[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sumsq_partials_final_kernel(const float* __restrict__ scratch, int nb, float* __restrict__ out) {
    __shared__ float red[16];
    float s = threadIdx.x < nb ? scratch[threadIdx.x] : 0.f;
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(out, s);
}
// partials_f32 must have room for 256 more floats behind its n entries (scratch of the first level)
extern "C" int tr1_sumsq_partials_accum(const void* partials_f32, int64_t n, void* out_scalar, void* stream) {
    if (n == 0) return 0;
    const int nb = (int)(n < 256 * 256 ? (n + 255) / 256 : 256);
    float* scratch = (float*)partials_f32 + n;
    hipLaunchKernelGGL(sumsq_partials_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const float*)partials_f32, n, scratch);
    hipLaunchKernelGGL(sumsq_partials_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)scratch, nb, (float*)out_scalar);
    TR1_LAUNCH_CHECK();
}

// out += sum of g[base + l * stride + r]^2 over l < count and r in the (<= 8) half-open ranges: the SMALL per-layer gradient tensors next to the large matrices
// whose squared norm came out of the weight-gradient epilogues
__global__ __launch_bounds__(256) void sumsq_periodic_kernel(const float* __restrict__ g, int64_t base, int64_t stride, ZeroRanges zr, float* __restrict__ out) {
    __shared__ float red[16];
    const float* gl = g + base + (int64_t)blockIdx.y * stride;
    float s = 0.f;
    for (int r = 0; r < zr.n; ++r)
        for (int64_t i = zr.a[r] + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < zr.b[r]; i += (int64_t)gridDim.x * blockDim.x) { const float v = gl[i]; s += v * v; }
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(out, s);
}
extern "C" int tr1_sumsq_ranges_periodic(const void* g_f32, int64_t base, int64_t stride, int64_t count, const int64_t* rel_ranges, int64_t n_ranges, void* out_scalar,
                                         void* stream) {
    TR1_CHECK_ARG(n_ranges >= 0 && n_ranges <= 8 && (n_ranges == 0 || rel_ranges) && out_scalar, "sumsq_ranges_periodic: at most 8 ranges");
    if (count <= 0 || n_ranges == 0) return 0;
    ZeroRanges zr; zr.n = (int)n_ranges;
    int64_t longest = 1;
    for (int i = 0; i < zr.n; ++i) {
        zr.a[i] = rel_ranges[2 * i]; zr.b[i] = rel_ranges[2 * i + 1];
        TR1_CHECK_ARG(zr.a[i] >= 0 && zr.b[i] >= zr.a[i] && zr.b[i] <= stride, "sumsq_ranges_periodic: range outside the period");
        if (zr.b[i] - zr.a[i] > longest) longest = zr.b[i] - zr.a[i];
    }
    const unsigned gx = (unsigned)((longest + 255) / 256 < 64 ? (longest + 255) / 256 : 64);
    hipLaunchKernelGGL(sumsq_periodic_kernel, dim3(gx, (unsigned)count), dim3(256), 0, (hipStream_t)stream, (const float*)g_f32, base, stride, zr, (float*)out_scalar);
    TR1_LAUNCH_CHECK();
}
