// Vocabulary-side kernels of the GRPO step for gfx950: per-token log-prob + entropy over the lm_head logits (forward and
// backward), the GRPO loss/gradient on [G, C] token grids, and the rollout sampler (temperature + top-k + inverse CDF, Philox).
//
// Reference semantics:
//   logp / entropy : src/time_r1/rl/timer1_trainer.py:458-481  (log_softmax, gather, H = -sum p log p)
//   KL (k3)        : :635-639        loss (both branches): :713-737       sampling: HF generate(do_sample, temperature, top_k)
#include "tr1_common.h"

// ---------------------------------------------------------------------------------------------------------------------
// One 256-thread block per row; single pass, online (max, sum exp, sum exp*x).  logits bf16 [R, V] with row stride ld.
// ---------------------------------------------------------------------------------------------------------------------
struct OnlineSE { float m, z, s; };  // running max, sum e^(x-m), sum e^(x-m)*x
TR1_DEV void ose_add(OnlineSE& a, float x) {
    if (x > a.m) { const float f = __expf(a.m - x); a.z = a.z * f + 1.f; a.s = a.s * f + x; a.m = x; }
    else { const float e = __expf(x - a.m); a.z += e; a.s += e * x; }
}
TR1_DEV void ose_merge(OnlineSE& a, const OnlineSE& b) {
    const float m = fmaxf(a.m, b.m);
    if (m == -INFINITY) return;
    const float fa = __expf(a.m - m), fb = __expf(b.m - m);
    a.z = a.z * fa + b.z * fb; a.s = a.s * fa + b.s * fb; a.m = m;
}

__global__ __launch_bounds__(256) void logp_entropy_fwd_kernel(const bf16_t* __restrict__ logits, int64_t ld, const int* __restrict__ targets,
                                                               float* __restrict__ logp, float* __restrict__ entropy, float* __restrict__ lse_out,
                                                               int V) {
    __shared__ float sm[3][4];
    const int r = blockIdx.x;
    const bf16_t* row = logits + (int64_t)r * ld;
    OnlineSE a = {-INFINITY, 0.f, 0.f};
    const int nch = V >> 3;
    for (int c = threadIdx.x; c < nch; c += 256) {
        const u32x4_t p = *reinterpret_cast<const u32x4_t*>(row + c * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) { ose_add(a, bflo(p[j])); ose_add(a, bfhi(p[j])); }
    }
    for (int i = nch * 8 + threadIdx.x; i < V; i += 256) ose_add(a, bf2f(row[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        OnlineSE b = {__shfl_xor(a.m, o, 64), __shfl_xor(a.z, o, 64), __shfl_xor(a.s, o, 64)};
        ose_merge(a, b);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sm[0][wave] = a.m; sm[1][wave] = a.z; sm[2][wave] = a.s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        OnlineSE t = {sm[0][0], sm[1][0], sm[2][0]};
        for (int w = 1; w < 4; ++w) { OnlineSE b = {sm[0][w], sm[1][w], sm[2][w]}; ose_merge(t, b); }
        const float lse = t.m + __logf(t.z);
        const int tg = targets[r];
        logp[r] = bf2f(row[tg]) - lse;
        if (entropy) entropy[r] = lse - t.s / t.z;   // H = lse - E_p[x]
        if (lse_out) lse_out[r] = lse;
    }
}

// dlogits[r, v] = dlogp[r] * (1[v == target] - exp(x - lse))     (bf16 out, same shape/stride as logits; may alias logits)
__global__ void logp_bwd_kernel(const bf16_t* __restrict__ logits, int64_t ld, const int* __restrict__ targets, const float* __restrict__ lse,
                                const float* __restrict__ dlogp, bf16_t* __restrict__ dlogits, int64_t ld_out, int V) {
    const int r = blockIdx.y;
    const float g = dlogp[r], ls = lse[r];
    const int tg = targets[r];
    const bf16_t* row = logits + (int64_t)r * ld;
    bf16_t* orow = dlogits + (int64_t)r * ld_out;
    const int nch = V >> 3;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < nch; c += gridDim.x * blockDim.x) {
        const u32x4_t p = *reinterpret_cast<const u32x4_t*>(row + c * 8);
        u32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int v0 = c * 8 + 2 * j;
            const float a = g * ((v0 == tg ? 1.f : 0.f) - __expf(bflo(p[j]) - ls));
            const float b = g * ((v0 + 1 == tg ? 1.f : 0.f) - __expf(bfhi(p[j]) - ls));
            o[j] = pack2bf(a, b);
        }
        *reinterpret_cast<u32x4_t*>(orow + c * 8) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// GRPO loss + d loss / d logp on the [G, C] grid (one block; G*C is a few thousand elements).
//   mode 1 (use_grpo): l = -(rho*A - beta*kl), loss = mean_g( sum_t l*m / sum_t m )
//   mode 0 (clip)    : l = -min(rho*A, clamp(rho,1-el,1+eh)*A) + beta*kl, loss = sum l*m / sum m
// rho = exp(logp - logp.detach()) == 1 in value; its gradient wrt logp is 1 (and the min/clamp pair passes the full
// gradient at rho == 1), so dl/dlogp = -A + beta*(1 - exp(ref - logp)).
// out[0] = loss, out[1] = mean_g(masked-mean kl), out[2] = sum of mask; row_len[g] = sum_t m.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grpo_loss_kernel(const float* __restrict__ logp, const float* __restrict__ ref_logp,
                                                        const int* __restrict__ mask, const float* __restrict__ adv, float* __restrict__ dlogp,
                                                        float* __restrict__ out, float* __restrict__ row_len, float* __restrict__ row_kl, int G,
                                                        int C, float beta, int use_grpo, float grad_scale) {
    __shared__ float red[16];
    __shared__ float s_len[64], s_l[64], s_kl[64];
    float tot_mask = 0.f;
    for (int g = 0; g < G; ++g) {
        float len = 0.f, sl = 0.f, skl = 0.f;
        for (int t = threadIdx.x; t < C; t += blockDim.x) {
            const int i = g * C + t;
            const float m = (float)mask[i];
            float kl = 0.f;
            if (ref_logp) { const float d = ref_logp[i] - logp[i]; kl = __expf(d) - d - 1.f; }
            const float l = -adv[g] + beta * kl;
            len += m; sl += l * m; skl += kl * m;
        }
        len = block_sum(len, red); sl = block_sum(sl, red); skl = block_sum(skl, red);
        if (threadIdx.x == 0) { s_len[g] = len; s_l[g] = sl; s_kl[g] = skl; row_len[g] = len; if (row_kl) row_kl[g] = skl; }
        tot_mask += len;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < G * C; i += blockDim.x) {
        const int g = i / C;
        const float m = (float)mask[i];
        float dkl = 0.f;
        if (ref_logp) dkl = 1.f - __expf(ref_logp[i] - logp[i]);
        const float w = use_grpo ? (m / s_len[g] / (float)G) : (m / tot_mask);
        dlogp[i] = (m > 0.f) ? (-adv[g] + beta * dkl) * w * grad_scale : 0.f;
    }
    if (threadIdx.x == 0) {
        float loss = 0.f, klm = 0.f, num = 0.f;
        for (int g = 0; g < G; ++g) {
            if (use_grpo) loss += s_l[g] / s_len[g] / (float)G; else num += s_l[g];
            klm += s_kl[g] / s_len[g] / (float)G;
        }
        if (!use_grpo) loss = num / tot_mask;
        out[0] = loss; out[1] = klm; out[2] = tot_mask;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Sampler: one 1024-thread block per rollout row.  x = logits / temperature; keep the top_k largest (ties with the k-th
// value kept, as HF's TopKLogitsWarper does); softmax over the kept set; inverse-CDF draw in vocabulary order with a
// Philox4x32-10 uniform keyed by (seed, row, step).  top_k <= 0 disables the filter.
// ---------------------------------------------------------------------------------------------------------------------
TR1_DEV unsigned mulhi32(unsigned a, unsigned b) { return __umulhi(a, b); }
TR1_DEV void philox4x32_10(unsigned c[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const unsigned h0 = mulhi32(0xD2511F53u, c[0]), l0 = 0xD2511F53u * c[0];
        const unsigned h1 = mulhi32(0xCD9E8D57u, c[2]), l1 = 0xCD9E8D57u * c[2];
        const unsigned n0 = h1 ^ c[1] ^ k0, n1 = l1, n2 = h0 ^ c[3] ^ k1, n3 = l0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
TR1_DEV unsigned f2key(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }  // order preserving
TR1_DEV float key2f(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__global__ __launch_bounds__(1024) void sample_kernel(const bf16_t* __restrict__ logits, int64_t ld, int V, float inv_temp, int top_k,
                                                      unsigned long long seed, const int* __restrict__ step_ptr, int* __restrict__ tokens,
                                                      int64_t tok_ld, int* __restrict__ finished, int eos_id, int pad_id, int stop_at_eos,
                                                      float* __restrict__ u_out) {
    __shared__ unsigned hist[256];
    __shared__ float red[16];
    __shared__ float chunk_sum[1024];
    __shared__ unsigned s_prefix; __shared__ int s_remaining; __shared__ int s_token;
    const int r = blockIdx.x, tid = threadIdx.x;
    const int step = step_ptr ? *step_ptr : 0;
    int* tok_out = tokens + (int64_t)r * tok_ld + step;
    if (finished && finished[r] && stop_at_eos) { if (tid == 0) *tok_out = pad_id; return; }
    const bf16_t* row = logits + (int64_t)r * ld;

    // ---- radix select of the k-th largest key (MSB first)
    unsigned thr_key = 0u;  // keep everything
    if (top_k > 0 && top_k < V) {
        if (tid == 0) { s_prefix = 0u; s_remaining = top_k; }
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            if (tid < 256) hist[tid] = 0u;
            __syncthreads();
            const unsigned prefix = s_prefix;
            const unsigned pmask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
            for (int i = tid; i < V; i += 1024) {
                const unsigned k = f2key(bf2f(row[i]) * inv_temp);
                if ((k & pmask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                int rem = s_remaining; int b = 255;
                for (; b > 0; --b) { if ((int)hist[b] >= rem) break; rem -= (int)hist[b]; }
                s_prefix = prefix | ((unsigned)b << shift); s_remaining = rem;
            }
            __syncthreads();
        }
        thr_key = s_prefix;
    }
    const float thr = (thr_key == 0u) ? -INFINITY : key2f(thr_key);

    // ---- max and the per-thread chunk sums of exp(x - M) over kept entries (vocabulary order)
    const int per = (V + 1023) / 1024;
    const int i0 = tid * per, i1 = min(V, i0 + per);
    float mx = -INFINITY;
    for (int i = i0; i < i1; ++i) mx = fmaxf(mx, bf2f(row[i]) * inv_temp);
    mx = block_max(mx, red);
    float cs = 0.f;
    for (int i = i0; i < i1; ++i) { const float x = bf2f(row[i]) * inv_temp; if (x >= thr) cs += __expf(x - mx); }
    chunk_sum[tid] = cs;
    __syncthreads();
    if (tid == 0) {
        float Z = 0.f;
        for (int i = 0; i < 1024; ++i) Z += chunk_sum[i];
        unsigned c[4] = {(unsigned)r, (unsigned)step, 0u, 0u};
        philox4x32_10(c, (unsigned)(seed & 0xffffffffu), (unsigned)(seed >> 32));
        const float uu = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        if (u_out) u_out[r] = uu;
        const float target = uu * Z;
        float acc = 0.f; int ch = 0;
        for (; ch < 1023; ++ch) { if (acc + chunk_sum[ch] >= target) break; acc += chunk_sum[ch]; }
        // walk inside the chunk; fall back to the last kept token for rounding slack
        int tok = -1, last_kept = -1;
        const int a0 = ch * per, a1 = min(V, a0 + per);
        for (int i = a0; i < a1; ++i) {
            const float x = bf2f(row[i]) * inv_temp;
            if (x >= thr) { last_kept = i; acc += __expf(x - mx); if (acc >= target) { tok = i; break; } }
        }
        if (tok < 0) {
            if (last_kept >= 0) tok = last_kept;
            else { for (int i = V - 1; i >= 0; --i) { if (bf2f(row[i]) * inv_temp >= thr) { tok = i; break; } } }
        }
        s_token = tok;
        *tok_out = tok;
        if (finished && tok == eos_id) finished[r] = 1;
    }
}

extern "C" int tr1_logp_entropy_fwd(const void* logits, int64_t ld, const void* targets, void* logp, void* entropy, void* lse, int64_t R,
                                    int64_t V, void* stream) {
    TR1_CHECK_ARG(ld % 8 == 0, "logp_entropy: ld must be a multiple of 8");
    if (R == 0) return 0;
    hipLaunchKernelGGL(logp_entropy_fwd_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)logits, ld,
                       (const int*)targets, (float*)logp, (float*)entropy, (float*)lse, (int)V);
    TR1_LAUNCH_CHECK();
}
extern "C" int tr1_logp_bwd(const void* logits, int64_t ld, const void* targets, const void* lse, const void* dlogp, void* dlogits,
                            int64_t ld_out, int64_t R, int64_t V, void* stream) {
    TR1_CHECK_ARG(ld % 8 == 0 && ld_out % 8 == 0 && V % 8 == 0, "logp_bwd: ld and V must be multiples of 8");
    if (R == 0) return 0;
    dim3 grid((unsigned)tr1_grid_1d(V / 8, 256, 64), (unsigned)R);
    hipLaunchKernelGGL(logp_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)logits, ld, (const int*)targets,
                       (const float*)lse, (const float*)dlogp, (bf16_t*)dlogits, ld_out, (int)V);
    TR1_LAUNCH_CHECK();
}
extern "C" int tr1_grpo_loss(const void* logp, const void* ref_logp, const void* mask, const void* adv, void* dlogp, void* out3,
                             void* row_len, void* row_kl, int64_t G, int64_t C, float beta, int use_grpo, float grad_scale, void* stream) {
    TR1_CHECK_ARG(G >= 1 && G <= 64, "grpo_loss: G must be in [1, 64]");
    hipLaunchKernelGGL(grpo_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)logp, (const float*)ref_logp,
                       (const int*)mask, (const float*)adv, (float*)dlogp, (float*)out3, (float*)row_len, (float*)row_kl, (int)G, (int)C, beta,
                       use_grpo, grad_scale);
    TR1_LAUNCH_CHECK();
}
extern "C" int tr1_sample_tokens(const void* logits, int64_t ld, int64_t rows, int64_t V, float temperature, int64_t top_k,
                                 uint64_t seed, const void* step_ptr, void* tokens, int64_t tok_ld, void* finished, int64_t eos_id,
                                 int64_t pad_id, int stop_at_eos, void* u_out, void* stream) {
    TR1_CHECK_ARG(temperature > 0.f, "sample: temperature must be > 0");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(sample_kernel, dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream, (const bf16_t*)logits, ld, (int)V,
                       1.0f / temperature, (int)top_k, (unsigned long long)seed, (const int*)step_ptr, (int*)tokens, tok_ld, (int*)finished,
                       (int)eos_id, (int)pad_id, stop_at_eos, (float*)u_out);
    TR1_LAUNCH_CHECK();
}
