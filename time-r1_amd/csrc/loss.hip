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
// 16-bit order-preserving key of a bf16 logit (temperature > 0 does not change the order, so top-k is selected on raw logits)
TR1_DEV unsigned bfkey(bf16_t b) { return (b & 0x8000u) ? ((~(unsigned)b) & 0xffffu) : ((unsigned)b | 0x8000u); }
TR1_DEV float key_logit(unsigned k) { const unsigned b = (k & 0x8000u) ? (k & 0x7fffu) : ((~k) & 0xffffu); return bf2f((bf16_t)b); }

// Multi-block sampler.  The vocabulary row (152k bf16 logits, L2 resident) is cut into SAMP_S slices, one 256-thread block each:
//   1. hist_hi   : 256-bin histogram of the key's high byte (+ row max)           -> finds the byte holding the k-th largest logit
//   2. hist_lo   : histogram of the low byte among logits in that high-byte bin    -> exact k-th largest 16-bit key = threshold
//   3. slice_sum : sum exp((x - max)/T) over kept logits per slice
//   4. pick      : Philox uniform, locate the slice and the token by an inverse-CDF walk in vocabulary order
// Workspace per row (uint32 words): hist_hi[256] | hist_lo[256] | misc[8] (0: max key) | slice sums[SAMP_S] (float)
#define SAMP_S 32
#define SAMP_WS_WORDS (256 + 256 + 8 + SAMP_S)

struct SampleArgs {
    const bf16_t* logits; int64_t ld; int V; float inv_temp; int top_k; unsigned long long seed; int group_rows; unsigned long long seed_stride; const int* step_ptr; int* tokens; int64_t tok_ld;
    int* finished; int eos_id, pad_id, stop_at_eos; float* u_out; unsigned* ws;
    int* next_ids;      // optional [rows]: the drawn token once more, where the next decode step's embedding gather reads it (no copy kernel in between)
    int ws_clean;       // the caller zero-filled ws once: the fused pick kernel re-zeroes what the histogram kernels dirtied (no memset per call)
};

TR1_DEV bool samp_row_done(const SampleArgs& a, int r) { return a.finished && a.stop_at_eos && a.finished[r]; }

// threshold search over a 256-bin histogram: largest bin b with (count of keys in bins > b) < need <= (count in bins >= b)
TR1_DEV void samp_find_bin(const unsigned* hist, int need, int& bin, int& rem) {
    int acc = 0; int b = 255;
    for (; b > 0; --b) { if (acc + (int)hist[b] >= need) break; acc += (int)hist[b]; }
    bin = b; rem = need - acc;
}

// The same search by the first 256 threads of a block (one bin each, from the top): a suffix scan replaces the up-to-255 dependent reads of the
// serial walk (12 us of a 26 us launch).  hist: 256 bins in LDS; scr: 8 unsigned of LDS scratch; result in *bin_out / *rem_out (LDS).
TR1_DEV void samp_find_bin_par(const unsigned* hist, int need, unsigned* scr, int* bin_out, int* rem_out) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    unsigned val = 0u, incl = 0u;
    if (tid < 256) {
        val = hist[255 - tid];
        incl = val;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        if (lane == 63) scr[w] = incl;
    }
    if (tid == 0) { *bin_out = 0; *rem_out = -1; }
    __syncthreads();
    if (tid < 256) {
        unsigned off = 0u;
        for (int j = 0; j < w; ++j) off += scr[j];
        incl += off;
        const unsigned excl = incl - val;
        if (tid < 255 && (int)incl >= need && (int)excl < need) { *bin_out = 255 - tid; *rem_out = need - (int)excl; }   // exactly one thread
        if (tid == 254) scr[4] = incl;                                            // count in bins 255 .. 1 (fallback: bin 0)
    }
    __syncthreads();
    if (tid == 0 && *rem_out < 0) *rem_out = need - (int)scr[4];
    __syncthreads();
}

__global__ __launch_bounds__(256) void samp_hist_hi_kernel(SampleArgs a) {
    __shared__ unsigned h[256];
    __shared__ unsigned smax;
    const int r = blockIdx.y;
    if (samp_row_done(a, r)) return;
    h[threadIdx.x] = 0u; if (threadIdx.x == 0) smax = 0u;
    __syncthreads();
    const bf16_t* row = a.logits + (int64_t)r * a.ld;
    const int per = (a.V + SAMP_S - 1) / SAMP_S, i0 = blockIdx.x * per, i1 = min(a.V, i0 + per);
    unsigned mx = 0u;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) { const unsigned k = bfkey(row[i]); mx = max(mx, k); atomicAdd(&h[k >> 8], 1u); }
    atomicMax(&smax, mx);
    __syncthreads();
    unsigned* ws = a.ws + (int64_t)r * SAMP_WS_WORDS;
    if (h[threadIdx.x]) atomicAdd(&ws[threadIdx.x], h[threadIdx.x]);
    if (threadIdx.x == 0) atomicMax(&ws[512], smax);
}

__global__ __launch_bounds__(256) void samp_hist_lo_kernel(SampleArgs a) {
    __shared__ unsigned h[256];
    __shared__ int sbin;
    const int r = blockIdx.y;
    if (samp_row_done(a, r)) return;
    unsigned* ws = a.ws + (int64_t)r * SAMP_WS_WORDS;
    __shared__ unsigned hh[256], scr[8];
    __shared__ int srem;
    h[threadIdx.x] = 0u;
    hh[threadIdx.x] = ws[threadIdx.x];
    __syncthreads();
    samp_find_bin_par(hh, a.top_k, scr, &sbin, &srem);
    const unsigned bin = (unsigned)sbin;
    const bf16_t* row = a.logits + (int64_t)r * a.ld;
    const int per = (a.V + SAMP_S - 1) / SAMP_S, i0 = blockIdx.x * per, i1 = min(a.V, i0 + per);
    for (int i = i0 + threadIdx.x; i < i1; i += 256) { const unsigned k = bfkey(row[i]); if ((k >> 8) == bin) atomicAdd(&h[k & 255u], 1u); }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&ws[256 + threadIdx.x], h[threadIdx.x]);
}

TR1_DEV unsigned samp_threshold(const SampleArgs& a, const unsigned* ws) {   // 16-bit key of the k-th largest logit (0 = keep all)
    if (a.top_k <= 0 || a.top_k >= a.V) return 0u;
    int bin, rem, lo, rem2;
    samp_find_bin(ws, a.top_k, bin, rem);
    samp_find_bin(ws + 256, rem, lo, rem2);
    return ((unsigned)bin << 8) | (unsigned)lo;
}

__global__ __launch_bounds__(256) void samp_slice_sum_kernel(SampleArgs a) {
    __shared__ float red[16];
    __shared__ unsigned sthr;
    const int r = blockIdx.y;
    if (samp_row_done(a, r)) return;
    unsigned* ws = a.ws + (int64_t)r * SAMP_WS_WORDS;
    if (threadIdx.x == 0) sthr = samp_threshold(a, ws);
    __syncthreads();
    const unsigned thr = sthr;
    const float mx = key_logit(ws[512]);
    const bf16_t* row = a.logits + (int64_t)r * a.ld;
    const int per = (a.V + SAMP_S - 1) / SAMP_S, i0 = blockIdx.x * per, i1 = min(a.V, i0 + per);
    float acc = 0.f;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) { const bf16_t b = row[i]; if (bfkey(b) >= thr) acc += __expf((bf2f(b) - mx) * a.inv_temp); }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) reinterpret_cast<float*>(ws + 520)[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void samp_pick_kernel(SampleArgs a) {
    __shared__ float part[256];
    __shared__ unsigned sthr; __shared__ int sslice; __shared__ float sbase, starget;
    const int r = blockIdx.x, tid = threadIdx.x;
    const int step = a.step_ptr ? *a.step_ptr : 0;
    int* tok_out = a.tokens + (int64_t)r * a.tok_ld + step;
    if (samp_row_done(a, r)) { if (tid == 0) { *tok_out = a.pad_id; if (a.next_ids) a.next_ids[r] = a.pad_id; } return; }
    unsigned* ws = a.ws + (int64_t)r * SAMP_WS_WORDS;
    const float* sums = reinterpret_cast<const float*>(ws + 520);
    if (tid == 0) {
        sthr = samp_threshold(a, ws);
        float Z = 0.f;
        for (int i = 0; i < SAMP_S; ++i) Z += sums[i];
        // several prompts in one launch: rows [b*group_rows, (b+1)*group_rows) use seed + b*seed_stride and their row index inside the group,
        // i.e. exactly the stream a separate launch per prompt would draw
        const int grp = a.group_rows > 0 ? r / a.group_rows : 0;
        const unsigned long long sd = a.seed + (unsigned long long)grp * a.seed_stride;
        unsigned c[4] = {(unsigned)(a.group_rows > 0 ? r % a.group_rows : r), (unsigned)step, 0u, 0u};
        philox4x32_10(c, (unsigned)(sd & 0xffffffffu), (unsigned)(sd >> 32));
        const float uu = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        if (a.u_out) a.u_out[r] = uu;
        const float target = uu * Z;
        float acc = 0.f; int sl = 0;
        for (; sl < SAMP_S - 1; ++sl) { if (acc + sums[sl] >= target) break; acc += sums[sl]; }
        sslice = sl; sbase = acc; starget = target;
    }
    __syncthreads();
    const unsigned thr = sthr;
    const float mx = key_logit(ws[512]);
    const bf16_t* row = a.logits + (int64_t)r * a.ld;
    const int per = (a.V + SAMP_S - 1) / SAMP_S, i0 = sslice * per, i1 = min(a.V, i0 + per);
    const int tper = (per + 255) / 256, j0 = i0 + tid * tper, j1 = min(i1, j0 + tper);   // contiguous run per thread: vocabulary order
    float acc = 0.f;
    for (int i = j0; i < j1; ++i) { const bf16_t b = row[i]; if (bfkey(b) >= thr) acc += __expf((bf2f(b) - mx) * a.inv_temp); }
    part[tid] = acc;
    __syncthreads();
    if (tid == 0) {
        float c = sbase; int t = 0;
        for (; t < 255; ++t) { if (c + part[t] >= starget) break; c += part[t]; }
        int tok = -1, last_kept = -1;
        const int k0 = i0 + t * tper, k1 = min(i1, k0 + tper);
        for (int i = k0; i < k1; ++i) {
            const bf16_t b = row[i];
            if (bfkey(b) >= thr) { last_kept = i; c += __expf((bf2f(b) - mx) * a.inv_temp); if (c >= starget) { tok = i; break; } }
        }
        if (tok < 0) {   // rounding slack: fall back to the last kept token at or before this point
            if (last_kept >= 0) tok = last_kept;
            else { for (int i = min(k1, a.V) - 1; i >= 0; --i) { if (bfkey(row[i]) >= thr) { tok = i; break; } } }
            if (tok < 0) { for (int i = 0; i < a.V; ++i) { if (bfkey(row[i]) >= thr) { tok = i; break; } } }
        }
        *tok_out = tok;
        if (a.next_ids) a.next_ids[r] = tok;
        if (a.finished && tok == a.eos_id) a.finished[r] = 1;
    }
}

// slice_sum + pick in ONE launch, one 1024-thread block per row (the separate pair cost 15 + 33 us per decode step, most of it the serial
// walks of thread 0).  The 16 waves own contiguous segments of the row; a wave reads its segment coalesced (64 lanes x 16 bytes per
// iteration) and keeps the per-iteration wave sums, so the inverse-CDF walk in vocabulary order is a three-level search (segment ->
// iteration -> lane) with one lane finally walking 8 logits.  Needs V % 8 == 0, ld % 8 == 0 and V <= SAMP_FUSED_MAXV.
#define SAMP_MAXIT 20
#define SAMP_FUSED_MAXV (16 * 64 * SAMP_MAXIT * 8)
__global__ __launch_bounds__(1024) void samp_sum_pick_kernel(SampleArgs a) {
    __shared__ float wsum[16];
    __shared__ unsigned sthr; __shared__ int sseg; __shared__ float sbase, starget;
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int step = a.step_ptr ? *a.step_ptr : 0;
    int* tok_out = a.tokens + (int64_t)r * a.tok_ld + step;
    if (samp_row_done(a, r)) { if (tid == 0) { *tok_out = a.pad_id; if (a.next_ids) a.next_ids[r] = a.pad_id; } return; }
    unsigned* ws = a.ws + (int64_t)r * SAMP_WS_WORDS;
    __shared__ unsigned shist[512];                 // both histograms staged once: the threshold search of thread 0 then walks LDS, not global memory
    __shared__ unsigned scr[8];
    __shared__ int sbin, srem, slo, srem2;
    if (tid < 512) shist[tid] = ws[tid];
    __syncthreads();
    if (a.top_k <= 0 || a.top_k >= a.V) { if (tid == 0) sthr = 0u; __syncthreads(); }
    else {
        samp_find_bin_par(shist, a.top_k, scr, &sbin, &srem);
        samp_find_bin_par(shist + 256, srem, scr, &slo, &srem2);
        if (tid == 0) sthr = ((unsigned)sbin << 8) | (unsigned)slo;
        __syncthreads();
    }
    const unsigned thr = sthr;
    const float mx = key_logit(ws[512]);
    const bf16_t* row = a.logits + (int64_t)r * a.ld;
    const int nch = a.V >> 3, seg = (nch + 15) >> 4, c0 = wave * seg, c1 = min(nch, c0 + seg);
    const int nit = (seg + 63) >> 6;
    // all loads of the segment are issued up front (clamped addresses, no branches around them): one L2 round trip instead of one per iteration
    u32x4_t vals[SAMP_MAXIT];
#pragma unroll
    for (int it = 0; it < SAMP_MAXIT; ++it) {
        int ch = c0 + it * 64 + lane; if (ch > nch - 1) ch = nch - 1;
        vals[it] = *reinterpret_cast<const u32x4_t*>(row + (int64_t)ch * 8);
    }
    float lsum[SAMP_MAXIT], itsum[SAMP_MAXIT];
    float wtot = 0.f;
    // Chunk-level reject for top-k: the two 16-bit keys of every dword are formed with 32-bit ops (key = b ^ (sign ? 0xffff : 0x8000)) and the
    // chunk's largest key is compared with the threshold key - one test per 8 logits instead of 8 key transforms + 8 divergent branches (one
    // CU handles the whole row, so this loop is instruction-issue bound); only the ~k chunks that hold a kept logit take the exact path.
    const bool filt = thr != 0u;
#pragma unroll
    for (int it = 0; it < SAMP_MAXIT; ++it) {
        const int ch = c0 + it * 64 + lane;
        const bool okc = it < nit && ch < c1;
        bool any = okc;
        if (filt) {
            unsigned mk = 0u;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const unsigned w = vals[it][d];
                const unsigned neg = (w >> 15) & 0x00010001u;                 // 1 in each half that holds a negative value
                const unsigned kw = w ^ (((neg << 15) - neg) | 0x80008000u);  // negative half: ^ 0xffff, positive half: ^ 0x8000
                mk = max(mk, max(kw >> 16, kw & 0xffffu));
            }
            any = okc & (mk >= thr);
        }
        float sacc = 0.f;
        if (any) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {        // vocabulary order inside the chunk
                const bf16_t b = (bf16_t)((e & 1) ? (vals[it][e >> 1] >> 16) : (vals[it][e >> 1] & 0xffffu));
                if (bfkey(b) >= thr) sacc += __expf((bf2f(b) - mx) * a.inv_temp);
            }
        }
        lsum[it] = sacc;
        itsum[it] = wave_sum(sacc);
        wtot += itsum[it];
    }
    if (lane == 0) wsum[wave] = wtot;
    __syncthreads();
    if (a.ws_clean && tid < SAMP_WS_WORDS) ws[tid] = 0u;        // every thread has read the histograms and the row max: leave the row's workspace zero for the next call
    if (tid == 0) {
        float Z = 0.f;
        for (int w = 0; w < 16; ++w) Z += wsum[w];
        const int grp = a.group_rows > 0 ? r / a.group_rows : 0;
        const unsigned long long sd = a.seed + (unsigned long long)grp * a.seed_stride;
        unsigned c[4] = {(unsigned)(a.group_rows > 0 ? r % a.group_rows : r), (unsigned)step, 0u, 0u};
        philox4x32_10(c, (unsigned)(sd & 0xffffffffu), (unsigned)(sd >> 32));
        const float uu = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        if (a.u_out) a.u_out[r] = uu;
        const float target = uu * Z;
        const int nv = (nch + seg - 1) / seg;        // non-empty segments (small vocabularies leave the last waves without chunks)
        float acc = 0.f; int sg = 0;
        for (; sg < nv - 1; ++sg) { if (acc + wsum[sg] >= target) break; acc += wsum[sg]; }
        sseg = sg; sbase = acc; starget = target;
    }
    __syncthreads();
    if (wave != sseg) return;
    const float target = starget;
    float c = sbase;
    int it = 0;
    float sl = lsum[0];
#pragma unroll
    for (int j = 0; j < SAMP_MAXIT - 1; ++j) {      // wave-uniform walk over the iteration sums; `sl` follows the selected iteration
        if (it == j && j + 1 < nit && !(c + itsum[j] >= target)) { c += itsum[j]; it = j + 1; sl = lsum[j + 1]; }
    }
    float incl = sl;                                // inclusive scan over the lanes (vocabulary order inside the iteration)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const float t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
    const float excl = incl - sl;
    const int last_ch = min(c1, c0 + (it + 1) * 64) - 1;                 // last valid chunk of this iteration
    const unsigned long long hit = __ballot(c + incl >= target && c0 + it * 64 + lane <= last_ch);
    const int L = hit ? (__ffsll((long long)hit) - 1) : (last_ch - (c0 + it * 64));
    if (lane != L) return;
    const int ch = c0 + it * 64 + L;
    float cc = c + excl;
    int tok = -1, last_kept = -1;
    for (int i = ch * 8; i < ch * 8 + 8; ++i) {
        const bf16_t b = row[i];
        if (bfkey(b) >= thr) { last_kept = i; cc += __expf((bf2f(b) - mx) * a.inv_temp); if (cc >= target) { tok = i; break; } }
    }
    if (tok < 0) {   // rounding slack: fall back to the last kept token at or before this point
        if (last_kept >= 0) tok = last_kept;
        else { for (int i = ch * 8 + 7; i >= 0; --i) { if (bfkey(row[i]) >= thr) { tok = i; break; } } }
        if (tok < 0) { for (int i = 0; i < a.V; ++i) { if (bfkey(row[i]) >= thr) { tok = i; break; } } }
    }
    *tok_out = tok;
    if (a.next_ids) a.next_ids[r] = tok;
    if (a.finished && tok == a.eos_id) a.finished[r] = 1;
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
extern "C" int64_t tr1_sample_workspace_words(int64_t rows) { return rows * SAMP_WS_WORDS; }

static int sample_tokens_impl(const void* logits, int64_t ld, int64_t rows, int64_t V, float temperature, int64_t top_k,
                                 uint64_t seed, int64_t group_rows, uint64_t seed_stride, const void* step_ptr, void* tokens, int64_t tok_ld,
                                 void* finished, int64_t eos_id, int64_t pad_id, int stop_at_eos, void* u_out, void* ws_u32, int64_t ws_words,
                                 void* next_ids, int ws_zeroed, void* stream) {
    TR1_CHECK_ARG(temperature > 0.f, "sample: temperature must be > 0");
    TR1_CHECK_ARG(ws_u32 && ws_words >= rows * SAMP_WS_WORDS, "sample: workspace too small (tr1_sample_workspace_words)");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    SampleArgs a;
    a.logits = (const bf16_t*)logits; a.ld = ld; a.V = (int)V; a.inv_temp = 1.0f / temperature; a.top_k = (int)top_k; a.seed = seed; a.group_rows = (int)group_rows; a.seed_stride = seed_stride;
    a.step_ptr = (const int*)step_ptr; a.tokens = (int*)tokens; a.tok_ld = tok_ld; a.finished = (int*)finished; a.eos_id = (int)eos_id;
    a.pad_id = (int)pad_id; a.stop_at_eos = stop_at_eos; a.u_out = (float*)u_out; a.ws = (unsigned*)ws_u32;
    a.next_ids = (int*)next_ids;
    const bool fused = V % 8 == 0 && ld % 8 == 0 && V <= SAMP_FUSED_MAXV && (reinterpret_cast<uintptr_t>(logits) & 15) == 0;
    a.ws_clean = (ws_zeroed && fused) ? 1 : 0;
    if (!a.ws_clean) hipMemsetAsync(ws_u32, 0, (size_t)rows * SAMP_WS_WORDS * 4, s);
    dim3 grid(SAMP_S, (unsigned)rows);
    hipLaunchKernelGGL(samp_hist_hi_kernel, grid, dim3(256), 0, s, a);     // also yields the row max (needed without top-k too)
    if (top_k > 0 && top_k < V) hipLaunchKernelGGL(samp_hist_lo_kernel, grid, dim3(256), 0, s, a);
    if (fused) {
        hipLaunchKernelGGL(samp_sum_pick_kernel, dim3((unsigned)rows), dim3(1024), 0, s, a);
    } else {
        hipLaunchKernelGGL(samp_slice_sum_kernel, grid, dim3(256), 0, s, a);
        hipLaunchKernelGGL(samp_pick_kernel, dim3((unsigned)rows), dim3(256), 0, s, a);
        if (ws_zeroed) hipMemsetAsync(ws_u32, 0, (size_t)rows * SAMP_WS_WORDS * 4, s);      // keep the caller's "zero between calls" contract on this path too
    }
    TR1_LAUNCH_CHECK();
}

extern "C" int tr1_sample_tokens(const void* logits, int64_t ld, int64_t rows, int64_t V, float temperature, int64_t top_k,
                                 uint64_t seed, int64_t group_rows, uint64_t seed_stride, const void* step_ptr, void* tokens, int64_t tok_ld,
                                 void* finished, int64_t eos_id, int64_t pad_id, int stop_at_eos, void* u_out, void* ws_u32, int64_t ws_words,
                                 void* stream) {
    return sample_tokens_impl(logits, ld, rows, V, temperature, top_k, seed, group_rows, seed_stride, step_ptr, tokens, tok_ld, finished, eos_id, pad_id,
                              stop_at_eos, u_out, ws_u32, ws_words, nullptr, 0, stream);
}

// The decode loop's form: next_ids[row] (optional) receives the drawn token as well - the buffer the next step's embedding gather reads, so no copy
// kernel runs between two steps - and ws_zeroed != 0 promises a workspace that was zero-filled ONCE and is only ever used through this entry point:
// the pick kernel then re-zeroes what the histogram kernels dirtied instead of a memset in front of every call.
extern "C" int tr1_sample_tokens_step(const void* logits, int64_t ld, int64_t rows, int64_t V, float temperature, int64_t top_k,
                                      uint64_t seed, int64_t group_rows, uint64_t seed_stride, const void* step_ptr, void* tokens, int64_t tok_ld,
                                      void* finished, int64_t eos_id, int64_t pad_id, int stop_at_eos, void* u_out, void* ws_u32, int64_t ws_words,
                                      void* next_ids, int ws_zeroed, void* stream) {
    return sample_tokens_impl(logits, ld, rows, V, temperature, top_k, seed, group_rows, seed_stride, step_ptr, tokens, tok_ld, finished, eos_id, pad_id,
                              stop_at_eos, u_out, ws_u32, ws_words, next_ids, ws_zeroed, stream);
}
