// Shared device/host helpers for the timer1 HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef unsigned short bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA A/B fragment (8 bf16, 4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // MFMA 16x16 C/D fragment
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

#define TR1_DEV __device__ __forceinline__

TR1_DEV float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }

// round-to-nearest-even via the gfx950 hardware conversion (v_cvt_pk_bf16_f32; the compiler selects it for __bf16 casts),
// same rounding as torch's float->bfloat16 cast
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_native_t;
TR1_DEV bf16_t f2bf(float f) { const __bf16 b = (__bf16)f; return __builtin_bit_cast(bf16_t, b); }
TR1_DEV unsigned pack2bf(float lo, float hi) {
    const bf16x2_native_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}
TR1_DEV float bflo(unsigned w) { return __uint_as_float(w << 16); }
TR1_DEV float bfhi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// ---- cross-lane exchanges without the LDS (round 6).  __shfl_xor lowers to ds_bpermute_b32: an LDS round trip plus an lgkmcnt(0) wait that also drains every LDS read
// in flight; a 6-step wave reduction is six DEPENDENT round trips (~500 cycles in the latency-bound decode kernels).  gfx950 exchanges the halves (v_permlane32_swap)
// and the odd / even 16-lane rows (v_permlane16_swap) of a wave in the vector unit, and DPP row rotations cover the steps inside a row.  The reductions below take
// the SAME partners in the SAME order as the xor butterfly 32, 16, 8, 4, 2, 1 they replace: after step k the values have period k inside a row, so the lane that a
// rotation by k/2 reads holds exactly what lane ^ (k/2) holds - and a + b, max(a, b) do not care which side the partner is on: bit-identical results.
// (Assembly for the swaps: the builtin's second result is mis-assigned by this hipcc; the leading nops are the VALU-write -> permlane-read wait states.)
TR1_DEV void tr1_halves32(float x, float& a, float& b) { a = x; b = x; asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }   // a = lane & 31 of the lower half, b = of the upper half
TR1_DEV void tr1_halves16(float x, float& a, float& b) { a = x; b = x; asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }   // a = the even row of each row pair, b = the odd row
template <int N> TR1_DEV float tr1_row_ror(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + N, 0xf, 0xf, false)); }
TR1_DEV float tr1_sum_xor32(float v) { float a, b; tr1_halves32(v, a, b); return a + b; }      // == v + __shfl_xor(v, 32, 64)
TR1_DEV float tr1_sum_xor16(float v) { float a, b; tr1_halves16(v, a, b); return a + b; }      // == v + __shfl_xor(v, 16, 64)
TR1_DEV float wave_sum(float v) {
    v = tr1_sum_xor32(v); v = tr1_sum_xor16(v);
    v += tr1_row_ror<8>(v); v += tr1_row_ror<4>(v); v += tr1_row_ror<2>(v); v += tr1_row_ror<1>(v);
    return v;
}
TR1_DEV float wave_max(float v) {
    float a, b;
    tr1_halves32(v, a, b); v = fmaxf(a, b);
    tr1_halves16(v, a, b); v = fmaxf(a, b);
    v = fmaxf(v, tr1_row_ror<8>(v)); v = fmaxf(v, tr1_row_ror<4>(v)); v = fmaxf(v, tr1_row_ror<2>(v)); v = fmaxf(v, tr1_row_ror<1>(v));
    return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); red must hold >= 16 floats
TR1_DEV float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = (lane < nw) ? red[lane] : 0.f;
    t = wave_sum(t);
    return t;
}
TR1_DEV float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = (lane < nw) ? red[lane] : -INFINITY;
    t = wave_max(t);
    return t;
}

// ---- host side error plumbing (C ABI returns int; message via tr1_last_error) ----
extern "C" void tr1_set_error_(const char* msg);

#define TR1_CHECK_ARG(cond, msg)                                  \
    do {                                                          \
        if (!(cond)) {                                            \
            tr1_set_error_(msg);                                  \
            return 1000;                                          \
        }                                                         \
    } while (0)

#define TR1_LAUNCH_CHECK()                                        \
    do {                                                          \
        hipError_t e__ = hipGetLastError();                       \
        if (e__ != hipSuccess) {                                  \
            tr1_set_error_(hipGetErrorString(e__));               \
            return (int)e__;                                      \
        }                                                         \
        return 0;                                                 \
    } while (0)

static inline int tr1_grid_1d(int64_t work_items, int per_block, int cap = 8192) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// ---- decode: q/k/v projection epilogue (shared by the bf16 and fp8 decode GEMMs) ---------------------------------------------------
// The fused QKV kernels give each block the column pair (d, d + hd/2) of one head, so rotate-half RoPE closes inside the block and the
// results go straight to their consumers: roped q -> q_out, roped k -> K cache row slots[m], v -> V^T cache column slots[m]
// (= tr1_decode_qkv_post folded into the projection; reference: Qwen2VLAttention.forward TF:521-556 + DynamicCache.update).
struct QkvEpi {
    const float* cosb; const float* sinb;            // [R, hd/2] fp32 tables of the rows' positions
    bf16_t* q_out; int64_t ld_q;
    bf16_t* kcache; int64_t k_ld;
    bf16_t* vtcache; int64_t vt_ld;
    const int* slots;
    int n_heads, n_kv, hd;
};
// rotate-half pair with explicitly rounded products (no fma contraction): the fused and the two-kernel decode paths must agree bit for bit
TR1_DEV void rope_pair(float a, float b, float c, float s, float& oa, float& ob) {
    oa = __fsub_rn(__fmul_rn(a, c), __fmul_rn(b, s));
    ob = __fadd_rn(__fmul_rn(b, c), __fmul_rn(a, s));
}
// vA / vB: projection outputs (bias included) of row m at columns h*hd + d and h*hd + hd/2 + d, already rounded to bf16 like the unfused path
TR1_DEV void qkv_epilogue_store(const QkvEpi& e, int m, int h, int d, float vA, float vB) {
    const int half = e.hd >> 1;
    if (h < e.n_heads + e.n_kv) {
        const float c = e.cosb[(int64_t)m * half + d], s = e.sinb[(int64_t)m * half + d];
        float fa, fb;
        rope_pair(vA, vB, c, s, fa, fb);
        const bf16_t oa = f2bf(fa), ob = f2bf(fb);
        bf16_t* dst = (h < e.n_heads) ? (e.q_out + (int64_t)m * e.ld_q + (int64_t)h * e.hd + d)
                                      : (e.kcache + (int64_t)e.slots[m] * e.k_ld + (int64_t)(h - e.n_heads) * e.hd + d);
        dst[0] = oa; dst[half] = ob;
    } else {
        bf16_t* col = e.vtcache + ((int64_t)(h - e.n_heads - e.n_kv) * e.hd + d) * e.vt_ld + e.slots[m];
        col[0] = f2bf(vA); col[(int64_t)half * e.vt_ld] = f2bf(vB);
    }
}
