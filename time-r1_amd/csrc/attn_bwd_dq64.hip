// dQ of the attention backward, 64 packed query rows per wave on ONE wave per SIMD (head dim 128; round 6).
// Same arithmetic as attn_bwd_dq32_kernel (attn_bwd.hip) - bit for bit: P = exp2(S scale - lse2), dS = P (dP - delta) per element, the same accumulation order of
// dQ^T += K^T dS^T over 16-key chunks - on the machine shape of attn_fwd64_kernel (attn_fwd64.hip, which explains the register-file pinning):
//   * a block is 4 waves x 64 rows (q-blocks A and B of 32 rows); every K / V fragment (b128) feeds both q-blocks' S / dP MFMAs, every K^T fragment
//     (two transposing reads) both dQ MFMAs: half the LDS bytes per MFMA of the 32-row kernel;
//   * the unit of the software pipeline is a 32-key HALF tile: body (t, kb) issues the 16 dQ MFMAs of the previous half and the 32 S / dP MFMAs of the next
//     half, and between consecutive MFMAs three vector instructions of the current half's P / dS chain (fma, exp2, subtract, multiply, bf16 pack):
//     the vector work no longer waits for - or makes wait - the wave's own matrix work;
//   * S / dP live in arch VGPRs (asm MFMAs), dQ's accumulators, Q and dO in the accumulator file.
// One barrier per 64-key tile; K | V row tiles in a 4-deep ring, tile t+2 requested behind tile t's barrier (tile t-1's K rows are still read - transposed, for
// the dQ product of its second half - during tile t's first body).
// Reference semantics: the backward of flash_attn_varlen_func / SDPA as autograd runs it under accelerator.backward (src/time_r1/rl/timer1_trainer.py:452-457).
#include "attn_common.h"
#include <stdlib.h>

#define ATT_QMETA 8         // ints per 64-row query tile in the mask summary (= attn_bwd.hip)

TR1_DEV void dq64_mfma0(f32x16_t& d, bf16x8_t k, bf16x8_t q) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(k), "a"(q)); }
TR1_DEV void dq64_mfma(f32x16_t& d, bf16x8_t k, bf16x8_t q) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(k), "a"(q)); }

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_bwd_dq64_kernel(AttnParams p, const float* __restrict__ lse2) {
    constexpr int D = 128, NB = 4, TILE = 64 * 256, BUF = 2 * TILE, AH = 2, TH = 2;
    extern __shared__ __attribute__((aligned(256))) char dyn_lds[];  // [NB][K rows | V rows] + block mask summary [8][3]
    int* lds_meta = reinterpret_cast<int*>(dyn_lds + NB * BUF);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), c32 = lane & 31, h = lane >> 5;
    const int kvh = blockIdx.y;
    const unsigned nR = (unsigned)p.T * (unsigned)p.group;
    const unsigned Rw0 = (unsigned)(gridDim.x - 1 - blockIdx.x) * 256u + (unsigned)wave * 64u;      // heaviest query blocks first
    const bool fused_delta = p.lse2_out != nullptr;                   // wave-uniform (kernel argument)
    int pre_e[2], lo_e[2], hi_d[2];                                   // visible(kv) = kv < pre_e | (unsigned)(kv - lo_e) <= hi_d   (see attn_fwd64_kernel)
    int wminpre[2], wmaxlo[2], wminhi[2];
    int tsum[6];                                                      // this wave's 64-row tile: max pre, min lo, max hi, min pre, max lo, min hi
    float lse[2], dlt[2];                                             // log2-scaled LSE (+inf: no visible key / padding row -> P = 0), delta
    bf16x8_t qf[2][D / 16], dof[2][D / 16];                           // Q / dO rows of this lane (B operands)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const unsigned R = Rw0 + (unsigned)b * 32u + (unsigned)c32;
        const bool valid = R < nR;
        int tq, hq;
        att_split_row(p, valid ? R : nR - 1, tq, hq);
        const int pre_b = valid ? p.pre[tq] : 0, lo_b = valid ? p.lo[tq] : 1, hi_b = valid ? p.hi[tq] : 0;
        const int64_t si = (int64_t)(kvh * p.group + hq) * p.T + tq;
        lse[b] = INFINITY; dlt[b] = 0.f;
        if (!fused_delta) { lse[b] = valid ? lse2[si] : INFINITY; dlt[b] = valid ? p.delta[si] : 0.f; }
        {
            const int hc = hi_b < p.n_slots ? hi_b : p.n_slots - 1;
            pre_e[b] = pre_b < p.n_slots ? pre_b : p.n_slots;
            lo_e[b] = hc >= lo_b ? lo_b : 0x7fffffff; hi_d[b] = hc >= lo_b ? hc - lo_b : 0;
        }
        int a0 = valid ? pre_b : 0, a1 = valid ? pre_b : 0x7fffffff;
        int a2 = (valid && hi_b >= lo_b) ? lo_b : 0x7fffffff, a3 = (valid && hi_b >= lo_b) ? hi_b : -1;
        int a4 = valid ? (hi_b >= lo_b ? lo_b : 0x7fffffff) : -1, a5 = valid ? (hi_b >= lo_b ? hi_b : -1) : 0x7fffffff;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            a0 = max(a0, __shfl_xor(a0, o, 64)); a1 = min(a1, __shfl_xor(a1, o, 64));
            a2 = min(a2, __shfl_xor(a2, o, 64)); a3 = max(a3, __shfl_xor(a3, o, 64));
            a4 = max(a4, __shfl_xor(a4, o, 64)); a5 = min(a5, __shfl_xor(a5, o, 64));
        }
        a0 = __builtin_amdgcn_readfirstlane(a0); a1 = __builtin_amdgcn_readfirstlane(a1); a2 = __builtin_amdgcn_readfirstlane(a2);
        a3 = __builtin_amdgcn_readfirstlane(a3); a4 = __builtin_amdgcn_readfirstlane(a4); a5 = __builtin_amdgcn_readfirstlane(a5);
        wminpre[b] = a1; wmaxlo[b] = a4; wminhi[b] = a5;
        if (lane == 0) { lds_meta[(wave * 2 + b) * 3 + 0] = a0; lds_meta[(wave * 2 + b) * 3 + 1] = a2; lds_meta[(wave * 2 + b) * 3 + 2] = a3; }
        if (b == 0) { tsum[0] = a0; tsum[1] = a2; tsum[2] = a3; tsum[3] = a1; tsum[4] = a4; tsum[5] = a5; }
        else { tsum[0] = max(tsum[0], a0); tsum[1] = min(tsum[1], a2); tsum[2] = max(tsum[2], a3); tsum[3] = min(tsum[3], a1); tsum[4] = max(tsum[4], a4); tsum[5] = min(tsum[5], a5); }
        const int64_t hoff = (int64_t)(kvh * p.group + hq) * D;
        const bf16_t* qrow = p.Q + (int64_t)tq * p.q_ld + hoff;
        const bf16_t* drow = p.dO + (int64_t)tq * p.do_ld + hoff;
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) { qf[b][ks] = load_row_frag(qrow, ks * 16 + h * 8, D, valid); dof[b][ks] = load_row_frag(drow, ks * 16 + h * 8, D, valid); }
        if (fused_delta) {
            // what attn_delta_kernel computed for this row (as attn_bwd_dq32_kernel does): delta = sum_d dO[d] O[d] over the lane's 64 features + its partner
            // lane's, the log2-scaled LSE; both also go to global memory for the dK/dV kernel that follows on the stream
            const bf16_t* orow = p.O + (int64_t)tq * p.o_ld + hoff;
            float s = 0.f;
#pragma unroll
            for (int ks = 0; ks < D / 16; ++ks) {
                const u32x4_t x = __builtin_bit_cast(u32x4_t, dof[b][ks]), y = __builtin_bit_cast(u32x4_t, load_row_frag(orow, ks * 16 + h * 8, D, valid));
#pragma unroll
                for (int j = 0; j < 4; ++j) s += bflo(x[j]) * bflo(y[j]) + bfhi(x[j]) * bfhi(y[j]);
            }
            s += __shfl_xor(s, 32, 64);
            const float l0 = valid ? p.lse[si] : NEG_INF;
            lse[b] = (l0 == NEG_INF) ? INFINITY : l0 * 1.4426950408889634f;
            dlt[b] = valid ? s : 0.f;
            if (valid && h == 0) { p.delta[si] = dlt[b]; p.lse2_out[si] = lse[b]; }
        }
    }
    if (p.qmeta_out && kvh == 0 && lane == 0 && Rw0 < nR) {           // the mask summary of this wave's 64-row tile (what the dK/dV kernel skips tiles with)
        int* qm = p.qmeta_out + (Rw0 >> 6) * ATT_QMETA;
#pragma unroll
        for (int j = 0; j < 6; ++j) qm[j] = tsum[j];
    }
    f32x16_t acc[2][4];                                               // dQ^T[q-block][feature block][C layout]
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int db = 0; db < 4; ++db) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][db][r] = 0.f;
            asm volatile("" : "+a"(acc[b][db]));                      // loop-carried tiles start in the accumulator file
        }
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) asm volatile("" ::"v"(qf[b][ks]), "v"(dof[b][ks]));      // hipcc places the wait for the operand loads here
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // (stores of delta / lse2 / qmeta included: from here on vmcnt counts DMA only)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) { asm volatile("" : "+a"(qf[b][ks])); asm volatile("" : "+a"(dof[b][ks])); }
    __syncthreads();
    int bmaxpre = 0, bminlo = 0x7fffffff, bmaxhi = -1;
#pragma unroll
    for (int w = 0; w < 8; ++w) { bmaxpre = max(bmaxpre, lds_meta[w * 3]); bminlo = min(bminlo, lds_meta[w * 3 + 1]); bmaxhi = max(bmaxhi, lds_meta[w * 3 + 2]); }
    TileRange tr = att_tile_range(bmaxpre, bminlo, bmaxhi, p.n_slots);
    tr.pre_tiles = __builtin_amdgcn_readfirstlane(tr.pre_tiles); tr.start2 = __builtin_amdgcn_readfirstlane(tr.start2);
    const int n_my = __builtin_amdgcn_readfirstlane(tr.n_rel);

    const unsigned lds_base = (unsigned)(uintptr_t)(att_lptr_t)dyn_lds;
    const char* kbase = reinterpret_cast<const char*>(p.K) + (int64_t)kvh * 256;
    const char* vbase = reinterpret_cast<const char*>(p.V) + (int64_t)kvh * 256;
    const unsigned k_ldb = (unsigned)p.k_ld * 2u, v_ldb = (unsigned)p.v_ld * 2u;
    unsigned koff[4], voff[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const unsigned row = 4u * (unsigned)(wave * 4 + d) + ((unsigned)lane >> 4);
        const unsigned ch = (unsigned)(((lane & 15) ^ skey(row & 15)) << 4);
        koff[d] = row * k_ldb + ch; voff[d] = row * v_ldb + ch;
    }
    // rows past the cache's last slot: clamped to the last chunk of the last row (finite data; those keys are masked to P = 0)
    const unsigned klim = ((unsigned)p.n_slots - 1u) * k_ldb + 240u, vlim = ((unsigned)p.n_slots - 1u) * v_ldb + 240u;
#define DQ64_DMA16(voff_, sbase, m0v) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff_), "s"(sbase) : "memory", "m0")
    auto dma_one = [&](int d, int it) {                               // d < 4: K row group wave*4 + d of tile index it (clamped), else V row group wave*4 + d - 4
        const int i = it < n_my ? it : n_my - 1;
        const unsigned t64 = (unsigned)att_tile_at(tr, i) * 64u, buf = lds_base + (unsigned)(it % NB) * BUF;
        if (d < 4) {
            unsigned off = koff[d] + t64 * k_ldb; off = off < klim ? off : klim;
            DQ64_DMA16(off, kbase, buf + (unsigned)(wave * 4 + d) * 1024u);
        } else {
            unsigned off = voff[d - 4] + t64 * v_ldb; off = off < vlim ? off : vlim;
            DQ64_DMA16(off, vbase, buf + TILE + (unsigned)(wave * 4 + d - 4) * 1024u);
        }
    };
    if (n_my > 0) {
#pragma unroll
        for (int d = 0; d < 8; ++d) dma_one(d, 0);
#pragma unroll
        for (int d = 0; d < 8; ++d) dma_one(d, 1);
    }

    typedef const __attribute__((address_space(3))) bf16x8_t* lds_b128_t;
#define LDS_B128(addr) (*(lds_b128_t)(uintptr_t)(addr))
#define LDS_TR16(addr) __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(uintptr_t)(addr)))
#define P2_LD(ya, n) make_frag(LDS_TR16(((ya) ^ (((n) & 3) * 64)) + ((n) >> 2) * 4096), LDS_TR16(((ya) ^ (((n) & 3) * 64 + 32)) + ((n) >> 2) * 4096 + 2048))
#define DQ64_PIN(x) asm volatile("" : "+v"(x))
    const int ti = lane & 15, tgrp = (lane >> 4) & 1;
    const unsigned a_lane = (unsigned)(c32 * 256 + ((h ^ skey(c32 & 15)) << 4));
    const unsigned t_lane = (unsigned)((4 * h + (ti >> 2)) * 256 + (ti & 1) * 8 + (((tgrp * 2 + ((ti & 3) >> 1)) ^ (((ti >> 2) << 2) | h)) << 4));
    float nlse[2] = {-lse[0], -lse[1]};

    // ---- the vector program of one half tile: 16 element pairs (q-block q >> 3, registers 2 (q & 7), +1) x 9 instructions, three per MFMA gap
    auto vstep = [&](int s, f32x16_t (&cs)[2], f32x16_t (&cp)[2], u32x4_t (&ds)[2][2]) {
#pragma unroll
        for (int o = 3 * s; o < 3 * s + 3; ++o) {
            const int q = o / 9, k = o - 9 * q, b = q >> 3, r = (q & 7) * 2;
            if (k < 2) cs[b][r + k] = __builtin_fmaf(cs[b][r + k], p.scale_log2, nlse[b]);
            else if (k < 4) cs[b][r + k - 2] = __builtin_amdgcn_exp2f(cs[b][r + k - 2]);
            else if (k < 6) cp[b][r + k - 4] = cp[b][r + k - 4] - dlt[b];
            else if (k < 8) cs[b][r + k - 6] = cs[b][r + k - 6] * cp[b][r + k - 6];
            else { unsigned w_ = pack2bf(cs[b][r], cs[b][r + 1]); DQ64_PIN(w_); ds[b][r >> 3][(r & 7) >> 1] = w_; }
        }
    };
    // keys of half kb of the tile at list index `it` that a q-block's rows cannot all see: S = -inf (P = 0 exactly) before the chain touches it
    auto mask_half = [&](int it, int kb, f32x16_t (&cs)[2]) {
        const int kv0 = att_tile_at(tr, it) * 64;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const bool full = (kv0 + 64 <= p.n_slots) && ((kv0 + 64 <= wminpre[b]) || (wmaxlo[b] <= kv0 && kv0 + 63 <= wminhi[b]));
            if (!full) {
                const int base = kv0 + kb * 32 + 4 * h;
                const unsigned A = (unsigned)(base - lo_e[b]), Dm = (unsigned)hi_d[b];
                const int B = pre_e[b] - base;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = (r & 3) + 8 * (r >> 2);
                    const float sv = cs[b][r];
                    float x = (A + (unsigned)c <= Dm) ? sv : NEG_INF;
                    asm volatile("" : "+v"(x));
                    x = (c < B) ? sv : x;
                    cs[b][r] = x;
                }
            }
        }
    };
    // S^T / dP^T of half kb of tile index it, both q-blocks: four independent chains; slot0 >= 0: the vector slices slot0 .. slot0 + 31 run in the gaps
    auto sp_product = [&](int it, int kb, f32x16_t (&csN)[2], f32x16_t (&cpN)[2], int slot0, f32x16_t (&cs)[2], f32x16_t (&cp)[2], u32x4_t (&ds)[2][2]) {
        const unsigned xa = lds_base + (unsigned)(it % NB) * BUF + kb * 8192 + a_lane;
        bf16x8_t ka[AH + 1], va[AH + 1];
#pragma unroll
        for (int ks = 0; ks < AH; ++ks) { ka[ks] = LDS_B128(xa ^ (ks * 32)); va[ks] = LDS_B128((xa ^ (ks * 32)) + TILE); }
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) {
            if (ks + AH < D / 16) { ka[(ks + AH) % (AH + 1)] = LDS_B128(xa ^ ((ks + AH) * 32)); va[(ks + AH) % (AH + 1)] = LDS_B128((xa ^ ((ks + AH) * 32)) + TILE); }
#pragma unroll
            for (int w = 0; w < 4; ++w) {                             // (S_A, dP_A, S_B, dP_B)
                f32x16_t& d = (w & 1) ? cpN[w >> 1] : csN[w >> 1];
                const bf16x8_t a = (w & 1) ? va[ks % (AH + 1)] : ka[ks % (AH + 1)], bq = (w & 1) ? dof[w >> 1][ks] : qf[w >> 1][ks];
                if (ks == 0) dq64_mfma0(d, a, bq); else dq64_mfma(d, a, bq);
                if (slot0 >= 0) vstep(slot0 + ks * 4 + w, cs, cp, ds);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_nop 7" ::: "memory");                         // asm MFMA D -> vector reader (see attn_fwd64.hip)
    };
    // dQ^T[feature][q] += K^T[feature][kv] dS^T[kv][q] over the two 16-key chunks of half kb of tile index it (K^T fragments: transposing reads of the K rows)
    bf16x8_t knext[TH];                                               // the first K^T fragments of the NEXT body's dQ product, read at the end of this one (the half they
    auto dq_prefetch = [&](int it, int kb) {                          // come from landed a tile ago and is not requested again for two more)
        const unsigned ya = lds_base + (unsigned)(it % NB) * BUF + kb * 8192 + t_lane;
#pragma unroll
        for (int n = 0; n < TH; ++n) knext[n] = P2_LD(ya, n);
    };
    auto dq_product = [&](int it, int kb, u32x4_t (&dsP)[2][2], bool with_steps, int it_dma, int d0, f32x16_t (&cs)[2], f32x16_t (&cp)[2], u32x4_t (&ds)[2][2]) {
        const unsigned ya = lds_base + (unsigned)(it % NB) * BUF + kb * 8192 + t_lane;
        bf16x8_t a[TH + 1];
#pragma unroll
        for (int n = 0; n < TH; ++n) a[n] = knext[n];
#pragma unroll
        for (int n = 0; n < 8; ++n) {                                 // n = chunk * 4 + feature block
            if (n + TH < 8) a[(n + TH) % (TH + 1)] = P2_LD(ya, n + TH);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                acc[b][n & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[n % (TH + 1)], __builtin_bit_cast(bf16x8_t, dsP[b][n >> 2]), acc[b][n & 3], 0, 0, 0);
                if (with_steps) {
                    if (d0 >= 0 && 2 * n + b < 8) dma_one(d0 + 2 * n + b, it_dma);
                    vstep(2 * n + b, cs, cp, ds);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // body (it, kb): current half's chain in the gaps of dQ(previous half) + S / dP(next half).  cX / pX: this half's S / dP; cY / pY: the next half's; dsP / dsC
    auto body = [&](int it, int kb, f32x16_t (&cX)[2], f32x16_t (&pX)[2], f32x16_t (&cY)[2], f32x16_t (&pY)[2], u32x4_t (&dsP)[2][2], u32x4_t (&dsC)[2][2]) {
        if (kb == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's share of tile it+1 (requested one tile ago) has landed
            __builtin_amdgcn_s_barrier();                             // ... and everybody's; everybody is done with tile it-2 (and with tile it-1 but for its K rows)
            asm volatile("" ::: "memory");
        }
        mask_half(it, kb, cX);
        // previous half: (it, 0) for kb = 1, (it-1, 1) for kb = 0 (it = 0: tile 0 with dS = 0); the 8 DMA instructions of tile it+2 ride on the first body's dQ phase
        if (kb == 0) dq_product(it > 0 ? it - 1 : 0, it > 0 ? 1 : 0, dsP, true, it + 2, 0, cX, pX, dsC);
        else dq_product(it, 0, dsP, true, 0, -1, cX, pX, dsC);
        if (kb == 0) sp_product(it, 1, cY, pY, 16, cX, pX, dsC);
        else sp_product(it + 1, 0, cY, pY, 16, cX, pX, dsC);
        dq_prefetch(it, kb);                                          // this half is the next body's "previous half"
    };

    f32x16_t c0[2], p0[2], c1[2], p1[2];
    u32x4_t da[2][2], db_[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 2; ++c) { da[b][c] = (u32x4_t){0, 0, 0, 0}; db_[b][c] = (u32x4_t){0, 0, 0, 0}; }
    if (n_my > 0) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");              // tile 0
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        sp_product(0, 0, c0, p0, -1, c0, p0, da);
        dq_prefetch(0, 0);
        for (int it = 0; it < n_my; ++it) {
            body(it, 0, c0, p0, c1, p1, db_, da);                     // chain(it, 0) -> da;  dQ(it-1, 1) reads db_ (zero at it = 0)
            body(it, 1, c1, p1, c0, p0, da, db_);                     // chain(it, 1) -> db_; dQ(it, 0) reads da
        }
        dq_product(n_my - 1, 1, db_, false, 0, -1, c0, p0, da);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (clamped requests of the last tiles)
    }
#undef LDS_B128
#undef LDS_TR16
#undef P2_LD
#undef DQ64_DMA16
#undef DQ64_PIN
    // lane holds dQ^T[feature = db*32 + 8i + 4h + j][its query row of q-block b]
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int e_h = (tid2 >> 5) & 1;
    const float scale = p.scale_log2 * 0.6931471805599453f;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const unsigned e_R = Rw0 + (unsigned)b * 32u + (unsigned)(tid2 & 31);
        if (e_R < nR) {
            int t2, hq2;
            att_split_row(p, e_R, t2, hq2);
            bf16_t* row = p.dQ + (int64_t)t2 * p.dq_ld + (int64_t)(kvh * p.group + hq2) * D;
            if (p.rope_cos) {
                // M-RoPE backward in the epilogue, exactly as attn_bwd_dq32_kernel: dQ rounded to bf16, rotated by the transposed rotary matrix in fp32, rounded again
                const float* cr = p.rope_cos + (int64_t)t2 * 64;
                const float* sr = p.rope_sin + (int64_t)t2 * 64;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int f = db * 32 + 8 * i + 4 * e_h;
                        const f32x4_t c4 = *reinterpret_cast<const f32x4_t*>(cr + f), s4 = *reinterpret_cast<const f32x4_t*>(sr + f);
                        float oa[4], ob[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float a = bf2f(f2bf(acc[b][db][4 * i + j] * scale)), bb = bf2f(f2bf(acc[b][db + 2][4 * i + j] * scale));
                            const float sn = -s4[j];
                            oa[j] = a * c4[j] - bb * sn; ob[j] = bb * c4[j] + a * sn;
                        }
                        const u32x2_t wa = {pack2bf(oa[0], oa[1]), pack2bf(oa[2], oa[3])}, wb = {pack2bf(ob[0], ob[1]), pack2bf(ob[2], ob[3])};
                        *reinterpret_cast<u32x2_t*>(row + f) = wa;
                        *reinterpret_cast<u32x2_t*>(row + f + 64) = wb;
                    }
            } else {
#pragma unroll
                for (int db = 0; db < 4; ++db)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const u32x2_t w = {pack2bf(acc[b][db][4 * i] * scale, acc[b][db][4 * i + 1] * scale), pack2bf(acc[b][db][4 * i + 2] * scale, acc[b][db][4 * i + 3] * scale)};
                        *reinterpret_cast<u32x2_t*>(row + db * 32 + 8 * i + 4 * e_h) = w;
                    }
            }
        }
    }
}

// launched by launch_bwd (attn_bwd.hip) where attn_bwd_dq32_kernel qualifies; same grid (256 packed rows per block, kv head in y)
int tr1_launch_attn_bwd_dq64(const AttnParams& p, unsigned blocks_x, hipStream_t s, const float* lse2) {
    const size_t dyn = 4 * (2 * 64 * 256) + 256;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); attr = true; }
    hipLaunchKernelGGL(attn_bwd_dq64_kernel, dim3(blocks_x, (unsigned)p.n_kv), dim3(256), dyn, s, p, lse2);
    return 0;
}
