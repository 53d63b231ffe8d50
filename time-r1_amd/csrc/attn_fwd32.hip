// Attention forward on 32x32x16 MFMA tiles (head dim 128, training / prefill / reference-policy forward; round 3).
// Same two-interval mask model and GQA row packing as attn_fwd.hip (see attn_common.h); K AND V are read ROW-major ([slots, n_kv*128]) - the
// V^T operand of O^T = V^T P^T comes from transposing LDS reads of the V row tile, so no V^T copy is needed for these launches.
//
// Block = 8 waves x 32 packed query rows.  Q stays in registers as the B operand of S^T[kv][q] = K Q^T, so a lane owns ONE query row and half of
// a tile's 64 keys: the online softmax is in-lane (32 values per lane + one exchange with lane ^ 32), and P^T comes out of the accumulator in the
// layout that is the B operand of the PV product after the k-permutation (attn_common.h: pack8).  64-key K / V row tiles stream through a 4-deep
// LDS ring by asm-issued LDS DMA (hipcc would otherwise serialise every LDS read behind the pending DMA; see csrc/attn_bwd.hip), swizzled on
// the source address with skey().  32 MFMAs per wave and tile against 16 b128 + 32 transposing LDS reads.
#include "attn_common.h"
#include <stdlib.h>

__global__ __launch_bounds__(512) void attn_fwd32_kernel(AttnParams p) {
    constexpr int D = 128, NB = 4, TILE = 64 * 256, BUF = 2 * TILE;
    extern __shared__ __attribute__((aligned(256))) char dyn_lds[];  // [NB][K rows | V rows] + block mask summary
    int* lds_meta = reinterpret_cast<int*>(dyn_lds + NB * BUF);      // [8][3]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), c32 = lane & 31, h = lane >> 5;
    const unsigned nR = (unsigned)p.T * (unsigned)p.group;
    const int nqb = (int)((nR + 255u) / 256u);
    // block -> (query block, kv head).  The hardware deals consecutive workgroups to the 8 XCDs round-robin; with 8 % n_kv == 0 every XCD serves ONE
    // kv head (its K / V stay in that XCD's L2) and walks the query blocks from the last (heaviest: most visible keys) to the first.
    int kvh, qb;
    if (p.xcd_pad) {
        const int id = (int)blockIdx.x, xcd = id & 7, per = 8 / p.n_kv;
        kvh = xcd % p.n_kv; qb = (id >> 3) * per + xcd / p.n_kv;
        if (qb >= nqb) return;
    } else { qb = (int)blockIdx.x % nqb; kvh = (int)blockIdx.x / nqb; }
    const unsigned Rw0 = (unsigned)(nqb - 1 - qb) * 256u + (unsigned)wave * 32u;
    const unsigned R = Rw0 + (unsigned)c32;
    const bool valid = R < nR;
    int tq, hq;
    att_split_row(p, valid ? R : nR - 1, tq, hq);
    const int pre = valid ? p.pre[tq] : 0, lo = valid ? p.lo[tq] : 1, hi = valid ? p.hi[tq] : 0;
    int wmaxpre = valid ? pre : 0, wminpre = valid ? pre : 0x7fffffff;
    int wminlo = (valid && hi >= lo) ? lo : 0x7fffffff, wmaxhi = (valid && hi >= lo) ? hi : -1;
    int wmaxlo = valid ? (hi >= lo ? lo : 0x7fffffff) : -1, wminhi = valid ? (hi >= lo ? hi : -1) : 0x7fffffff;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        wmaxpre = max(wmaxpre, __shfl_xor(wmaxpre, o, 64)); wminpre = min(wminpre, __shfl_xor(wminpre, o, 64));
        wminlo = min(wminlo, __shfl_xor(wminlo, o, 64)); wmaxhi = max(wmaxhi, __shfl_xor(wmaxhi, o, 64));
        wmaxlo = max(wmaxlo, __shfl_xor(wmaxlo, o, 64)); wminhi = min(wminhi, __shfl_xor(wminhi, o, 64));
    }
    wmaxpre = __builtin_amdgcn_readfirstlane(wmaxpre); wminpre = __builtin_amdgcn_readfirstlane(wminpre);
    wminlo = __builtin_amdgcn_readfirstlane(wminlo); wmaxhi = __builtin_amdgcn_readfirstlane(wmaxhi);
    wmaxlo = __builtin_amdgcn_readfirstlane(wmaxlo); wminhi = __builtin_amdgcn_readfirstlane(wminhi);
    const bool wave_rows_all = Rw0 + 32u <= nR;
    if (lane == 0) { lds_meta[wave * 3 + 0] = wmaxpre; lds_meta[wave * 3 + 1] = wminlo; lds_meta[wave * 3 + 2] = wmaxhi; }
    bf16x8_t qf[D / 16];                                              // Q row of this lane, features ks*16 + h*8 .. +7 (B operand)
    {
        const bf16_t* qrow = p.Q + (int64_t)tq * p.q_ld + (int64_t)(kvh * p.group + hq) * D;
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) qf[ks] = load_row_frag(qrow, ks * 16 + h * 8, D, valid);
    }
    f32x16_t acc[4];                                                  // O^T[feature block][C layout]
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[db][r] = 0.f;
    float m = NEG_INF, l = 0.f;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) asm volatile("" ::"v"(qf[ks]));        // hipcc places the wait for the Q loads here (see attn_bwd.hip)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int bmaxpre = 0, bminlo = 0x7fffffff, bmaxhi = -1;
#pragma unroll
    for (int w = 0; w < 8; ++w) { bmaxpre = max(bmaxpre, lds_meta[w * 3]); bminlo = min(bminlo, lds_meta[w * 3 + 1]); bmaxhi = max(bmaxhi, lds_meta[w * 3 + 2]); }
    const TileRange tr = att_tile_range(bmaxpre, bminlo, bmaxhi, p.n_slots);
    const int n_my = tr.n_rel;

    const unsigned lds_base = (unsigned)(uintptr_t)(att_lptr_t)dyn_lds;
    const char* kbase = reinterpret_cast<const char*>(p.K) + (int64_t)kvh * 256;
    const char* vbase = reinterpret_cast<const char*>(p.V) + (int64_t)kvh * 256;
    const unsigned k_ldb = (unsigned)p.k_ld * 2u, v_ldb = (unsigned)p.v_ld * 2u;
#define DMA16(voff, sbase, m0v) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff), "s"(sbase) : "memory", "m0")
    auto issue_tile = [&](int tile, int slot) {
        const unsigned buf = lds_base + slot * BUF;
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned row = 4u * (wave * 2 + j) + ((unsigned)ln >> 4);
            unsigned kv = (unsigned)tile * 64u + row; kv = kv < (unsigned)p.n_slots ? kv : (unsigned)p.n_slots - 1u;
            const unsigned ch = (unsigned)(((ln & 15) ^ skey(row & 15)) << 4);
            const unsigned dst = buf + (wave * 2 + j) * 1024;
            DMA16(kv * k_ldb + ch, kbase, dst);
            DMA16(kv * v_ldb + ch, vbase, dst + TILE);
        }
    };
#undef DMA16
#pragma unroll
    for (int j = 0; j < NB - 1; ++j)
        if (j < n_my) issue_tile(att_tile_at(tr, j), j);

    typedef const __attribute__((address_space(3))) bf16x8_t* lds_b128_t;
#define LDS_B128(addr) (*(lds_b128_t)(uintptr_t)(addr))
#define LDS_TR16(addr) __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(uintptr_t)(addr)))
    const int ti = lane & 15, tgrp = (lane >> 4) & 1;
    const unsigned a_lane = (unsigned)(c32 * 256 + ((h ^ skey(c32 & 15)) << 4));
    const unsigned t_lane = (unsigned)((4 * h + (ti >> 2)) * 256 + (ti & 1) * 8 + (((tgrp * 2 + ((ti & 3) >> 1)) ^ (((ti >> 2) << 2) | h)) << 4));

    for (int it = 0; it < n_my; ++it) {
        {
            const int after = (n_my - 1 - it) < (NB - 2) ? (n_my - 1 - it) : (NB - 2);
            if (after >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (after == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                                 // tile `it` is complete for everybody; everybody is done with tile it-1
        asm volatile("" ::: "memory");
        if (it + NB - 1 < n_my) issue_tile(att_tile_at(tr, it + NB - 1), (it + NB - 1) % NB);
        const int kv0 = att_tile_at(tr, it) * 64;
        const bool any = (kv0 < wmaxpre) || (kv0 + 63 >= wminlo && kv0 <= wmaxhi);
        const bool full = wave_rows_all && (kv0 + 64 <= p.n_slots) && ((kv0 + 64 <= wminpre) || (wmaxlo <= kv0 && kv0 + 63 <= wminhi));
        if (any) {
            const unsigned kb_ = lds_base + (unsigned)(it % NB) * BUF;
            // S^T[kv][q]: two independent chains (the tile's 32-key halves)
            f32x16_t cs[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) { cs[0][r] = 0.f; cs[1][r] = 0.f; }
            {
                const unsigned xa = kb_ + a_lane;
                constexpr int AH = 2;
                bf16x8_t k0[AH + 1], k1[AH + 1];
#pragma unroll
                for (int ks = 0; ks < AH; ++ks) { k0[ks] = LDS_B128(xa ^ (ks * 32)); k1[ks] = LDS_B128((xa ^ (ks * 32)) + 8192); }
#pragma unroll
                for (int ks = 0; ks < D / 16; ++ks) {
                    if (ks + AH < D / 16) { k0[(ks + AH) % (AH + 1)] = LDS_B128(xa ^ ((ks + AH) * 32)); k1[(ks + AH) % (AH + 1)] = LDS_B128((xa ^ ((ks + AH) * 32)) + 8192); }
                    cs[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0[ks % (AH + 1)], qf[ks], cs[0], 0, 0, 0);
                    cs[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1[ks % (AH + 1)], qf[ks], cs[1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // lane holds S^T[kv = kv0 + kb*32 + 8i + 4h + j][its query row] in cs[kb][4i + j]; lane ^ 32 holds the other 32 keys of the row
            float mx = NEG_INF;
            if (full) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, cs[kb][r]);
            } else {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kv = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const bool ok = (kv < p.n_slots) & att_visible_nb(kv, pre, lo, hi);
                        const float v = ok ? cs[kb][r] : NEG_INF;
                        cs[kb][r] = v; mx = fmaxf(mx, v);
                    }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m, mx * p.scale_log2);          // max over RAW scores (scale > 0 commutes with max)
            const float m_safe = (m_new == NEG_INF) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m - m_safe);
            float rs = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(cs[kb][r], p.scale_log2, -m_safe));
                    cs[kb][r] = e; rs += e;
                }
            rs += __shfl_xor(rs, 32, 64);
            l = l * alpha + rs; m = m_new;
            if (!__all(alpha == 1.0f)) {      // exact: the running maximum did not move for any row of the wave -> no rescale needed
#pragma unroll
                for (int db = 0; db < 4; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[db][r] *= alpha;
            }
            // O^T[feature][q] += V^T[feature][kv] P^T[kv][q], 16 keys per MFMA; V^T fragments are transposing reads of the V row tile
            const unsigned ya = kb_ + TILE + t_lane;
            constexpr int TH = 2;
            bf16x8_t a[TH + 1];
#define P2_LD(n) make_frag(LDS_TR16((ya ^ (((n) & 3) * 64)) + ((n) >> 2) * 4096), LDS_TR16((ya ^ (((n) & 3) * 64 + 32)) + ((n) >> 2) * 4096 + 2048))
#pragma unroll
            for (int n = 0; n < TH; ++n) a[n] = P2_LD(n);
#pragma unroll
            for (int n = 0; n < 16; ++n) {                            // n = chunk (16 keys) * 4 + feature block
                if (n + TH < 16) a[(n + TH) % (TH + 1)] = P2_LD(n + TH);
                const bf16x8_t f = pack8(cs[n >> 3], ((n >> 2) & 1) * 8);
                acc[n & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[n % (TH + 1)], f, acc[n & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#undef P2_LD
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // all LDS reads of this tile have returned before the barrier that frees its slot
    }
#undef LDS_B128
#undef LDS_TR16
    // lane holds O^T[feature = db*32 + 8i + 4h + j][its query row]
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const unsigned e_R = Rw0 + (unsigned)(tid2 & 31);
    const int e_h = (tid2 >> 5) & 1;
    if (e_R < nR) {
        int t2, hq2;
        att_split_row(p, e_R, t2, hq2);
        const float inv = l > 0.f ? 1.f / l : 0.f;
        bf16_t* row = p.O + (int64_t)t2 * p.o_ld + (int64_t)(kvh * p.group + hq2) * D;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x2_t w = {pack2bf(acc[db][4 * i] * inv, acc[db][4 * i + 1] * inv), pack2bf(acc[db][4 * i + 2] * inv, acc[db][4 * i + 3] * inv)};
                *reinterpret_cast<u32x2_t*>(row + db * 32 + 8 * i + 4 * e_h) = w;
            }
        if (e_h == 0 && p.lse) p.lse[(int64_t)(kvh * p.group + hq2) * p.T + t2] = l > 0.f ? (m + log2f(l)) * 0.6931471805599453f : NEG_INF;
    }
}

// Q/O: [T, n_heads*128]; K, V: [n_slots, n_kv*128] row-major (any leading dims that are multiples of 8); lse (optional): fp32 [n_heads, T].
extern "C" int tr1_attn_fwd_rows(const void* Q, int64_t q_ld, const void* K, int64_t k_ld, const void* V, int64_t v_ld, void* O, int64_t o_ld, void* lse,
                                 const void* pre, const void* lo, const void* hi, int64_t T, int64_t n_heads, int64_t n_kv, int64_t n_slots,
                                 int64_t head_dim, float scale, void* stream) {
    AttnParams p; memset(&p, 0, sizeof(p));
    TR1_CHECK_ARG(head_dim == 128, "attention forward (row-major K / V form): head dim must be 128");
    TR1_CHECK_ARG(n_kv > 0 && n_heads % n_kv == 0, "attention: n_heads must be a multiple of n_kv");
    TR1_CHECK_ARG(q_ld % 8 == 0 && k_ld % 8 == 0 && v_ld % 8 == 0 && o_ld % 4 == 0, "attention: leading dims must be multiples of 8");
    TR1_CHECK_ARG((uint64_t)n_slots * (uint64_t)(k_ld > v_ld ? k_ld : v_ld) * 2ull < 0xffffffffull, "attention: K / V too large for 32-bit DMA offsets");
    p.Q = (const bf16_t*)Q; p.q_ld = q_ld; p.K = (const bf16_t*)K; p.k_ld = k_ld; p.V = (const bf16_t*)V; p.v_ld = v_ld;
    p.O = (bf16_t*)O; p.o_ld = o_ld; p.lse = (float*)lse; p.pre = (const int*)pre; p.lo = (const int*)lo; p.hi = (const int*)hi;
    p.T = (int)T; const bool magic_ok = att_set_group(p, T, (int)(n_heads / n_kv)); p.n_kv = (int)n_kv; p.n_slots = (int)n_slots; p.d_real = 128; p.nsplit = 1;
    p.n_batch = 1;
    p.scale_log2 = scale * 1.4426950408889634f;
    TR1_CHECK_ARG(magic_ok, "attention: T * group^2 must stay below 2^32");
    if (T == 0) return 0;
    const int64_t nR = T * p.group;
    const int nqb = (int)((nR + 255) / 256);
    static int xcd_map = -1;
    if (xcd_map < 0) { const char* e = getenv("TR1_ATTN_XCD"); xcd_map = e ? atoi(e) : 1; }
    unsigned blocks = (unsigned)(nqb * n_kv);
    if (xcd_map && 8 % n_kv == 0) {
        const int per = 8 / (int)n_kv;
        p.xcd_pad = ((nqb + per - 1) / per) * 8;
        blocks = (unsigned)p.xcd_pad;
    }
    const size_t dyn = 4 * (2 * 64 * 256) + 128;
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); attr = true; }
    hipLaunchKernelGGL(attn_fwd32_kernel, dim3(blocks), dim3(512), dyn, (hipStream_t)stream, p);
    TR1_LAUNCH_CHECK();
}
