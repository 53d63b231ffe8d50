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
#ifndef FWD_AH
#define FWD_AH 2        // K fragments / V^T fragments requested ahead of the MFMA that uses them (forward kernel)
#endif
#ifndef FWD_TH
#define FWD_TH 2
#endif
#ifndef TR1_ABL
#define TR1_ABL 0        // timing-ablation bits for the forward loop (tools/build_variant.py <name> -DTR1_ABL=<bits>; results are WRONG by design):
#endif                   // 1 no exp2, 2 no PV MFMAs, 4 no S MFMAs, 8 no per-tile barrier

#ifdef TR1_PROBE
// wave-timeline probe (tools/bench_attn.py --probe-fwd against tools/_probe_lib.so): the stamps of a tile stay in scalar registers, go to a spare LDS
// area at the end of the tile (no global store inside the loop: stores share vmcnt with the tile DMA) and are dumped once when the block is done
__device__ unsigned long long* tr1_fwd_probe = nullptr;
extern "C" int probe_fwd_set_ptr(void* ptr) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(tr1_fwd_probe), &ptr, sizeof(ptr)); }
#define FWD_PROBE_LDS (8 * 48 * 8 * 8)
#define FWD_STAMPS unsigned long long fwd_st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define FWD_STAMP(slot) do { fwd_st_[slot] = __builtin_amdgcn_s_memtime(); } while (0)
#define FWD_FLUSH(it) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (it) < 48) { \
    unsigned long long* pl_ = reinterpret_cast<unsigned long long*>(dyn_lds + 4 * (2 * 64 * 256) + 128); \
    _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) pl_[(((threadIdx.x >> 6) * 48 + (it)) * 8 + s_)] = fwd_st_[s_]; } } while (0)
#define FWD_DUMP() do { if (tr1_fwd_probe && blockIdx.x == 0) { __syncthreads(); \
    const unsigned long long* pl_ = reinterpret_cast<const unsigned long long*>(dyn_lds + 4 * (2 * 64 * 256) + 128); \
    for (int i_ = threadIdx.x; i_ < 8 * 48 * 8; i_ += 512) tr1_fwd_probe[i_] = pl_[i_]; } } while (0)
#else
#define FWD_PROBE_LDS 0
#define FWD_STAMPS do { } while (0)
#define FWD_STAMP(slot) do { } while (0)
#define FWD_FLUSH(it) do { } while (0)
#define FWD_DUMP() do { } while (0)
#endif

#ifndef FWD_LAZY_MAX
#define FWD_LAZY_MAX 6      // log2 units; 0 = rescale whenever a row's maximum moves (round-3 behaviour)
#endif
// KS = k-steps (16 features each) of the S product that can be non-zero, NDB = KS / 2 = 32-feature blocks of O that are computed and stored.  KS = 8: head dim 128.
// KS = 6 (round 6, the vision towers' head dim 80 on 128-wide zero-padded heads, features d < 40 at d and d + 40 at 48 + d): features 96..127 of Q / K / V are zero
// by construction, so 12 + 12 of the 16 + 16 MFMAs of a tile (and the matching LDS fragment reads) carry everything; O columns 96..127 are not written.
template <int KS>
__global__ __launch_bounds__(512) void attn_fwd32_kernel(AttnParams p) {
    constexpr int D = 128, NB = 4, TILE = 64 * 256, BUF = 2 * TILE, NDB = KS / 2;
    static_assert(KS == 8 || KS == 6, "k-steps");
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
    } else if ((p.n_kv & 7) == 0) {
        // >= 8 kv heads (the vision towers: 16 heads, group 1): head h is served by XCD h % 8, which walks that head's query blocks in order - a segment's
        // K / V (0.4 MB at config 3) is then fetched into ONE L2 instead of into all eight (the plain map dealt the query blocks of a head round-robin
        // over the XCDs: 8 x the operand bytes from HBM, the round-2 finding for the 96-wide kernel)
        const int id = (int)blockIdx.x, xcd = id & 7, sq = id >> 3;
        kvh = xcd + 8 * (sq / nqb); qb = sq - (sq / nqb) * nqb;
    } else { qb = (int)blockIdx.x % nqb; kvh = (int)blockIdx.x / nqb; }
    const unsigned Rw0 = (unsigned)(nqb - 1 - qb) * 256u + (unsigned)wave * 32u;
    const unsigned R = Rw0 + (unsigned)c32;
    const bool valid = R < nR;
    int tq, hq;
    att_split_row(p, valid ? R : nR - 1, tq, hq);
    const int pre = valid ? p.pre[tq] : 0, lo = valid ? p.lo[tq] : 1, hi = valid ? p.hi[tq] : 0;
    // the row's two visible intervals clamped to the cache's slots: visible(kv) = kv < pre_e | (unsigned)(kv - lo_e) <= hi_d  (lo_e = INT_MAX: no second interval)
    const int hi_c = hi < p.n_slots ? hi : p.n_slots - 1;
    const int pre_e = pre < p.n_slots ? pre : p.n_slots, lo_e = hi_c >= lo ? lo : 0x7fffffff, hi_d = hi_c >= lo ? hi_c - lo : 0;
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
    // (rows past nR - the padding of the last block - never force the masked path: they compute finite garbage that is not stored.  With an
    //  `all 32 rows valid` term in `full`, the ONE partially valid wave of the heaviest block took the per-element mask path on every tile:
    //  3 900 instead of 1 200 cycles of softmax, all other waves waiting for it at the barrier - wave timeline in DESIGN.md)
    if (lane == 0) { lds_meta[wave * 3 + 0] = wmaxpre; lds_meta[wave * 3 + 1] = wminlo; lds_meta[wave * 3 + 2] = wmaxhi; }
    bf16x8_t qf[KS];                                                  // Q row of this lane, features ks*16 + h*8 .. +7 (B operand)
    {
        const bf16_t* qrow = p.Q + (int64_t)tq * p.q_ld + (int64_t)(kvh * p.group + hq) * D;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = load_row_frag(qrow, ks * 16 + h * 8, D, valid);
    }
    f32x16_t acc[NDB];                                                // O^T[feature block][C layout]
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[db][r] = 0.f;
    float m = NEG_INF, l = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(qf[ks]));            // hipcc places the wait for the Q loads here (see attn_bwd.hip)
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
        FWD_STAMPS;
        FWD_STAMP(0);
        {
            const int after = (n_my - 1 - it) < (NB - 2) ? (n_my - 1 - it) : (NB - 2);
            if (after >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (after == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#if !(TR1_ABL & 8)
        __builtin_amdgcn_s_barrier();                                 // tile `it` is complete for everybody; everybody is done with tile it-1
#endif
        asm volatile("" ::: "memory");
        FWD_STAMP(1);
        if (it + NB - 1 < n_my) issue_tile(att_tile_at(tr, it + NB - 1), (it + NB - 1) % NB);
        const int kv0 = att_tile_at(tr, it) * 64;
        const bool any = (kv0 < wmaxpre) || (kv0 + 63 >= wminlo && kv0 <= wmaxhi);
        const bool full = (kv0 + 64 <= p.n_slots) && ((kv0 + 64 <= wminpre) || (wmaxlo <= kv0 && kv0 + 63 <= wminhi));
        if (any) {
            const unsigned kb_ = lds_base + (unsigned)(it % NB) * BUF;
            // S^T[kv][q]: two independent chains (the tile's 32-key halves)
            f32x16_t cs[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) { cs[0][r] = 0.f; cs[1][r] = 0.f; }
            {
                const unsigned xa = kb_ + a_lane;
                constexpr int AH = FWD_AH;
                bf16x8_t k0[AH + 1], k1[AH + 1];
#pragma unroll
                for (int ks = 0; ks < AH; ++ks) { k0[ks] = LDS_B128(xa ^ (ks * 32)); k1[ks] = LDS_B128((xa ^ (ks * 32)) + 8192); }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks + AH < KS) { k0[(ks + AH) % (AH + 1)] = LDS_B128(xa ^ ((ks + AH) * 32)); k1[(ks + AH) % (AH + 1)] = LDS_B128((xa ^ ((ks + AH) * 32)) + 8192); }
#if !(TR1_ABL & 4)
                    cs[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0[ks % (AH + 1)], qf[ks], cs[0], 0, 0, 0);
                    cs[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1[ks % (AH + 1)], qf[ks], cs[1], 0, 0, 0);
#else
                    cs[0][ks] += bf2f((bf16_t)k0[ks % (AH + 1)][0]); cs[1][ks] += bf2f((bf16_t)k1[ks % (AH + 1)][0]);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            FWD_STAMP(2);
            // lane holds S^T[kv = kv0 + kb*32 + 8i + 4h + j][its query row] in cs[kb][4i + j]; lane ^ 32 holds the other 32 keys of the row
            // (vector-ALU work is what this kernel pays for next to its MFMAs - the two do not overlap on a SIMD, see DESIGN.md: the masked path
            //  only rewrites cs in place, so both paths share ONE copy of the code below and no register copies appear at the join; the max runs
            //  as two independent v_max3 chains; scale / subtract and the row sum are packed-fp32 instructions, two values each)
            if (!full) {
                // (round 6) vector instructions only - two compares + two selects per score on the clamped intervals [0, pre_e) and [lo_e, lo_e + hi_d]: the lane-mask form
                // (v_cmp -> s_and / s_or on the masks -> v_cndmask) put scalar instructions that wait for vector results between the selects (attn_fwd64.hip measured
                // 4 600 against 2 600 cycles per masked tile); same values
                const int base = kv0 + 4 * h;
                const unsigned mA = (unsigned)(base - lo_e), mD = (unsigned)hi_d;
                const int mB = pre_e - base;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = kb * 32 + (r & 3) + 8 * (r >> 2);
                        const float sv = cs[kb][r];
                        float x = (mA + (unsigned)c <= mD) ? sv : NEG_INF;
                        asm volatile("" : "+v"(x));
                        x = (c < mB) ? sv : x;
                        cs[kb][r] = x;
                    }
            }
            float mxa = NEG_INF, mxb = NEG_INF;
#pragma unroll
            for (int r = 0; r < 16; r += 2) { mxa = att_max3(mxa, cs[0][r], cs[0][r + 1]); mxb = att_max3(mxb, cs[1][r], cs[1][r + 1]); }
            float mx = fmaxf(mxa, mxb);
            { float mx0_, mx1_; att_halves(mx, mx0_, mx1_); mx = fmaxf(mx0_, mx1_); }
            // Lazy running maximum: when no row of the wave would raise its maximum by more than FWD_LAZY_MAX (log2 units), the OLD maximum stays the
            // reference point of this tile - alpha == 1 for every row, so the 64 accumulator multiplies per lane are skipped and P = exp2(s - m_old) <= 2^6;
            // l, the accumulators and the LSE (m + log2 l) stay mutually consistent because all three use the same reference.  With the exact test
            // (maximum unchanged) the skip only fired late in a row's key range: a tile of 64 new keys holds a new row maximum with probability 1 / (t + 1).
            const float m_cand = fmaxf(m, mx * p.scale_log2);         // max over RAW scores (scale > 0 commutes with max)
#if FWD_LAZY_MAX > 0
            const float m_new = __all(m_cand - m <= (float)FWD_LAZY_MAX) ? m : m_cand;      // (m = -inf: the difference is inf or NaN -> false -> take the new maximum)
#else
            const float m_new = m_cand;
#endif
            const float m_safe = (m_new == NEG_INF) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m - m_safe);
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            const f32x2_t sc2 = {p.scale_log2, p.scale_log2}, nm2 = {-m_safe, -m_safe};
            f32x2_t rs2 = {0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f32x2_t x = {cs[kb][r], cs[kb][r + 1]};
                    x = __builtin_elementwise_fma(x, sc2, nm2);
#if !(TR1_ABL & 1)
                    const f32x2_t e = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
#else
                    const f32x2_t e = x;
#endif
                    cs[kb][r] = e[0]; cs[kb][r + 1] = e[1];
                    rs2 += e;
                }
            float rs = rs2[0] + rs2[1];
            { float rs0_, rs1_; att_halves(rs, rs0_, rs1_); rs = rs0_ + rs1_; }
            l = l * alpha + rs; m = m_new;
            if (!__all(alpha == 1.0f)) {      // exact: the running maximum did not move for any row of the wave -> no rescale needed
#pragma unroll
                for (int db = 0; db < NDB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[db][r] *= alpha;
            }
            FWD_STAMP(3);
            // O^T[feature][q] += V^T[feature][kv] P^T[kv][q], 16 keys per MFMA; V^T fragments are transposing reads of the V row tile
            const unsigned ya = kb_ + TILE + t_lane;
            constexpr int TH = FWD_TH;
            bf16x8_t a[TH + 1];
#define P2_LD(n) make_frag(LDS_TR16((ya ^ (((n) % NDB) * 64)) + ((n) / NDB) * 4096), LDS_TR16((ya ^ (((n) % NDB) * 64 + 32)) + ((n) / NDB) * 4096 + 2048))
#pragma unroll
            for (int n = 0; n < TH; ++n) a[n] = P2_LD(n);
#pragma unroll
            for (int n = 0; n < 4 * NDB; ++n) {                       // n = chunk (16 keys) * NDB + feature block
                if (n + TH < 4 * NDB) a[(n + TH) % (TH + 1)] = P2_LD(n + TH);
                const bf16x8_t f = pack8(cs[(n / NDB) >> 1], ((n / NDB) & 1) * 8);
#if !(TR1_ABL & 2)
                acc[n % NDB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[n % (TH + 1)], f, acc[n % NDB], 0, 0, 0);
#else
                acc[n % NDB][n] += bf2f((bf16_t)a[n % (TH + 1)][0]) * bf2f((bf16_t)f[0]);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
#undef P2_LD
            FWD_STAMP(4);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // all LDS reads of this tile have returned before the barrier that frees its slot
        FWD_STAMP(5);
        FWD_FLUSH(it);
    }
    FWD_DUMP();
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
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x2_t w = {pack2bf(acc[db][4 * i] * inv, acc[db][4 * i + 1] * inv), pack2bf(acc[db][4 * i + 2] * inv, acc[db][4 * i + 3] * inv)};
                *reinterpret_cast<u32x2_t*>(row + db * 32 + 8 * i + 4 * e_h) = w;
            }
        if (e_h == 0 && p.lse) p.lse[(int64_t)(kvh * p.group + hq2) * p.T + t2] = l > 0.f ? (m + log2f(l)) * 0.6931471805599453f : NEG_INF;
    }
}

// ---------------------------------------------------------------------------------------------------------- split-KV decode (head dim 128)
// One block = 64 packed query rows (2 waves x 32) of one (prompt, kv head) x one split of the key tiles; launched for the layers of a decode step
// that read the per-step tile plan (plan_mode 2).  The old decode kernel stages K / V^T tiles through registers, where hipcc drains every load
// before each tile is written to LDS; here all of a block's tiles (2-3 at config 3) are requested at once by asm-issued LDS DMA and consumed
// behind hand-counted vmcnt waits, so a block costs ONE memory round trip.  K tile: 64 rows x 256 B (swizzled with skey); V^T tile: 128 feature
// rows x 128 B straight from the cache's V^T layout (chunk ^ (row & 7)).  Partials go to the same workspace the combine kernel reads.
#define DEC32_CAP 1024      // = ATT_LIST_CAP of attn_fwd.hip (plan entry layout: [CAP ids | count])
#ifdef TR1_PROBE
__device__ unsigned long long* tr1_dec_probe = nullptr;          // [blocks][4 waves][12 stamps]  (tools/bench_attn_decode.py PROBE=1, -DTR1_PROBE build)
extern "C" int probe_dec_set_ptr(void* ptr) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(tr1_dec_probe), &ptr, sizeof(ptr)); }
#define DEC_STAMPS unsigned long long dst_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define DEC_STAMP(i) do { dst_[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define DEC_FLUSH() do { if (tr1_dec_probe && (threadIdx.x & 63) == 0) { const size_t bid_ = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; \
    _Pragma("unroll") for (int s_ = 0; s_ < 12; ++s_) tr1_dec_probe[(bid_ * 4 + (threadIdx.x >> 6)) * 12 + s_] = dst_[s_]; } } while (0)
#else
#define DEC_STAMPS do { } while (0)
#define DEC_STAMP(i) do { } while (0)
#define DEC_FLUSH() do { } while (0)
#endif
__global__ __launch_bounds__(512) void attn_dec32_kernel(AttnParams p) {
    constexpr int D = 128, NB = 4, KT = 64 * 256, VT = 128 * 128, BUF = KT + VT;
    DEC_STAMPS;
    DEC_STAMP(0);
    extern __shared__ __attribute__((aligned(256))) char dyn_lds[];  // [NB][K rows | V^T rows] | Q rows (64 x 256 B) | pre, lo, hi [64] | meta
    constexpr int QOFF = NB * BUF, MOFF = QOFF + 64 * 256;
    int* lds_meta = reinterpret_cast<int*>(dyn_lds + MOFF + 768);    // [2][3]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), c32 = lane & 31, h = lane >> 5;
    const int gx = gridDim.x, gy = gridDim.y, by = blockIdx.y, split = blockIdx.z;
    const int b = by / p.n_kv, kvh = by - b * p.n_kv;
    const int qtile = gx - 1 - (int)blockIdx.x;
    p.Q += (int64_t)b * p.T * p.q_ld; p.pre += (int64_t)b * p.T; p.lo += (int64_t)b * p.T; p.hi += (int64_t)b * p.T;
    const unsigned nR = (unsigned)p.T * (unsigned)p.group;
    const int* plan_blk = p.plan + ((int64_t)b * gx + qtile) * (DEC32_CAP + 1);
    // the plan words this block can need for its first NB tiles, requested TOGETHER at kernel entry (independent scalar loads, one wait much
    // later): read one by one where they are used, they were four dependent L2 round trips (~3 us) in front of the late tiles' DMA
    int plan_w[NB + 1];
    plan_w[NB] = plan_blk[DEC32_CAP];
#pragma unroll
    for (int j = 0; j < NB; ++j) { const int idx = split + j * p.nsplit; plan_w[j] = plan_blk[idx < DEC32_CAP ? idx : DEC32_CAP - 1]; }

    const unsigned lds_base = (unsigned)(uintptr_t)(att_lptr_t)dyn_lds;
    const char* kbase = reinterpret_cast<const char*>(p.K) + ((int64_t)b * p.kv_batch_slots * p.k_ld + (int64_t)kvh * D) * 2;
    const char* vbase = reinterpret_cast<const char*>(p.VT) + ((int64_t)kvh * D * p.vt_ld + (int64_t)b * p.kv_batch_slots) * 2;
    const unsigned k_ldb = (unsigned)p.k_ld * 2u, vt_ldb = (unsigned)p.vt_ld * 2u;
    // 8 waves: waves 0..3 compute - (row half rw, key half kh) of every 64 x 64 tile, so a wave's share of a tile is 8 + 8 MFMAs and 16 score
    // elements per lane (the block is a latency chain: per-wave work is what the tiles cost) - and all 8 issue DMA, 4 instructions per tile and
    // wave (an LDS-DMA instruction costs 60-280 issue cycles).  Lane constants of the DMA are hoisted; a tile adds one scalar multiple.
    unsigned koff[2], voff[2], kdst[2], vdst[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = wave * 2 + j;
        const unsigned krow = 4u * i + ((unsigned)lane >> 4);        // K rows 4i .. 4i+3 (256 B each): lane -> row, physical chunk lane%16
        koff[j] = krow * k_ldb + (unsigned)(((lane & 15) ^ skey(krow & 15)) << 4);
        const unsigned vrow = 8u * i + ((unsigned)lane >> 3);        // V^T rows 8i .. 8i+7 (128 B = 64 slots each): physical chunk lane%8 = logical ^ (row & 7)
        voff[j] = vrow * vt_ldb + (unsigned)(((lane & 7) ^ (vrow & 7)) << 4);
        kdst[j] = (unsigned)i * 1024u; vdst[j] = (unsigned)KT + (unsigned)i * 1024u;
    }
#define DMA16(voff_, sbase, m0v) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff_), "s"(sbase) : "memory", "m0")
    auto issue_tile = [&](int tile, int slot) {                       // (cache regions are whole tiles: s_cap % 64 == 0, checked by the launcher)
        const unsigned buf = lds_base + slot * BUF;
        const unsigned tk = (unsigned)tile * 64u * k_ldb, tv = (unsigned)tile * 128u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            DMA16(koff[j] + tk, kbase, buf + __builtin_amdgcn_readfirstlane(kdst[j]));
            DMA16(voff[j] + tv, vbase, buf + __builtin_amdgcn_readfirstlane(vdst[j]));
        }
    };
#undef DMA16
    const bool cw = wave < 4;                                         // compute wave
    const int rw = wave & 1, kh = (wave >> 1) & 1;
    const unsigned Rw0 = (unsigned)qtile * 64u + (unsigned)rw * 32u;
    const unsigned R = Rw0 + (unsigned)c32;
    const bool valid = R < nR && cw;
    int tq, hq;
    att_split_row(p, R < nR ? R : nR - 1, tq, hq);
    // Order of the block's memory requests (everything by hand: any load hipcc tracks itself would make it wait with vmcnt(0), i.e. for all
    // the DMA as well): 1. the block's FIRST tile, speculated as tile `split` of the shared prefix (true for every decode row set whose prefix
    // has more than nsplit tiles) - requested before the plan is even read; 2. Q fragments + the three mask words of the lane's row;
    // 3. the plan (two dependent scalar loads); 4. the remaining tiles.  Tile 0 is then computed under the flight of the others.
    const int spec_tile = split;
#ifdef DEC_NO_SPEC
    const bool spec = false;
#else
    const bool spec = spec_tile * 64 < p.n_slots;
#endif
    if (spec) issue_tile(spec_tile, 0);
    DEC_STAMP(1);
    // Q rows and the row masks travel by LDS DMA like the tiles (2 + (wave 0: 3) instructions): EVERY input is then ordered by the hand-counted
    // vmcnt waits, and hipcc sees LDS reads only.  (Plain loads written in asm were tried: hipcc may copy their destination registers before the
    // hand-placed wait - it believes them defined at the asm statement - and the copies then hold stale data.)
    {
#define DMA16Q(voff_, sbase, m0v) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff_), "s"(sbase) : "memory", "m0")
#define DMA4Q(voff_, sbase, m0v) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(m0v), "v"(voff_), "s"(sbase) : "memory", "m0")
        const char* qbase = reinterpret_cast<const char*>(p.Q) + (int64_t)kvh * p.group * 256;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = wave * 2 + j;                               // packed rows 4i .. 4i+3 of the block's 64
            const unsigned row = 4u * i + ((unsigned)lane >> 4);
            unsigned Rr = (unsigned)qtile * 64u + row; Rr = Rr < nR ? Rr : nR - 1u;
            const unsigned tu = p.group == 1 ? Rr : __umulhi(Rr, p.group_magic);
            const unsigned off = tu * ((unsigned)p.q_ld * 2u) + (Rr - tu * (unsigned)p.group) * 256u + (unsigned)(((lane & 15) ^ skey(row & 15)) << 4);
            DMA16Q(off, qbase, lds_base + QOFF + (unsigned)i * 1024u);
        }
        if (wave == 0) {
            unsigned Rr = (unsigned)qtile * 64u + (unsigned)lane; Rr = Rr < nR ? Rr : nR - 1u;
            const unsigned to = (p.group == 1 ? Rr : __umulhi(Rr, p.group_magic)) * 4u;
            DMA4Q(to, p.pre, lds_base + MOFF);
            DMA4Q(to, p.lo, lds_base + MOFF + 256);
            DMA4Q(to, p.hi, lds_base + MOFF + 512);
        }
#undef DMA16Q
#undef DMA4Q
    }
    DEC_STAMP(2);
    const int plan_n = __builtin_amdgcn_readfirstlane(plan_w[NB]);
    DEC_STAMP(3);
    int n_my; TileRange tr{0, 0, 0};
    bool spec_hit = false;
    int first_late = 0;                                               // tiles requested after the plain loads (their DMA is younger than Q / masks)
    if (plan_n >= 0) {
        n_my = (split < plan_n) ? (plan_n - split + p.nsplit - 1) / p.nsplit : 0;
        const int t0 = n_my > 0 ? __builtin_amdgcn_readfirstlane(plan_w[0]) : -1;
        spec_hit = spec && t0 == spec_tile;
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (j < n_my && !(j == 0 && spec_hit)) { issue_tile(__builtin_amdgcn_readfirstlane(plan_w[j]), j); ++first_late; }
    } else {
        // no plan for this step (tile list too long for a reader block): the block walks its contiguous tile ranges; needs the masks first
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const bool v0 = valid && kh == 0;
        const int* mk = reinterpret_cast<const int*>(dyn_lds + MOFF) + rw * 32 + c32;
        const int pre_ = v0 ? mk[0] : 0, lo_ = v0 ? mk[64] : 1, hi_ = v0 ? mk[128] : 0;
        int wmaxpre = pre_, wminlo = (v0 && hi_ >= lo_) ? lo_ : 0x7fffffff, wmaxhi = (v0 && hi_ >= lo_) ? hi_ : -1;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            wmaxpre = max(wmaxpre, __shfl_xor(wmaxpre, o, 64)); wminlo = min(wminlo, __shfl_xor(wminlo, o, 64)); wmaxhi = max(wmaxhi, __shfl_xor(wmaxhi, o, 64));
        }
        if (lane == 0 && wave < 2) { lds_meta[wave * 3 + 0] = wmaxpre; lds_meta[wave * 3 + 1] = wminlo; lds_meta[wave * 3 + 2] = wmaxhi; }
        __syncthreads();
        tr = att_tile_range(max(lds_meta[0], lds_meta[3]), min(lds_meta[1], lds_meta[4]), max(lds_meta[2], lds_meta[5]), p.n_slots);
        n_my = (split < tr.n_rel) ? (tr.n_rel - split + p.nsplit - 1) / p.nsplit : 0;
        spec_hit = spec && n_my > 0 && att_tile_at(tr, split) == spec_tile;
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (j < n_my && !(j == 0 && spec_hit)) { issue_tile(att_tile_at(tr, split + j * p.nsplit), j); ++first_late; }
    }
    DEC_STAMP(4);
    const int pw0 = __builtin_amdgcn_readfirstlane(plan_w[0]), pw1 = __builtin_amdgcn_readfirstlane(plan_w[1]), pw2 = __builtin_amdgcn_readfirstlane(plan_w[2]),
              pw3 = __builtin_amdgcn_readfirstlane(plan_w[3]);
    auto dec_tile = [&](int i) -> int {      // tile id of the block's i-th tile (scalar selects: no register-relative indexing)
        if (plan_n < 0) return att_tile_at(tr, split + i * p.nsplit);
        return i == 0 ? pw0 : i == 1 ? pw1 : i == 2 ? pw2 : i == 3 ? pw3 : __builtin_amdgcn_readfirstlane(plan_blk[split + i * p.nsplit]);
    };
#define DEC_TILE(i) dec_tile(i)
    f32x16_t acc[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[db][r] = 0.f;
    float m = NEG_INF, l = 0.f;
    // Q, masks and (on a speculation hit) tile 0 have landed once only the late tiles' DMA is outstanding: 4 instructions per tile and wave
    if (spec_hit) {
        if (first_late >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (first_late == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (first_late == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DEC_STAMP(5);
    __builtin_amdgcn_s_barrier();                                     // Q rows, masks (and tile 0) of every wave's DMA share are in LDS
    asm volatile("" ::: "memory");
    bf16x8_t qf[D / 16];
    int pre, lo, hi;
    {
        const unsigned qa = lds_base + QOFF + (unsigned)(rw * 8192 + c32 * 256 + ((h ^ skey(c32 & 15)) << 4));
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) qf[ks] = *(const __attribute__((address_space(3))) bf16x8_t*)(uintptr_t)(qa ^ (ks * 32));
        const int* mk = reinterpret_cast<const int*>(dyn_lds + MOFF) + rw * 32 + c32;
        pre = mk[0]; lo = mk[64]; hi = mk[128];
    }
    if (!valid) { pre = 0; lo = 1; hi = 0; }
    const int hi_c = hi < p.n_slots ? hi : p.n_slots - 1;             // the row's two visible intervals clamped to the cache's slots
    const int pre_e = pre < p.n_slots ? pre : p.n_slots, lo_e = hi_c >= lo ? lo : 0x7fffffff, hi_d = hi_c >= lo ? hi_c - lo : 0;
    int landed = spec_hit ? 1 : (n_my < NB ? n_my : NB);             // tiles whose DMA (this wave's share) is known complete
    const int first_batch = n_my < NB ? n_my : NB;                    // tiles requested before the loop

    typedef const __attribute__((address_space(3))) bf16x8_t* lds_b128_t;
    typedef const __attribute__((address_space(3))) u32x2_t* lds_b64_t;
#define LDS_B128(addr) (*(lds_b128_t)(uintptr_t)(addr))
#define LDS_B64(addr) (*(lds_b64_t)(uintptr_t)(addr))
    const unsigned a_lane = (unsigned)(kh * 8192 + c32 * 256 + ((h ^ skey(c32 & 15)) << 4));      // K rows kh*32 + c32
    // V^T A fragment of feature row d = db*32 + c32, 16-key chunk cc (of this wave's key half: chunks 2kh, 2kh+1): keys 4h..4h+3 (first 8 bytes)
    // and 8+4h..8+4h+3 (second): byte cc*32 + 8h (+16) of the 128-byte row -> logical 16-byte chunk 2cc (+1), half h; physical = logical ^ (row & 7)
    const unsigned v_lane = (unsigned)(c32 * 128 + 8 * h), v_key = (unsigned)((c32 & 7) << 4);

    for (int it = 0; it < n_my; ++it) {
        if (it >= landed) {                                           // this wave's share of tile `it`: wait until only younger tiles are outstanding
            const int issued = it < first_batch ? first_batch : (it + NB - 2 < n_my ? it + NB - 1 : n_my);      // ring refills trail the loop by one tile
            const int younger = issued - 1 - it;
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            landed = it + 1;
        }
        if (it > 0) __builtin_amdgcn_s_barrier();                     // (tile 0: the barrier in front of the Q / mask reads already covered it)
        asm volatile("" ::: "memory");
        if (it < 3) DEC_STAMP(6 + it);
        if (it >= 1 && it + NB - 1 < n_my) issue_tile(DEC_TILE(it + NB - 1), (it + NB - 1) % NB);
        if (!cw) { continue; }
        const int kv0 = DEC_TILE(it) * 64 + kh * 32;                  // first key of this wave's half tile
        const unsigned kb_ = lds_base + (unsigned)(it % NB) * BUF;
        // S^T[32 keys][32 rows]: all 8 K fragments requested up front, two accumulators (even / odd k-steps)
        bf16x8_t kf[D / 16];
        {
            const unsigned xa = kb_ + a_lane;
#pragma unroll
            for (int ks = 0; ks < D / 16; ++ks) kf[ks] = LDS_B128(xa ^ (ks * 32));
        }
        f32x16_t cs, c1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { cs[r] = 0.f; c1[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < D / 16; ks += 2) {
            cs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], cs, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks + 1], qf[ks + 1], c1, 0, 0, 0);
        }
        // the V^T fragments of the wave's two 16-key chunks, requested under the QK MFMAs
        bf16x8_t vf[8];
        {
            const unsigned va = kb_ + KT + v_lane;
#pragma unroll
            for (int n = 0; n < 8; ++n) {                             // n = chunk * 4 + feature block
                const int cc = 2 * kh + (n >> 2), db = n & 3;
                const unsigned ra = va + db * 4096;
                vf[n] = make_frag(LDS_B64(ra + (((unsigned)(2 * cc) << 4) ^ v_key)), LDS_B64(ra + (((unsigned)(2 * cc + 1) << 4) ^ v_key)));
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) cs[r] += c1[r];
        // does every row of the wave see all 32 keys (every prefix tile of a decode step)?  One compare chain + a ballot per tile - a wave
        // reduction of the masks up front cost 15 dependent cross-lane steps (~1 us) in front of the first tile
#ifdef DEC_NO_FULL
        const bool full = false;
#else
        const bool full = (kv0 + 32 <= p.n_slots) && __all(!valid | (kv0 + 32 <= pre) | ((lo <= kv0) & (kv0 + 31 <= hi)));
#endif
        float mx = NEG_INF;
        if (full) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, cs[r]);
        } else {
            // (round 6) vector-only interval mask (two compares + two selects on the clamped intervals; see attn_fwd32_kernel): every suffix tile of a decode step takes this path
            const int base = kv0 + 4 * h;
            const unsigned mA = (unsigned)(base - lo_e), mD = (unsigned)hi_d;
            const int mB = pre_e - base;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = (r & 3) + 8 * (r >> 2);
                const float sv = cs[r];
                float v = (mA + (unsigned)c <= mD) ? sv : NEG_INF;
                asm volatile("" : "+v"(v));
                v = (c < mB) ? sv : v;
                cs[r] = v; mx = fmaxf(mx, v);
            }
        }
        { float mx0_, mx1_; att_halves(mx, mx0_, mx1_); mx = fmaxf(mx0_, mx1_); }
        const float m_new = fmaxf(m, mx * p.scale_log2);
        const float m_safe = (m_new == NEG_INF) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m - m_safe);
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(cs[r], p.scale_log2, -m_safe));
            cs[r] = e; rs += e;
        }
        { float rs0_, rs1_; att_halves(rs, rs0_, rs1_); rs = rs0_ + rs1_; }
        l = l * alpha + rs; m = m_new;
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[db][r] *= alpha;
        }
        const bf16x8_t f0 = pack8(cs, 0), f1 = pack8(cs, 8);
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[n & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[n], (n >> 2) ? f1 : f0, acc[n & 3], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#undef LDS_B128
#undef LDS_B64
#undef DEC_TILE
    DEC_STAMP(9);
    // merge the two key halves of every row (waves rw + 2 hand their running state to waves rw through LDS; ring slot 0 is free by now)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    f32x4_t* xch = reinterpret_cast<f32x4_t*>(dyn_lds) + (size_t)rw * (17 * 64);      // [16 accumulator quads + (m, l)][64 lanes] per row half, 16 B per lane
    if (cw && kh == 1) {
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int i = 0; i < 4; ++i) xch[(db * 4 + i) * 64 + lane] = (f32x4_t){acc[db][4 * i], acc[db][4 * i + 1], acc[db][4 * i + 2], acc[db][4 * i + 3]};
        xch[16 * 64 + lane] = (f32x4_t){m, l, 0.f, 0.f};
    }
    __syncthreads();
    if (valid && kh == 0) {
        const f32x4_t ml = xch[16 * 64 + lane];
        const float m2 = ml[0], l2 = ml[1];
        const float M = fmaxf(m, m2), Ms = (M == NEG_INF) ? 0.f : M;
        const float w1 = __builtin_amdgcn_exp2f(m - Ms), w2 = __builtin_amdgcn_exp2f(m2 - Ms);
        const float L = l * w1 + l2 * w2;
        // partials: O^T accumulators (fp32) + running (m, l), in the workspace layout attn_combine_kernel reads
        const int64_t nRpad = (int64_t)gx * 64;
        const int64_t slot = ((int64_t)split * gy + by) * nRpad + R;
        float* op = p.Opart + slot * D;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4_t o2 = xch[(db * 4 + i) * 64 + lane];
                f32x4_t v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[db][4 * i + j] * w1 + o2[j] * w2;
                *reinterpret_cast<f32x4_t*>(op + db * 32 + 8 * i + 4 * h) = v;
            }
        if (h == 0) { p.mpart[slot] = M; p.lpart[slot] = L; }
    }
    DEC_STAMP(10);
    DEC_FLUSH();
}

// launched by attn_fwd_impl (attn_fwd.hip) for plan_mode 2 launches at head dim 128; returns 0 when the shape does not qualify, 1 when launched
// (partials written: attn_combine_kernel follows)
int tr1_launch_attn_dec32(AttnParams& p, dim3 grid, hipStream_t s) {
    if (p.d_real != 128 || p.plan_mode != 2 || !p.plan || p.n_slots % 64 != 0) return 0;
    const uint64_t kbytes = (uint64_t)p.kv_batch_slots * (uint64_t)p.k_ld * 2ull, vbytes = (uint64_t)128 * (uint64_t)p.vt_ld * 2ull;
    if (kbytes >= 0xffffffffull || vbytes >= 0xffffffffull || (uint64_t)p.n_slots * (uint64_t)p.k_ld * 2ull >= 0xffffffffull) return 0;
    const size_t dyn = 4 * (64 * 256 + 128 * 128) + 64 * 256 + 768 + 64;
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_dec32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); attr = true; }
    hipLaunchKernelGGL(attn_dec32_kernel, grid, dim3(512), dyn, s, p);
    return 1;
}

int tr1_launch_attn_fwd64(AttnParams& p, unsigned blocks, hipStream_t s);      // attn_fwd64.hip
// Q/O: [T, n_heads*128]; K, V: [n_slots, n_kv*128] row-major (any leading dims that are multiples of 8); lse (optional): fp32 [n_heads, T].
static int attn_fwd_rows_impl(const void* Q, int64_t q_ld, const void* K, int64_t k_ld, const void* V, int64_t v_ld, void* O, int64_t o_ld, void* lse,
                              const void* pre, const void* lo, const void* hi, int64_t T, int64_t n_heads, int64_t n_kv, int64_t n_slots,
                              int64_t head_dim, float scale, int live96, void* stream) {
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
    unsigned blocks = (unsigned)(nqb * n_kv);
    if (8 % n_kv == 0) {
        const int per = 8 / (int)n_kv;
        p.xcd_pad = ((nqb + per - 1) / per) * 8;
        blocks = (unsigned)p.xcd_pad;
    }
    const size_t dyn = 4 * (2 * 64 * 256) + 128 + FWD_PROBE_LDS;
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd32_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd32_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        attr = true;
    }
    // head dim 128, all 128 features live, long key ranges: the 64-rows-per-wave kernel (attn_fwd64.hip, bit-identical results).  Its software pipeline pays one
    // extra body per block for fill / drain, so it wins from ~30 key tiles per block on (tools/sweep_fwd64.py, profiles/r06_sweep_fwd64.txt: -8 % at 3 072 prompt
    // tokens, -3 % at 2 048, +1 % at 1 536, +12 % at 256) and loses on the vision towers' 13-tile segments (live-96 form built and measured: 195 against 173 us).
    // TR1_FWD64 = 1 / 0 forces / forbids it (A/B runs, the bit-identity test) - read per call, so a test can switch inside one process.
    const char* f64 = getenv("TR1_FWD64");
    const bool use64 = !live96 && (f64 ? f64[0] != '0' : n_slots >= 3072);
    if (use64) tr1_launch_attn_fwd64(p, blocks, (hipStream_t)stream);
    else if (live96) hipLaunchKernelGGL(attn_fwd32_kernel<6>, dim3(blocks), dim3(512), dyn, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(attn_fwd32_kernel<8>, dim3(blocks), dim3(512), dyn, (hipStream_t)stream, p);
    TR1_LAUNCH_CHECK();
}

extern "C" int tr1_attn_fwd_rows(const void* Q, int64_t q_ld, const void* K, int64_t k_ld, const void* V, int64_t v_ld, void* O, int64_t o_ld, void* lse,
                                 const void* pre, const void* lo, const void* hi, int64_t T, int64_t n_heads, int64_t n_kv, int64_t n_slots,
                                 int64_t head_dim, float scale, void* stream) {
    return attn_fwd_rows_impl(Q, q_ld, K, k_ld, V, v_ld, O, o_ld, lse, pre, lo, hi, T, n_heads, n_kv, n_slots, head_dim, scale, 0, stream);
}

// The same launch for 128-wide heads whose features 96..127 are ZERO in Q, K and V (the vision towers' head dim 80 padded as d -> d, d + 40 -> 48 + d by
// tr1_gemm_qkv_rope_vit_bf16): 24 instead of 32 MFMAs per wave and key tile.  O columns 96..127 of every head are NOT written (the caller keeps them zero).
// ref: VisionAttention.forward TF:379-396 (Qwen2-VL) / the windowed form of Qwen2.5-VL, frozen towers (src/time_r1/rl/timer1_trainer.py:264-269).
extern "C" int tr1_attn_fwd_rows_live96(const void* Q, int64_t q_ld, const void* K, int64_t k_ld, const void* V, int64_t v_ld, void* O, int64_t o_ld, void* lse,
                                        const void* pre, const void* lo, const void* hi, int64_t T, int64_t n_heads, int64_t n_kv, int64_t n_slots,
                                        int64_t head_dim, float scale, void* stream) {
    return attn_fwd_rows_impl(Q, q_ld, K, k_ld, V, v_ld, O, o_ld, lse, pre, lo, hi, T, n_heads, n_kv, n_slots, head_dim, scale, 1, stream);
}
