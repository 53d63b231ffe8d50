// Attention forward, 64 packed query rows per wave, ONE wave per SIMD (head dim 128; round 6).
// Same mask model, GQA row packing, tile images and results as attn_fwd32_kernel (attn_fwd32.hip) - bit for bit: same 32-row softmax groups, same
// lazy-maximum decisions, same summation orders - on a different machine shape.  Round-3/4 timing ablations of that kernel showed its floor to be LDS operand
// traffic plus vector work that never overlapped the MFMAs of the SIMD's other wave: a wave of 32 rows reads the whole K and V tile for 32 MFMAs.
// Here a block is 4 waves x 64 rows (two 32-row "q-blocks" A and B per wave) with the 512-register file of a lone wave:
//   * every K fragment (b128) feeds FOUR MFMAs (2 key halves x 2 q-blocks -> 4 independent accumulator chains), every V^T fragment (2 transposing reads) two:
//     half the LDS bytes per MFMA;
//   * the tile loop is software-pipelined ACROSS tiles inside the wave: body t issues the MFMAs of  O += V^T P^T (tile t-1)  and of  S = K Q^T (tile t+1)
//     - 64 of them, none depending on this body's vector work - and between consecutive MFMAs a fixed slice of the online softmax of tile t
//     (max -> lazy running maximum -> exp2 / row sums / bf16 packing), the LDS fragment reads and the LDS-DMA requests of tiles t+4 (K) / t+2 (V):
//     ~4.5 vector instructions per MFMA gap, placed by hand (one scheduling fence per MFMA: the order below IS the instruction order);
//   * K and V tiles travel in separate 4-deep rings (K is consumed one body before, V one body after the tile's softmax), requests are unconditional
//     (the tile index is clamped to the block's last tile) so every wait is the same counted vmcnt.
// Reference semantics: flash_attn_varlen_func via attn_implementation=flash_attention_2 (/root/reference/scripts/posttrain/train_rl.sh:33), Qwen2VLAttention TF:521-556.
#include "attn_common.h"
#include <stdlib.h>

#ifndef F64_AH
#define F64_AH 2
#endif
#ifndef F64_TH
#define F64_TH 2
#endif
#ifndef F64_KREG
#define F64_KREG "v"
#endif
#ifndef FWD64_LAZY_MAX
#define FWD64_LAZY_MAX 6      // = FWD_LAZY_MAX of attn_fwd32.hip (the two kernels must take the same rescale decisions)
#endif

#ifdef TR1_PROBE
// wave-timeline probe (tools/check_fwd64.py --probe against tools/_probe_lib.so): stamps stay in scalar registers, go to a spare LDS area per tile, one dump at the end
__device__ unsigned long long* tr1_fwd64_probe = nullptr;          // [4 waves][64 tiles][8 stamps]
extern "C" int probe_fwd64_set_ptr(void* ptr) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(tr1_fwd64_probe), &ptr, sizeof(ptr)); }
#define F64_PROBE_LDS (4 * 64 * 8 * 8)
#define F64_STAMPS unsigned long long f64_st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define F64_STAMP(slot) do { f64_st_[slot] = __builtin_amdgcn_s_memtime(); } while (0)
#define F64_FLUSH(it) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (it) < 64) { \
    unsigned long long* pl_ = reinterpret_cast<unsigned long long*>(dyn_lds + 8 * (64 * 256) + 128); \
    _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) pl_[(((threadIdx.x >> 6) * 64 + (it)) * 8 + s_)] = f64_st_[s_]; } } while (0)
#define F64_DUMP() do { if (tr1_fwd64_probe && blockIdx.x == 0) { __syncthreads(); \
    const unsigned long long* pl_ = reinterpret_cast<const unsigned long long*>(dyn_lds + 8 * (64 * 256) + 128); \
    for (int i_ = threadIdx.x; i_ < 4 * 64 * 8; i_ += 256) tr1_fwd64_probe[i_] = pl_[i_]; } } while (0)
#else
#define F64_PROBE_LDS 0
#define F64_STAMPS do { } while (0)
#define F64_STAMP(slot) do { } while (0)
#define F64_FLUSH(it) do { } while (0)
#define F64_DUMP() do { } while (0)
#endif
// both key halves of a query row meet: x -> (value of the row's lane < 32, value of its lane >= 32) in every lane.  v_permlane32_swap is a vector instruction;
// __shfl_xor(x, 32) is a ds_bpermute whose lgkmcnt(0) wait also drains every fragment read in flight - four of those per tile cost 600 of the body's 4 000 cycles
// (profiles/r06_fwd64_timeline.txt).  (The builtin's second result is mis-assigned by this hipcc - both halves read back vdst - hence assembly; the leading
// nops are the VALU-write -> permlane read wait states.)
TR1_DEV void f64_halves(float x, float& a, float& b) {
    a = x; b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
// max of two without the canonicalising v_max x, x that fmaxf() costs per operand (no NaN reaches the statistics: -inf masks, finite inputs)
TR1_DEV float f64_max(float a, float b) { float o; asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b)); return o; }
// The S product's MFMAs are written in assembly to pin their register files: hipcc gives a kernel with a 512-register budget the accumulator-file form of
// EVERY MFMA (C / D in AGPRs), and the softmax - vector instructions cannot read AGPRs - then costs 200+ v_accvgpr moves per tile.  S lives in arch VGPRs
// (C / D "v"), the K fragment in VGPRs, the Q fragment in the accumulator file ("a": it is only ever an MFMA operand); O's accumulators stay with the builtin
// (AGPRs).  Nothing reads an S tile before the next body (>= 12 wait states after its last MFMA: s_product ends with explicit nops).
TR1_DEV void f64_mfma_s0(f32x16_t& d, bf16x8_t k, bf16x8_t q) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : F64_KREG(k), "a"(q)); }
TR1_DEV void f64_mfma_s(f32x16_t& d, bf16x8_t k, bf16x8_t q) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : F64_KREG(k), "a"(q)); }

template <int KS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_fwd64_kernel(AttnParams p) {
    static_assert(KS == 8, "head dim 128");
    constexpr int D = 128, NB = 4, TILE = 64 * 256, AH = F64_AH, TH = F64_TH;
    extern __shared__ __attribute__((aligned(256))) char dyn_lds[];  // [NB] K row tiles | [NB] V row tiles | block mask summary [8][3]
    int* lds_meta = reinterpret_cast<int*>(dyn_lds + 2 * NB * TILE);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), c32 = lane & 31, h = lane >> 5;
    const unsigned nR = (unsigned)p.T * (unsigned)p.group;
    const int nqb = (int)((nR + 255u) / 256u);
    int kvh, qblk;                                                    // block -> (query block, kv head): the map of attn_fwd32_kernel
    if (p.xcd_pad) {
        const int id = (int)blockIdx.x, xcd = id & 7, per = 8 / p.n_kv;
        kvh = xcd % p.n_kv; qblk = (id >> 3) * per + xcd / p.n_kv;
        if (qblk >= nqb) return;
    } else if ((p.n_kv & 7) == 0) {
        const int id = (int)blockIdx.x, xcd = id & 7, sq = id >> 3;
        kvh = xcd + 8 * (sq / nqb); qblk = sq - (sq / nqb) * nqb;
    } else { qblk = (int)blockIdx.x % nqb; kvh = (int)blockIdx.x / nqb; }
    const unsigned Rw0 = (unsigned)(nqb - 1 - qblk) * 256u + (unsigned)wave * 64u;
    int pre_e[2], lo_e[2], hi_d[2];                                   // per row: visible(kv) = kv < pre_e | (unsigned)(kv - lo_e) <= hi_d   (clamped to the cache's slots; lo_e = INT_MAX: no second interval)
    int wmaxpre[2], wminpre[2], wmaxlo[2], wminhi[2];
    bf16x8_t qf[2][KS];                                               // Q rows of this lane (q-blocks A / B), features ks*16 + h*8 .. +7 (B operands)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const unsigned R = Rw0 + (unsigned)b * 32u + (unsigned)c32;
        const bool valid = R < nR;
        int tq, hq;
        att_split_row(p, valid ? R : nR - 1, tq, hq);
        const int pre_b = valid ? p.pre[tq] : 0, lo_b = valid ? p.lo[tq] : 1, hi_b = valid ? p.hi[tq] : 0;
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
        wmaxpre[b] = __builtin_amdgcn_readfirstlane(a0); wminpre[b] = __builtin_amdgcn_readfirstlane(a1);
        const int wminlo = __builtin_amdgcn_readfirstlane(a2), wmaxhi = __builtin_amdgcn_readfirstlane(a3);
        wmaxlo[b] = __builtin_amdgcn_readfirstlane(a4); wminhi[b] = __builtin_amdgcn_readfirstlane(a5);
        if (lane == 0) { lds_meta[(wave * 2 + b) * 3 + 0] = wmaxpre[b]; lds_meta[(wave * 2 + b) * 3 + 1] = wminlo; lds_meta[(wave * 2 + b) * 3 + 2] = wmaxhi; }
        const bf16_t* qrow = p.Q + (int64_t)tq * p.q_ld + (int64_t)(kvh * p.group + hq) * D;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[b][ks] = load_row_frag(qrow, ks * 16 + h * 8, D, valid);
    }
    f32x16_t acc[2][4];                                               // O^T[q-block][feature block][C layout]
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][db][r] = 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int db = 0; db < 4; ++db) asm volatile("" : "+a"(acc[b][db]));      // the loop-carried tiles start in the accumulator file (else the loop header copies all 128 out and back in)
    float m[2] = {NEG_INF, NEG_INF}, l[2] = {0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(qf[b][ks]));      // hipcc places the wait for the Q loads here: from now on vmcnt counts DMA only
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+a"(qf[b][ks]));    // Q lives in the accumulator file from here on (else: 64 v_accvgpr_write per tile in front of the S MFMAs)
    __syncthreads();
    int bmaxpre = 0, bminlo = 0x7fffffff, bmaxhi = -1;
#pragma unroll
    for (int w = 0; w < 8; ++w) { bmaxpre = max(bmaxpre, lds_meta[w * 3]); bminlo = min(bminlo, lds_meta[w * 3 + 1]); bmaxhi = max(bmaxhi, lds_meta[w * 3 + 2]); }
    TileRange tr = att_tile_range(bmaxpre, bminlo, bmaxhi, p.n_slots);
    tr.pre_tiles = __builtin_amdgcn_readfirstlane(tr.pre_tiles); tr.start2 = __builtin_amdgcn_readfirstlane(tr.start2);      // tile ids stay in scalar registers
    const int n_my = __builtin_amdgcn_readfirstlane(tr.n_rel);

    const unsigned lds_base = (unsigned)(uintptr_t)(att_lptr_t)dyn_lds;
    const char* kbase = reinterpret_cast<const char*>(p.K) + (int64_t)kvh * 256;
    const char* vbase = reinterpret_cast<const char*>(p.V) + (int64_t)kvh * 256;
    const unsigned k_ldb = (unsigned)p.k_ld * 2u, v_ldb = (unsigned)p.v_ld * 2u;
    // LDS DMA of one tile half (K or V rows): 16 instructions of 1 KiB (4 rows x 256 B), 4 per wave; lane constants hoisted (a lone wave has the registers)
    unsigned koff[4], voff[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const unsigned row = 4u * (unsigned)(wave * 4 + d) + ((unsigned)lane >> 4);
        const unsigned ch = (unsigned)(((lane & 15) ^ skey(row & 15)) << 4);
        koff[d] = row * k_ldb + ch; voff[d] = row * v_ldb + ch;
    }
    // rows past the cache's last slot (the one partial tile): the offset is clamped to the LAST 16-byte chunk of the last row - wrong chunk, real (finite) bf16 data,
    // and those keys are masked to P = 0 exactly (kv >= n_slots)
    const unsigned klim = ((unsigned)p.n_slots - 1u) * k_ldb + 240u, vlim = ((unsigned)p.n_slots - 1u) * v_ldb + 240u;
#define F64_DMA16(voff_, sbase, m0v) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff_), "s"(sbase) : "memory", "m0")
    // one DMA instruction: d < 4 -> K row group wave*4 + d of tile index ik, else V row group wave*4 + d - 4 of tile index iv (indices into the block's tile list, clamped)
    auto dma_one = [&](int d, int ik, int iv) {
        if (d < 4) {
            const int i = ik < n_my ? ik : n_my - 1;
            const unsigned t64 = (unsigned)att_tile_at(tr, i) * 64u;
            unsigned off = koff[d] + t64 * k_ldb; off = off < klim ? off : klim;
            F64_DMA16(off, kbase, lds_base + (unsigned)(ik % NB) * TILE + (unsigned)(wave * 4 + d) * 1024u);
        } else {
            const int i = iv < n_my ? iv : n_my - 1;
            const unsigned t64 = (unsigned)att_tile_at(tr, i) * 64u;
            unsigned off = voff[d - 4] + t64 * v_ldb; off = off < vlim ? off : vlim;
            F64_DMA16(off, vbase, lds_base + (unsigned)(NB + iv % NB) * TILE + (unsigned)(wave * 4 + d - 4) * 1024u);
        }
    };
    if (n_my > 0) {       // request order K0 V0 K1 V1 K2 K3 (24 instructions per wave); body t adds K(t+4), V(t+2)
#pragma unroll
        for (int d = 0; d < 4; ++d) dma_one(d, 0, 0);
#pragma unroll
        for (int d = 4; d < 8; ++d) dma_one(d, 0, 0);
#pragma unroll
        for (int d = 0; d < 4; ++d) dma_one(d, 1, 0);
#pragma unroll
        for (int d = 4; d < 8; ++d) dma_one(d, 0, 1);
#pragma unroll
        for (int d = 0; d < 4; ++d) dma_one(d, 2, 0);
#pragma unroll
        for (int d = 0; d < 4; ++d) dma_one(d, 3, 0);
    }

    typedef const __attribute__((address_space(3))) bf16x8_t* lds_b128_t;
#define LDS_B128(addr) (*(lds_b128_t)(uintptr_t)(addr))
#define LDS_TR16(addr) __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(uintptr_t)(addr)))
#define P2_LD(ya, n) make_frag(LDS_TR16(((ya) ^ (((n) & 3) * 64)) + ((n) >> 2) * 4096), LDS_TR16(((ya) ^ (((n) & 3) * 64 + 32)) + ((n) >> 2) * 4096 + 2048))
    const int ti = lane & 15, tgrp = (lane >> 4) & 1;
    const unsigned a_lane = (unsigned)(c32 * 256 + ((h ^ skey(c32 & 15)) << 4));
    const unsigned t_lane = (unsigned)((4 * h + (ti >> 2)) * 256 + (ti & 1) * 8 + (((tgrp * 2 + ((ti & 3) >> 1)) ^ (((ti >> 2) << 2) | h)) << 4));

    // (F64_PIN: an empty asm that makes a value opaque where it stands - without it the IR passes, which do not see the scheduling fences, sink and SLP-pack the
    //  row sums into one late v_pk_add chain and keep all 64 exponentials alive for it)
#define F64_PIN(x) asm volatile("" : "+v"(x))
    // ---- the vector program of one tile's softmax, cut into 64 slices (slot s runs between MFMA s and MFMA s + 1 of the body)
    float mxa[2], mxb[2], mcand[2], alpha[2] = {1.f, 1.f}, negm[2], rsE[2], rsO[2];
    auto vstep = [&](int s, f32x16_t (&sC)[2][2], u32x4_t (&pC)[2][4]) {
#ifdef F64_ABL
        if (F64_ABL & 1) return;
        if ((F64_ABL & 2) && s >= 12 && s < 60) return;
        if ((F64_ABL & 4) && s < 12) return;
        if ((F64_ABL & 8) && s >= 8 && s < 12) return;
        if ((F64_ABL & 16) && s < 8) return;
        if ((F64_ABL & 32) && s >= 60) return;
#endif
        if (s < 8) {                                                  // row maximum over the lane's 32 keys: two v_max3 chains per q-block (key halves), ONE asm statement per slot
            const int b = s >> 2, r0 = (s & 3) * 4;                   // (hipcc pads every asm statement whose output the next vector instruction reads with an s_nop)
            if (r0 == 0) { mxa[b] = NEG_INF; mxb[b] = NEG_INF; }
            asm("v_max3_f32 %0, %0, %2, %3\n\tv_max3_f32 %1, %1, %6, %7\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %1, %1, %8, %9"
                : "+v"(mxa[b]), "+v"(mxb[b])
                : "v"(sC[b][0][r0]), "v"(sC[b][0][r0 + 1]), "v"(sC[b][0][r0 + 2]), "v"(sC[b][0][r0 + 3]),
                  "v"(sC[b][1][r0]), "v"(sC[b][1][r0 + 1]), "v"(sC[b][1][r0 + 2]), "v"(sC[b][1][r0 + 3]));
        } else if (s == 8 || s == 10) {
            const int b = (s - 8) >> 1;
            float mx = f64_max(mxa[b], mxb[b]), mx0, mx1;
            f64_halves(mx, mx0, mx1);
            mx = f64_max(mx0, mx1);
            mcand[b] = f64_max(m[b], mx * p.scale_log2);               // max over RAW scores (scale > 0 commutes with max)
            F64_PIN(mcand[b]);
        } else if (s == 9 || s == 11) {
            const int b = (s - 9) >> 1;
            // lazy running maximum, exactly as attn_fwd32_kernel: the old maximum stays the reference while no row of the 32-row group would move by more than 2^6
#if FWD64_LAZY_MAX > 0
            const float m_new = __all(mcand[b] - m[b] <= (float)FWD64_LAZY_MAX) ? m[b] : mcand[b];
#else
            const float m_new = mcand[b];
#endif
            const float m_safe = (m_new == NEG_INF) ? 0.f : m_new;
            alpha[b] = __builtin_amdgcn_exp2f(m[b] - m_safe);
            negm[b] = -m_safe; m[b] = m_new; rsE[b] = 0.f; rsO[b] = 0.f;
            F64_PIN(alpha[b]); F64_PIN(negm[b]); F64_PIN(m[b]);
        } else if (s < 60) {                                          // 32 element pairs x (2 fma, 2 exp2, 2 adds, 1 pack) over 48 slots: 5 / 5 / 4 instructions
            const int j = s - 12, T = j / 3, pos = j - 3 * T, q0 = 2 * T, q1 = 2 * T + 1;
#define F64_REF(q, i) sC[(q) >> 4][((q) >> 3) & 1][((q) & 7) * 2 + (i)]
#define F64_FMA(q, i) F64_REF(q, i) = __builtin_fmaf(F64_REF(q, i), p.scale_log2, negm[(q) >> 4])
#if defined(F64_ABL) && (F64_ABL & 128)
#define F64_EXP(q, i) do { F64_REF(q, i) = F64_REF(q, i) + 1.0f; F64_PIN(F64_REF(q, i)); } while (0)
#else
#define F64_EXP(q, i) do { F64_REF(q, i) = __builtin_amdgcn_exp2f(F64_REF(q, i)); } while (0)      // (not pinned: its consumers - the pinned row sums and packed words - hold it in place)
#endif
#define F64_ADD(q) do { rsE[(q) >> 4] += F64_REF(q, 0); rsO[(q) >> 4] += F64_REF(q, 1); F64_PIN(rsE[(q) >> 4]); F64_PIN(rsO[(q) >> 4]); } while (0)
#define F64_CVT(q) do { unsigned w_ = pack2bf(F64_REF(q, 0), F64_REF(q, 1)); F64_PIN(w_); pC[(q) >> 4][(((q) >> 3) & 1) * 2 + (((q) & 7) >> 2)][(q) & 3] = w_; } while (0)
            if (pos == 0) { F64_FMA(q0, 0); F64_FMA(q0, 1); F64_EXP(q0, 0); F64_EXP(q0, 1); F64_FMA(q1, 0); }
            else if (pos == 1) { F64_ADD(q0); F64_CVT(q0); F64_FMA(q1, 1); F64_EXP(q1, 0); }
            else { F64_EXP(q1, 1); F64_ADD(q1); F64_CVT(q1); }
#undef F64_REF
#undef F64_FMA
#undef F64_EXP
#undef F64_ADD
#undef F64_CVT
        } else if (s < 62) {
            const int b = s - 60;
            float rs = rsE[b] + rsO[b], rs0, rs1;
            f64_halves(rs, rs0, rs1);
            rs = rs0 + rs1;
            l[b] = l[b] * alpha[b] + rs;
            F64_PIN(l[b]);
        }
    };
    // keys of the tile at list index `it` that a q-block's rows cannot all see: -inf before the softmax touches them
    auto mask_tile = [&](int it, f32x16_t (&sC)[2][2]) {
        const int kv0 = att_tile_at(tr, it) * 64;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const bool full = (kv0 + 64 <= p.n_slots) && ((kv0 + 64 <= wminpre[b]) || (wmaxlo[b] <= kv0 && kv0 + 63 <= wminhi[b]));
            if (!full) {
                // vector instructions only (two compares + two selects per score): combining the intervals as lane masks (v_cmp -> s_or_b64 -> v_cndmask) put a
                // scalar instruction that depends on vector results between every pair of selects - 4 600 cycles per masked tile, more than a whole unmasked body
                const int base = kv0 + 4 * h;
                const unsigned A = (unsigned)(base - lo_e[b]), Dm = (unsigned)hi_d[b];
                const int B = pre_e[b] - base;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = kb * 32 + (r & 3) + 8 * (r >> 2);
                        const float sv = sC[b][kb][r];
                        float x = (A + (unsigned)c <= Dm) ? sv : NEG_INF;
                        asm volatile("" : "+v"(x));
                        x = (c < B) ? sv : x;
                        sC[b][kb][r] = x;
                    }
            }
        }
    };
    // S^T[kv][q] of tile index ik for both q-blocks: four independent chains; slot0 >= 0: the softmax slices slot0 .. slot0 + 31 run in the gaps
    auto s_product = [&](int ik, f32x16_t (&sN)[2][2], int slot0, f32x16_t (&sC)[2][2], u32x4_t (&pC)[2][4]) {
        const unsigned xa = lds_base + (unsigned)(ik % NB) * TILE + a_lane;
        bf16x8_t k0[AH + 1], k1[AH + 1];
#pragma unroll
        for (int ks = 0; ks < AH; ++ks) { k0[ks] = LDS_B128(xa ^ (ks * 32)); k1[ks] = LDS_B128((xa ^ (ks * 32)) + 8192); }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + AH < KS) { k0[(ks + AH) % (AH + 1)] = LDS_B128(xa ^ ((ks + AH) * 32)); k1[(ks + AH) % (AH + 1)] = LDS_B128((xa ^ ((ks + AH) * 32)) + 8192); }
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (ks == 0) f64_mfma_s0(sN[w >> 1][w & 1], (w & 1) ? k1[ks % (AH + 1)] : k0[ks % (AH + 1)], qf[w >> 1][ks]);
                else f64_mfma_s(sN[w >> 1][w & 1], (w & 1) ? k1[ks % (AH + 1)] : k0[ks % (AH + 1)], qf[w >> 1][ks]);
                if (slot0 >= 0) vstep(slot0 + ks * 4 + w, sC, pC);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_nop 7" ::: "memory");                         // MFMA D -> vector reader: 12 wait states, none inserted for an asm MFMA (8 here + the fragment
                                                                      // reads, wait, barrier and scalar tile logic that separate the last S MFMA from the next body's first score read)
    };
    // O^T[feature][q] += V^T[feature][kv] P^T[kv][q] of tile index iv (16 keys per MFMA, V^T fragments = transposing reads of the V row tile, each feeding both q-blocks)
    bf16x8_t vnext[TH];                                               // the first V^T fragments of the NEXT body's PV product, read before the barrier in between
    auto pv_prefetch = [&](int iv) {
        const unsigned ya = lds_base + (unsigned)(NB + iv % NB) * TILE + t_lane;
#pragma unroll
        for (int n = 0; n < TH; ++n) vnext[n] = P2_LD(ya, n);
    };
    auto pv_product = [&](int iv, u32x4_t (&pP)[2][4], bool with_steps, int ik_dma, int iv_dma, f32x16_t (&sC)[2][2], u32x4_t (&pC)[2][4]) {
        const unsigned ya = lds_base + (unsigned)(NB + iv % NB) * TILE + t_lane;
        bf16x8_t a[TH + 1];
#pragma unroll
        for (int n = 0; n < TH; ++n) a[n] = vnext[n];
#pragma unroll
        for (int n = 0; n < 16; ++n) {                                // n = chunk (16 keys) * 4 + feature block
            if (n + TH < 16) a[(n + TH) % (TH + 1)] = P2_LD(ya, n + TH);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                acc[b][n & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[n % (TH + 1)], __builtin_bit_cast(bf16x8_t, pP[b][n >> 2]), acc[b][n & 3], 0, 0, 0);
                if (with_steps) {
                    if (2 * n + b < 8) dma_one(2 * n + b, ik_dma, iv_dma);
                    vstep(2 * n + b, sC, pC);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // body t: [wait + barrier] mask(t) | PV(t-1) + S(t+1) MFMAs with softmax(t) in their gaps | rescale when a running maximum moved
    auto body = [&](int it, f32x16_t (&sC)[2][2], f32x16_t (&sN)[2][2], u32x4_t (&pP)[2][4], u32x4_t (&pC)[2][4]) {
        F64_STAMPS;
        F64_STAMP(0);
        // this wave's share of K(it+1) and V(it) has landed: everything but the most recent body's requests (it < 2: K3 + body 0's / K3, K4, V2).  V(it) is only
        // multiplied in the NEXT body; waiting for it here lets this body's end read that product's first fragments ahead of the barrier in between
        if (it >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                 // ... and everybody's; everybody is done with K(it) and V(it-2)
        asm volatile("" ::: "memory");
        F64_STAMP(1);
        mask_tile(it, sC);
        F64_STAMP(2);
        pv_product(it > 0 ? it - 1 : 0, pP, true, it + 4, it + 2, sC, pC);
        F64_STAMP(3);
        s_product(it + 1, sN, 32, sC, pC);
        F64_STAMP(4);
        if (it > 0 && !__all((alpha[0] == 1.0f) & (alpha[1] == 1.0f))) {      // (it = 0: the accumulators are still zero)
            // cold path (lazy maximum: a running maximum moved by more than 2^6).  Written in assembly, one register at a time through ONE vector temporary:
            // as C++ (acc *= alpha) hipcc reads tiles out of the accumulator file wholesale, and the extra live VGPRs of this rarely taken block made it
            // spill the hot path's registers for the whole loop (137 spilled VGPRs, scratch traffic inside the tile loop)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                if (!__all(alpha[b] == 1.0f))                         // (per 32-row q-block: the other one's accumulators are left alone - multiplying by 1 is the identity)
#pragma unroll
                for (int db = 0; db < 4; ++db)
#pragma unroll
                    for (int r = 0; r < 16; r += 4) {                 // four registers per statement: the reads, multiplies and writes of a group overlap their latencies
                        float x0 = acc[b][db][r], x1 = acc[b][db][r + 1], x2 = acc[b][db][r + 2], x3 = acc[b][db][r + 3], t0, t1, t2, t3;
                        asm volatile("v_accvgpr_read_b32 %4, %0\n\tv_accvgpr_read_b32 %5, %1\n\tv_accvgpr_read_b32 %6, %2\n\tv_accvgpr_read_b32 %7, %3\n\t"
                                     "v_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %7, %7, %8\n\t"
                                     "v_accvgpr_write_b32 %0, %4\n\tv_accvgpr_write_b32 %1, %5\n\tv_accvgpr_write_b32 %2, %6\n\tv_accvgpr_write_b32 %3, %7"
                                     : "+a"(x0), "+a"(x1), "+a"(x2), "+a"(x3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(alpha[b]));
                        acc[b][db][r] = x0; acc[b][db][r + 1] = x1; acc[b][db][r + 2] = x2; acc[b][db][r + 3] = x3;
                    }
            asm volatile("s_nop 3" ::: "memory");
        }
        pv_prefetch(it);                                              // V(it) for body it+1 (its slot is not requested again before body it+2's barrier)
        F64_STAMP(5);
        F64_FLUSH(it);
    };

    f32x16_t s0[2][2], s1[2][2];
    u32x4_t pa[2][4], pb[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) { pa[b][c] = (u32x4_t){0, 0, 0, 0}; pb[b][c] = (u32x4_t){0, 0, 0, 0}; }
    if (n_my > 0) {
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");             // K0, V0
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        s_product(0, s0, -1, s0, pa);
        pv_prefetch(0);
        int it = 0;
        for (; it + 1 < n_my; it += 2) {
            body(it, s0, s1, pb, pa);                                 // softmax(it) -> pa;  PV(it-1) reads pb (zero at it = 0)
            body(it + 1, s1, s0, pa, pb);
        }
        if (it < n_my) {
            body(it, s0, s1, pb, pa);
            pv_product(it, pa, false, 0, 0, s0, pa);                  // V(n-1) was complete at the last body's barrier
        } else {
            pv_product(it - 1, pb, false, 0, 0, s0, pa);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (the clamped requests of the last bodies: nothing may be in flight into LDS when the block ends)
    }
    F64_DUMP();
#undef LDS_B128
#undef LDS_TR16
#undef P2_LD
#undef F64_DMA16
#undef F64_PIN
    // lane holds O^T[feature = db*32 + 8i + 4h + j][its query row of q-block b]
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int e_h = (tid2 >> 5) & 1;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const unsigned e_R = Rw0 + (unsigned)b * 32u + (unsigned)(tid2 & 31);
        if (e_R < nR) {
            int t2, hq2;
            att_split_row(p, e_R, t2, hq2);
            const float inv = l[b] > 0.f ? 1.f / l[b] : 0.f;
            bf16_t* row = p.O + (int64_t)t2 * p.o_ld + (int64_t)(kvh * p.group + hq2) * D;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32x2_t w = {pack2bf(acc[b][db][4 * i] * inv, acc[b][db][4 * i + 1] * inv), pack2bf(acc[b][db][4 * i + 2] * inv, acc[b][db][4 * i + 3] * inv)};
                    *reinterpret_cast<u32x2_t*>(row + db * 32 + 8 * i + 4 * e_h) = w;
                }
            if (e_h == 0 && p.lse) p.lse[(int64_t)(kvh * p.group + hq2) * p.T + t2] = l[b] > 0.f ? (m[b] + log2f(l[b])) * 0.6931471805599453f : NEG_INF;
        }
    }
}

// launched by attn_fwd_rows_impl (attn_fwd32.hip) for head-dim-128 row-major launches; same grid as attn_fwd32_kernel (256 packed rows per block)
int tr1_launch_attn_fwd64(AttnParams& p, unsigned blocks, hipStream_t s) {
    const size_t dyn = 8 * (64 * 256) + 128 + F64_PROBE_LDS;
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd64_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); attr = true; }
    hipLaunchKernelGGL(attn_fwd64_kernel<8>, dim3(blocks), dim3(256), dyn, s, p);
    return 0;
}
