// Attention backward: delta, dQ kernel (per query block, loops key tiles) and dK/dV kernel (per key block, loops query tiles).
// Recompute-based (FlashAttention-2 style), no atomics. See attn_common.h for the mask model and MFMA orientation.
//
//   dV = P^T dO,  dP = dO V^T,  dS = P o (dP - delta),  dQ = scale * dS K,  dK = scale * dS^T Q,   delta = rowsum(dO o O)
//
// Gradient flow through the shared prompt prefix falls out of the mask: every completion row of every group sees the prefix
// keys, so the dK/dV kernel accumulates all G suffixes' contributions into the single prefix K/V (hard part 1).
//
// Both kernels double-buffer their operand tiles in LDS and stage them through registers one tile ahead (branch-free loads,
// masks applied at the LDS write), one barrier per tile.  The dK/dV kernel additionally splits the query-tile range over
// gridDim.z blocks (the prefix key blocks see ~all query tiles, the suffix blocks only a few): partial dK/dV go to an fp32
// workspace and a small reduce kernel sums and rounds them.
#include "attn_common.h"
#include <stdlib.h>

// per 64 packed rows: (max pre, min lo, max hi) over rows with a non-empty [lo,hi].  One wave per tile; runs inside attn_delta_kernel (its
// first n_qtiles blocks): under a weight-gradient GEMM on the side stream every extra small launch of the main stream waits ~150 us for
// wave slots, so the 6 us of work used to cost 170 us per layer as a launch of its own.
TR1_DEV void attn_qmeta_tile(const int* __restrict__ pre, const int* __restrict__ lo, const int* __restrict__ hi, int* __restrict__ qmeta,
                             int T, int group, int tile, int lane) {
    const int64_t R = (int64_t)tile * 64 + lane;
    int mp = 0, ml = 0x7fffffff, mh = -1, mnp = 0x7fffffff;
    if (R < (int64_t)T * group) {
        const int t = (int)((unsigned)R / (unsigned)group);
        mp = pre[t]; mnp = pre[t];
        if (hi[t] >= lo[t]) { ml = lo[t]; mh = hi[t]; }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        mp = max(mp, __shfl_xor(mp, o, 64)); ml = min(ml, __shfl_xor(ml, o, 64)); mh = max(mh, __shfl_xor(mh, o, 64)); mnp = min(mnp, __shfl_xor(mnp, o, 64));
    }
    // [3]: the smallest prefix length among the tile's rows - keys below it are visible to EVERY row of the tile (mask-free fast path)
    if (lane == 0) { qmeta[tile * 4 + 0] = mp; qmeta[tile * 4 + 1] = ml; qmeta[tile * 4 + 2] = mh; qmeta[tile * 4 + 3] = mnp; }
}

// delta[h][t] = sum_d dO[t,h,d] * O[t,h,d].  16 lanes per (t, h) row, 16 bytes per lane: consecutive rows are consecutive in memory, so a
// wave instruction reads 4 rows = 1 KiB contiguous (was one thread per row: 64 different lines per load, 72 us for 73 MB at config 3).
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ dO, int64_t do_ld, const bf16_t* __restrict__ O, int64_t o_ld,
                                                         float* __restrict__ delta, int T, int n_heads, int d, const int* __restrict__ pre,
                                                         const int* __restrict__ lo, const int* __restrict__ hi, int* __restrict__ qmeta, int group,
                                                         int n_qtiles, const float* __restrict__ lse, float* __restrict__ lse2) {
    if ((int)blockIdx.x < n_qtiles && threadIdx.x < 64) attn_qmeta_tile(pre, lo, hi, qmeta, T, group, (int)blockIdx.x, (int)threadIdx.x);
    const int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    const bool ok = i < (int64_t)T * n_heads;
    const int t = ok ? (int)(i / n_heads) : 0, h = ok ? (int)(i - (int64_t)t * n_heads) : 0;
    const bf16_t* a = dO + (int64_t)t * do_ld + (int64_t)h * d;
    const bf16_t* b = O + (int64_t)t * o_ld + (int64_t)h * d;
    float s = 0.f;
    if (ok)
        for (int c = sub * 8; c < d; c += 128) {
            const u32x4_t x = *reinterpret_cast<const u32x4_t*>(a + c), y = *reinterpret_cast<const u32x4_t*>(b + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) s += bflo(x[j]) * bflo(y[j]) + bfhi(x[j]) * bfhi(y[j]);
        }
    s += __shfl_xor(s, 8, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 1, 64);
    if (ok && sub == 0) {
        delta[(int64_t)h * T + t] = s;
        // log2-scaled LSE for the DMA-staged dK/dV kernel (exp2 argument = s * scale_log2 - lse2); rows that saw no key carry +inf -> p = 0
        const float l0 = lse[(int64_t)h * T + t];
        lse2[(int64_t)h * T + t] = (l0 == NEG_INF) ? INFINITY : l0 * 1.4426950408889634f;
    }
}

// Transposed MFMA operands straight from a ROW-major LDS tile: ds_read_b64_tr_b16.  Every lane supplies its own 8-byte address; inside a 16-lane
// group lane i supplies row i/4, columns 4(i%4)..+3 of a 4 x 16 block and receives column i of that block (probed on MI355X).
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
TR1_DEV u32x2_t lds_read_tr16(const char* p) {
    const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
    return __builtin_bit_cast(u32x2_t, v);
}

// ---------------------------------------------------------------------------------------------------------------- dQ
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnParams p) {
    constexpr int KSTR = 2 * D + 16;
    constexpr int RB = ATT_KV * KSTR, BUF = 2 * RB;
    constexpr int PF = 2;                                             // K/V tiles in flight (register ring)
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];   // [2][K rows | V rows] + meta; K^T fragments come from the K rows (tr16 reads)
    int* lds_meta = reinterpret_cast<int*>(dyn_lds + 2 * BUF);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, u = lane & 15, g = lane >> 4;
    const int kvh = blockIdx.y;
    const int64_t nR = (int64_t)p.T * p.group;
    const int64_t R0 = (int64_t)(gridDim.x - 1 - blockIdx.x) * 128 + wave * 32;      // heaviest query tiles first (see attn_fwd_kernel)

    int tq[2], hq[2], pre[2], lo[2], hi[2]; bool valid[2];
    float lse2[2], dlt[2];
    int wmaxpre = 0, wminlo = 0x7fffffff, wmaxhi = -1;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int64_t R = R0 + cb * 16 + u;
        valid[cb] = R < nR;
        const int64_t Rc = valid[cb] ? R : nR - 1;
        att_split_row(p, Rc, tq[cb], hq[cb]);
        pre[cb] = valid[cb] ? p.pre[tq[cb]] : 0;
        lo[cb] = valid[cb] ? p.lo[tq[cb]] : 1;
        hi[cb] = valid[cb] ? p.hi[tq[cb]] : 0;
        if (valid[cb]) { wmaxpre = max(wmaxpre, pre[cb]); if (hi[cb] >= lo[cb]) { wminlo = min(wminlo, lo[cb]); wmaxhi = max(wmaxhi, hi[cb]); } }
        const int64_t si = (int64_t)(kvh * p.group + hq[cb]) * p.T + tq[cb];
        const float ls = valid[cb] ? p.lse[si] : NEG_INF;
        lse2[cb] = (ls == NEG_INF) ? INFINITY : ls * 1.4426950408889634f;
        dlt[cb] = valid[cb] ? p.delta[si] : 0.f;
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        wmaxpre = max(wmaxpre, __shfl_xor(wmaxpre, o, 64)); wminlo = min(wminlo, __shfl_xor(wminlo, o, 64)); wmaxhi = max(wmaxhi, __shfl_xor(wmaxhi, o, 64));
    }
    if (lane == 0) { lds_meta[wave * 3 + 0] = wmaxpre; lds_meta[wave * 3 + 1] = wminlo; lds_meta[wave * 3 + 2] = wmaxhi; }
    __syncthreads();
    int bmaxpre = 0, bminlo = 0x7fffffff, bmaxhi = -1;
#pragma unroll
    for (int w = 0; w < 4; ++w) { bmaxpre = max(bmaxpre, lds_meta[w * 3]); bminlo = min(bminlo, lds_meta[w * 3 + 1]); bmaxhi = max(bmaxhi, lds_meta[w * 3 + 2]); }
    const TileRange tr = att_tile_range(bmaxpre, bminlo, bmaxhi, p.n_slots);
    const bool wave_active = __builtin_amdgcn_readfirstlane((int)(R0 < nR)) != 0;

    bf16x8_t qf[2][D / 32], dof[2][D / 32];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int64_t hoff = (int64_t)(kvh * p.group + hq[cb]) * p.d_real;
        const bf16_t* qrow = p.Q + (int64_t)tq[cb] * p.q_ld + hoff;
        const bf16_t* drow = p.dO + (int64_t)tq[cb] * p.do_ld + hoff;
#pragma unroll
        for (int ks = 0; ks < D / 32; ++ks) {
            qf[cb][ks] = load_row_frag(qrow, ks * 32 + g * 8, p.d_real, valid[cb]);
            dof[cb][ks] = load_row_frag(drow, ks * 32 + g * 8, p.d_real, valid[cb]);
        }
    }
    f32x4_t dq[D / 16][2];
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt) { dq[dt][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dq[dt][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }

    const int n_my = tr.n_rel;
    const int64_t kcol = (int64_t)kvh * p.d_real;
    struct KVRegs { TReg<D> rk, rv; };
    KVRegs rg[PF];
#define DQ_KV0(i) ((int64_t)att_tile_at(tr, (i)) * ATT_KV)
#define DQ_LOAD(r, i) do { rows_load<D>((r).rk, p.K, p.k_ld, kcol, DQ_KV0(i), p.n_slots, p.d_real); rows_load<D>((r).rv, p.V, p.v_ld, kcol, DQ_KV0(i), p.n_slots, p.d_real); } while (0)
#define DQ_STORE(r, i, buf) do { rows_store<D>((r).rk, (buf), DQ_KV0(i), p.n_slots, p.d_real); rows_store<D>((r).rv, (buf) + RB, DQ_KV0(i), p.n_slots, p.d_real); } while (0)
#pragma unroll
    for (int j = 0; j < PF; ++j)
        if (j < n_my) DQ_LOAD(rg[j], j);
    if (n_my > 0) {
        DQ_STORE(rg[0], 0, dyn_lds);
        if (PF < n_my) DQ_LOAD(rg[0], PF);
    }
    __syncthreads();

    for (int it0 = 0; it0 < n_my; it0 += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        const int it = it0 + j;
        if (it >= n_my) break;
        const int kv0 = att_tile_at(tr, it) * ATT_KV;
        const char* lds_k = dyn_lds + (it & 1) * BUF;
        const char* lds_v = lds_k + RB;
        if (wave_active) {
            f32x4_t s[4][2], dp[4][2];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) { s[kt][cb] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dp[kt][cb] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int ks = 0; ks < D / 32; ++ks) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(lds_k + (kt * 16 + u) * KSTR + (ks * 4 + g) * 16);
                    const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(lds_v + (kt * 16 + u) * KSTR + (ks * 4 + g) * 16);
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        s[kt][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[cb][ks], s[kt][cb], 0, 0, 0);
                        dp[kt][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[cb][ks], dp[kt][cb], 0, 0, 0);
                    }
                }
            }
            bool full = kv0 + ATT_KV <= p.n_slots;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
                full = full && (!valid[cb] || (kv0 + ATT_KV - 1 < pre[cb]) || (kv0 >= lo[cb] && kv0 + ATT_KV - 1 <= hi[cb]));
            const bool wave_full = __all(full);
            bf16x8_t dsf[2][2];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                if (wave_full) {
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][cb][r], p.scale_log2, -lse2[cb]));
                            s[kt][cb][r] = pr * (dp[kt][cb][r] - dlt[cb]);
                        }
                } else {
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int kv = kv0 + kt * 16 + g * 4 + r;
                            const bool ok = kv < p.n_slots && att_visible(kv, pre[cb], lo[cb], hi[cb]);
                            const float pr = ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][cb][r], p.scale_log2, -lse2[cb])) : 0.f;
                            s[kt][cb][r] = pr * (dp[kt][cb][r] - dlt[cb]);
                        }
                }
                dsf[0][cb] = pack_frag(s[0][cb], s[1][cb]);
                dsf[1][cb] = pack_frag(s[2][cb], s[3][cb]);
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int dt = 0; dt < D / 16; ++dt) {
                    // K^T[d = dt*16 + u][kv = kk*32 + g*4 .. +3 | kk*32 + 16 + g*4 .. +3] from the K rows
                    const char* base = lds_k + (kk * 32 + g * 4 + (u >> 2)) * KSTR + dt * 32 + (u & 3) * 8;
                    const bf16x8_t ktf = make_frag(lds_read_tr16(base), lds_read_tr16(base + 16 * KSTR));
                    dq[dt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf[kk][0], dq[dt][0], 0, 0, 0);
                    dq[dt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf[kk][1], dq[dt][1], 0, 0, 0);
                }
            }
        }
        if (it + 1 < n_my) {
            char* nb = dyn_lds + ((it + 1) & 1) * BUF;
            DQ_STORE(rg[(j + 1) % PF], it + 1, nb);
            if (it + 1 + PF < n_my) DQ_LOAD(rg[(j + 1) % PF], it + 1 + PF);
        }
        __syncthreads();
    }
    }
#undef DQ_KV0
#undef DQ_LOAD
#undef DQ_STORE
    const float scale = p.scale_log2 * 0.6931471805599453f;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        if (!valid[cb]) continue;
        bf16_t* row = p.dQ + (int64_t)tq[cb] * p.dq_ld + (int64_t)(kvh * p.group + hq[cb]) * p.d_real;
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) {
            const int d = dt * 16 + g * 4;
            if (d < p.d_real) {
                u32x2_t w = {pack2bf(dq[dt][cb][0] * scale, dq[dt][cb][1] * scale), pack2bf(dq[dt][cb][2] * scale, dq[dt][cb][3] * scale)};
                *reinterpret_cast<u32x2_t*>(row + d) = w;
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------------------- dK/dV
// Block = 64 keys of one kv head x one slice of the query tiles (gridDim.z); wave owns 16 keys (K/V fragments stay in registers).
#define DKDV_MAXT 1024
struct RowMeta { float lse, dlt; int pre, lo, hi; };

// NW waves per block, each owning 16 keys (block = 16*NW keys) and sharing the staged 64-query tile.  NW = 8 (D = 64 / 128): the tile is
// filled by 512 threads, so one staging set is 32 VGPRs and TWO sets fit (two query tiles in flight), the global->LDS traffic per key
// halves, and the block - alone on its CU with 146 KB of LDS - keeps 8 waves of MFMA work per staged tile instead of 4.
// TR (the 8-wave form): Q^T / dO^T fragments are read from the ROW-major tiles with ds_read_b64_tr_b16 (each 16-lane group reads a 4 x 16 block:
// lane i supplies row i/4, columns 4(i%4)..+3, and receives column i of the block) - no transposed copies in LDS or in global memory, half
// the staging traffic and registers, which pays for THREE query tiles in flight.

// KT = 16-key tiles per wave.  <8, 1>: 8 waves x 16 keys; <4, 2>: 4 waves x 32 keys - every Q / dO / Q^T / dO^T fragment read from LDS then
// feeds two MFMAs, which halves the LDS instructions per MFMA (the 16-key form reads one fragment per MFMA and is LDS-issue bound).
template <int D, int NW, int KT = 1>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dkdv_kernel(AttnParams p, int n_qtiles, float* __restrict__ part_k, float* __restrict__ part_v) {
    constexpr int NT = NW * 64, KB = NW * 16 * KT;
    constexpr bool TR = (NW == 8) || (KT == 2);
    constexpr int KSTR = 2 * D + 16;
    constexpr int RB = 64 * KSTR, TB = TR ? 0 : D * 144, BUF = 2 * RB + 2 * TB + 64 * 5 * 4 + 16;
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];   // [2][Q rows | dO rows | Q^T | dO^T | row meta] + tile list
    int* lds_tiles = reinterpret_cast<int*>(dyn_lds + 2 * BUF);      // [DKDV_MAXT + 1]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, u = lane & 15, g = lane >> 4;
    // grid = (n_kv * QS, 1, key blocks): the dispatcher walks x fastest and z slowest, so blocks leave in order of their key block.  Low key
    // blocks are the heavy ones (every later query tile sees them; the completion keys at the end see a single group), so the long blocks
    // start first and the short ones fill the tail (with key blocks on x the last query slice's heavy blocks started last: 34 % idle CUs)
    const int QS = gridDim.x / p.n_kv;
    const int kvh = blockIdx.x % p.n_kv, qz = blockIdx.x / p.n_kv;
    const int kvb0 = blockIdx.z * KB;
    int kv[KT]; bool kv_ok[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) { kv[kt] = kvb0 + (wave * KT + kt) * 16 + u; kv_ok[kt] = kv[kt] < p.n_slots; }
    const int64_t nR = (int64_t)p.T * p.group;

    // ---- this block's list of relevant query tiles (qi = qz, qz+QS, ...), compacted by wave 0
    if (wave == 0) {
        int count = 0;
        const int n_cand = (n_qtiles - qz + QS - 1) / QS;
        for (int base = 0; base < n_cand; base += 64) {
            const int c = base + lane, qi = qz + c * QS;
            bool rel = false;
            if (c < n_cand) {
                const int mp = p.qmeta[qi * 4], ml = p.qmeta[qi * 4 + 1], mh = p.qmeta[qi * 4 + 2];
                rel = (kvb0 < mp) || (kvb0 + KB - 1 >= ml && kvb0 <= mh);
            }
            const unsigned long long mask = __ballot(rel);
            if (rel) lds_tiles[count + __popcll(mask & ((1ull << lane) - 1ull))] = qi;
            count += __popcll(mask);
        }
        if (lane == 0) lds_tiles[DKDV_MAXT] = count;
    }
    bf16x8_t kf[KT][D / 32], vf[KT][D / 32];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const bf16_t* krow = p.K + (int64_t)(kv_ok[kt] ? kv[kt] : 0) * p.k_ld + (int64_t)kvh * p.d_real;
        const bf16_t* vrow = p.V + (int64_t)(kv_ok[kt] ? kv[kt] : 0) * p.v_ld + (int64_t)kvh * p.d_real;
#pragma unroll
        for (int ks = 0; ks < D / 32; ++ks) {
            kf[kt][ks] = load_row_frag(krow, ks * 32 + g * 8, p.d_real, kv_ok[kt]);
            vf[kt][ks] = load_row_frag(vrow, ks * 32 + g * 8, p.d_real, kv_ok[kt]);
        }
    }
    f32x4_t dk[KT][D / 16], dv[KT][D / 16];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) { dk[kt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dv[kt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
    __syncthreads();
    const int n_my = lds_tiles[DKDV_MAXT];

    // PF register sets: the global loads of PF query tiles are in flight while one is computed.  The block is alone on its CU (146 KB
    // of LDS), so with one set each iteration is an exposed L2/HBM round trip (measured: 2.7 ms per call, 7B config 3, NW = 4).
    constexpr int PF = TR ? (KT == 2 ? 1 : 3) : 1;      // tiles in flight (8 waves: 2 or 4 make hipcc spill heavily at the 256-VGPR cap)
    struct QTileRegs { TReg<D, NT> rq, rdo, rqt, rdot; RowMeta rm; };
    QTileRegs rg[PF];
    auto load_tile = [&](QTileRegs& r, int qi) {
        const int64_t Rq0 = (int64_t)qi * 64;
        prows_load<D, NT>(r.rq, p.Q, p.q_ld, kvh, p.group, Rq0, nR, p.d_real, p.group_magic);
        prows_load<D, NT>(r.rdo, p.dO, p.do_ld, kvh, p.group, Rq0, nR, p.d_real, p.group_magic);
        if (!TR) {
            T_load<D, NT>(r.rqt, p.QT, p.qt_ld, kvh, Rq0, nR, p.d_real);
            T_load<D, NT>(r.rdot, p.dOT, p.dot_ld, kvh, Rq0, nR, p.d_real);
        }
        if (threadIdx.x < 64) {
            const int64_t R = Rq0 + threadIdx.x;
            r.rm = RowMeta{INFINITY, 0.f, 0, 1, 0};
            if (R < nR) {
                int t, hq;
                att_split_row(p, R, t, hq);
                const int64_t si = (int64_t)(kvh * p.group + hq) * p.T + t;
                const float l0 = p.lse[si];
                r.rm.lse = (l0 == NEG_INF) ? INFINITY : l0 * 1.4426950408889634f;
                r.rm.dlt = p.delta[si]; r.rm.pre = p.pre[t]; r.rm.lo = p.lo[t]; r.rm.hi = p.hi[t];
            }
        }
    };
    auto store_tile = [&](const QTileRegs& r, int qi, char* buf) {
        const int64_t Rq0 = (int64_t)qi * 64;
        rows_store<D, NT>(r.rq, buf, Rq0, nR, p.d_real);
        rows_store<D, NT>(r.rdo, buf + RB, Rq0, nR, p.d_real);
        if (!TR) {
            T_store<D, NT>(r.rqt, buf + 2 * RB, Rq0, nR, p.d_real);
            T_store<D, NT>(r.rdot, buf + 2 * RB + TB, Rq0, nR, p.d_real);
        }
        if (threadIdx.x < 64) {
            float* mf = reinterpret_cast<float*>(buf + 2 * RB + 2 * TB);
            int* mi = reinterpret_cast<int*>(mf + 128);
            mf[threadIdx.x] = r.rm.lse; mf[64 + threadIdx.x] = r.rm.dlt;
            mi[threadIdx.x] = r.rm.pre; mi[64 + threadIdx.x] = r.rm.lo; mi[128 + threadIdx.x] = r.rm.hi;
            int mp = r.rm.lse == INFINITY ? 0x7fffffff : r.rm.pre;      // rows without keys (padding) carry p = 0 through lse = +inf
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) mp = min(mp, __shfl_xor(mp, o, 64));
            if (threadIdx.x == 0) mi[192] = mp;                          // min prefix length of the tile: keys below it are visible to every row
        }
    };
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        rg[j].rm = RowMeta{INFINITY, 0.f, 0, 1, 0};
        if (j < n_my) load_tile(rg[j], lds_tiles[j]);
    }
    if (n_my > 0) {
        store_tile(rg[0], lds_tiles[0], dyn_lds);
        if (PF < n_my) load_tile(rg[0], lds_tiles[PF]);
    }
    __syncthreads();

    for (int it0 = 0; it0 < n_my; it0 += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        const int it = it0 + j;
        if (it >= n_my) break;
        const char* buf = dyn_lds + (it & 1) * BUF;
        const char* lds_q = buf;
        const char* lds_do = buf + RB;
        const char* lds_qt = buf + 2 * RB;
        const char* lds_dot = lds_qt + TB;
        const float* lds_lse = reinterpret_cast<const float*>(buf + 2 * RB + 2 * TB);
        const float* lds_dlt = lds_lse + 64;
        const int* lds_pre = reinterpret_cast<const int*>(lds_lse + 128);
        const int* lds_lo = lds_pre + 64;
        const int* lds_hi = lds_pre + 128;

        f32x4_t s[KT][4], dp[KT][4];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) { s[kt][qt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dp[kt][qt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < D / 32; ++ks) {
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                const bf16x8_t qa = *reinterpret_cast<const bf16x8_t*>(lds_q + (qt * 16 + u) * KSTR + (ks * 4 + g) * 16);
                const bf16x8_t da = *reinterpret_cast<const bf16x8_t*>(lds_do + (qt * 16 + u) * KSTR + (ks * 4 + g) * 16);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    s[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[kt][ks], s[kt][qt], 0, 0, 0);
                    dp[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[kt][ks], dp[kt][qt], 0, 0, 0);
                }
            }
        }
        // lane holds S[q = qt*16 + g*4 + r][kv = u-th key of this wave].  Row statistics come as 4-wide LDS reads issued unconditionally (no
        // short-circuit guards around them); a tile whose rows all see every key of this block (keys below the tile's smallest prefix
        // length: the common completion-rows x prompt-keys case) skips the interval tests.
        typedef __attribute__((ext_vector_type(4))) int i32x4_t;
        bf16x8_t pf0[KT], pf1[KT], df0[KT], df1[KT];
        const bool full = (kvb0 + KB <= lds_pre[192]) && (kvb0 + KB <= p.n_slots);      // block-uniform
        if (full) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {                    // query halves: fragment 0 = query sub-tiles 0, 1; fragment 1 = 2, 3
                f32x4_t l4[2], d4[2];
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    l4[q2] = *reinterpret_cast<const f32x4_t*>(lds_lse + (2 * h + q2) * 16 + g * 4);
                    d4[q2] = *reinterpret_cast<const f32x4_t*>(lds_dlt + (2 * h + q2) * 16 + g * 4);
                }
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    f32x4_t pr[2], ds[2];
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][2 * h + q2][r], p.scale_log2, -l4[q2][r]));
                            pr[q2][r] = pv; ds[q2][r] = pv * (dp[kt][2 * h + q2][r] - d4[q2][r]);
                        }
                    if (h == 0) { pf0[kt] = pack_frag(pr[0], pr[1]); df0[kt] = pack_frag(ds[0], ds[1]); }
                    else { pf1[kt] = pack_frag(pr[0], pr[1]); df1[kt] = pack_frag(ds[0], ds[1]); }
                }
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4_t l4[2], d4[2];
                i32x4_t p4[2], lo4[2], hi4[2];
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const int o = (2 * h + q2) * 16 + g * 4;
                    l4[q2] = *reinterpret_cast<const f32x4_t*>(lds_lse + o); d4[q2] = *reinterpret_cast<const f32x4_t*>(lds_dlt + o);
                    p4[q2] = *reinterpret_cast<const i32x4_t*>(lds_pre + o); lo4[q2] = *reinterpret_cast<const i32x4_t*>(lds_lo + o);
                    hi4[q2] = *reinterpret_cast<const i32x4_t*>(lds_hi + o);
                }
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    f32x4_t pr[2], ds[2];
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool ok = kv_ok[kt] & att_visible_nb(kv[kt], p4[q2][r], lo4[q2][r], hi4[q2][r]);
                            const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][2 * h + q2][r], p.scale_log2, -l4[q2][r]));
                            const float pv = ok ? e : 0.f;
                            pr[q2][r] = pv; ds[q2][r] = pv * (dp[kt][2 * h + q2][r] - d4[q2][r]);
                        }
                    if (h == 0) { pf0[kt] = pack_frag(pr[0], pr[1]); df0[kt] = pack_frag(ds[0], ds[1]); }
                    else { pf1[kt] = pack_frag(pr[0], pr[1]); df1[kt] = pack_frag(ds[0], ds[1]); }
                }
            }
        }
        if (TR) {
            // lane (u, g) of a transposed fragment: rows qb + g*4 + (u >> 2), 8 bytes at feature dt*16 + (u & 3)*4; it receives Q[qb + g*4 + r][dt*16 + u]
            const char* tq = lds_q + (g * 4 + (u >> 2)) * KSTR + (u & 3) * 8;
            const char* to = lds_do + (g * 4 + (u >> 2)) * KSTR + (u & 3) * 8;
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) {
                const bf16x8_t q0 = make_frag(lds_read_tr16(tq + dt * 32), lds_read_tr16(tq + 16 * KSTR + dt * 32));
                const bf16x8_t o0 = make_frag(lds_read_tr16(to + dt * 32), lds_read_tr16(to + 16 * KSTR + dt * 32));
                _Pragma("unroll") for (int kt = 0; kt < KT; ++kt) dv[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(o0, pf0[kt], dv[kt][dt], 0, 0, 0);
                _Pragma("unroll") for (int kt = 0; kt < KT; ++kt) dk[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0, df0[kt], dk[kt][dt], 0, 0, 0);
            }
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) {
                const bf16x8_t q1 = make_frag(lds_read_tr16(tq + 32 * KSTR + dt * 32), lds_read_tr16(tq + 48 * KSTR + dt * 32));
                const bf16x8_t o1 = make_frag(lds_read_tr16(to + 32 * KSTR + dt * 32), lds_read_tr16(to + 48 * KSTR + dt * 32));
                _Pragma("unroll") for (int kt = 0; kt < KT; ++kt) dv[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(o1, pf1[kt], dv[kt][dt], 0, 0, 0);
                _Pragma("unroll") for (int kt = 0; kt < KT; ++kt) dk[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1, df1[kt], dk[kt][dt], 0, 0, 0);
            }
        } else {
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) {
            const char* bq = lds_qt + (dt * 16 + u) * 144 + g * 8;
            const char* bo = lds_dot + (dt * 16 + u) * 144 + g * 8;
            const bf16x8_t q0 = make_frag(*reinterpret_cast<const u32x2_t*>(bq), *reinterpret_cast<const u32x2_t*>(bq + 32));
            const bf16x8_t o0 = make_frag(*reinterpret_cast<const u32x2_t*>(bo), *reinterpret_cast<const u32x2_t*>(bo + 32));
            _Pragma("unroll") for (int kt = 0; kt < KT; ++kt) dv[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(o0, pf0[kt], dv[kt][dt], 0, 0, 0);
            _Pragma("unroll") for (int kt = 0; kt < KT; ++kt) dk[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0, df0[kt], dk[kt][dt], 0, 0, 0);
        }
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) {
            const char* bq = lds_qt + (dt * 16 + u) * 144 + g * 8;
            const char* bo = lds_dot + (dt * 16 + u) * 144 + g * 8;
            const bf16x8_t q1 = make_frag(*reinterpret_cast<const u32x2_t*>(bq + 64), *reinterpret_cast<const u32x2_t*>(bq + 96));
            const bf16x8_t o1 = make_frag(*reinterpret_cast<const u32x2_t*>(bo + 64), *reinterpret_cast<const u32x2_t*>(bo + 96));
            _Pragma("unroll") for (int kt = 0; kt < KT; ++kt) dv[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(o1, pf1[kt], dv[kt][dt], 0, 0, 0);
            _Pragma("unroll") for (int kt = 0; kt < KT; ++kt) dk[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1, df1[kt], dk[kt][dt], 0, 0, 0);
        }
        }
        if (it + 1 < n_my) {
            store_tile(rg[(j + 1) % PF], lds_tiles[it + 1], dyn_lds + ((it + 1) & 1) * BUF);
            if (it + 1 + PF < n_my) load_tile(rg[(j + 1) % PF], lds_tiles[it + 1 + PF]);
        }
        __syncthreads();
    }
    }
    // lane holds dK^T/dV^T[d = dt*16 + g*4 + r][kv]
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
    if (kv_ok[kt]) {
        if (part_k) {   // split over query tiles: fp32 partials, summed / scaled / rounded by attn_bwd_reduce_kernel
            const int64_t kvd = (int64_t)p.n_kv * p.d_real;
            float* pk = part_k + ((int64_t)qz * p.n_slots + kv[kt]) * kvd + (int64_t)kvh * p.d_real;
            float* pv = part_v + ((int64_t)qz * p.n_slots + kv[kt]) * kvd + (int64_t)kvh * p.d_real;
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) {
                const int d = dt * 16 + g * 4;
                if (d < p.d_real) { *reinterpret_cast<f32x4_t*>(pk + d) = dk[kt][dt]; *reinterpret_cast<f32x4_t*>(pv + d) = dv[kt][dt]; }
            }
        } else {
            const float scale = p.scale_log2 * 0.6931471805599453f;
            bf16_t* kr = p.dK + (int64_t)kv[kt] * p.dk_ld + (int64_t)kvh * p.d_real;
            bf16_t* vr = p.dV + (int64_t)kv[kt] * p.dv_ld + (int64_t)kvh * p.d_real;
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) {
                const int d = dt * 16 + g * 4;
                if (d < p.d_real) {
                    u32x2_t wk = {pack2bf(dk[kt][dt][0] * scale, dk[kt][dt][1] * scale), pack2bf(dk[kt][dt][2] * scale, dk[kt][dt][3] * scale)};
                    u32x2_t wv = {pack2bf(dv[kt][dt][0], dv[kt][dt][1]), pack2bf(dv[kt][dt][2], dv[kt][dt][3])};
                    *reinterpret_cast<u32x2_t*>(kr + d) = wk;
                    *reinterpret_cast<u32x2_t*>(vr + d) = wv;
                }
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------- dK/dV, LDS-DMA staged (head dim 128)
// Same decomposition (block = KB keys of one kv head x one slice of the query tiles, K/V fragments stationary in registers, transposed
// operands read from the row-major tiles with ds_read_b64_tr_b16), but the 64-row Q / dO tiles and their row statistics travel
// global -> LDS with global_load_lds (no staging registers, no ds_write): a ring of NB tile buffers keeps NB-1 query tiles in flight per
// block whatever the register budget, which is what lets a wave own 32 keys (KT = 2: every Q / dO / Q^T / dO^T fragment read from LDS
// feeds two MFMAs - half the LDS traffic per FLOP of the 16-key form, whose LDS read time equalled its MFMA time).
// Tile image: 64 rows x 256 bytes, no padding; row r keeps its logical 16-byte chunk c at c ^ dkey(r & 15).  dkey spreads the 16 rows of a
// b128 fragment read over all 16 chunk positions and the 8 rows x 2 chunks of a 32-lane transposing read over 16 distinct positions:
// both read shapes are bank-conflict free.  One DMA instruction = 4 rows (1 KiB, lane -> row lane / 16, physical chunk lane % 16).
// Row statistics (log2-scaled LSE from attn_delta_kernel, delta, pre / lo / hi) follow as five 256-byte dword DMAs issued by wave 0.
// Ordering: each wave counts its own DMA instructions (s_waitcnt vmcnt(n)), then ONE barrier per tile publishes the tile to the block and
// retires the buffer consumed in the previous iteration, which is refilled right behind the barrier.
TR1_DEV int dkey(int row) { return ((row & 7) << 1) | ((row >> 3) & 1); }
typedef const __attribute__((address_space(1))) void* att_gptr_t;
typedef __attribute__((address_space(3))) void* att_lptr_t;

template <int NW, int KT, int NB>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dkdv_dma_kernel(AttnParams p, int n_qtiles, const float* __restrict__ lse2, float* __restrict__ part_k,
                                                                    float* __restrict__ part_v) {
    constexpr int D = 128, KB = NW * 16 * KT;
    constexpr int TILE = 64 * 256, META = 64 * 5 * 4, BUF = 2 * TILE + META;
    constexpr int IPW = 16 / NW;                                      // 4-row groups per wave: IPW Q + IPW dO instructions per tile
    constexpr int PER = 2 * IPW, PER0 = PER + 5;                      // DMA instructions per tile: waves 1.., wave 0 (+ row statistics)
    static_assert(NB >= 3 && NB <= 5 && (NB - 2) * PER0 <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];   // [NB][Q rows | dO rows | lse2, delta, pre, lo, hi] + tile list
    int* lds_tiles = reinterpret_cast<int*>(dyn_lds + NB * BUF);      // [DKDV_MAXT + 1] tile ids, then [DKDV_MAXT] their smallest prefix length
    int* lds_minpre = lds_tiles + DKDV_MAXT + 1;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, u = lane & 15, g = lane >> 4;
    // grid = (n_kv * QS, 1, key blocks): the dispatcher walks x fastest and z slowest, so blocks leave in order of their key block.  Low key
    // blocks are the heavy ones (every later query tile sees them; the completion keys at the end see a single group), so the long blocks
    // start first and the short ones fill the tail (with key blocks on x the last query slice's heavy blocks started last: 34 % idle CUs)
    const int QS = gridDim.x / p.n_kv;
    const int kvh = blockIdx.x % p.n_kv, qz = blockIdx.x / p.n_kv;
    const int kvb0 = blockIdx.z * KB;
    int kv[KT]; bool kv_ok[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) { kv[kt] = kvb0 + (wave * KT + kt) * 16 + u; kv_ok[kt] = kv[kt] < p.n_slots; }
    const int64_t nR = (int64_t)p.T * p.group;

    if (wave == 0) {      // this block's list of relevant query tiles (qi = qz, qz+QS, ...)
        int count = 0;
        const int n_cand = (n_qtiles - qz + QS - 1) / QS;
        for (int base = 0; base < n_cand; base += 64) {
            const int c = base + lane, qi = qz + c * QS;
            bool rel = false;
            int mnp = 0;
            if (c < n_cand) {
                const int mp = p.qmeta[qi * 4], ml = p.qmeta[qi * 4 + 1], mh = p.qmeta[qi * 4 + 2];
                mnp = p.qmeta[qi * 4 + 3];
                rel = (kvb0 < mp) || (kvb0 + KB - 1 >= ml && kvb0 <= mh);
            }
            const unsigned long long mask = __ballot(rel);
            if (rel) { const int at = count + __popcll(mask & ((1ull << lane) - 1ull)); lds_tiles[at] = qi; lds_minpre[at] = mnp; }
            count += __popcll(mask);
        }
        if (lane == 0) lds_tiles[DKDV_MAXT] = count;
    }
    bf16x8_t kf[KT][D / 32], vf[KT][D / 32];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const bf16_t* krow = p.K + (int64_t)(kv_ok[kt] ? kv[kt] : 0) * p.k_ld + (int64_t)kvh * D;
        const bf16_t* vrow = p.V + (int64_t)(kv_ok[kt] ? kv[kt] : 0) * p.v_ld + (int64_t)kvh * D;
#pragma unroll
        for (int ks = 0; ks < D / 32; ++ks) {
            kf[kt][ks] = load_row_frag(krow, ks * 32 + g * 8, D, kv_ok[kt]);
            vf[kt][ks] = load_row_frag(vrow, ks * 32 + g * 8, D, kv_ok[kt]);
        }
    }
    f32x4_t dk[KT][D / 16], dv[KT][D / 16];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) { dk[kt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dv[kt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the K / V fragments are in registers: from here on vmcnt counts DMA only
    __syncthreads();
    const int n_my = lds_tiles[DKDV_MAXT];

    // ---- DMA of one query tile into ring slot `slot`
    auto issue_tile = [&](int qi, int slot) {
        char* buf = dyn_lds + slot * BUF;
        const int64_t Rq0 = (int64_t)qi * 64;
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            const int i = wave * IPW + j;                             // rows 4i .. 4i+3
            const int row = 4 * i + (lane >> 4);
            int64_t R = Rq0 + row; if (R > nR - 1) R = nR - 1;
            const unsigned ru = (unsigned)R, tu = p.group == 1 ? ru : __umulhi(ru, p.group_magic);
            const int hq = (int)(ru - tu * (unsigned)p.group);
            const int64_t hoff = (int64_t)(kvh * p.group + hq) * D + (((lane & 15) ^ dkey(row & 15)) << 3);
            __builtin_amdgcn_global_load_lds((att_gptr_t)(p.Q + (int64_t)tu * p.q_ld + hoff), (att_lptr_t)(buf + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((att_gptr_t)(p.dO + (int64_t)tu * p.do_ld + hoff), (att_lptr_t)(buf + TILE + i * 1024), 16, 0, 0);
        }
        if (wave == 0) {
            int64_t R = Rq0 + lane; if (R > nR - 1) R = nR - 1;
            const unsigned ru = (unsigned)R, tu = p.group == 1 ? ru : __umulhi(ru, p.group_magic);
            const int hq = (int)(ru - tu * (unsigned)p.group);
            const int64_t si = (int64_t)(kvh * p.group + hq) * p.T + tu;
            char* mb = buf + 2 * TILE;
            __builtin_amdgcn_global_load_lds((att_gptr_t)(lse2 + si), (att_lptr_t)(mb), 4, 0, 0);
            __builtin_amdgcn_global_load_lds((att_gptr_t)(p.delta + si), (att_lptr_t)(mb + 256), 4, 0, 0);
            __builtin_amdgcn_global_load_lds((att_gptr_t)(p.pre + tu), (att_lptr_t)(mb + 512), 4, 0, 0);
            __builtin_amdgcn_global_load_lds((att_gptr_t)(p.lo + tu), (att_lptr_t)(mb + 768), 4, 0, 0);
            __builtin_amdgcn_global_load_lds((att_gptr_t)(p.hi + tu), (att_lptr_t)(mb + 1024), 4, 0, 0);
        }
    };
#pragma unroll
    for (int j = 0; j < NB - 1; ++j)
        if (j < n_my) issue_tile(__builtin_amdgcn_readfirstlane(lds_tiles[j]), j);

    // per-lane fragment addresses inside a tile (swizzled); qt / dt / half offsets are added in the loops
    const int keyu = dkey(u);
    const int trow = g * 4 + (u >> 2);                                // row (mod 16) of this lane's transposing reads
    const int keyt = dkey(trow);
    const int tr_lo = trow * 256 + (u & 1) * 8, tr_c = (u & 3) >> 1;  // + ((dt*2 + tr_c) ^ keyt) * 16 + qb * 256

    for (int it = 0; it < n_my; ++it) {
        // my DMA share of tile `it` has landed when at most (tiles issued after it) x (my instructions per tile) are outstanding
        {
            const int after = (n_my - 1 - it) < (NB - 2) ? (n_my - 1 - it) : (NB - 2);
            if (wave == 0) {
                if (after >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER0 > 63 ? 63 : 3 * PER0) : "memory");
                else if (after == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER0) : "memory");
                else if (after == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER0) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                if (after >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory");
                else if (after == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
                else if (after == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __builtin_amdgcn_s_barrier();                                 // tile `it` is complete for everybody; everybody is done with tile it-1
        asm volatile("" ::: "memory");
        if (it + NB - 1 < n_my) issue_tile(__builtin_amdgcn_readfirstlane(lds_tiles[it + NB - 1]), (it + NB - 1) % NB);     // = the slot of tile it-1

        const int qi = __builtin_amdgcn_readfirstlane(lds_tiles[it]);
        const int tile_minpre = __builtin_amdgcn_readfirstlane(lds_minpre[it]);
        const char* buf = dyn_lds + (it % NB) * BUF;
        const char* lds_q = buf;
        const char* lds_do = buf + TILE;
        const float* lds_lse = reinterpret_cast<const float*>(buf + 2 * TILE);
        const float* lds_dlt = lds_lse + 64;
        const int* lds_pre = reinterpret_cast<const int*>(lds_lse + 128);
        const int* lds_lo = lds_pre + 64;
        const int* lds_hi = lds_pre + 128;

        f32x4_t s[KT][4], dp[KT][4];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) { s[kt][qt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dp[kt][qt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < D / 32; ++ks) {
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                const int off = (qt * 16 + u) * 256 + (((ks * 4 + g) ^ keyu) << 4);
                const bf16x8_t qa = *reinterpret_cast<const bf16x8_t*>(lds_q + off);
                const bf16x8_t da = *reinterpret_cast<const bf16x8_t*>(lds_do + off);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    s[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[kt][ks], s[kt][qt], 0, 0, 0);
                    dp[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[kt][ks], dp[kt][qt], 0, 0, 0);
                }
            }
        }
        typedef __attribute__((ext_vector_type(4))) int i32x4_t;
        bf16x8_t pf0[KT], pf1[KT], df0[KT], df1[KT];
        const int rows_valid = (int)((nR - (int64_t)qi * 64) < 64 ? (nR - (int64_t)qi * 64) : 64);
        const bool full = (kvb0 + KB <= tile_minpre) && (kvb0 + KB <= p.n_slots) && rows_valid == 64;      // block-uniform
        if (full) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4_t l4[2], d4[2];
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    l4[q2] = *reinterpret_cast<const f32x4_t*>(lds_lse + (2 * h + q2) * 16 + g * 4);
                    d4[q2] = *reinterpret_cast<const f32x4_t*>(lds_dlt + (2 * h + q2) * 16 + g * 4);
                }
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    f32x4_t pr[2], ds[2];
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][2 * h + q2][r], p.scale_log2, -l4[q2][r]));
                            pr[q2][r] = pv; ds[q2][r] = pv * (dp[kt][2 * h + q2][r] - d4[q2][r]);
                        }
                    if (h == 0) { pf0[kt] = pack_frag(pr[0], pr[1]); df0[kt] = pack_frag(ds[0], ds[1]); }
                    else { pf1[kt] = pack_frag(pr[0], pr[1]); df1[kt] = pack_frag(ds[0], ds[1]); }
                }
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4_t l4[2], d4[2];
                i32x4_t p4[2], lo4[2], hi4[2];
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const int o = (2 * h + q2) * 16 + g * 4;
                    l4[q2] = *reinterpret_cast<const f32x4_t*>(lds_lse + o); d4[q2] = *reinterpret_cast<const f32x4_t*>(lds_dlt + o);
                    p4[q2] = *reinterpret_cast<const i32x4_t*>(lds_pre + o); lo4[q2] = *reinterpret_cast<const i32x4_t*>(lds_lo + o);
                    hi4[q2] = *reinterpret_cast<const i32x4_t*>(lds_hi + o);
                }
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    f32x4_t pr[2], ds[2];
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool ok = kv_ok[kt] & att_visible_nb(kv[kt], p4[q2][r], lo4[q2][r], hi4[q2][r]) & ((2 * h + q2) * 16 + g * 4 + r < rows_valid);
                            const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][2 * h + q2][r], p.scale_log2, -l4[q2][r]));
                            const float pv = ok ? e : 0.f;
                            pr[q2][r] = pv; ds[q2][r] = ok ? pv * (dp[kt][2 * h + q2][r] - d4[q2][r]) : 0.f;
                        }
                    if (h == 0) { pf0[kt] = pack_frag(pr[0], pr[1]); df0[kt] = pack_frag(ds[0], ds[1]); }
                    else { pf1[kt] = pack_frag(pr[0], pr[1]); df1[kt] = pack_frag(ds[0], ds[1]); }
                }
            }
        }
        // transposed fragments: rows qb + trow (first 8 bytes) and qb + 16 + trow (second), feature chunk dt*2 + tr_c
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) {
                const int off = hh * 32 * 256 + tr_lo + (((dt * 2 + tr_c) ^ keyt) << 4);
                const bf16x8_t q0 = make_frag(lds_read_tr16(lds_q + off), lds_read_tr16(lds_q + off + 16 * 256));
                const bf16x8_t o0 = make_frag(lds_read_tr16(lds_do + off), lds_read_tr16(lds_do + off + 16 * 256));
                _Pragma("unroll") for (int kt = 0; kt < KT; ++kt) dv[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(o0, hh ? pf1[kt] : pf0[kt], dv[kt][dt], 0, 0, 0);
                _Pragma("unroll") for (int kt = 0; kt < KT; ++kt) dk[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0, hh ? df1[kt] : df0[kt], dk[kt][dt], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // all LDS reads of this tile have returned before the barrier that frees its slot
    }
    // lane holds dK^T/dV^T[d = dt*16 + g*4 + r][kv]
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
    if (kv_ok[kt]) {
        if (part_k) {
            const int64_t kvd = (int64_t)p.n_kv * D;
            float* pk = part_k + ((int64_t)qz * p.n_slots + kv[kt]) * kvd + (int64_t)kvh * D;
            float* pv = part_v + ((int64_t)qz * p.n_slots + kv[kt]) * kvd + (int64_t)kvh * D;
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) {
                const int d = dt * 16 + g * 4;
                *reinterpret_cast<f32x4_t*>(pk + d) = dk[kt][dt]; *reinterpret_cast<f32x4_t*>(pv + d) = dv[kt][dt];
            }
        } else {
            const float scale = p.scale_log2 * 0.6931471805599453f;
            bf16_t* kr = p.dK + (int64_t)kv[kt] * p.dk_ld + (int64_t)kvh * D;
            bf16_t* vr = p.dV + (int64_t)kv[kt] * p.dv_ld + (int64_t)kvh * D;
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) {
                const int d = dt * 16 + g * 4;
                u32x2_t wk = {pack2bf(dk[kt][dt][0] * scale, dk[kt][dt][1] * scale), pack2bf(dk[kt][dt][2] * scale, dk[kt][dt][3] * scale)};
                u32x2_t wv = {pack2bf(dv[kt][dt][0], dv[kt][dt][1]), pack2bf(dv[kt][dt][2], dv[kt][dt][3])};
                *reinterpret_cast<u32x2_t*>(kr + d) = wk;
                *reinterpret_cast<u32x2_t*>(vr + d) = wv;
            }
        }
    }
}

// dK = scale * sum_z part_k[z], dV = sum_z part_v[z]  -> bf16
__global__ void attn_bwd_reduce_kernel(const float* __restrict__ part_k, const float* __restrict__ part_v, bf16_t* __restrict__ dK, int64_t dk_ld,
                                       bf16_t* __restrict__ dV, int64_t dv_ld, int n_slots, int kvd, int QS, float scale) {
    const int64_t nch = (int64_t)n_slots * (kvd / 4);
    const int64_t stride = (int64_t)n_slots * kvd;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nch; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t slot = i / (kvd / 4); const int c = (int)(i - slot * (kvd / 4)) * 4;
        f32x4_t ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < QS; ++z) {
            ak += *reinterpret_cast<const f32x4_t*>(part_k + z * stride + slot * kvd + c);
            av += *reinterpret_cast<const f32x4_t*>(part_v + z * stride + slot * kvd + c);
        }
        u32x2_t wk = {pack2bf(ak[0] * scale, ak[1] * scale), pack2bf(ak[2] * scale, ak[3] * scale)};
        u32x2_t wv = {pack2bf(av[0], av[1]), pack2bf(av[2], av[3])};
        *reinterpret_cast<u32x2_t*>(dK + slot * dk_ld + c) = wk;
        *reinterpret_cast<u32x2_t*>(dV + slot * dv_ld + c) = wv;
    }
}

static int dkdv_keys_per_block(int d_pad) { return (d_pad == 64 || d_pad == 128) ? 128 : 64; }   // 8-wave blocks where 8*D/512 is integral

static int dkdv_qsplit(int64_t T, int group, int n_kv, int64_t n_slots, int kb) {
    const int64_t n_qtiles = (T * group + 63) / 64;
    const int64_t kvblocks = ((n_slots + kb - 1) / kb) * n_kv;
    int64_t qs = (1024 + kvblocks - 1) / kvblocks;
    if (qs > 8) qs = 8;
    {   // tuning hook (A/B runs): TR1_DKDV_QS=<n> fixes the number of query slices
        static int force = -1;
        if (force < 0) { const char* e = getenv("TR1_DKDV_QS"); force = e ? atoi(e) : 0; }
        if (force > 0) qs = force;
    }
    if (qs > n_qtiles) qs = n_qtiles;
    if (qs < 1) qs = 1;
    while ((n_qtiles + qs - 1) / qs > DKDV_MAXT) ++qs;
    return (int)qs;
}

template <int D>
static int launch_bwd(const AttnParams& p, hipStream_t s, float* ws, int64_t ws_floats, const float* lse2) {
    const int64_t nR = (int64_t)p.T * p.group;
    const int n_qtiles = (int)((nR + 63) / 64);
    constexpr int KSTR = 2 * D + 16;
    constexpr int NW = (D == 64 || D == 128) ? 8 : 4;
    constexpr int KB = NW * 16;
    // TR1_DKDV_KT=2 selects the 4-wave x 32-key form (half the LDS reads per MFMA, but one wave per SIMD and - at the 512-register
    // limit - a single query tile in flight: 1.03 ms against 0.87 ms for the 8-wave form at config 3, so it stays an experiment)
    static int kt2 = -1;
    if (kt2 < 0) { const char* e = getenv("TR1_DKDV_KT"); kt2 = (e ? atoi(e) : 1) == 2 && NW == 8; }
    const size_t dyn_dq = 2 * (2 * ATT_KV * KSTR) + 64;
    const size_t dyn_kv = 2 * (2 * 64 * KSTR + (NW == 8 ? 0 : 2 * D * 144) + 64 * 5 * 4 + 16) + (DKDV_MAXT + 1) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_dq);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_kernel<D, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_kv);
        if (NW == 8) hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_kernel<D, 4, (NW == 8 ? 2 : 1)>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_kv);
        attr_set = true;
    }
    hipLaunchKernelGGL(attn_bwd_dq_kernel<D>, dim3((unsigned)((nR + 127) / 128), p.n_kv), dim3(256), dyn_dq, s, p);
    // head dim 128: the LDS-DMA staged forms.  TR1_DKDV_DMA = 0: register-staged 8 waves x 16 keys; 1: DMA, 4 waves x 32 keys; 2: DMA, 8 waves x 16 keys
    static int dma = -1;
    if (dma < 0) { const char* e = getenv("TR1_DKDV_DMA"); dma = e ? atoi(e) : 2; }
    constexpr int DMA_NB = 4;
    const size_t dyn_dma = DMA_NB * (2 * 64 * 256 + 64 * 5 * 4) + (2 * DKDV_MAXT + 2) * 4;
    static bool dma_attr = false;
    if (D == 128 && !dma_attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_dma_kernel<4, 2, DMA_NB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_dma);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_dma_kernel<8, 1, DMA_NB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_dma);
        dma_attr = true;
    }
    const bool use_dma = D == 128 && p.d_real == 128 && dma > 0 && lse2 != nullptr;
    const int QS = dkdv_qsplit(p.T, p.group, p.n_kv, p.n_slots, KB);
    const int64_t kvd = (int64_t)p.n_kv * p.d_real;
    float *pk = nullptr, *pv = nullptr;
    if (QS > 1) {
        const int64_t need = 2 * (int64_t)QS * p.n_slots * kvd;
        if (!ws || ws_floats < need) { tr1_set_error_("attention bwd: workspace too small"); return 1000; }
        pk = ws; pv = ws + (int64_t)QS * p.n_slots * kvd;
    }
    if (use_dma && dma == 2) hipLaunchKernelGGL((attn_bwd_dkdv_dma_kernel<8, 1, DMA_NB>), dim3((unsigned)(p.n_kv * QS), 1, (unsigned)((p.n_slots + 127) / 128)), dim3(512), dyn_dma, s, p, n_qtiles, lse2, pk, pv);
    else if (use_dma) hipLaunchKernelGGL((attn_bwd_dkdv_dma_kernel<4, 2, DMA_NB>), dim3((unsigned)(p.n_kv * QS), 1, (unsigned)((p.n_slots + 127) / 128)), dim3(256), dyn_dma, s, p, n_qtiles, lse2, pk, pv);
    else if (kt2) hipLaunchKernelGGL((attn_bwd_dkdv_kernel<D, 4, (NW == 8 ? 2 : 1)>), dim3((unsigned)(p.n_kv * QS), 1, (unsigned)((p.n_slots + KB - 1) / KB)), dim3(256), dyn_kv, s, p, n_qtiles, pk, pv);
    else hipLaunchKernelGGL((attn_bwd_dkdv_kernel<D, NW>), dim3((unsigned)(p.n_kv * QS), 1, (unsigned)((p.n_slots + KB - 1) / KB)), dim3(NW * 64), dyn_kv, s, p, n_qtiles, pk, pv);
    if (QS > 1) {
        const float scale = p.scale_log2 * 0.6931471805599453f;
        hipLaunchKernelGGL(attn_bwd_reduce_kernel, dim3(tr1_grid_1d(p.n_slots * kvd / 4, 256, 2048)), dim3(256), 0, s, pk, pv, p.dK, p.dk_ld, p.dV, p.dv_ld,
                           p.n_slots, (int)kvd, QS, scale);
    }
    return 0;
}

extern "C" int64_t tr1_attn_bwd_workspace_floats(int64_t T, int64_t n_heads, int64_t n_kv, int64_t n_slots, int64_t head_dim) {
    if (n_kv <= 0 || T <= 0) return 0;
    const int QS = dkdv_qsplit(T, (int)(n_heads / n_kv), (int)n_kv, n_slots, dkdv_keys_per_block((int)((head_dim + 31) / 32 * 32)));
    return QS > 1 ? 2 * (int64_t)QS * n_slots * n_kv * head_dim : 0;
}

// Scratch: qmeta_ws int32 [4*ceil(T*group/64)], delta fp32 [2*n_heads*T] (delta | log2-scaled LSE), ws_f32 of tr1_attn_bwd_workspace_floats() floats.
extern "C" int tr1_attn_bwd(const void* Q, int64_t q_ld, const void* K, int64_t k_ld, const void* V, int64_t v_ld, const void* KT,
                            int64_t kt_ld, const void* QT, int64_t qt_ld, const void* dOT, int64_t dot_ld, const void* O, int64_t o_ld,
                            const void* dO, int64_t do_ld, const void* lse, void* delta, void* dQ, int64_t dq_ld, void* dK, int64_t dk_ld,
                            void* dV, int64_t dv_ld, const void* pre, const void* lo, const void* hi, void* qmeta_ws, void* ws_f32,
                            int64_t ws_floats, int64_t T, int64_t n_heads, int64_t n_kv, int64_t n_slots, int64_t head_dim, float scale,
                            void* stream) {
    AttnParams p; memset(&p, 0, sizeof(p));
    TR1_CHECK_ARG(n_kv > 0 && n_heads % n_kv == 0, "attention bwd: n_heads must be a multiple of n_kv");
    p.Q = (const bf16_t*)Q; p.q_ld = q_ld; p.K = (const bf16_t*)K; p.k_ld = k_ld; p.V = (const bf16_t*)V; p.v_ld = v_ld;
    p.KT = (const bf16_t*)KT; p.kt_ld = kt_ld; p.QT = (const bf16_t*)QT; p.qt_ld = qt_ld; p.dOT = (const bf16_t*)dOT; p.dot_ld = dot_ld;
    p.dO = (const bf16_t*)dO; p.do_ld = do_ld; p.dQ = (bf16_t*)dQ; p.dq_ld = dq_ld; p.dK = (bf16_t*)dK; p.dk_ld = dk_ld;
    p.dV = (bf16_t*)dV; p.dv_ld = dv_ld; p.lse = (float*)lse; p.delta = (float*)delta; p.pre = (const int*)pre; p.lo = (const int*)lo;
    p.hi = (const int*)hi; p.qmeta = (const int*)qmeta_ws;
    p.T = (int)T; const bool magic_ok = att_set_group(p, T, (int)(n_heads / n_kv)); p.n_kv = (int)n_kv; p.n_slots = (int)n_slots; p.d_real = (int)head_dim; p.nsplit = 1;
    p.scale_log2 = scale * 1.4426950408889634f;
    const int d_pad = (int)((head_dim + 31) / 32 * 32);
    TR1_CHECK_ARG(d_pad == 32 || d_pad == 64 || d_pad == 96 || d_pad == 128, "attention bwd: padded head dim must be 32/64/96/128");
    TR1_CHECK_ARG(magic_ok, "attention bwd: T * group^2 must stay below 2^32");
    TR1_CHECK_ARG(head_dim % 8 == 0 && q_ld % 8 == 0 && k_ld % 8 == 0 && v_ld % 8 == 0 && do_ld % 8 == 0 && o_ld % 8 == 0,
                  "attention bwd: dims must be multiples of 8");
    // Q^T / dO^T are only read by the 4-wave dK/dV form (head dims padded to 32 / 96); the 8-wave form transposes in its LDS reads
    const bool need_qt = dkdv_keys_per_block(d_pad) == 64;
    TR1_CHECK_ARG(!need_qt || (QT && dOT && qt_ld % 8 == 0 && qt_ld >= T * p.group && dot_ld % 8 == 0 && dot_ld >= T * p.group),
                  "attention bwd: Q^T / dO^T missing or leading dims too small");
    TR1_CHECK_ARG(dk_ld % 4 == 0 && dv_ld % 4 == 0, "attention bwd: dk/dv leading dims must be multiples of 4");
    if (T == 0 || n_slots == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int n_qtiles = (int)((T * p.group + 63) / 64);
    float* lse2 = (float*)delta + T * n_heads;                                   // second half of the delta scratch: log2-scaled LSE
    const unsigned delta_blocks = (unsigned)((T * n_heads + 15) / 16);          // >= 4 * n_qtiles (n_heads >= group)
    hipLaunchKernelGGL(attn_delta_kernel, dim3(delta_blocks > (unsigned)n_qtiles ? delta_blocks : (unsigned)n_qtiles), dim3(256), 0, s,
                       (const bf16_t*)dO, do_ld, (const bf16_t*)O, o_ld, (float*)delta, (int)T, (int)n_heads, (int)head_dim, p.pre, p.lo, p.hi,
                       (int*)qmeta_ws, p.group, n_qtiles, (const float*)lse, lse2);
    int rc = 0;
    switch (d_pad) {
        case 32: rc = launch_bwd<32>(p, s, (float*)ws_f32, ws_floats, lse2); break;
        case 64: rc = launch_bwd<64>(p, s, (float*)ws_f32, ws_floats, lse2); break;
        case 96: rc = launch_bwd<96>(p, s, (float*)ws_f32, ws_floats, lse2); break;
        default: rc = launch_bwd<128>(p, s, (float*)ws_f32, ws_floats, lse2); break;
    }
    if (rc) return rc;
    TR1_LAUNCH_CHECK();
}
