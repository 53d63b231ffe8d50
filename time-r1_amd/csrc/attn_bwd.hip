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

#define DKDV32_MAXT 512     // query tiles per block of the 32x32x16 kernel (its LDS budget is spent on the tile ring + P exchange)
#define ATT_QMETA 8         // ints per 64-row query tile in the mask summary written by attn_delta_kernel

// optional wave-timeline probe (tools/bench_attn.py --probe against a -DTR1_PROBE build of the library; not compiled into the product):
// lane 0 of every wave of ONE block (blockIdx.x == 0, blockIdx.z == 2) stamps s_memtime at up to 8 points of its first 64 tiles
#ifdef TR1_PROBE
__device__ unsigned long long* tr1_bwd_probe = nullptr;
extern "C" int probe_bwd_set_ptr(void* ptr) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(tr1_bwd_probe), &ptr, sizeof(ptr)); }
// the stamps of a tile stay in scalar registers and are written in one burst by BWD_FLUSH: reading an s_memtime result costs an
// s_waitcnt lgkmcnt(0), which would drain the LDS reads in flight at the stamped point
#define BWD_STAMPS unsigned long long bwd_st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define BWD_STAMP(it, slot) do { bwd_st_[slot] = __builtin_amdgcn_s_memtime(); } while (0)
#define BWD_FLUSH(it) do { if (tr1_bwd_probe && blockIdx.x == 0 && blockIdx.z == 2 && (threadIdx.x & 63) == 0 && (it) < 64) { \
    _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) tr1_bwd_probe[(((threadIdx.x >> 6) * 64 + (it)) * 8 + s_)] = bwd_st_[s_]; } } while (0)
#else
#define BWD_STAMPS do { } while (0)
#define BWD_STAMP(it, slot) do { } while (0)
#define BWD_FLUSH(it) do { } while (0)
#endif

// per 64 packed rows: (max pre, min lo, max hi) over rows with a non-empty [lo,hi].  One wave per tile; runs inside attn_delta_kernel (its
// first n_qtiles blocks): under a weight-gradient GEMM on the side stream every extra small launch of the main stream waits ~150 us for
// wave slots, so the 6 us of work used to cost 170 us per layer as a launch of its own.
TR1_DEV void attn_qmeta_tile(const int* __restrict__ pre, const int* __restrict__ lo, const int* __restrict__ hi, int* __restrict__ qmeta,
                             int T, int group, int tile, int lane) {
    const int64_t R = (int64_t)tile * 64 + lane;
    int mp = 0, ml = 0x7fffffff, mh = -1, mnp = 0x7fffffff, mxl = -1, mnh = 0x7fffffff;
    if (R < (int64_t)T * group) {
        const int t = (int)((unsigned)R / (unsigned)group);
        mp = pre[t]; mnp = pre[t];
        if (hi[t] >= lo[t]) { ml = lo[t]; mh = hi[t]; mxl = lo[t]; mnh = hi[t]; }
        else { mxl = 0x7fffffff; mnh = -1; }                  // a row with an empty [lo, hi] sees no key through its interval
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        mp = max(mp, __shfl_xor(mp, o, 64)); ml = min(ml, __shfl_xor(ml, o, 64)); mh = max(mh, __shfl_xor(mh, o, 64)); mnp = min(mnp, __shfl_xor(mnp, o, 64));
        mxl = max(mxl, __shfl_xor(mxl, o, 64)); mnh = min(mnh, __shfl_xor(mnh, o, 64));
    }
    // [3]: the smallest prefix length among the tile's rows - keys below it are visible to EVERY row of the tile (mask-free fast path);
    // [4], [5]: largest lo / smallest hi - keys in [max lo, min hi] are visible to every row as well (causal prompt rows below the diagonal)
    if (lane == 0) {
        int* q = qmeta + tile * ATT_QMETA;
        q[0] = mp; q[1] = ml; q[2] = mh; q[3] = mnp; q[4] = mxl; q[5] = mnh;
    }
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
                const int mp = p.qmeta[qi * ATT_QMETA], ml = p.qmeta[qi * ATT_QMETA + 1], mh = p.qmeta[qi * ATT_QMETA + 2];
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
                const int mp = p.qmeta[qi * ATT_QMETA], ml = p.qmeta[qi * ATT_QMETA + 1], mh = p.qmeta[qi * ATT_QMETA + 2];
                mnp = p.qmeta[qi * ATT_QMETA + 3];
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

// ------------------------------------------------------------------------------ dK/dV on 32x32x16 MFMA tiles, role-split wave pairs (head dim 128)
// Round 3.  The 16x16x32 forms above read one LDS fragment per MFMA and were LDS- / issue-bound (MFMA busy 17-20 %).  Here a wave owns 32 keys
// and works on v_mfma_f32_32x32x16_bf16 tiles: an operand fragment read from LDS feeds twice the MACs (half the LDS bytes per FLOP), and the
// register cost of 32 keys (K + V fragments 64, dK + dV accumulators 128, S + dP 64) is split over a PAIR of waves:
//   role 0:  S = Q K^T  ->  P = exp2(S * scale_log2 - lse2) (masked)  ->  P to its pair partner through LDS  ->  dV^T += dO^T P
//   role 1:  dP = dO V^T                         (barrier)  ->  dS = P o (dP - delta)                          ->  dK^T += Q^T dS
// Both roles run the SAME instruction skeleton on swapped tiles (16 MFMAs against the stationary K / V fragments, 16 MFMAs against the
// transposed tile), ~170 registers each, two waves per SIMD - one of each role (waves w and w + 4 share a SIMD), so one wave's exp2 / pack
// work runs beside the other's MFMAs.  S and dP come out in the SAME accumulator layout (lane = key, registers = query rows), so the P hand-off
// is lane-private: each lane writes its 32 values and its partner's same lane reads them back (b128, conflict-free, no shuffles).
// The C layout of a 32x32 tile (lane (n, h): rows (r&3) + 8(r>>2) + 4h) IS a legal B operand for the next product after a k-permutation:
// registers 8c..8c+7 = the 16 query rows q = 16c + (j&3) + 8(j>>2) + 4h (j = k-slot of lane half h), and the transposed A operand
// (dO^T / Q^T: 32 features x those 16 rows) is two ds_read_b64_tr_b16 per lane from the ROW-major tile (rows 4h..4h+3 and 8+4h..8+4h+3).
// Tile images as in the DMA kernel above (64 rows x 256 B, unpadded, global_load_lds with the swizzle applied on the SOURCE address), but
// keyed with skey(row) = (row&3)<<2 | (row>>2)&3: the 16 rows of a b128 service group get 16 distinct chunk positions, and the 4 rows x 4
// chunks of a 32-lane transposing read land in 4 disjoint aligned chunk groups - both read shapes are bank-conflict free.

template <int NB, int NP>
__global__ __launch_bounds__(NP * 128) void attn_bwd_dkdv32_kernel(AttnParams p, int n_qtiles, const float* __restrict__ lse2, float* __restrict__ part_k,
                                                              float* __restrict__ part_v) {
    constexpr int D = 128, KB = NP * 32, SW = 2 * NP - 1;             // NP wave pairs x 32 keys; SW: the wave that also carries the row statistics
    constexpr int TILE = 64 * 256, META = 64 * 5 * 4, BUF = 2 * TILE + META, PEX = 64 * 32 * 2;      // P exchange: bf16, one buffer per tile parity
    static_assert(NB == 3, "ring depth (the top-of-tile wait is vmcnt(0): tile it+1 was requested one iteration ago)");
    extern __shared__ __attribute__((aligned(256))) char dyn_lds[];  // [NB][Q rows | dO rows | lse2, delta, pre, lo, hi] | [2][NP] P exchange | tile list
    char* lds_pex = dyn_lds + NB * BUF;
    int* lds_tiles = reinterpret_cast<int*>(lds_pex + 2 * NP * PEX);
    int* lds_full = lds_tiles + DKDV32_MAXT + 1;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), c32 = lane & 31, h = lane >> 5;
    const int role = wave / NP, pair = wave - role * NP;              // waves w, w + 4, w + 8 share a SIMD: every SIMD hosts both roles
    const bool dma_wave = wave < 8;                                   // the 32 row-group DMA instructions of a tile: 4 each for waves 0..7
    const int QS = gridDim.x / p.n_kv;
    const int kvh = blockIdx.x % p.n_kv, qz = blockIdx.x / p.n_kv;
    const int kvb0 = blockIdx.z * KB;
    const int kv = kvb0 + pair * 32 + c32;
    const bool kv_ok = kv < p.n_slots;
    const unsigned nR = (unsigned)p.T * (unsigned)p.group;

    if (wave == 0) {      // this block's list of relevant query tiles (qi = qz, qz+QS, ...) and whether every (row, key) pair of the visit is visible
        int count = 0;
        const int n_cand = (n_qtiles - qz + QS - 1) / QS;
        const bool keys_in = kvb0 + KB <= p.n_slots;
        for (int base = 0; base < n_cand; base += 64) {
            const int c = base + lane, qi = qz + c * QS;
            bool rel = false, full = false;
            if (c < n_cand) {
                const int* qm = p.qmeta + qi * ATT_QMETA;
                const int mp = qm[0], ml = qm[1], mh = qm[2], mnp = qm[3], mxl = qm[4], mnh = qm[5];
                rel = (kvb0 < mp) || (kvb0 + KB - 1 >= ml && kvb0 <= mh);
                full = keys_in && ((unsigned)qi * 64u + 64u <= nR) && ((kvb0 + KB <= mnp) || (mxl <= kvb0 && kvb0 + KB - 1 <= mnh));
            }
            const unsigned long long mask = __ballot(rel);
            if (rel) { const int at = count + __popcll(mask & ((1ull << lane) - 1ull)); lds_tiles[at] = qi; lds_full[at] = full ? 1 : 0; }
            count += __popcll(mask);
        }
        if (lane == 0) lds_tiles[DKDV32_MAXT] = count;
    }
    // stationary B fragments of this wave's 32 keys: K rows (role 0) or V rows (role 1); lane (key c32, half h) holds features ks*16 + h*8 .. +7
    bf16x8_t sf[D / 16];
    {
        const bf16_t* base = role == 0 ? p.K : p.V;
        const int64_t ld = role == 0 ? p.k_ld : p.v_ld;
        const bf16_t* srow = base + (int64_t)(kv_ok ? kv : 0) * ld + (int64_t)kvh * D;
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) sf[ks] = load_row_frag(srow, ks * 16 + h * 8, D, kv_ok);
    }
    f32x16_t acc[4];                                                  // dV^T (role 0) / dK^T (role 1): [32-feature block][C layout]
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[db][r] = 0.f;
    // the stationary fragments are in registers before the loop - and hipcc must KNOW it (an asm that reads them makes it place the wait itself):
    // with the loads still pending in its model it would put an s_waitcnt vmcnt(0) in front of their first use in every iteration, which
    // also waits for the hand-issued DMA.  From here on vmcnt counts DMA only.
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) asm volatile("" ::"v"(sf[ks]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int n_my = lds_tiles[DKDV32_MAXT];

    // ---- DMA of one query tile: 32-bit byte offsets from the uniform bases (the host checked T * ld * 2 < 2^32).  The offsets of a tile are
    // computed BEFORE the barrier that frees its ring slot (they only depend on the tile id), so that behind the barrier the wave issues its
    // DMA instructions and nothing else.
    const char* qbase = reinterpret_cast<const char*>(p.Q) + (int64_t)kvh * p.group * 256;
    const char* dobase = reinterpret_cast<const char*>(p.dO) + (int64_t)kvh * p.group * 256;
    const unsigned q_ldb = (unsigned)p.q_ld * 2u, do_ldb = (unsigned)p.do_ld * 2u;
    struct TileAddr { unsigned q[2], o[2]; unsigned st_t, st_si; };      // byte offsets (the statistics: element indices, n_heads * T < 2^30)
    auto tile_addr = [&](int qi, TileAddr& ta) {
        const unsigned Rq0 = (unsigned)qi * 64u;
        int ln = lane;
        asm volatile("" : "+v"(ln));                                  // lane constants are rebuilt here (a handful of VALU), not kept live / spilled
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (!dma_wave) break;
            const unsigned row = 4u * (wave * 2 + j) + ((unsigned)ln >> 4);
            unsigned R = Rq0 + row; R = R < nR ? R : nR - 1;
            const unsigned tu = p.group == 1 ? R : __umulhi(R, p.group_magic);
            const unsigned hb = (R - tu * (unsigned)p.group) * 256u + (unsigned)(((ln & 15) ^ skey(row & 15)) << 4);
            ta.q[j] = tu * q_ldb + hb; ta.o[j] = tu * do_ldb + hb;
        }
        if (wave == SW) {
            unsigned R = Rq0 + (unsigned)ln; R = R < nR ? R : nR - 1;
            const unsigned tu = p.group == 1 ? R : __umulhi(R, p.group_magic);
            ta.st_t = tu;
            ta.st_si = (unsigned)(kvh * p.group + (int)(R - tu * (unsigned)p.group)) * (unsigned)p.T + tu;
        }
    };
    // The DMA instructions are written in assembly: for the builtin, hipcc tracks the asynchronous LDS write and puts an s_waitcnt vmcnt(0)
    // in front of the next LDS read it cannot prove disjoint (all of them: one dynamic LDS array) - the "prefetch" then completes before the
    // first operand read of the tile, i.e. nothing is prefetched.  Landing is ordered by the hand-placed vmcnt wait + barrier at the loop top.
    const unsigned lds_base = (unsigned)(uintptr_t)(att_lptr_t)dyn_lds;
    const float* dlt_base = p.delta;
    const int *pre_base = p.pre, *lo_base = p.lo, *hi_base = p.hi;
#define DMA16(voff, sbase, m0v) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff), "s"(sbase) : "memory", "m0")
#define DMA4(voff, sbase, m0v) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(m0v), "v"(voff), "s"(sbase) : "memory", "m0")
    auto issue_tile = [&](const TileAddr& ta, int slot) {
        const unsigned buf = lds_base + slot * BUF;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (!dma_wave) break;
            const unsigned dst = buf + (wave * 2 + j) * 1024;
            DMA16(ta.q[j], qbase, dst);
            DMA16(ta.o[j], dobase, dst + TILE);
        }
        if (wave == SW) {                                             // row statistics (a role-1 wave: its tile work is the lighter one)
            const unsigned mb = buf + 2 * TILE;
            const unsigned so = ta.st_si * 4u, to = ta.st_t * 4u;
            DMA4(so, lse2, mb);
            DMA4(so, dlt_base, mb + 256);
            DMA4(to, pre_base, mb + 512);
            DMA4(to, lo_base, mb + 768);
            DMA4(to, hi_base, mb + 1024);
        }
    };
#undef DMA16
#undef DMA4
    TileAddr ta;
    if (n_my > 0) { tile_addr(__builtin_amdgcn_readfirstlane(lds_tiles[0]), ta); issue_tile(ta, 0); }
    if (n_my > 1) { tile_addr(__builtin_amdgcn_readfirstlane(lds_tiles[1]), ta); issue_tile(ta, 1); }

    // ---- LDS addressing.  All reads go through 32-bit LDS addresses built by hand: every tile image, ring slot and half-tile offset is a
    // multiple of 256 bytes and the swizzle only touches address bits 4..7, so a fragment address is (per-tile base) ^ (compile-time constant)
    // + an immediate offset - ONE v_xor per distinct fragment column (8 + 8 per tile).  Written as pointer arithmetic, hipcc spent ~3 VALU on
    // every one of the 60 LDS reads of a tile (v_add3 / v_xad / v_subrev against the rotating slot base): 395 VALU per wave and tile made
    // the kernel VALU-bound (44 % VALU busy against 29 % MFMA busy).
    typedef const __attribute__((address_space(3))) bf16x8_t* lds_b128_t;
    typedef const __attribute__((address_space(3))) f32x4_t* lds_f128_t;
    typedef __attribute__((ext_vector_type(4))) int i32x4_t;
    typedef const __attribute__((address_space(3))) i32x4_t* lds_i128_t;
#define LDS_B128(addr) (*(lds_b128_t)(uintptr_t)(addr))
#define LDS_F128(addr) (*(lds_f128_t)(uintptr_t)(addr))
#define LDS_I128(addr) (*(lds_i128_t)(uintptr_t)(addr))
#define LDS_TR16(addr) __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(uintptr_t)(addr)))
    const int ti = lane & 15, tgrp = (lane >> 4) & 1;
    // b128 A fragment of row c32 (+32 per half): chunk (ks*2 + h) ^ skey(row)      -> byte (ks*32) ^ a_lane inside the row
    const unsigned a_lane = (unsigned)(c32 * 256 + ((h ^ skey(c32 & 15)) << 4));
    // transposing read: row 4h + ti/4 (+8: second half, chunk key ^ 2), chunk (db*4 + tgrp*2 + (ti&3)/2) ^ skey(row), 8 bytes at (ti&1)*8
    const unsigned t_lane = (unsigned)((4 * h + (ti >> 2)) * 256 + (ti & 1) * 8 + (((tgrp * 2 + ((ti & 3) >> 1)) ^ (((ti >> 2) << 2) | h)) << 4));
    const unsigned pex_lane = lds_base + NB * BUF + pair * PEX + lane * 16;
    const unsigned st_lane = (unsigned)(16 * h);                      // row statistics: 4 floats / ints from row 4h (+ 8i + 32 qb)

    // phase-1 product of one 32-row half of a tile: c = X[rows qb*32..+31] . sf  (S for role 0 with X = Q, dP for role 1 with X = dO).
    // Operand reads run a fixed distance ahead of their MFMAs (explicit register ring + scheduling fences): left alone, hipcc hoists every
    // LDS read of the phase to its top and spills.
    auto product1 = [&](unsigned xa, f32x16_t& c) {                   // xa: tile rows + qb * 8192 + a_lane (LDS byte address)
        // 8 waves (2 per SIMD): two accumulators (even / odd k-steps) - a single chain of 8 dependent 32x32x16 MFMAs runs at the instruction's
        // latency (64 cycles), not its issue rate (32).  12 waves (3 per SIMD, 168 registers): one chain, the other waves fill the pipe.
        constexpr bool TWO = NP == 4;
        f32x16_t c1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { c[r] = 0.f; if (TWO) c1[r] = 0.f; }
        constexpr int AH = NP == 4 ? 3 : 2;                           // k-steps of lookahead
        bf16x8_t a[AH + 1];
#pragma unroll
        for (int ks = 0; ks < AH; ++ks) a[ks] = LDS_B128(xa ^ (ks * 32));
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) {
            if (ks + AH < D / 16) a[(ks + AH) % (AH + 1)] = LDS_B128(xa ^ ((ks + AH) * 32));
            if (TWO && (ks & 1)) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks % (AH + 1)], sf[ks], c1, 0, 0, 0);
            else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks % (AH + 1)], sf[ks], c, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (TWO) {
#pragma unroll
            for (int r = 0; r < 16; ++r) c[r] += c1[r];
        }
    };
    // phase-2 product over two 16-row chunks: acc^T[feature][key] += Y^T[feature][q] * frag[q][key]
    auto product2 = [&](unsigned ya, const bf16x8_t& f0, const bf16x8_t& f1) {      // ya: tile rows + first chunk * 4096 + t_lane
        constexpr int TH = NP == 4 ? 3 : 2;                           // MFMAs of lookahead
        bf16x8_t a[TH + 1];
#define P2_LD(n) make_frag(LDS_TR16((ya ^ (((n) & 3) * 64)) + ((n) >> 2) * 4096), LDS_TR16((ya ^ (((n) & 3) * 64 + 32)) + ((n) >> 2) * 4096 + 2048))
#pragma unroll
        for (int n = 0; n < TH; ++n) a[n] = P2_LD(n);
#pragma unroll
        for (int n = 0; n < 8; ++n) {                                 // n = chunk * 4 + db
            if (n + TH < 8) a[(n + TH) % (TH + 1)] = P2_LD(n + TH);
            acc[n & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[n % (TH + 1)], (n >> 2) ? f1 : f0, acc[n & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef P2_LD
    };
    // role 0: S of one 32-row half (in c) -> P, packed to two B fragments of the dV product and written to the exchange buffer
    auto make_p = [&](f32x16_t& c, int qb, unsigned sa, bool full, int rows_valid, bf16x8_t& f0, bf16x8_t& f1, unsigned pex) {   // sa: statistics + st_lane
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4_t l4 = LDS_F128(sa + (qb * 32 + 8 * i) * 4);
            if (full) {
#pragma unroll
                for (int j = 0; j < 4; ++j) c[4 * i + j] = __builtin_amdgcn_exp2f(__builtin_fmaf(c[4 * i + j], p.scale_log2, -l4[j]));
            } else {
                const i32x4_t p4 = LDS_I128(sa + 512 + (qb * 32 + 8 * i) * 4), lo4 = LDS_I128(sa + 768 + (qb * 32 + 8 * i) * 4),
                              hi4 = LDS_I128(sa + 1024 + (qb * 32 + 8 * i) * 4);
                // (round 6) vector instructions only: four compare + select pairs instead of four compares whose lane masks were combined by scalar instructions
                // (each of which waits for the vector results in front of it: attn_fwd64.hip's timeline, 4 600 against 2 600 cycles per masked tile); same values
                const int q1 = qb * 32 + 8 * i + 4 * h;
                const int rv_eff = kv_ok ? rows_valid : 0;            // key beyond the cache / row beyond the packed rows: nothing visible
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(c[4 * i + j], p.scale_log2, -l4[j]));
                    float x = (kv <= hi4[j]) ? e : 0.f;
                    asm volatile("" : "+v"(x));
                    x = (kv >= lo4[j]) ? x : 0.f;
                    asm volatile("" : "+v"(x));
                    x = (kv < p4[j]) ? e : x;
                    asm volatile("" : "+v"(x));
                    c[4 * i + j] = (q1 + j < rv_eff) ? x : 0.f;
                }
            }
        }
        f0 = pack8(c, 0); f1 = pack8(c, 8);
        *(__attribute__((address_space(3))) bf16x8_t*)(uintptr_t)(pex + (2 * qb) * 1024) = f0;
        *(__attribute__((address_space(3))) bf16x8_t*)(uintptr_t)(pex + (2 * qb + 1) * 1024) = f1;
    };
    // role 1: dP of one 32-row half (in c) and the partner's P -> dS, packed to two B fragments of the dK product
    auto make_ds = [&](const f32x16_t& c, int qb, unsigned sa, unsigned pex, bf16x8_t& f0, bf16x8_t& f1) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const u32x4_t pw = __builtin_bit_cast(u32x4_t, LDS_B128(pex + (2 * qb + hf) * 1024));     // P[q rows of the chunk][key]: 8 bf16 in k-slot order
            const f32x4_t d0 = LDS_F128(sa + 256 + (qb * 32 + hf * 16) * 4), d1 = LDS_F128(sa + 256 + (qb * 32 + hf * 16 + 8) * 4);
            const int b = hf * 8;
            u32x4_t w;
#pragma unroll
            for (int e = 0; e < 2; ++e) {                             // masked entries carry P = 0
                w[e] = pack2bf(bflo(pw[e]) * (c[b + 2 * e] - d0[2 * e]), bfhi(pw[e]) * (c[b + 2 * e + 1] - d0[2 * e + 1]));
                w[2 + e] = pack2bf(bflo(pw[2 + e]) * (c[b + 4 + 2 * e] - d1[2 * e]), bfhi(pw[2 + e]) * (c[b + 4 + 2 * e + 1] - d1[2 * e + 1]));
            }
            if (hf == 0) f0 = __builtin_bit_cast(bf16x8_t, w); else f1 = __builtin_bit_cast(bf16x8_t, w);
        }
    };

    // prologue: tile 0 has landed (this wave's share; tile 1 may still be in flight: 4 instructions, wave 7 carries 5 statistics rows more)
    if (n_my > 1) {                                                   // (NP = 4: wave 7 issues 4 + 5 instructions per tile; NP = 6: wave 11 only the 5)
        if (wave == SW) { if (SW < 8) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(9) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5) : "memory"); }
        else if (dma_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4) : "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (role == 0 && n_my > 0) {                                      // P of tile 0
        const int qi0 = __builtin_amdgcn_readfirstlane(lds_tiles[0]);
        const int rv = (int)(nR - (unsigned)qi0 * 64u < 64u ? nR - (unsigned)qi0 * 64u : 64u);
        const bool full0 = __builtin_amdgcn_readfirstlane(lds_full[0]) != 0;
        f32x16_t c;
        bf16x8_t f0, f1;
        product1(lds_base + a_lane, c);
        make_p(c, 0, lds_base + 2 * TILE + st_lane, full0, rv, f0, f1, pex_lane);
        product1(lds_base + 8192 + a_lane, c);
        make_p(c, 1, lds_base + 2 * TILE + st_lane, full0, rv, f0, f1, pex_lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    // One tile loop PER ROLE (same trip count, same barriers): with both roles inside one loop body hipcc gave the accumulators different
    // registers on the two paths and moved all 64 of them at the join - 130-190 v_mov per tile.
#define TILE_TOP()                                                                                                                        \
        BWD_STAMPS;                                                                                                                       \
        BWD_STAMP(it, 0);                                                                                                                 \
        if (it + 2 < n_my) tile_addr(__builtin_amdgcn_readfirstlane(lds_tiles[it + 2]), ta);   /* DMA addresses of the tile requested behind the barrier */ \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          /* this wave's share of tile it+1 (requested one iteration ago) has landed */ \
        BWD_STAMP(it, 1);                                                                                                                 \
        __builtin_amdgcn_s_barrier();                              /* tile it+1 and P(it) are published; everybody is done with tile it-1 */ \
        asm volatile("" ::: "memory");                                                                                                    \
        BWD_STAMP(it, 2);                                                                                                                 \
        if (it + 2 < n_my) issue_tile(ta, (it + 2) % NB);         /* = the slot of tile it-1 */                                           \
        BWD_STAMP(it, 3);                                                                                                                 \
        const unsigned buf = lds_base + (unsigned)(it % NB) * BUF;                                                                        \
        const unsigned pex = pex_lane + (unsigned)(it & 1) * (NP * PEX);   /* P(it): written by the pair's role-0 wave one iteration ago */ \
        f32x16_t c;                                                                                                                       \
        bf16x8_t f0, f1
#define TILE_BOTTOM()                                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        /* all LDS traffic of this tile is complete before the barrier that publishes / frees */ \
        BWD_STAMP(it, 7);                                                                                                                 \
        BWD_FLUSH(it)
    if (role == 0) {
        for (int it = 0; it < n_my; ++it) {
            TILE_TOP();
            // per 32-row half: S(it+1) -> dV(it) over P(it) read back from the exchange buffer (the MFMAs the exp / pack work of S hides under) -> P(it+1)
            const bool more = it + 1 < n_my;
            const unsigned nbuf = lds_base + (unsigned)((it + 1) % NB) * BUF;
            const unsigned npex = pex_lane + (unsigned)((it + 1) & 1) * (NP * PEX);
            int rv = 64; bool fulln = false;
            if (more) {
                const int qn = __builtin_amdgcn_readfirstlane(lds_tiles[it + 1]);
                rv = (int)(nR - (unsigned)qn * 64u < 64u ? nR - (unsigned)qn * 64u : 64u);
                fulln = __builtin_amdgcn_readfirstlane(lds_full[it + 1]) != 0;
            }
            const unsigned xa = nbuf + a_lane, ya = buf + TILE + t_lane, sa = nbuf + 2 * TILE + st_lane;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                if (more) product1(xa + hf * 8192, c);
                if (hf == 0) BWD_STAMP(it, 4);
                f0 = LDS_B128(pex + (2 * hf) * 1024);
                f1 = LDS_B128(pex + (2 * hf + 1) * 1024);
                product2(ya + hf * 8192, f0, f1);
                if (more) make_p(c, hf, sa, fulln, rv, f0, f1, npex);
                if (hf == 0) BWD_STAMP(it, 5);
            }
            BWD_STAMP(it, 6);
            TILE_BOTTOM();
        }
    } else {
        for (int it = 0; it < n_my; ++it) {
            TILE_TOP();
            const unsigned xa = buf + TILE + a_lane, ya = buf + t_lane, sa = buf + 2 * TILE + st_lane;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                product1(xa + hf * 8192, c);                          // dP = dO V^T, 32 rows
                if (hf == 0) BWD_STAMP(it, 4);
                make_ds(c, hf, sa, pex, f0, f1);
                product2(ya + hf * 8192, f0, f1);                     // dK^T += Q^T dS
                if (hf == 0) BWD_STAMP(it, 5);
            }
            BWD_STAMP(it, 6);
            TILE_BOTTOM();
        }
    }
#undef TILE_TOP
#undef TILE_BOTTOM
#undef LDS_B128
#undef LDS_F128
#undef LDS_I128
#undef LDS_TR16
    // lane holds acc^T[feature = db*32 + 8i + 4h + j][key c32]: role 0 -> dV, role 1 -> dK (scaled).  The lane coordinates are rebuilt from a
    // laundered thread id: values kept live across the tile loop for the epilogue get spilled, and ANY scratch access makes hipcc wait on
    // vmcnt inside the loop - which would also wait for the hand-issued DMA
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int e_c32 = tid2 & 31, e_h = (tid2 >> 5) & 1;
    const int e_kv = kvb0 + pair * 32 + e_c32;
    if (e_kv < p.n_slots) {
        const float scale = role == 0 ? 1.f : p.scale_log2 * 0.6931471805599453f;
        if (part_k) {
            const int64_t kvd = (int64_t)p.n_kv * D;
            float* pp = (role == 0 ? part_v : part_k) + ((int64_t)qz * p.n_slots + e_kv) * kvd + (int64_t)kvh * D;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4_t v = {acc[db][4 * i], acc[db][4 * i + 1], acc[db][4 * i + 2], acc[db][4 * i + 3]};
                    *reinterpret_cast<f32x4_t*>(pp + db * 32 + 8 * i + 4 * e_h) = v;     // unscaled partials; the reduce kernel scales dK
                }
        } else {
            bf16_t* rowp = (role == 0 ? p.dV + (int64_t)e_kv * p.dv_ld : p.dK + (int64_t)e_kv * p.dk_ld) + (int64_t)kvh * D;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32x2_t w = {pack2bf(acc[db][4 * i] * scale, acc[db][4 * i + 1] * scale), pack2bf(acc[db][4 * i + 2] * scale, acc[db][4 * i + 3] * scale)};
                    *reinterpret_cast<u32x2_t*>(rowp + db * 32 + 8 * i + 4 * e_h) = w;
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------- dQ on 32x32x16 MFMA tiles (head dim 128)
// Round 3.  A wave owns 32 packed query rows (block = 8 waves = 256 rows), Q and dO stay in registers as B operands, 64-key K / V row tiles
// stream through a 4-deep LDS ring (asm-issued LDS DMA, swizzled on the source address like the dK/dV kernel above).  Scores are computed
// transposed, S^T[kv][q] = K Q^T and dP^T = V dO^T (A = the K / V rows as stored, two independent MFMA chains), so a lane holds ONE query row:
// lse2 / delta / the mask intervals are lane-local scalars, and dS^T comes out in the accumulator layout that is the B operand of
// dQ^T[d][q] += K^T[d][kv] dS^T[kv][q] after the k-permutation (K^T fragments: transposing reads of the same K row tile).
// 48 MFMAs per wave and tile against 32 b128 + 32 transposing reads (half the LDS bytes per FLOP of the 16x16x32 form), 2 waves per SIMD.
__global__ __launch_bounds__(512) void attn_bwd_dq32_kernel(AttnParams p, const float* __restrict__ lse2) {
    constexpr int D = 128, NB = 4, TILE = 64 * 256, BUF = 2 * TILE;
    extern __shared__ __attribute__((aligned(256))) char dyn_lds[];  // [NB][K rows | V rows] + block mask summary
    int* lds_meta = reinterpret_cast<int*>(dyn_lds + NB * BUF);      // [8][6]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), c32 = lane & 31, h = lane >> 5;
    const int kvh = blockIdx.y;
    const unsigned nR = (unsigned)p.T * (unsigned)p.group;
    const unsigned Rw0 = (unsigned)(gridDim.x - 1 - blockIdx.x) * 256u + (unsigned)wave * 32u;      // heaviest query blocks first
    const unsigned R = Rw0 + (unsigned)c32;
    const bool valid = R < nR;
    int tq, hq;
    att_split_row(p, valid ? R : nR - 1, tq, hq);
    const int pre = valid ? p.pre[tq] : 0, lo = valid ? p.lo[tq] : 1, hi = valid ? p.hi[tq] : 0;
    const int hi_c = hi < p.n_slots ? hi : p.n_slots - 1;             // the two visible intervals clamped to the cache's slots (see attn_fwd32_kernel)
    const int pre_e = pre < p.n_slots ? pre : p.n_slots, lo_e = hi_c >= lo ? lo : 0x7fffffff, hi_d = hi_c >= lo ? hi_c - lo : 0;
    const int64_t si = (int64_t)(kvh * p.group + hq) * p.T + tq;
    const bool fused_delta = p.lse2_out != nullptr;                   // wave-uniform (kernel argument)
    float lse = INFINITY, dlt = 0.f;                                  // log2-scaled LSE; +inf (no visible key / padding row) -> P = 0
    if (!fused_delta) { lse = valid ? lse2[si] : INFINITY; dlt = valid ? p.delta[si] : 0.f; }
    // wave summary of the masks: which tiles every row of the wave sees completely, which it sees at all
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
    if (lane == 0) {
        lds_meta[wave * 6 + 0] = wmaxpre; lds_meta[wave * 6 + 1] = wminlo; lds_meta[wave * 6 + 2] = wmaxhi;
        lds_meta[wave * 6 + 3] = wminpre; lds_meta[wave * 6 + 4] = wmaxlo; lds_meta[wave * 6 + 5] = wminhi;
    }
    // stationary B fragments: Q / dO row of this lane, features ks*16 + h*8 .. +7
    bf16x8_t qf[D / 16], dof[D / 16];
    {
        const int64_t hoff = (int64_t)(kvh * p.group + hq) * D;
        const bf16_t* qrow = p.Q + (int64_t)tq * p.q_ld + hoff;
        const bf16_t* drow = p.dO + (int64_t)tq * p.do_ld + hoff;
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) { qf[ks] = load_row_frag(qrow, ks * 16 + h * 8, D, valid); dof[ks] = load_row_frag(drow, ks * 16 + h * 8, D, valid); }
        if (fused_delta) {
            // what attn_delta_kernel computed for this row: delta = sum_d dO[d] O[d] (the lane's 64 features + its partner lane's), the log2-scaled LSE;
            // both also go to global memory for the dK/dV kernel that follows on the stream
            const bf16_t* orow = p.O + (int64_t)tq * p.o_ld + hoff;
            float s = 0.f;
#pragma unroll
            for (int ks = 0; ks < D / 16; ++ks) {
                const u32x4_t x = __builtin_bit_cast(u32x4_t, dof[ks]), y = __builtin_bit_cast(u32x4_t, load_row_frag(orow, ks * 16 + h * 8, D, valid));
#pragma unroll
                for (int j = 0; j < 4; ++j) s += bflo(x[j]) * bflo(y[j]) + bfhi(x[j]) * bfhi(y[j]);
            }
            s += __shfl_xor(s, 32, 64);
            const float l0 = valid ? p.lse[si] : NEG_INF;
            lse = (l0 == NEG_INF) ? INFINITY : l0 * 1.4426950408889634f;
            dlt = valid ? s : 0.f;
            if (valid && h == 0) { p.delta[si] = dlt; p.lse2_out[si] = lse; }
        }
    }
    f32x16_t acc[4];                                                  // dQ^T[feature block][C layout]
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[db][r] = 0.f;
    // (an asm that reads the fragments makes hipcc place the wait for their loads itself - see the dK/dV kernel; from here on vmcnt counts DMA only)
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) asm volatile("" ::"v"(qf[ks]), "v"(dof[ks]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int bmaxpre = 0, bminlo = 0x7fffffff, bmaxhi = -1;
#pragma unroll
    for (int w = 0; w < 8; ++w) { bmaxpre = max(bmaxpre, lds_meta[w * 6]); bminlo = min(bminlo, lds_meta[w * 6 + 1]); bmaxhi = max(bmaxhi, lds_meta[w * 6 + 2]); }
    const TileRange tr = att_tile_range(bmaxpre, bminlo, bmaxhi, p.n_slots);
    const int n_my = tr.n_rel;
    if (p.qmeta_out && kvh == 0 && (wave & 1) == 0 && lane < 6) {
        // the mask summary of 64-row tile (this wave's 32 rows + the next wave's): attn_qmeta_tile's six values (max pre, min lo, max hi, min pre, max lo, min hi)
        const unsigned tile = Rw0 >> 6;
        if (tile * 64u < nR) {
            const int a = lds_meta[wave * 6 + lane], b = lds_meta[(wave + 1) * 6 + lane];
            const bool is_max = lane == 0 || lane == 2 || lane == 4;
            p.qmeta_out[tile * ATT_QMETA + lane] = is_max ? max(a, b) : min(a, b);
        }
    }

    // ---- DMA of one key tile (K rows + V rows): 4 instructions per wave, 32-bit byte offsets from uniform bases
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
        {   // this wave's share of tile `it` has landed when at most (tiles requested after it) x 4 instructions are outstanding
            const int after = (n_my - 1 - it) < (NB - 2) ? (n_my - 1 - it) : (NB - 2);
            if (after >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (after == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                                 // tile `it` is complete for everybody; everybody is done with tile it-1
        asm volatile("" ::: "memory");
        if (it + NB - 1 < n_my) issue_tile(att_tile_at(tr, it + NB - 1), (it + NB - 1) % NB);
        const int kv0 = att_tile_at(tr, it) * 64;
        // wave-uniform: does any row of the wave see a key of the tile / does every row see every key
        const bool any = (kv0 < wmaxpre) || (kv0 + 63 >= wminlo && kv0 <= wmaxhi);
        const bool full = (kv0 + 64 <= p.n_slots) && ((kv0 + 64 <= wminpre) || (wmaxlo <= kv0 && kv0 + 63 <= wminhi));
        if (any) {
            const unsigned kb_ = lds_base + (unsigned)(it % NB) * BUF;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {                          // 32-key halves of the tile
                const unsigned xa = kb_ + kb * 8192 + a_lane;
                f32x16_t cs, cp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { cs[r] = 0.f; cp[r] = 0.f; }
                constexpr int AH = 2;
                bf16x8_t ka[AH + 1], va[AH + 1];
#pragma unroll
                for (int ks = 0; ks < AH; ++ks) { ka[ks] = LDS_B128(xa ^ (ks * 32)); va[ks] = LDS_B128((xa ^ (ks * 32)) + TILE); }
#pragma unroll
                for (int ks = 0; ks < D / 16; ++ks) {
                    if (ks + AH < D / 16) { ka[(ks + AH) % (AH + 1)] = LDS_B128(xa ^ ((ks + AH) * 32)); va[(ks + AH) % (AH + 1)] = LDS_B128((xa ^ ((ks + AH) * 32)) + TILE); }
                    cs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[ks % (AH + 1)], qf[ks], cs, 0, 0, 0);
                    cp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[ks % (AH + 1)], dof[ks], cp, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // lane holds S^T / dP^T [kv = kv0 + kb*32 + 8i + 4h + j][q = its row] in register 4i + j
                if (full) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(cs[r], p.scale_log2, -lse));
                        cs[r] = pv * (cp[r] - dlt);
                    }
                } else {
                    // (round 6) the interval mask as vector instructions only (two compares + two selects on the clamped intervals, see attn_fwd32_kernel): a masked
                    // score becomes -inf, so P = exp2(-inf) = +0 - the value the select on the lane masks produced
                    const int base = kv0 + kb * 32 + 4 * h;
                    const unsigned mA = (unsigned)(base - lo_e), mD = (unsigned)hi_d;
                    const int mB = pre_e - base;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = (r & 3) + 8 * (r >> 2);
                        const float sv = cs[r];
                        float x = (mA + (unsigned)c <= mD) ? sv : NEG_INF;
                        asm volatile("" : "+v"(x));
                        x = (c < mB) ? sv : x;
                        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(x, p.scale_log2, -lse));
                        cs[r] = pv * (cp[r] - dlt);
                    }
                }
                const bf16x8_t f0 = pack8(cs, 0), f1 = pack8(cs, 8);
                const unsigned ya = kb_ + kb * 8192 + t_lane;
                constexpr int TH = 2;
                bf16x8_t a[TH + 1];
#define P2_LD(n) make_frag(LDS_TR16((ya ^ (((n) & 3) * 64)) + ((n) >> 2) * 4096), LDS_TR16((ya ^ (((n) & 3) * 64 + 32)) + ((n) >> 2) * 4096 + 2048))
#pragma unroll
                for (int n = 0; n < TH; ++n) a[n] = P2_LD(n);
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    if (n + TH < 8) a[(n + TH) % (TH + 1)] = P2_LD(n + TH);
                    acc[n & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[n % (TH + 1)], (n >> 2) ? f1 : f0, acc[n & 3], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef P2_LD
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // all LDS reads of this tile have returned before the barrier that frees its slot
    }
#undef LDS_B128
#undef LDS_TR16
    // lane holds dQ^T[feature = db*32 + 8i + 4h + j][its query row]
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const unsigned e_R = Rw0 + (unsigned)(tid2 & 31);
    const int e_h = (tid2 >> 5) & 1;
    if (e_R < nR) {
        int t2, hq2;
        att_split_row(p, e_R, t2, hq2);
        const float scale = p.scale_log2 * 0.6931471805599453f;
        bf16_t* row = p.dQ + (int64_t)t2 * p.dq_ld + (int64_t)(kvh * p.group + hq2) * D;
        if (p.rope_cos) {
            // M-RoPE backward in the epilogue: feature f < 64 and its rotate-half partner f + 64 (db + 2) sit in the SAME lane.  As the separate
            // kernels did: dQ is rounded to bf16 first, then da' = da c + db s, db' = db c - da s in fp32, rounded once more.
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
                        const float a = bf2f(f2bf(acc[db][4 * i + j] * scale)), b = bf2f(f2bf(acc[db + 2][4 * i + j] * scale));
                        const float sn = -s4[j];                     // rope_apply_kernel's sgn = -1
                        oa[j] = a * c4[j] - b * sn; ob[j] = b * c4[j] + a * sn;
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
                const u32x2_t w = {pack2bf(acc[db][4 * i] * scale, acc[db][4 * i + 1] * scale), pack2bf(acc[db][4 * i + 2] * scale, acc[db][4 * i + 3] * scale)};
                *reinterpret_cast<u32x2_t*>(row + db * 32 + 8 * i + 4 * e_h) = w;
            }
        }
    }
}

// dK = scale * sum_z part_k[z], dV = sum_z part_v[z]  -> bf16.  ROPE (head dim 128): a thread owns 4 features f < 64 of a head AND their rotate-half partners
// f + 64, and dK leaves rotated by the transposed rotary matrix of its slot (rounded to bf16 before and after, as the separate rope kernel did).
template <bool ROPE>
__global__ void attn_bwd_reduce_kernel(const float* __restrict__ part_k, const float* __restrict__ part_v, bf16_t* __restrict__ dK, int64_t dk_ld,
                                       bf16_t* __restrict__ dV, int64_t dv_ld, int n_slots, int kvd, int QS, float scale,
                                       const float* __restrict__ cosb, const float* __restrict__ sinb) {
    const int64_t stride = (int64_t)n_slots * kvd;
    if (ROPE) {
        const int64_t nch = (int64_t)n_slots * (kvd / 8);
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nch; i += (int64_t)gridDim.x * blockDim.x) {
            const int64_t slot = i / (kvd / 8); const int cc = (int)(i - slot * (kvd / 8));
            const int f = (cc & 15) * 4, c = (cc >> 4) * 128 + f;
            f32x4_t ka = {0.f, 0.f, 0.f, 0.f}, kb = ka, va = ka, vb = ka;
            for (int z = 0; z < QS; ++z) {
                const float* pk = part_k + z * stride + slot * kvd + c;
                const float* pv = part_v + z * stride + slot * kvd + c;
                ka += *reinterpret_cast<const f32x4_t*>(pk); kb += *reinterpret_cast<const f32x4_t*>(pk + 64);
                va += *reinterpret_cast<const f32x4_t*>(pv); vb += *reinterpret_cast<const f32x4_t*>(pv + 64);
            }
            const f32x4_t c4 = *reinterpret_cast<const f32x4_t*>(cosb + slot * 64 + f), s4 = *reinterpret_cast<const f32x4_t*>(sinb + slot * 64 + f);
            float oa[4], ob[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = bf2f(f2bf(ka[j] * scale)), b = bf2f(f2bf(kb[j] * scale));
                const float sn = -s4[j];
                oa[j] = a * c4[j] - b * sn; ob[j] = b * c4[j] + a * sn;
            }
            *reinterpret_cast<u32x2_t*>(dK + slot * dk_ld + c) = (u32x2_t){pack2bf(oa[0], oa[1]), pack2bf(oa[2], oa[3])};
            *reinterpret_cast<u32x2_t*>(dK + slot * dk_ld + c + 64) = (u32x2_t){pack2bf(ob[0], ob[1]), pack2bf(ob[2], ob[3])};
            *reinterpret_cast<u32x2_t*>(dV + slot * dv_ld + c) = (u32x2_t){pack2bf(va[0], va[1]), pack2bf(va[2], va[3])};
            *reinterpret_cast<u32x2_t*>(dV + slot * dv_ld + c + 64) = (u32x2_t){pack2bf(vb[0], vb[1]), pack2bf(vb[2], vb[3])};
        }
        return;
    }
    const int64_t nch = (int64_t)n_slots * (kvd / 4);
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

extern "C" int tr1_rope_apply(const void* in, int64_t ld_in, void* out, int64_t ld_out, const void* cosb, const void* sinb, int64_t T, int64_t n_heads,
                              int64_t head_dim, int backward, void* stream);

// wave pairs per block of the 32x32x16 kernel: 6 pairs = 12 waves = 3 per SIMD, 192 keys per block (4 pairs / 128 keys measured slower in round 3)
static int dkdv32_pairs() { return 6; }
static int dkdv_keys_per_block(int d_pad) {      // 8-wave blocks where 8*D/512 is integral; head dim 128: the 32x32x16 kernel's pairs x 32
    if (d_pad == 128) return dkdv32_pairs() * 32;
    return (d_pad == 64 || d_pad == 128) ? 128 : 64;
}

static int dkdv_qsplit(int64_t T, int group, int n_kv, int64_t n_slots, int kb) {
    const int64_t n_qtiles = (T * group + 63) / 64;
    const int64_t kvblocks = ((n_slots + kb - 1) / kb) * n_kv;
    int64_t qs = (1024 + kvblocks - 1) / kvblocks;
    if (qs > 8) qs = 8;
    if (qs > n_qtiles) qs = n_qtiles;
    if (qs < 1) qs = 1;
    while ((n_qtiles + qs - 1) / qs > DKDV32_MAXT) ++qs;
    return (int)qs;
}

// both head-dim-128 kernels on the 32x32x16 forms (32-bit DMA byte offsets fit): the dQ kernel's prologue then also does attn_delta_kernel's work (round 6)
static bool bwd32_both(const AttnParams& p) {
    return p.d_real == 128 && (uint64_t)p.n_slots * (uint64_t)(p.k_ld > p.v_ld ? p.k_ld : p.v_ld) * 2ull < 0xffffffffull &&
           (uint64_t)p.T * (uint64_t)(p.q_ld > p.do_ld ? p.q_ld : p.do_ld) * 2ull < 0xffffffffull;
}

template <int D>
static int launch_bwd(const AttnParams& p, hipStream_t s, float* ws, int64_t ws_floats, const float* lse2) {
    const int64_t nR = (int64_t)p.T * p.group;
    const int n_qtiles = (int)((nR + 63) / 64);
    constexpr int KSTR = 2 * D + 16;
    constexpr int NW = (D == 64 || D == 128) ? 8 : 4;
    constexpr int KB = NW * 16;
    const size_t dyn_dq = 2 * (2 * ATT_KV * KSTR) + 64;
    const size_t dyn_kv = 2 * (2 * 64 * KSTR + (NW == 8 ? 0 : 2 * D * 144) + 64 * 5 * 4 + 16) + (DKDV_MAXT + 1) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_dq);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_kernel<D, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_kv);
        attr_set = true;
    }
    // head dim 128: the 32x32x16-MFMA dQ kernel (round 3)
    const size_t dyn_dq32 = 4 * (2 * 64 * 256) + 256;
    static bool dq32_attr = false;
    if (D == 128 && !dq32_attr) { hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_dq32); dq32_attr = true; }
    const bool use_dq32 = D == 128 && p.d_real == 128 && lse2 != nullptr &&
                          (uint64_t)p.n_slots * (uint64_t)(p.k_ld > p.v_ld ? p.k_ld : p.v_ld) * 2ull < 0xffffffffull;      // 32-bit DMA byte offsets
    if (use_dq32) hipLaunchKernelGGL(attn_bwd_dq32_kernel, dim3((unsigned)((nR + 255) / 256), p.n_kv), dim3(512), dyn_dq32, s, p, lse2);
    else hipLaunchKernelGGL(attn_bwd_dq_kernel<D>, dim3((unsigned)((nR + 127) / 128), p.n_kv), dim3(256), dyn_dq, s, p);
    // head dim 128 beyond the 32-bit DMA offsets of the 32x32x16 kernel: the LDS-DMA staged 16x16x32 form, 8 waves x 16 keys
    constexpr int DMA_NB = 4;
    const size_t dyn_dma = DMA_NB * (2 * 64 * 256 + 64 * 5 * 4) + (2 * DKDV_MAXT + 2) * 4;
    static bool dma_attr = false;
    if (D == 128 && !dma_attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_dma_kernel<8, 1, DMA_NB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_dma);
        dma_attr = true;
    }
    // the 32x32x16-MFMA role-split kernel (round 3) for head dim 128
    constexpr int V32_NB = 3;
    const int v32_np = dkdv32_pairs();
    const size_t dyn_v32 = V32_NB * (2 * 64 * 256 + 64 * 5 * 4) + 2 * v32_np * (64 * 32 * 2) + (2 * DKDV32_MAXT + 2) * 4;
    static bool v32_attr = false;
    if (D == 128 && !v32_attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv32_kernel<V32_NB, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        v32_attr = true;
    }
    const bool use_v32 = D == 128 && p.d_real == 128 && lse2 != nullptr &&
                         (uint64_t)p.T * (uint64_t)(p.q_ld > p.do_ld ? p.q_ld : p.do_ld) * 2ull < 0xffffffffull;      // 32-bit DMA byte offsets
    const bool use_dma = D == 128 && p.d_real == 128 && lse2 != nullptr;
    const int QS = dkdv_qsplit(p.T, p.group, p.n_kv, p.n_slots, use_v32 ? v32_np * 32 : KB);
    const int64_t kvd = (int64_t)p.n_kv * p.d_real;
    float *pk = nullptr, *pv = nullptr;
    if (QS > 1) {
        const int64_t need = 2 * (int64_t)QS * p.n_slots * kvd;
        if (!ws || ws_floats < need) { tr1_set_error_("attention bwd: workspace too small"); return 1000; }
        pk = ws; pv = ws + (int64_t)QS * p.n_slots * kvd;
    }
    if (use_v32) hipLaunchKernelGGL((attn_bwd_dkdv32_kernel<V32_NB, 6>), dim3((unsigned)(p.n_kv * QS), 1, (unsigned)((p.n_slots + 191) / 192)), dim3(768), dyn_v32, s, p, n_qtiles, lse2, pk, pv);
    else if (use_dma) hipLaunchKernelGGL((attn_bwd_dkdv_dma_kernel<8, 1, DMA_NB>), dim3((unsigned)(p.n_kv * QS), 1, (unsigned)((p.n_slots + 127) / 128)), dim3(512), dyn_dma, s, p, n_qtiles, lse2, pk, pv);
    else hipLaunchKernelGGL((attn_bwd_dkdv_kernel<D, NW>), dim3((unsigned)(p.n_kv * QS), 1, (unsigned)((p.n_slots + KB - 1) / KB)), dim3(NW * 64), dyn_kv, s, p, n_qtiles, pk, pv);
    const bool rope = p.rope_cos != nullptr;
    bool dk_rotated = false;
    if (QS > 1) {
        const float scale = p.scale_log2 * 0.6931471805599453f;
        if (rope && p.d_real == 128) {
            hipLaunchKernelGGL(attn_bwd_reduce_kernel<true>, dim3(tr1_grid_1d(p.n_slots * kvd / 8, 256, 2048)), dim3(256), 0, s, pk, pv, p.dK, p.dk_ld, p.dV, p.dv_ld,
                               p.n_slots, (int)kvd, QS, scale, p.rope_cos, p.rope_sin);
            dk_rotated = true;
        } else
        hipLaunchKernelGGL(attn_bwd_reduce_kernel<false>, dim3(tr1_grid_1d(p.n_slots * kvd / 4, 256, 2048)), dim3(256), 0, s, pk, pv, p.dK, p.dk_ld, p.dV, p.dv_ld,
                           p.n_slots, (int)kvd, QS, scale, (const float*)nullptr, (const float*)nullptr);
    }
    if (rope) {      // forms without the fused rotation: the separate kernel, in place (a thread reads its pair before writing it)
        int rc = 0;
        if (!use_dq32) rc = tr1_rope_apply(p.dQ, p.dq_ld, p.dQ, p.dq_ld, p.rope_cos, p.rope_sin, p.T, (int64_t)p.n_kv * p.group, p.d_real, 1, s);
        if (!rc && !dk_rotated) rc = tr1_rope_apply(p.dK, p.dk_ld, p.dK, p.dk_ld, p.rope_cos, p.rope_sin, p.n_slots, p.n_kv, p.d_real, 1, s);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int64_t tr1_attn_bwd_workspace_floats(int64_t T, int64_t n_heads, int64_t n_kv, int64_t n_slots, int64_t head_dim) {
    if (n_kv <= 0 || T <= 0) return 0;
    const int d_pad = (int)((head_dim + 31) / 32 * 32);
    // the launch picks its key-block size (and with it the number of query slices) from the shape; size the partials for every form it may take
    int64_t need = 0;
    for (int kb : {dkdv_keys_per_block(d_pad), d_pad == 128 ? 128 : 64}) {
        const int QS = dkdv_qsplit(T, (int)(n_heads / n_kv), (int)n_kv, n_slots, kb);
        const int64_t n = QS > 1 ? 2 * (int64_t)QS * n_slots * n_kv * head_dim : 0;
        if (n > need) need = n;
    }
    return need;
}

// Scratch: qmeta_ws int32 [8*ceil(T*group/64)], delta fp32 [2*n_heads*T] (delta | log2-scaled LSE), ws_f32 of tr1_attn_bwd_workspace_floats() floats.
static int attn_bwd_impl(const void* Q, int64_t q_ld, const void* K, int64_t k_ld, const void* V, int64_t v_ld, const void* KT,
                            int64_t kt_ld, const void* QT, int64_t qt_ld, const void* dOT, int64_t dot_ld, const void* O, int64_t o_ld,
                            const void* dO, int64_t do_ld, const void* lse, void* delta, void* dQ, int64_t dq_ld, void* dK, int64_t dk_ld,
                            void* dV, int64_t dv_ld, const void* pre, const void* lo, const void* hi, void* qmeta_ws, void* ws_f32,
                            int64_t ws_floats, int64_t T, int64_t n_heads, int64_t n_kv, int64_t n_slots, int64_t head_dim, float scale,
                            const void* rope_cos, const void* rope_sin, void* stream) {
    AttnParams p; memset(&p, 0, sizeof(p));
    p.rope_cos = (const float*)rope_cos; p.rope_sin = (const float*)rope_sin;
    TR1_CHECK_ARG(!rope_cos || (rope_sin && n_slots == T && head_dim % 16 == 0), "attention bwd: rope tables need sin, n_slots == T and head_dim % 16 == 0");
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
    static int fuse_delta = -1;                                                  // TR1_BWD_FUSE_DELTA=0: the separate attn_delta_kernel launch (A/B measurements)
    if (fuse_delta < 0) { const char* e = getenv("TR1_BWD_FUSE_DELTA"); fuse_delta = e ? atoi(e) : 1; }
    if (fuse_delta && d_pad == 128 && bwd32_both(p)) {
        p.O = (bf16_t*)const_cast<void*>(O); p.o_ld = o_ld; p.lse2_out = lse2; p.qmeta_out = (int*)qmeta_ws;      // delta, lse2 and qmeta come out of the dQ kernel's prologue
    } else
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

extern "C" int tr1_attn_bwd(const void* Q, int64_t q_ld, const void* K, int64_t k_ld, const void* V, int64_t v_ld, const void* KT,
                            int64_t kt_ld, const void* QT, int64_t qt_ld, const void* dOT, int64_t dot_ld, const void* O, int64_t o_ld,
                            const void* dO, int64_t do_ld, const void* lse, void* delta, void* dQ, int64_t dq_ld, void* dK, int64_t dk_ld,
                            void* dV, int64_t dv_ld, const void* pre, const void* lo, const void* hi, void* qmeta_ws, void* ws_f32,
                            int64_t ws_floats, int64_t T, int64_t n_heads, int64_t n_kv, int64_t n_slots, int64_t head_dim, float scale,
                            void* stream) {
    return attn_bwd_impl(Q, q_ld, K, k_ld, V, v_ld, KT, kt_ld, QT, qt_ld, dOT, dot_ld, O, o_ld, dO, do_ld, lse, delta, dQ, dq_ld, dK, dk_ld, dV, dv_ld, pre, lo,
                         hi, qmeta_ws, ws_f32, ws_floats, T, n_heads, n_kv, n_slots, head_dim, scale, nullptr, nullptr, stream);
}

// The same backward with the M-RoPE backward folded in: dQ and dK are returned with respect to the UN-rotated q / k (what the q|k|v projection produced),
// i.e. multiplied by the transposed rotary matrix of their row (cos / sin fp32 [T, head_dim / 2]; n_slots == T).  Head dim 128 rotates in the dQ kernel's
// epilogue and in the partial-sum kernel of dK / dV; other shapes run the rotation kernel in place.  Bit-identical to tr1_attn_bwd + tr1_rope_apply(backward).
extern "C" int tr1_attn_bwd_rope(const void* Q, int64_t q_ld, const void* K, int64_t k_ld, const void* V, int64_t v_ld, const void* KT,
                                 int64_t kt_ld, const void* QT, int64_t qt_ld, const void* dOT, int64_t dot_ld, const void* O, int64_t o_ld,
                                 const void* dO, int64_t do_ld, const void* lse, void* delta, void* dQ, int64_t dq_ld, void* dK, int64_t dk_ld,
                                 void* dV, int64_t dv_ld, const void* pre, const void* lo, const void* hi, void* qmeta_ws, void* ws_f32,
                                 int64_t ws_floats, int64_t T, int64_t n_heads, int64_t n_kv, int64_t n_slots, int64_t head_dim, float scale,
                                 const void* rope_cos, const void* rope_sin, void* stream) {
    TR1_CHECK_ARG(rope_cos && rope_sin, "attention bwd (rope): cos / sin tables required");
    return attn_bwd_impl(Q, q_ld, K, k_ld, V, v_ld, KT, kt_ld, QT, qt_ld, dOT, dot_ld, O, o_ld, dO, do_ld, lse, delta, dQ, dq_ld, dK, dk_ld, dV, dv_ld, pre, lo,
                         hi, qmeta_ws, ws_f32, ws_floats, T, n_heads, n_kv, n_slots, head_dim, scale, rope_cos, rope_sin, stream);
}
