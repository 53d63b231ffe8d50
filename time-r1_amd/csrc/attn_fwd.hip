// Attention forward (flash style, online softmax) + split-KV combine + layout helpers. See attn_common.h for the design.
#include "attn_common.h"

// Block: 256 threads = 4 waves; wave owns 16*CB packed query rows (CB 16-column blocks); block = 64*CB packed rows.
// grid = (ceil(T*group/(64*CB)), n_kv, nsplit).  K / V^T tiles are double-buffered in LDS and staged through registers: the global
// loads for tile i+1+PF are issued before tile i+1 is computed (guide: async-STAGE split), one barrier per tile.
//   CB = 2, PF = 1: training / prefill (MFMA-bound, 256 VGPRs at D = 128).
//   CB = 1, PF = 3: split-KV decode.  A decode block sees only 2-3 tiles, so its time is a chain of memory latencies: with PF = 3
//                   register sets all of its tiles are in flight at once (one latency instead of three), and 16 rows per wave keep
//                   all four waves busy for the 56 packed rows of G = 8 x group 7.
// optional block-timeline probe (tools/probe_attn.hip, -DTR1_PROBE); not compiled into the library
#ifdef TR1_PROBE
__device__ unsigned long long* tr1_probe = nullptr;
#define TR1_PROBE_AT(slot) do { if (tr1_probe && threadIdx.x == 0) tr1_probe[(size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define TR1_PROBE_AT(slot) do { } while (0)
#endif

// Split-KV decode: list of the tiles at least one row of the block can see.  The block's tile RANGE is one interval from the first group's
// suffix start to the last group's newest slot; with the cache laid out group by group (slot = P + g*C + s) that interval also spans the
// NOT YET GENERATED slots of every group but the last - at C = 1024 a 64-row block walked 144 suffix tiles at every decode step, 8 of them
// useful on average (config 4: 35 us per layer, constant over the rollout).  The list keeps the prefix tiles and, of the suffix range, the
// tiles that intersect some token's [lo, hi]; one wave evaluates 64 candidate tiles per pass against the <= 11 tokens of the block.
#define ATT_LIST_CAP 1024
template <int D, int CB, int PF>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnParams p) {
    TR1_PROBE_AT(0);
    constexpr int KSTR = 2 * D + 16;
    // logical block coordinates: the 3-D grid itself for split-KV decode, the XCD-contiguous re-mapping of a 1-D grid otherwise
    int bx = blockIdx.x, by_ = blockIdx.y, gx = gridDim.x, gy = gridDim.y;
    if (PF == 1 && p.xcd_pad > 0) {
        // chunks of 8 consecutive logical blocks (8 query tiles of one head: about one ViT segment, or neighbouring causal tiles of similar
        // weight) are dealt round-robin to the XCDs: sharing inside a chunk, balance across XCDs
        const int L = (int)blockIdx.x, xcd = L & 7, slot = L >> 3;
        const int logical = (((slot >> 3) << 3) + xcd) * 8 + (slot & 7);
        if (logical >= p.grid_x * p.grid_y) return;                         // padding block (the grid is rounded up to a multiple of 8)
        gx = p.grid_x; gy = p.grid_y; by_ = logical / gx; bx = logical - by_ * gx;
    }
    const int bidx = by_ / p.n_kv;
    {   // batched launch (decode over several prompts' caches): blockIdx.y = b * n_kv + kvh
        const int b = bidx;
        p.Q += (int64_t)b * p.T * p.q_ld; p.O += (int64_t)b * p.T * p.o_ld;
        p.pre += (int64_t)b * p.T; p.lo += (int64_t)b * p.T; p.hi += (int64_t)b * p.T;
        p.K += (int64_t)b * p.kv_batch_slots * p.k_ld; p.VT += (int64_t)b * p.kv_batch_slots;
        if (p.lse) p.lse += (int64_t)b * p.n_kv * p.group * p.T;
    }
    constexpr int KBYTES = ATT_KV * KSTR, VBYTES = D * 144, BUF = KBYTES + VBYTES;
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];      // [2][K tile | V^T tile] + meta
    int* lds_meta = reinterpret_cast<int*>(dyn_lds + 2 * BUF);          // [4][3]
    int* lds_list = lds_meta + 16;                                      // split-KV decode: [ATT_LIST_CAP] relevant tile ids + their count
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, u = lane & 15, g = lane >> 4;
    const int kvh = by_ % p.n_kv, split = blockIdx.z;
    const int64_t nR = (int64_t)p.T * p.group;
    // query tiles are walked from the LAST one down: later rows of the packed sequence see more keys (causal prompt rows, then the completion
    // rows with the whole prefix), so the heaviest blocks are dispatched first and the light ones fill the tail (longest-processing-time order)
    const int qtile = (int)(gx - 1 - bx);
    const int64_t R0 = (int64_t)qtile * (64 * CB) + wave * (16 * CB);
    // plan_mode 2: this block's tile list was stored by the first layer's launch of the same decode step.  The count is a scalar load; the
    // block's own entries (list[split + t * nsplit], t = thread) are requested at once, clamped into the plan - entries beyond the count are never used.
    int* plan_blk = (PF > 1 && p.plan) ? p.plan + ((int64_t)bidx * gx + qtile) * (ATT_LIST_CAP + 1) : nullptr;
    int plan_n = -1, plan_mine = 0;
    if (PF > 1 && p.plan_mode == 2) {
        plan_n = plan_blk[ATT_LIST_CAP];
        const int idx = (int)blockIdx.z + (int)threadIdx.x * p.nsplit;
        plan_mine = plan_blk[idx < ATT_LIST_CAP ? idx : ATT_LIST_CAP - 1];
    }
    const bool planned = plan_n >= 0;

    // Split-KV decode (PF > 1): the block's first tile is almost always tile `split` of the shared prefix.  Its K / V^T loads are issued
    // BEFORE the row masks are fetched and reduced (a dependent global round trip + a barrier, 1.5 us of a 10 us block); once the tile
    // range is known the speculation is checked and, if wrong (prompt rows, very short prefixes), the tile is simply loaded again.
    TileRegs<D> rg[PF];
    const int64_t spec_kv0 = (int64_t)blockIdx.z * ATT_KV;
    const bool spec = PF > 1 && spec_kv0 < p.n_slots;
    if (spec) tile_load_regs<D>(rg[0], p.K, p.k_ld, p.VT, p.vt_ld, kvh, spec_kv0, p.n_slots, p.d_real);

    int tq[CB], hq[CB], pre[CB], lo[CB], hi[CB]; bool valid[CB];
    int wmaxpre = 0, wminlo = 0x7fffffff, wmaxhi = -1;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const int64_t R = R0 + cb * 16 + u;
        valid[cb] = R < nR;
        const int64_t Rc = valid[cb] ? R : nR - 1;
        att_split_row(p, Rc, tq[cb], hq[cb]);
        pre[cb] = valid[cb] ? p.pre[tq[cb]] : 0;
        lo[cb] = valid[cb] ? p.lo[tq[cb]] : 1;
        hi[cb] = valid[cb] ? p.hi[tq[cb]] : 0;
    }
    // Q fragments (B operand): Q[q = u][d = ks*32 + g*8 .. +8].  q was written by the previous kernel (an L2 miss, ~2 us): requested here, in front
    // of the mask reduction / plan hand-off, so that it has landed when the first tile is staged (the compiler waits for ALL outstanding loads
    // there).  (Round 1, without the speculated first tile: issuing them early measured slower, 3.95 vs 3.86 ms per decode step.)
    bf16x8_t qf[CB][D / 32];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const bf16_t* qrow = p.Q + (int64_t)tq[cb] * p.q_ld + (int64_t)(kvh * p.group + hq[cb]) * p.d_real;
#pragma unroll
        for (int ks = 0; ks < D / 32; ++ks) qf[cb][ks] = load_row_frag(qrow, ks * 32 + g * 8, p.d_real, valid[cb]);
    }
    TileRange tr{0, 0, 0};
    if (!planned) {                                                         // block-uniform
        // (the first USE of the mask values: with a plan they are not touched before the tile loop, so the row masks, the speculated tile, the
        //  plan entries and the Q fragments are ONE memory round trip instead of two)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
            if (valid[cb]) { wmaxpre = max(wmaxpre, pre[cb]); if (hi[cb] >= lo[cb]) { wminlo = min(wminlo, lo[cb]); wmaxhi = max(wmaxhi, hi[cb]); } }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            wmaxpre = max(wmaxpre, __shfl_xor(wmaxpre, o, 64)); wminlo = min(wminlo, __shfl_xor(wminlo, o, 64)); wmaxhi = max(wmaxhi, __shfl_xor(wmaxhi, o, 64));
        }
        if (lane == 0) { lds_meta[wave * 3 + 0] = wmaxpre; lds_meta[wave * 3 + 1] = wminlo; lds_meta[wave * 3 + 2] = wmaxhi; }
        __syncthreads();
        int bmaxpre = 0, bminlo = 0x7fffffff, bmaxhi = -1;
#pragma unroll
        for (int w = 0; w < 4; ++w) { bmaxpre = max(bmaxpre, lds_meta[w * 3]); bminlo = min(bminlo, lds_meta[w * 3 + 1]); bmaxhi = max(bmaxhi, lds_meta[w * 3 + 2]); }
        tr = att_tile_range(bmaxpre, bminlo, bmaxhi, p.n_slots);
    }
    TR1_PROBE_AT(1);
    // a wave whose 32 rows are all out of range (decode: 56 live rows in a 128-row tile) only helps staging
    const bool wave_active = __builtin_amdgcn_readfirstlane((int)(R0 < nR)) != 0;



    f32x4_t o[D / 16][CB];
    float m[CB], l[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        m[cb] = NEG_INF; l[cb] = 0.f;
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) o[dt][cb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }

    const bool use_list = !planned && PF > 1 && tr.n_rel <= ATT_LIST_CAP;
    int n_rel = planned ? plan_n : tr.n_rel;
    if (planned) {      // own entries, compacted: lds_list[t] = list[split + t * nsplit]
        const int n_mine = (split < n_rel) ? (n_rel - split + p.nsplit - 1) / p.nsplit : 0;
        if ((int)threadIdx.x < n_mine) lds_list[threadIdx.x] = plan_mine;
        __syncthreads();
    }
    if (use_list) {
        if (wave == 0) {
            const int64_t Rb = (int64_t)qtile * (64 * CB);
            const int64_t Re = (Rb + 64 * CB - 1 < nR - 1) ? Rb + 64 * CB - 1 : nR - 1;
            int t0, t1, hq_;
            att_split_row(p, Rb, t0, hq_); att_split_row(p, Re, t1, hq_);
            const int ntok = t1 - t0 + 1;                                   // <= 64*CB / group + 1 tokens
            int mylo = 1, myhi = 0;
            if (lane < ntok && lane < 64) { mylo = p.lo[t0 + lane]; myhi = p.hi[t0 + lane]; }
            int count = 0;
            for (int base = 0; base < tr.n_rel; base += 64) {
                const int i = base + lane;
                const bool in = i < tr.n_rel;
                const int tile = att_tile_at(tr, in ? i : 0);
                const int kv0 = tile * ATT_KV;
                bool rel = i < tr.pre_tiles;
                for (int k = 0; k < ntok; ++k) {                            // every lane takes part in the shuffles (no divergence around them)
                    const int lk = __shfl(mylo, k, 64), hk = __shfl(myhi, k, 64);
                    rel = rel | ((lk <= kv0 + ATT_KV - 1) & (hk >= kv0));
                }
                rel = rel & in;
                const unsigned long long mask = __ballot(rel);
                if (rel) lds_list[count + __popcll(mask & ((1ull << lane) - 1ull))] = tile;
                count += __popcll(mask);
            }
            if (lane == 0) lds_list[ATT_LIST_CAP] = count;
        }
        __syncthreads();
        n_rel = lds_list[ATT_LIST_CAP];
        if (p.plan_mode == 1 && kvh == 0 && split == 0) {                   // one block per (batch entry, query tile) publishes the list
            for (int i = threadIdx.x; i < n_rel; i += 256) plan_blk[i] = lds_list[i];
            if (threadIdx.x == 0) plan_blk[ATT_LIST_CAP] = (n_rel <= 256 * p.nsplit) ? n_rel : -1;   // a reader block holds one entry per thread
        }
    } else if (PF > 1 && !planned && p.plan_mode == 1 && kvh == 0 && split == 0 && threadIdx.x == 0) {
        plan_blk[ATT_LIST_CAP] = -1;                                        // no list (range too long): the other layers take the full path as well
    }
    const int n_my = (split < n_rel) ? (n_rel - split + p.nsplit - 1) / p.nsplit : 0;
#define TILE_KV0(i) ((int64_t)(planned ? lds_list[i] : use_list ? lds_list[split + (i) * p.nsplit] : att_tile_at(tr, split + (i) * p.nsplit)) * ATT_KV)
    // Split-KV decode with the speculated first tile already in flight: stage it into LDS BEFORE the other tiles are requested.  The tiles
    // travel through registers, so the compiler places the s_waitcnt, and with loads on conditional paths it waits for EVERYTHING outstanding
    // (vmcnt(0)) in front of a tile_store_lds: requested first, tiles 1 and 2 would hold back tile 0 by a full memory latency.
    const bool spec_hit = PF > 1 && spec && n_my > 0 && TILE_KV0(0) == spec_kv0;        // block-uniform
    if (spec_hit) tile_store_lds<D>(rg[0], dyn_lds, dyn_lds + KBYTES, TILE_KV0(0), p.n_slots, p.d_real);
#pragma unroll
    for (int j = 0; j < PF; ++j)
        if (j < n_my && !(j == 0 && spec_hit))
            tile_load_regs<D>(rg[j], p.K, p.k_ld, p.VT, p.vt_ld, kvh, TILE_KV0(j), p.n_slots, p.d_real);
    if (n_my > 0) {
        if (!spec_hit) tile_store_lds<D>(rg[0], dyn_lds, dyn_lds + KBYTES, TILE_KV0(0), p.n_slots, p.d_real);
        if (PF < n_my) tile_load_regs<D>(rg[0], p.K, p.k_ld, p.VT, p.vt_ld, kvh, TILE_KV0(PF), p.n_slots, p.d_real);
    }
    __syncthreads();
    TR1_PROBE_AT(2);

    for (int it0 = 0; it0 < n_my; it0 += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        const int it = it0 + j;
        if (it >= n_my) break;
        const int kv0 = (int)TILE_KV0(it);
        const char* lds_k = dyn_lds + (it & 1) * BUF;
        const char* lds_vt = lds_k + KBYTES;
        if (wave_active) {
            f32x4_t s[4][CB];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) s[kt][cb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < D / 32; ++ks) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(lds_k + (kt * 16 + u) * KSTR + (ks * 4 + g) * 16);
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) s[kt][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[cb][ks], s[kt][cb], 0, 0, 0);
                }
            }
            if (it == 0) TR1_PROBE_AT(5);     // QK^T of the first tile issued
            bf16x8_t pf[2][CB];
            // Tiles that every row of this wave sees completely (all shared-prefix tiles of completion rows, everything below the
            // diagonal of prompt rows) skip the per-element visibility test - a wave-uniform branch.
            bool full = kv0 + ATT_KV <= p.n_slots;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
                full = full & (!valid[cb] | (kv0 + ATT_KV - 1 < pre[cb]) | ((kv0 >= lo[cb]) & (kv0 + ATT_KV - 1 <= hi[cb])));   // bitwise: no exec-mask chains
            const bool wave_full = __all(full);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                float mx = NEG_INF;
                // max over RAW scores (scale > 0 commutes with max); the scale is folded into the exp2 argument as one fma
                if (wave_full) {
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][cb][r]);
                } else {
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int kv = kv0 + kt * 16 + g * 4 + r;
                            const bool ok = (kv < p.n_slots) & att_visible_nb(kv, pre[cb], lo[cb], hi[cb]);   // bitwise: no control flow per element
                            const float v = ok ? s[kt][cb][r] : NEG_INF;
                            s[kt][cb][r] = v; mx = fmaxf(mx, v);
                        }
                }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float m_new = fmaxf(m[cb], mx * p.scale_log2);
                const float m_safe = (m_new == NEG_INF) ? 0.f : m_new;
                const float alpha = __builtin_amdgcn_exp2f(m[cb] - m_safe);
                float rs = 0.f;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][cb][r], p.scale_log2, -m_safe));
                        s[kt][cb][r] = e; rs += e;
                    }
                rs += __shfl_xor(rs, 16, 64); rs += __shfl_xor(rs, 32, 64);
                l[cb] = l[cb] * alpha + rs; m[cb] = m_new;
                if (!__all(alpha == 1.0f)) {   // exact: the running maximum did not move for any row of the wave -> no rescale needed
#pragma unroll
                    for (int dt = 0; dt < D / 16; ++dt) o[dt][cb] *= alpha;
                }
                pf[0][cb] = pack_frag(s[0][cb], s[1][cb]);
                pf[1][cb] = pack_frag(s[2][cb], s[3][cb]);
            }
            if (it == 0) TR1_PROBE_AT(6);     // softmax of the first tile done
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const char* base = lds_vt + (dt * 16 + u) * 144 + kk * 64 + g * 8;
                    const bf16x8_t vf = make_frag(*reinterpret_cast<const u32x2_t*>(base), *reinterpret_cast<const u32x2_t*>(base + 32));
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) o[dt][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[kk][cb], o[dt][cb], 0, 0, 0);
                }
            }
        }
        if (it == 0) TR1_PROBE_AT(7);         // PV of the first tile issued
        if (it + 1 < n_my) {
            char* nk = dyn_lds + ((it + 1) & 1) * BUF;
            tile_store_lds<D>(rg[(j + 1) % PF], nk, nk + KBYTES, TILE_KV0(it + 1), p.n_slots, p.d_real);
            if (it + 1 + PF < n_my)
                tile_load_regs<D>(rg[(j + 1) % PF], p.K, p.k_ld, p.VT, p.vt_ld, kvh, TILE_KV0(it + 1 + PF), p.n_slots, p.d_real);
        }
        __syncthreads();
    }
    }
#undef TILE_KV0

    TR1_PROBE_AT(3);
    // ---- epilogue. Lane holds O^T[d = dt*16 + g*4 + r][q = u].
    if (p.nsplit == 1) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            if (!valid[cb]) continue;
            const float inv = l[cb] > 0.f ? 1.f / l[cb] : 0.f;
            bf16_t* orow = p.O + (int64_t)tq[cb] * p.o_ld + (int64_t)(kvh * p.group + hq[cb]) * p.d_real;
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) {
                const int d = dt * 16 + g * 4;
                if (d < p.d_real) {
                    u32x2_t w = {pack2bf(o[dt][cb][0] * inv, o[dt][cb][1] * inv), pack2bf(o[dt][cb][2] * inv, o[dt][cb][3] * inv)};
                    *reinterpret_cast<u32x2_t*>(orow + d) = w;
                }
            }
            if (g == 0 && p.lse)
                p.lse[(int64_t)(kvh * p.group + hq[cb]) * p.T + tq[cb]] = l[cb] > 0.f ? (m[cb] + log2f(l[cb])) * 0.6931471805599453f : NEG_INF;
        }
    } else {
        const int64_t nRpad = (int64_t)gx * (64 * CB);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            if (!valid[cb]) continue;
            const int64_t R = R0 + cb * 16 + u;
            const int64_t slot = ((int64_t)split * gy + by_) * nRpad + R;
            float* op = p.Opart + slot * D;
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) *reinterpret_cast<f32x4_t*>(op + dt * 16 + g * 4) = o[dt][cb];
            if (g == 0) { p.mpart[slot] = m[cb]; p.lpart[slot] = l[cb]; }
        }
    }
    TR1_PROBE_AT(4);
    // (An in-kernel merge of the split-KV partials by the last-arriving block was tried and dropped: on ONE CU the merge is a chain of
    //  ~nsplit dependent load rounds, 38-80 us against 7.5 us for the separate 56-block combine launch below.)
}

// Merge split-KV partials. Block = 256 threads = 2 packed rows x 4 split groups x 32 lanes; each lane owns 4 features (D <= 128).
// The merge is a latency chain, not a bandwidth problem (7 MB at 7B): the splits of a row are spread over 4 lane groups and every
// lane issues all of its (<= 16) partial-row loads up front - clamped address, zero weight beyond nsplit - instead of walking the
// splits four at a time (7.5 us per layer before).  The (m, l) statistics of the block's rows are staged in LDS first.
#define ATT_COMBINE_ROWS 2
template <int D>
__global__ __launch_bounds__(256) void attn_combine_kernel(AttnParams p, int64_t nRpad) {
    __shared__ float sm_m[ATT_COMBINE_ROWS][64], sm_l[ATT_COMBINE_ROWS][64];
    __shared__ __attribute__((aligned(16))) f32x4_t part[ATT_COMBINE_ROWS][4][32];
    const int64_t nR = (int64_t)p.T * p.group;
    const int kvh = blockIdx.y % p.n_kv, by = blockIdx.y, nby = gridDim.y;
    {
        const int b = blockIdx.y / p.n_kv;
        p.O += (int64_t)b * p.T * p.o_ld;
        if (p.lse) p.lse += (int64_t)b * p.n_kv * p.group * p.T;
    }
    const int c = threadIdx.x & 31, sg = (threadIdx.x >> 5) & 3, rl = threadIdx.x >> 7;
    const int64_t Rb = (int64_t)blockIdx.x * ATT_COMBINE_ROWS;
    const int64_t R = Rb + rl;
    const bool act = R < nR;
    const int64_t Rc = act ? R : nR - 1;
    const int d = c * 4;
    // partial rows first (independent of the statistics): splits sg, sg + 4, ...
    f32x4_t ov[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int sp = sg + 4 * i;
        ov[i] = *reinterpret_cast<const f32x4_t*>(p.Opart + (((int64_t)(sp < p.nsplit ? sp : 0) * nby + by) * nRpad + Rc) * D + (d < D ? d : 0));
    }
    for (int i = threadIdx.x; i < ATT_COMBINE_ROWS * p.nsplit; i += 256) {
        const int r = i / p.nsplit, sp = i - r * p.nsplit;
        const int64_t Rr = Rb + r;
        const int64_t slot = ((int64_t)sp * nby + by) * nRpad + Rr;
        sm_m[r][sp] = (Rr < nR) ? p.mpart[slot] : NEG_INF;
        sm_l[r][sp] = (Rr < nR) ? p.lpart[slot] : 0.f;
    }
    __syncthreads();
    float M, Ms, L;
    att_merge_stats(sm_m[rl], sm_l[rl], p.nsplit, M, Ms, L);
    const float inv = L > 0.f ? 1.f / L : 0.f;
    f32x4_t acc = att_merge_weighted(ov, sm_m[rl], Ms, sg, p.nsplit);
    part[rl][sg][c] = acc;
    __syncthreads();
    if (!act || sg != 0) return;
    acc = (part[rl][0][c] + part[rl][1][c]) + (part[rl][2][c] + part[rl][3][c]);
    int t, hq;
    att_split_row(p, R, t, hq);
    if (d < p.d_real) {
        bf16_t* orow = p.O + (int64_t)t * p.o_ld + (int64_t)(kvh * p.group + hq) * p.d_real;
        u32x2_t w = {pack2bf(acc[0] * inv, acc[1] * inv), pack2bf(acc[2] * inv, acc[3] * inv)};
        *reinterpret_cast<u32x2_t*>(orow + d) = w;
    }
    if (c == 0 && p.lse) p.lse[(int64_t)(kvh * p.group + hq) * p.T + t] = L > 0.f ? (M + log2f(L)) * 0.6931471805599453f : NEG_INF;
}

// Head dim 128 (round 5): ONE WAVE per packed row.  attn_combine_kernel spends a quarter of its 6 us in per-thread scalar loops (every thread walks all
// nsplit statistics twice: ~600 dependent instructions on a one-wave-per-SIMD block); here lane j holds split j's (m, l), the maximum and the sum are two
// wave reductions, a lane group sg = lane / 16 takes splits sg, sg + 4, ... with 8 features per lane (two 16-byte loads per split, all in flight), the weights
// arrive by lane shuffles and the four groups meet through two more: no LDS, no barrier, ~150 instructions.  (Sums are trees instead of chains: same fp32
// arithmetic, different association - the split-KV result is compared with the oracle, not bit for bit with the old merge.)
__global__ __launch_bounds__(128) void attn_combine128_kernel(AttnParams p, int64_t nRpad) {
    const int64_t nR = (int64_t)p.T * p.group;
    const int kvh = blockIdx.y % p.n_kv, by = blockIdx.y, nby = gridDim.y;
    {
        const int b = blockIdx.y / p.n_kv;
        if (!p.o_frag) p.O += (int64_t)b * p.T * p.o_ld;
        if (p.lse) p.lse += (int64_t)b * p.n_kv * p.group * p.T;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, sg = lane >> 4, c = lane & 15;
    const int64_t R = (int64_t)blockIdx.x * 2 + wv;
    if (R >= nR) return;                                              // (wave-uniform)
    const int ni = (p.nsplit + 3) >> 2;                               // splits per lane group (the last ones may not exist for the higher groups)
    f32x4_t ov[16][2];
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < ni) {
            const int sp = sg + 4 * i;
            const float* src = p.Opart + (((int64_t)(sp < p.nsplit ? sp : 0) * nby + by) * nRpad + R) * 128 + c * 8;
            ov[i][0] = *reinterpret_cast<const f32x4_t*>(src); ov[i][1] = *reinterpret_cast<const f32x4_t*>(src + 4);
        }
    const int64_t sslot = ((int64_t)(lane < p.nsplit ? lane : 0) * nby + by) * nRpad + R;
    const float mj = lane < p.nsplit ? p.mpart[sslot] : NEG_INF, lj = lane < p.nsplit ? p.lpart[sslot] : 0.f;
    const float M = wave_max(mj), Ms = (M == NEG_INF) ? 0.f : M;
    const float e = exp2f(mj - Ms);                                   // (0 for the lanes without a split: their m is -inf)
    const float L = wave_sum(e * lj);
    const float inv = L > 0.f ? 1.f / L : 0.f;
    f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < ni) {
            const int sp = sg + 4 * i;
            float w = __shfl(e, sp & 63, 64);
            if (sp >= p.nsplit) w = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { a0[j] = __builtin_fmaf(w, ov[i][0][j], a0[j]); a1[j] = __builtin_fmaf(w, ov[i][1][j], a1[j]); }
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a0[j] = tr1_sum_xor16(a0[j]); a1[j] = tr1_sum_xor16(a1[j]);      // (vector-unit exchanges, same partners and order as the shuffles they replace: tr1_common.h)
        a0[j] = tr1_sum_xor32(a0[j]); a1[j] = tr1_sum_xor32(a1[j]);
    }
    if (sg != 0) return;
    int t, hq;
    att_split_row(p, R, t, hq);
    bf16_t* orow = p.O + (int64_t)t * p.o_ld + (int64_t)(kvh * p.group + hq) * 128 + c * 8;
    if (p.o_frag) {                                                   // fragment-major for the o projection (AttnParams::o_frag)
        const int m = (int)(blockIdx.y / p.n_kv) * p.T + t, kf = (kvh * p.group + hq) * 128 + c * 8;
        orow = p.O + ((((int64_t)(m >> 4) * (p.n_kv * p.group * 4) + (kf >> 5)) * 16 + (m & 15)) * 32 + (kf & 31));
    }
    const u32x4_t w = {pack2bf(a0[0] * inv, a0[1] * inv), pack2bf(a0[2] * inv, a0[3] * inv), pack2bf(a1[0] * inv, a1[1] * inv), pack2bf(a1[2] * inv, a1[3] * inv)};
    *reinterpret_cast<u32x4_t*>(orow) = w;
    if (c == 0 && p.lse) p.lse[(int64_t)(kvh * p.group + hq) * p.T + t] = L > 0.f ? (M + log2f(L)) * 0.6931471805599453f : NEG_INF;
}

// out[(kvh*d + dd) * ld_out + t*group + hq] = in[t*ld_in + (kvh*group + hq)*d + dd] ; columns [T*group, ld_out) zero-filled
// when zero_pad != 0. With group = 1 this is the K^T / V^T builder (optionally through a slot map: column = slots[t]).
__global__ __launch_bounds__(256) void pack_transpose_kernel(const bf16_t* __restrict__ in, int64_t ld_in, bf16_t* __restrict__ out,
                                                             int64_t ld_out, const int* __restrict__ slots, int64_t T, int n_kv, int group,
                                                             int d, int64_t zero_cols) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64][72];       // [packed row][feature], 144-byte rows
    const int kvh = blockIdx.z;
    const int64_t nR = T * group;
    const int64_t R0 = (int64_t)blockIdx.y * 64; const int d0 = blockIdx.x * 64;
    // 16-byte accesses on both sides (the scalar form ran at ~1 TB/s): 8 features per lane in, 8 packed rows per lane out
    const bool vec_in = (ld_in % 8 == 0) && (d % 8 == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = threadIdx.x + i * 256;
        const int rr = idx >> 3, cc = (idx & 7) * 8;
        const int64_t R = R0 + rr; const int dd = d0 + cc;
        u32x4_t v = {0, 0, 0, 0};
        if (R < nR && dd < d) {
            const int64_t t = (unsigned)R / (unsigned)group; const int hq = (int)(R - t * group);      // 32-bit division (R < 2^31)
            const bf16_t* src = in + t * ld_in + (int64_t)(kvh * group + hq) * d + dd;
            if (vec_in && dd + 8 <= d) v = *reinterpret_cast<const u32x4_t*>(src);
            else {
                bf16_t e8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) e8[e] = (dd + e < d) ? src[e] : (bf16_t)0;
                v = (u32x4_t){e8[0] | ((unsigned)e8[1] << 16), e8[2] | ((unsigned)e8[3] << 16), e8[4] | ((unsigned)e8[5] << 16), e8[6] | ((unsigned)e8[7] << 16)};
            }
        }
        *reinterpret_cast<u32x4_t*>(&tile[rr][cc]) = v;
    }
    __syncthreads();
    const int64_t limit = slots ? nR : (zero_cols > nR ? zero_cols : nR);      // rows >= nR of the tile are zero: they are the padding columns
    const bool vec_out = !slots && (ld_out % 8 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = threadIdx.x + i * 256;
        const int cc = idx & 63, rr = (idx >> 6) * 8;
        const int dd = d0 + cc; const int64_t R = R0 + rr;
        if (dd >= d || R >= limit) continue;
        bf16_t e8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) e8[e] = tile[rr + e][cc];
        bf16_t* orow = out + ((int64_t)kvh * d + dd) * ld_out;
        if (vec_out && R + 8 <= limit) {
            const u32x4_t v = {e8[0] | ((unsigned)e8[1] << 16), e8[2] | ((unsigned)e8[3] << 16), e8[4] | ((unsigned)e8[5] << 16), e8[6] | ((unsigned)e8[7] << 16)};
            *reinterpret_cast<u32x4_t*>(orow + R) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (R + e < limit) orow[slots ? (int64_t)slots[R + e] : R + e] = e8[e];
        }
    }
}

// rows of `src` [T, cols] -> dst[slots[t], :]   (KV-cache append; 16 bytes per lane)
__global__ void scatter_slots_kernel(const bf16_t* __restrict__ src, int64_t ld_src, bf16_t* __restrict__ dst, int64_t ld_dst,
                                     const int* __restrict__ slots, int64_t T, int cols) {
    const int nch = cols >> 3;
    const int64_t total = T * nch;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / nch; const int c = (int)(i - t * nch);
        *reinterpret_cast<u32x4_t*>(dst + (int64_t)slots[t] * ld_dst + c * 8) = *reinterpret_cast<const u32x4_t*>(src + t * ld_src + c * 8);
    }
}

int tr1_launch_attn_dec32(AttnParams& p, dim3 grid, hipStream_t s);      // attn_fwd32.hip: 0 not launched, 1 launched (partials: combine follows)

template <int D, int CB, int PF>
static void launch_fwd(dim3 grid, hipStream_t s, const AttnParams& p) {
    const size_t dyn = 2 * (ATT_KV * (2 * D + 16) + D * 144) + 64 + (PF > 1 ? (ATT_LIST_CAP + 1) * 4 : 0);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<D, CB, PF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        attr_set = true;
    }
    hipLaunchKernelGGL((attn_fwd_kernel<D, CB, PF>), grid, dim3(256), dyn, s, p);
}

static int attn_check(const AttnParams& p, int d_pad) {
    TR1_CHECK_ARG(d_pad == 32 || d_pad == 64 || d_pad == 96 || d_pad == 128, "attention: padded head dim must be 32/64/96/128");
    TR1_CHECK_ARG(p.d_real % 8 == 0 && p.d_real <= d_pad && p.d_real > d_pad - 32, "attention: head dim must be a multiple of 8");
    TR1_CHECK_ARG(p.q_ld % 8 == 0 && p.k_ld % 8 == 0 && p.o_ld % 4 == 0, "attention: leading dims must be multiples of 8");
    TR1_CHECK_ARG(p.group >= 1 && p.n_kv >= 1 && p.nsplit >= 1 && p.nsplit <= 64, "attention: bad group/n_kv/nsplit (nsplit <= 64)");
    return 0;
}

extern "C" int64_t tr1_attn_plan_ints(int64_t T, int64_t n_heads, int64_t n_kv, int64_t n_batch) {
    const int64_t nR = T * (n_kv > 0 ? n_heads / n_kv : 1);
    return n_batch * ((nR + 63) / 64) * (ATT_LIST_CAP + 1);
}

static int attn_fwd_impl(const void* Q, int64_t q_ld, const void* K, int64_t k_ld, const void* VT, int64_t vt_ld, void* O, int64_t o_ld,
                         void* lse, const void* pre, const void* lo, const void* hi, int64_t T, int64_t n_heads, int64_t n_kv,
                         int64_t n_slots, int64_t head_dim, float scale, int64_t nsplit, void* ws_f32, int64_t ws_floats, int64_t n_batch,
                         int64_t kv_batch_slots, void* plan, int plan_mode, void* stream, int o_frag = 0) {
    AttnParams p; memset(&p, 0, sizeof(p));
    p.o_frag = o_frag;
    TR1_CHECK_ARG(!o_frag || (nsplit > 1 && head_dim == 128), "attention: fragment-major output needs the split-KV merge at head dim 128");
    TR1_CHECK_ARG(plan_mode == 0 || (plan && nsplit > 1 && (plan_mode == 1 || plan_mode == 2)), "attention: plan_mode 1 / 2 needs a plan buffer and nsplit > 1");
    p.plan = (int*)plan; p.plan_mode = plan_mode;
    TR1_CHECK_ARG(n_batch >= 1, "attention: n_batch must be >= 1");
    p.n_batch = (int)n_batch; p.kv_batch_slots = kv_batch_slots;
    p.Q = (const bf16_t*)Q; p.q_ld = q_ld; p.K = (const bf16_t*)K; p.k_ld = k_ld; p.VT = (const bf16_t*)VT; p.vt_ld = vt_ld;
    p.O = (bf16_t*)O; p.o_ld = o_ld; p.lse = (float*)lse; p.pre = (const int*)pre; p.lo = (const int*)lo; p.hi = (const int*)hi;
    TR1_CHECK_ARG(n_kv > 0 && n_heads % n_kv == 0, "attention: n_heads must be a multiple of n_kv");
    p.T = (int)T; const bool magic_ok = att_set_group(p, T * (n_batch > 0 ? n_batch : 1), (int)(n_heads / n_kv)); p.n_kv = (int)n_kv; p.n_slots = (int)n_slots; p.d_real = (int)head_dim; p.nsplit = (int)nsplit;
    p.scale_log2 = scale * 1.4426950408889634f;
    const int d_pad = (int)((head_dim + 31) / 32 * 32);
    if (int e = attn_check(p, d_pad)) return e;
    TR1_CHECK_ARG(magic_ok, "attention: T * group^2 must stay below 2^32");
    TR1_CHECK_ARG(vt_ld % 8 == 0 && vt_ld >= n_slots, "attention: vt_ld must be a multiple of 8 and >= n_slots");
    if (T == 0) return 0;
    const int64_t nR = T * p.group;
    const bool decode = nsplit > 1;                 // split-KV: 64-row query tiles, 3 tiles in flight per block
    const int qrows = decode ? 64 : 128;
    const int qtiles = (int)((nR + qrows - 1) / qrows);
    const int64_t nRpad = (int64_t)qtiles * qrows;
    if (nsplit > 1) {
        const int64_t need = nsplit * n_batch * n_kv * nRpad * (d_pad + 2);
        TR1_CHECK_ARG(ws_f32 && ws_floats >= need, "attention: split-KV workspace too small");
        p.Opart = (float*)ws_f32; p.mpart = p.Opart + nsplit * n_batch * n_kv * nRpad * d_pad; p.lpart = p.mpart + nsplit * n_batch * n_kv * nRpad;
    }
    dim3 grid(qtiles, (unsigned)(n_kv * n_batch), (unsigned)nsplit);
    if (!decode) {      // XCD-aware block order for the nsplit == 1 launches
        p.grid_x = qtiles; p.grid_y = (int)(n_kv * n_batch);
        p.xcd_pad = (p.grid_x * p.grid_y + 63) / 64 * 64;
        grid = dim3((unsigned)p.xcd_pad, 1, 1);
    }
    hipStream_t s = (hipStream_t)stream;
    const int dec32 = decode ? tr1_launch_attn_dec32(p, grid, s) : 0;
    if (dec32) {
        // head dim 128 reading the per-step plan: the LDS-DMA kernel of attn_fwd32.hip (all of a block's tiles in flight at once)
    } else if (decode) {
        switch (d_pad) {
            case 32: launch_fwd<32, 1, 3>(grid, s, p); break;
            case 64: launch_fwd<64, 1, 3>(grid, s, p); break;
            case 96: launch_fwd<96, 1, 3>(grid, s, p); break;
            default: launch_fwd<128, 1, 3>(grid, s, p); break;
        }
    } else {
        switch (d_pad) {
            case 32: launch_fwd<32, 2, 1>(grid, s, p); break;
            case 64: launch_fwd<64, 2, 1>(grid, s, p); break;
            case 96: launch_fwd<96, 2, 1>(grid, s, p); break;
            default: launch_fwd<128, 2, 1>(grid, s, p); break;
        }
    }
    if (nsplit > 1) {
        dim3 cg((unsigned)((nR + ATT_COMBINE_ROWS - 1) / ATT_COMBINE_ROWS), (unsigned)(n_kv * n_batch));
        switch (d_pad) {
            case 32: hipLaunchKernelGGL(attn_combine_kernel<32>, cg, dim3(256), 0, s, p, nRpad); break;
            case 64: hipLaunchKernelGGL(attn_combine_kernel<64>, cg, dim3(256), 0, s, p, nRpad); break;
            case 96: hipLaunchKernelGGL(attn_combine_kernel<96>, cg, dim3(256), 0, s, p, nRpad); break;
            default:
                if (p.d_real == 128 && (p.o_ld % 8 == 0 || p.o_frag)) hipLaunchKernelGGL(attn_combine128_kernel, cg, dim3(128), 0, s, p, nRpad);      // one wave per row
                else hipLaunchKernelGGL(attn_combine_kernel<128>, cg, dim3(256), 0, s, p, nRpad);
                break;
        }
    }
    TR1_LAUNCH_CHECK();
}

extern "C" int tr1_attn_fwd(const void* Q, int64_t q_ld, const void* K, int64_t k_ld, const void* VT, int64_t vt_ld, void* O, int64_t o_ld,
                            void* lse, const void* pre, const void* lo, const void* hi, int64_t T, int64_t n_heads, int64_t n_kv,
                            int64_t n_slots, int64_t head_dim, float scale, int64_t nsplit, void* ws_f32, int64_t ws_floats, int64_t n_batch,
                            int64_t kv_batch_slots, void* stream) {
    return attn_fwd_impl(Q, q_ld, K, k_ld, VT, vt_ld, O, o_ld, lse, pre, lo, hi, T, n_heads, n_kv, n_slots, head_dim, scale, nsplit, ws_f32, ws_floats,
                         n_batch, kv_batch_slots, nullptr, 0, stream);
}

// Split-KV decode over the layers of ONE decode step: the masks are the same in every layer, so the launch of the first layer (plan_mode 1)
// publishes each query tile's relevant-tile list in `plan` (tr1_attn_plan_ints ints) and the launches of the other layers (plan_mode 2) start
// from it - no mask reduction, no list construction, two block barriers fewer in front of the first tile.  plan_mode 0 = tr1_attn_fwd.
extern "C" int tr1_attn_fwd_planned(const void* Q, int64_t q_ld, const void* K, int64_t k_ld, const void* VT, int64_t vt_ld, void* O, int64_t o_ld,
                                    void* lse, const void* pre, const void* lo, const void* hi, int64_t T, int64_t n_heads, int64_t n_kv,
                                    int64_t n_slots, int64_t head_dim, float scale, int64_t nsplit, void* ws_f32, int64_t ws_floats, int64_t n_batch,
                                    int64_t kv_batch_slots, void* plan, int plan_mode, void* stream) {
    return attn_fwd_impl(Q, q_ld, K, k_ld, VT, vt_ld, O, o_ld, lse, pre, lo, hi, T, n_heads, n_kv, n_slots, head_dim, scale, nsplit, ws_f32, ws_floats,
                         n_batch, kv_batch_slots, plan, plan_mode, stream);
}

// tr1_attn_fwd_planned whose merged rows leave FRAGMENT-MAJOR for tr1_gemm_oproj_frag (csrc/oproj.hip): O holds ceil(n_batch * T / 16) * 16 rows x n_heads * 128
// bf16, element (m, k) at ((m / 16) * (n_heads * 4) + k / 32) * 512 + (m % 16) * 32 + k % 32.  Split-KV (nsplit > 1), head dim 128.
extern "C" int tr1_attn_fwd_planned_frag(const void* Q, int64_t q_ld, const void* K, int64_t k_ld, const void* VT, int64_t vt_ld, void* Ofrag, const void* pre,
                                         const void* lo, const void* hi, int64_t T, int64_t n_heads, int64_t n_kv, int64_t n_slots, int64_t head_dim, float scale,
                                         int64_t nsplit, void* ws_f32, int64_t ws_floats, int64_t n_batch, int64_t kv_batch_slots, void* plan, int plan_mode, void* stream) {
    return attn_fwd_impl(Q, q_ld, K, k_ld, VT, vt_ld, Ofrag, n_heads * head_dim, nullptr, pre, lo, hi, T, n_heads, n_kv, n_slots, head_dim, scale, nsplit, ws_f32,
                         ws_floats, n_batch, kv_batch_slots, plan, plan_mode, stream, 1);
}

extern "C" int64_t tr1_attn_fwd_workspace_floats(int64_t T, int64_t n_heads, int64_t n_kv, int64_t head_dim, int64_t nsplit) {
    // per batch entry; multiply by n_batch for a batched launch
    if (nsplit <= 1 || n_kv <= 0) return 0;
    const int64_t nR = T * (n_heads / n_kv);
    const int64_t nRpad = (nR + 127) / 128 * 128;
    const int64_t d_pad = (head_dim + 31) / 32 * 32;
    return nsplit * n_kv * nRpad * (d_pad + 2);
}

extern "C" int tr1_pack_transpose(const void* in, int64_t ld_in, void* out, int64_t ld_out, const void* slots, int64_t T, int64_t n_heads,
                                  int64_t n_kv, int64_t head_dim, int64_t zero_cols, void* stream) {
    TR1_CHECK_ARG(n_kv > 0 && n_heads % n_kv == 0, "pack_transpose: n_heads must be a multiple of n_kv");
    const int group = (int)(n_heads / n_kv);
    TR1_CHECK_ARG(slots || ld_out >= T * group, "pack_transpose: ld_out too small");
    if (T == 0) return 0;
    TR1_CHECK_ARG(zero_cols <= ld_out, "pack_transpose: zero_cols exceeds the leading dimension");
    const int64_t span = (zero_cols > T * group && !slots) ? zero_cols : T * group;
    dim3 grid((unsigned)((head_dim + 63) / 64), (unsigned)((span + 63) / 64), (unsigned)n_kv);
    hipLaunchKernelGGL(pack_transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, ld_in, (bf16_t*)out, ld_out,
                       (const int*)slots, T, (int)n_kv, group, (int)head_dim, slots ? (int64_t)0 : zero_cols);
    TR1_LAUNCH_CHECK();
}

extern "C" int tr1_scatter_slots(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, const void* slots, int64_t T, int64_t cols,
                                 void* stream) {
    TR1_CHECK_ARG(cols % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0, "scatter_slots: cols/ld must be multiples of 8");
    if (T == 0) return 0;
    hipLaunchKernelGGL(scatter_slots_kernel, dim3(tr1_grid_1d(T * cols / 8, 256, 4096)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                       ld_src, (bf16_t*)dst, ld_dst, (const int*)slots, T, (int)cols);
    TR1_LAUNCH_CHECK();
}
