// Shared pieces of the attention kernels (gfx950).
//
// One kernel family serves every attention in the GRPO path:
//   * LLM causal GQA attention with ONE shared prompt prefix and G completion suffixes (training forward/backward,
//     rollout prefill, and - through split-KV - single-token decode over the KV cache);
//   * ViT varlen non-causal attention over cu_seqlens segments (Qwen2-VL frames, Qwen2.5-VL windows).
// The mask is "two intervals per query token":   key kv is visible to token t  iff
//        kv < pre[t]   or   lo[t] <= kv <= hi[t]
// (prompt token: pre=0, lo=0, hi=t;  completion token of group g at step s: pre=P, lo=P+g*C, hi=P+g*C+s;
//  ViT token in segment [a,b): pre=0, lo=a, hi=b-1).  Reference semantics: flash_attn_varlen_func / SDPA as called from
//  transformers/models/qwen2_vl/modeling_qwen2_vl.py:379-396 (vision) and :521-556 (LLM, softmax in fp32, GQA by repeat).
//
// Query rows are "GQA packed": row R = t*group + hq enumerates the `group` query heads that share kv head `kvh`, so a
// K/V tile staged in LDS is reused by all of them and decode (T = G rollout rows) still fills MFMA columns.
//
// MFMA orientation trick used throughout: scores are computed TRANSPOSED, S^T[kv][q] = K * Q^T, so that after
// v_mfma_f32_16x16x32_bf16 a lane holds, for ITS query column q = lane&15, the four keys kv = (lane>>4)*4 + r.
// Row statistics (max, sum, lse, delta) are then lane-local, and two S^T tiles concatenate into a legal B operand
// (k-slot (g,j): j<4 -> tile0 key g*4+j, j>=4 -> tile1 key g*4+j-4) for O^T = V^T * P^T with V^T read as 8-byte pairs.
#pragma once
#include "tr1_common.h"

struct AttnParams {
    const bf16_t* Q;  int64_t q_ld;    // [T, n_heads*d]   (row = token)
    const bf16_t* K;  int64_t k_ld;    // [slots, n_kv*d]  (row = kv slot)
    const bf16_t* V;  int64_t v_ld;    // [slots, n_kv*d]
    const bf16_t* KT; int64_t kt_ld;   // [n_kv*d, slots_pad]  (transposed copies, slot-contiguous)
    const bf16_t* VT; int64_t vt_ld;
    const bf16_t* QT; int64_t qt_ld;   // [n_kv*d, T*group padded]  (packed-row-contiguous)
    const bf16_t* dOT; int64_t dot_ld;
    bf16_t* O;        int64_t o_ld;    // [T, n_heads*d]
    const bf16_t* dO; int64_t do_ld;
    bf16_t* dQ;       int64_t dq_ld;
    bf16_t* dK;       int64_t dk_ld;   // [slots, n_kv*d]
    bf16_t* dV;       int64_t dv_ld;
    float* lse;                        // [n_heads, T] natural-log LSE of the scaled scores
    float* delta;                      // [n_heads, T] rowsum(dO*O)
    const int* pre; const int* lo; const int* hi;   // [T]
    const int* qmeta;                  // [n_qtiles64, 3] (max pre, min lo, max hi) per 64 packed rows (backward only)
    float* Opart; float* mpart; float* lpart;       // split-KV workspaces
    int T, group, n_kv, n_slots, d_real, nsplit;
    unsigned group_magic;              // ceil(2^32 / group): R / group == umulhi(R, group_magic) for R < 2^32 / group (att_set_group)
    int n_batch; int64_t kv_batch_slots;   // forward only: batch b uses Q/O/mask rows [b*T,(b+1)*T) and cache slots [b*kv_batch_slots, ...)
    float scale_log2;                  // softmax scale * log2(e)
    // split-KV decode: the relevant-tile lists of a decode step are the same in every layer (same pre / lo / hi), so the first layer's launch
    // (plan_mode 1) stores them - [n_batch][q tiles][ATT_LIST_CAP ids + count] - and the other layers' launches (plan_mode 2) read them
    int* plan; int plan_mode;
    // split-KV decode at head dim 128: the merge kernel writes O FRAGMENT-MAJOR for tr1_gemm_oproj_frag (csrc/oproj.hip) - element (row m, feature k) at
    // ((m / 16) * (n_heads * 4) + k / 32) * 512 + (m % 16) * 32 + k % 32, m = batch entry * T + token - instead of row-major [rows, n_heads * 128]
    int o_frag;
    // nsplit == 1 (training / prefill / ViT): the (query tile, kv head) grid is launched as ONE dimension of xcd_pad blocks and re-mapped so that
    // each of the 8 XCDs (blocks are dealt to them round-robin) walks chunks of 8 CONSECUTIVE blocks of it: the query tiles that share a segment's / a
    // head's K and V tiles then meet in one L2 instead of fetching them from HBM once per XCD.  grid_x / grid_y = the logical grid.
    int grid_x, grid_y, xcd_pad;
    // backward only: when set, dQ and dK leave the kernels already multiplied by the TRANSPOSED rotary matrix (the backward of M-RoPE, TF:212-222):
    // fp32 cos / sin tables [T, d / 2]; row t of dQ and slot t of dK use row t
    const float* rope_cos; const float* rope_sin;
    // backward at head dim 128 (round 6): the dQ kernel's prologue also produces what attn_delta_kernel did - delta = rowsum(dO o O) and the log2-scaled LSE
    // of its own rows (it holds the dO row in registers already) and the per-64-row mask summary `qmeta` the dK/dV kernel skips tiles with - so that launch is gone.
    // lse2_out / qmeta_out: where to store them (null: the separate kernel ran and `delta` / lse2 / `qmeta` are inputs)
    float* lse2_out; int* qmeta_out;
};

#define ATT_KV 64          // keys per tile
#define NEG_INF (-INFINITY)

// ---- merge of split-KV partials: attn_combine_kernel and the merged tail of attn_dec32_kernel run THESE operations in this order (bit-identical results)
TR1_DEV void att_merge_stats(const float* m_row, const float* l_row, int nsplit, float& M, float& Ms, float& L) {
    M = NEG_INF;
    for (int sp = 0; sp < nsplit; ++sp) M = fmaxf(M, m_row[sp]);
    Ms = (M == NEG_INF) ? 0.f : M;
    L = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) L = __builtin_fmaf(exp2f(m_row[sp] - Ms), l_row[sp], L);
}
// the share of split group sg (splits sg, sg + 4, ...) in one row's weighted sum; ov[i] = the partial row of split sg + 4 i (anything when that split does not exist)
TR1_DEV f32x4_t att_merge_weighted(const f32x4_t (&ov)[16], const float* m_row, float Ms, int sg, int nsplit) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int sp = sg + 4 * i;
        const float w = sp < nsplit ? exp2f(m_row[sp < 64 ? sp : 0] - Ms) : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(w, ov[i][j], acc[j]);
    }
    return acc;
}

TR1_DEV bf16x8_t make_frag(u32x2_t a, u32x2_t b) {
    u32x4_t w = {a[0], a[1], b[0], b[1]};
    return __builtin_bit_cast(bf16x8_t, w);
}
TR1_DEV bf16x8_t pack_frag(f32x4_t a, f32x4_t b) {
    u32x4_t w = {pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3])};
    return __builtin_bit_cast(bf16x8_t, w);
}
TR1_DEV bf16x8_t zero_frag() { u32x4_t w = {0, 0, 0, 0}; return __builtin_bit_cast(bf16x8_t, w); }

// 16-byte global load of 8 bf16 at column d0 of a row, zero beyond d_real / invalid row
TR1_DEV bf16x8_t load_row_frag(const bf16_t* row_ptr, int d0, int d_real, bool valid) {
    if (valid && d0 < d_real) return *reinterpret_cast<const bf16x8_t*>(row_ptr + d0);
    return zero_frag();
}

// Stage `NROWS` rows x D columns (row-major source) into LDS with a (2*D+16)-byte row stride (conflict-free b128 column reads).
// Row r of the tile is source row (row0 + r); rows >= nvalid and columns >= d_real are zero-filled.
template <int D, int NROWS>
TR1_DEV void stage_rows(char* lds, const bf16_t* src, int64_t ld, int64_t col0, int64_t row0, int64_t nvalid, int d_real) {
    constexpr int CH = D / 8;
    constexpr int STRIDE = 2 * D + 16;
    for (int idx = threadIdx.x; idx < NROWS * CH; idx += 256) {
        const int r = idx / CH, c = idx - r * CH;
        u32x4_t v = {0, 0, 0, 0};
        const int64_t row = row0 + r;
        if (row < nvalid && c * 8 < d_real) v = *reinterpret_cast<const u32x4_t*>(src + row * ld + col0 + c * 8);
        *reinterpret_cast<u32x4_t*>(lds + r * STRIDE + c * 16) = v;
    }
}
// Same, but the tile rows are GQA-packed query rows R = R0 + r  ->  token R/group, head R%group.
template <int D, int NROWS>
TR1_DEV void stage_packed_rows(char* lds, const bf16_t* src, int64_t ld, int kvh, int group, int64_t R0, int64_t nR, int d_real) {
    constexpr int CH = D / 8;
    constexpr int STRIDE = 2 * D + 16;
    for (int idx = threadIdx.x; idx < NROWS * CH; idx += 256) {
        const int r = idx / CH, c = idx - r * CH;
        u32x4_t v = {0, 0, 0, 0};
        const int64_t R = R0 + r;
        if (R < nR && c * 8 < d_real) {
            const int64_t t = R / group; const int hq = (int)(R - t * group);
            v = *reinterpret_cast<const u32x4_t*>(src + t * ld + (int64_t)(kvh * group + hq) * d_real + c * 8);
        }
        *reinterpret_cast<u32x4_t*>(lds + r * STRIDE + c * 16) = v;
    }
}
// Stage a transposed tile: D rows (feature d) x 64 columns (slot / packed row), source is [n_kv*d_real, ldT] with the
// column index contiguous. LDS row stride 144 bytes. Columns >= nvalid and rows >= d_real are zero-filled.
template <int D>
TR1_DEV void stage_T(char* lds, const bf16_t* srcT, int64_t ldT, int kvh, int64_t col0, int64_t nvalid, int d_real) {
    for (int idx = threadIdx.x; idx < D * 8; idx += 256) {
        const int d = idx >> 3, c = idx & 7;
        u32x4_t v = {0, 0, 0, 0};
        const int64_t col = col0 + c * 8;
        if (d < d_real && col < nvalid) {
            v = *reinterpret_cast<const u32x4_t*>(srcT + ((int64_t)kvh * d_real + d) * ldT + col);
            if (col + 8 > nvalid) {  // ragged tail: keep only the valid columns (stale cache slots must not leak)
                const int keep = (int)(nvalid - col);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (2 * e >= keep) v[e] = 0;
                    else if (2 * e + 1 >= keep) v[e] &= 0xffffu;
                }
            }
        }
        *reinterpret_cast<u32x4_t*>(lds + d * 144 + c * 16) = v;
    }
}

// ---- split staging (issue the global loads early, write LDS late): hides HBM/L2 latency under the current tile's MFMAs.
// A row-major 64 x D tile is D/8*64 16-byte chunks = (D/32) chunks per thread of a 256-thread block; same count for the D x 64 V^T tile.
template <int D>
struct TileRegs { u32x4_t k[D / 32]; u32x4_t v[D / 32]; };

template <int D>
TR1_DEV void tile_load_regs(TileRegs<D>& r, const bf16_t* K, int64_t k_ld, const bf16_t* VT, int64_t vt_ld, int kvh, int64_t kv0, int64_t n_slots,
                            int d_real) {
    // Branch-free: every lane issues all its loads from a clamped (always valid) address; out-of-range pieces are zeroed when
    // the registers are written to LDS (tile_store_lds), so no s_waitcnt lands between the loads and the MFMAs they overlap.
    constexpr int CH = D / 8;
    const int64_t last_slot = n_slots - 1;
    const int64_t last_chunk = (last_slot >> 3) << 3;
#pragma unroll
    for (int j = 0; j < D / 32; ++j) {
        const int idx = threadIdx.x + j * 256;
        const int row = idx / CH, c = idx - row * CH;
        int64_t slot = kv0 + row; if (slot > last_slot) slot = last_slot;
        const int cc = (c * 8 < d_real) ? c * 8 : 0;
        r.k[j] = *reinterpret_cast<const u32x4_t*>(K + slot * k_ld + (int64_t)kvh * d_real + cc);
        int d = idx >> 3; if (d >= d_real) d = d_real - 1;
        int64_t col = kv0 + (idx & 7) * 8; if (col > last_chunk) col = last_chunk;
        r.v[j] = *reinterpret_cast<const u32x4_t*>(VT + ((int64_t)kvh * d_real + d) * vt_ld + col);
    }
}
template <int D>
TR1_DEV void tile_store_lds(const TileRegs<D>& r, char* lds_k, char* lds_vt, int64_t kv0, int64_t n_slots, int d_real) {
    constexpr int CH = D / 8;
    constexpr int KSTR = 2 * D + 16;
    const u32x4_t zero = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < D / 32; ++j) {
        const int idx = threadIdx.x + j * 256;
        const int row = idx / CH, c = idx - row * CH;
        const bool kok = (kv0 + row < n_slots) && (c * 8 < d_real);
        *reinterpret_cast<u32x4_t*>(lds_k + row * KSTR + c * 16) = kok ? r.k[j] : zero;
        const int d = idx >> 3, c2 = idx & 7;
        const int64_t col = kv0 + c2 * 8;
        u32x4_t v = r.v[j];
        const int keep = (d < d_real) ? (int)min((int64_t)8, max((int64_t)0, n_slots - col)) : 0;   // valid slots in this chunk
        if (keep < 8) {     // ragged tail / padding: stale cache slots must not leak (wave-divergent only on the last tile)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (2 * e >= keep) v[e] = 0;
                else if (2 * e + 1 >= keep) v[e] &= 0xffffu;
            }
        }
        *reinterpret_cast<u32x4_t*>(lds_vt + d * 144 + c2 * 16) = v;
    }
}

// ---- generic single-tile versions of the split staging (used by the backward kernels) -------------------------------------
// NT = threads of the block (256 or 512); a 64 x D tile is 8*D 16-byte chunks, 8*D/NT per thread.
template <int D, int NT = 256>
struct TReg { u32x4_t v[8 * D / NT]; };

// row-major rows [row0, row0+64) x D of src (leading dim ld, column offset col0); clamped addresses, masks applied at store time
template <int D, int NT = 256>
TR1_DEV void rows_load(TReg<D, NT>& r, const bf16_t* src, int64_t ld, int64_t col0, int64_t row0, int64_t nvalid, int d_real) {
    constexpr int CH = D / 8;
#pragma unroll
    for (int j = 0; j < 8 * D / NT; ++j) {
        const int idx = threadIdx.x + j * NT;
        const int row = idx / CH, c = idx - row * CH;
        int64_t rr = row0 + row; if (rr > nvalid - 1) rr = nvalid - 1;
        const int cc = (c * 8 < d_real) ? c * 8 : 0;
        r.v[j] = *reinterpret_cast<const u32x4_t*>(src + rr * ld + col0 + cc);
    }
}
// GQA-packed query rows R = R0 + row -> token R/group, head kvh*group + R%group
template <int D, int NT = 256>
TR1_DEV void prows_load(TReg<D, NT>& r, const bf16_t* src, int64_t ld, int kvh, int group, int64_t R0, int64_t nR, int d_real, unsigned magic) {
    constexpr int CH = D / 8;
#pragma unroll
    for (int j = 0; j < 8 * D / NT; ++j) {
        const int idx = threadIdx.x + j * NT;
        const int row = idx / CH, c = idx - row * CH;
        int64_t R = R0 + row; if (R > nR - 1) R = nR - 1;
        const unsigned ru = (unsigned)R, tu = group == 1 ? ru : __umulhi(ru, magic);
        const int64_t t = tu; const int hq = (int)(ru - tu * (unsigned)group);
        const int cc = (c * 8 < d_real) ? c * 8 : 0;
        r.v[j] = *reinterpret_cast<const u32x4_t*>(src + t * ld + (int64_t)(kvh * group + hq) * d_real + cc);
    }
}
template <int D, int NT = 256>
TR1_DEV void rows_store(const TReg<D, NT>& r, char* lds, int64_t row0, int64_t nvalid, int d_real) {
    constexpr int CH = D / 8;
    constexpr int STRIDE = 2 * D + 16;
    const u32x4_t zero = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 8 * D / NT; ++j) {
        const int idx = threadIdx.x + j * NT;
        const int row = idx / CH, c = idx - row * CH;
        const bool ok = (row0 + row < nvalid) && (c * 8 < d_real);
        *reinterpret_cast<u32x4_t*>(lds + row * STRIDE + c * 16) = ok ? r.v[j] : zero;
    }
}
// transposed source [n_kv*d_real, ldT], columns [col0, col0+64)
template <int D, int NT = 256>
TR1_DEV void T_load(TReg<D, NT>& r, const bf16_t* srcT, int64_t ldT, int kvh, int64_t col0, int64_t nvalid, int d_real) {
    const int64_t last_chunk = ((nvalid - 1) >> 3) << 3;
#pragma unroll
    for (int j = 0; j < 8 * D / NT; ++j) {
        const int idx = threadIdx.x + j * NT;
        int d = idx >> 3; if (d >= d_real) d = d_real - 1;
        int64_t col = col0 + (idx & 7) * 8; if (col > last_chunk) col = last_chunk;
        r.v[j] = *reinterpret_cast<const u32x4_t*>(srcT + ((int64_t)kvh * d_real + d) * ldT + col);
    }
}
template <int D, int NT = 256>
TR1_DEV void T_store(const TReg<D, NT>& r, char* lds, int64_t col0, int64_t nvalid, int d_real) {
#pragma unroll
    for (int j = 0; j < 8 * D / NT; ++j) {
        const int idx = threadIdx.x + j * NT;
        const int d = idx >> 3, c2 = idx & 7;
        const int64_t col = col0 + c2 * 8;
        u32x4_t v = r.v[j];
        const int keep = (d < d_real) ? (int)min((int64_t)8, max((int64_t)0, nvalid - col)) : 0;
        if (keep < 8) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (2 * e >= keep) v[e] = 0;
                else if (2 * e + 1 >= keep) v[e] &= 0xffffu;
            }
        }
        *reinterpret_cast<u32x4_t*>(lds + d * 144 + c2 * 16) = v;
    }
}

TR1_DEV bool att_visible(int kv, int pre, int lo, int hi) { return (kv < pre) || (kv >= lo && kv <= hi); }
// branch-free form (bitwise, every operand already in a register): keeps hipcc from guarding operand loads behind short-circuit branches
TR1_DEV bool att_visible_nb(int kv, int pre, int lo, int hi) { return (kv < pre) | ((kv >= lo) & (kv <= hi)); }

// GQA-packed row R -> (token, query head inside the group) without the 64-bit software division of `R / group`
TR1_DEV void att_split_row(const AttnParams& p, int64_t R, int& t, int& hq) {
    const unsigned r = (unsigned)R;
    const unsigned q = p.group == 1 ? r : __umulhi(r, p.group_magic);
    t = (int)q; hq = (int)(r - q * (unsigned)p.group);
}
static inline bool att_set_group(AttnParams& p, int64_t T, int group) {
    p.group = group;
    p.group_magic = group > 1 ? (unsigned)((0x100000000ull + (unsigned)group - 1) / (unsigned)group) : 0u;
    return (uint64_t)T * (uint64_t)group * (uint64_t)group < 0x100000000ull;     // exactness range of the magic multiply
}

// Which 64-key tiles can a set of rows with (max_pre, min_lo, max_hi) see?  [0, pre_tiles) U [start2, end2]
struct TileRange { int pre_tiles, start2, n_rel; };
TR1_DEV TileRange att_tile_range(int max_pre, int min_lo, int max_hi, int n_slots) {
    TileRange tr;
    if (max_pre > n_slots) max_pre = n_slots;
    if (max_hi >= n_slots) max_hi = n_slots - 1;
    tr.pre_tiles = (max_pre + ATT_KV - 1) / ATT_KV;
    int s2 = min_lo / ATT_KV; if (s2 < tr.pre_tiles) s2 = tr.pre_tiles;
    const int e2 = (max_hi >= 0) ? max_hi / ATT_KV : -1;
    tr.start2 = s2;
    tr.n_rel = tr.pre_tiles + ((max_hi >= min_lo && e2 >= s2) ? (e2 - s2 + 1) : 0);
    return tr;
}
TR1_DEV int att_tile_at(const TileRange& tr, int i) { return i < tr.pre_tiles ? i : tr.start2 + (i - tr.pre_tiles); }

// ---- pieces shared by the 32x32x16-MFMA kernels (round 3: attn_bwd_dkdv32 / attn_bwd_dq32 / attn_fwd32) -------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16_t;       // 32x32 accumulator: lane (column n = lane & 31, half h = lane >> 5), register r -> row (r&3) + 8(r>>2) + 4h
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef const __attribute__((address_space(1))) void* att_gptr_t;
typedef __attribute__((address_space(3))) void* att_lptr_t;
// swizzle key of row (mod 16) of an unpadded 256-byte-row tile image: logical 16-byte chunk c of the row is stored at chunk c ^ skey(row)
TR1_DEV int skey(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
// max of three without the canonicalising v_max x, x that fmaxf() costs per operand (no NaN can reach the score tiles: -inf masks, finite inputs)
TR1_DEV float att_max3(float a, float b, float c) { float o; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c)); return o; }
// both halves of a wave meet (tr1_common.h: v_permlane32_swap instead of the ds_bpermute behind __shfl_xor(x, 32)): fmaxf(a, b) / a + b give what fmaxf(x, shfl) / x + shfl gave
TR1_DEV void att_halves(float x, float& a, float& b) { tr1_halves32(x, a, b); }
// registers b .. b+7 of a 32x32 accumulator -> one bf16 MFMA operand fragment (k-slot j of lane half h = accumulator row 16(b/8) + (j&3) + 8(j>>2) + 4h)
TR1_DEV bf16x8_t pack8(const f32x16_t& c, int b) {
    u32x4_t w = {pack2bf(c[b], c[b + 1]), pack2bf(c[b + 2], c[b + 3]), pack2bf(c[b + 4], c[b + 5]), pack2bf(c[b + 6], c[b + 7])};
    return __builtin_bit_cast(bf16x8_t, w);
}
