// bf16 GEMM for gfx950 (CDNA4): C[M,N] = A[M,K] * B[N,K]^T (+ bias[N]) (+ residual[M,N]), fp32 accumulate on MFMA.
//
// This is the projection workhorse behind every Linear in the GRPO path (reference call sites: the cuBLAS GEMMs under
// transformers/models/qwen2_vl/modeling_qwen2_vl.py:501-504 (q/k/v/o), :459-466 (MLP), :251-274 (patch embed as GEMM),
// :277-290 (patch merger), :1323 (lm_head)).  Both operands are K-contiguous ("NT"), which is what nn.Linear stores.
//
// Design (MI355X-first, see /opt/skills/guides/cdna_hip_programming.md section 5):
//   * 128x128x64 block tile, 256 threads = 4 waves (2x2), each wave owns 64x64 = 4x4 v_mfma_f32_16x16x32_bf16 tiles.
//   * operands go HBM -> LDS directly with global_load_lds_dwordx4 (no VGPR round trip), two LDS buffers,
//     one __syncthreads per K-tile (the barrier's vmcnt(0) retires the DMA issued one iteration earlier).
//   * the LDS image is lane-linear per DMA instruction, so the bank-conflict swizzle is applied on the SOURCE
//     address (which 16-byte chunk of the 128-byte row a lane fetches) and mirrored on the ds_read_b128 side.
//   * the MFMA is issued with the weight tile as the A operand and the activation tile as the B operand, with the
//     16 weight rows of tile j permuted so that every lane ends up owning 16 CONTIGUOUS output columns of one output
//     row: the epilogue is two 16-byte bf16 stores (or four fp32 ones) per row, no LDS transpose.
//   * 1-D grid with an XCD-aware, M-grouped tile order so the 32 blocks resident on one XCD share B panels in its L2.
#include "tr1_common.h"
#include <stdlib.h>

#define BM 128
#define BN 128
#define BK 64
#define TILE_BYTES (BM * BK * 2)  // 16 KiB per operand per buffer

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

TR1_DEV bf16x8_t zero_frag8() { u32x4_t w = {0, 0, 0, 0}; return __builtin_bit_cast(bf16x8_t, w); }
TR1_DEV int keyA(int row) { return (row >> 1) & 7; }
TR1_DEV int keyB(int row) { return (((row >> 4) & 3) << 1) | ((row >> 1) & 1); }

// Stage one 128x64 bf16 tile: 16 wave-instructions of 1 KiB, 4 per wave. Rows beyond `rows_valid` are clamped
// (their products land in rows/cols that are never stored).
template <bool IS_B>
TR1_DEV void stage_tile(const bf16_t* __restrict__ g, int64_t ld, int64_t row0, int64_t rows_valid, int64_t k0, char* lds_tile,
                        int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int inst = wave * 4 + i;
        const int row = inst * 8 + (lane >> 3);
        const int phys = lane & 7;
        const int logical = phys ^ (IS_B ? keyB(row) : keyA(row));
        int64_t grow = row0 + row;
        if (grow >= rows_valid) grow = rows_valid - 1;
        const bf16_t* src = g + grow * ld + k0 + logical * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds_tile + inst * 1024), 16, 0, 0);
    }
}

template <bool OUT_F32, bool ACCUM>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, void* __restrict__ Cv,
                                                         const bf16_t* __restrict__ bias, const bf16_t* __restrict__ residual,
                                                         int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                                                         int64_t ldr, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // [buf][A|B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- XCD-aware tile order: block b runs on XCD b%8; give each XCD a contiguous run of the grouped order.
    const int nwg = tiles_m * tiles_n;
    int wgid;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int GROUP_M = 8;
    const int group = wgid / (GROUP_M * tiles_n);
    const int first_m = group * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int in_group = wgid - group * GROUP_M * tiles_n;
    const int tm = first_m + in_group % gsz;
    const int tn = in_group / gsz;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = (int)(K / BK);
    stage_tile<false>(A, lda, m0, M, 0, smem, wave, lane);
    stage_tile<true>(B, ldb, n0, N, 0, smem + TILE_BYTES, wave, lane);

    const int u = lane & 15, g = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();  // retires this wave's DMA for tile kt (vmcnt(0)) and orders it against every reader
        char* curA = smem + (kt & 1) * 2 * TILE_BYTES;
        char* curB = curA + TILE_BYTES;
        if (kt + 1 < nk) {
            char* nxtA = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
            stage_tile<false>(A, lda, m0, M, (int64_t)(kt + 1) * BK, nxtA, wave, lane);
            stage_tile<true>(B, ldb, n0, N, (int64_t)(kt + 1) * BK, nxtA + TILE_BYTES, wave, lane);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t xa[4], wb[4];
            const int chunk = ks * 4 + g;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm * 64 + i * 16 + u;
                xa[i] = *reinterpret_cast<const bf16x8_t*>(curA + row * 128 + ((chunk ^ keyA(row)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wn * 64 + (u >> 2) * 16 + j * 4 + (u & 3);
                wb[j] = *reinterpret_cast<const bf16x8_t*>(curB + row * 128 + ((chunk ^ keyB(row)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: lane owns, for each i, row m = m0 + wm*64 + i*16 + u and columns n0 + wn*64 + g*16 + [0,16)
    const int64_t nbase = n0 + wn * 64 + g * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + wm * 64 + i * 16 + u;
        if (m >= M) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // two 8-column halves
            const int64_t n = nbase + h * 8;
            if (n + 8 > N) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[i][h * 2 + (e >> 2)][e & 3];
            if (bias) {
                const u32x4_t bv = *reinterpret_cast<const u32x4_t*>(bias + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[2 * e] += bflo(bv[e]); v[2 * e + 1] += bfhi(bv[e]); }
            }
            if (residual) {
                const u32x4_t rv = *reinterpret_cast<const u32x4_t*>(residual + m * ldr + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[2 * e] += bflo(rv[e]); v[2 * e + 1] += bfhi(rv[e]); }
            }
            if (OUT_F32) {
                float* cp = reinterpret_cast<float*>(Cv) + m * ldc + n;
                f32x4_t o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                if (ACCUM) {
                    const f32x4_t p0 = *reinterpret_cast<const f32x4_t*>(cp), p1 = *reinterpret_cast<const f32x4_t*>(cp + 4);
                    o0 += p0; o1 += p1;
                }
                *reinterpret_cast<f32x4_t*>(cp) = o0;
                *reinterpret_cast<f32x4_t*>(cp + 4) = o1;
            } else {
                bf16_t* cp = reinterpret_cast<bf16_t*>(Cv) + m * ldc + n;
                u32x4_t o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = pack2bf(v[2 * e], v[2 * e + 1]);
                *reinterpret_cast<u32x4_t*>(cp) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// 256x256x64 variant for large outputs: 512 threads = 8 waves (2 x 4), each wave owns 128x64 = 8x4 MFMA tiles, so every
// ds_read_b128 feeds more MFMAs (12 fragment reads per 32 MFMAs instead of 16) and the HBM/L2 traffic per FLOP halves.
// Same staging / swizzle / epilogue scheme as gemm_nt_kernel; 128 KiB of LDS (two buffers), one block per CU.
// ------------------------------------------------------------------------------------------------------------------
#define BN2 256
#define TILE2_BYTES (BN2 * BK * 2)  // 32 KiB: the B operand tile (and the A tile of the 256-row form)

// RT = 16-row MFMA tiles per wave along M: 7 / 8 / 9 / 10 -> 224 / 256 / 288 / 320 x 256 block.  The 288-row form exists for the M = 5074 (P + G*C)
// GEMMs with N = 3584: 18 x 14 = 252 blocks fill the 256 CUs in ONE round at 98 % padding efficiency, where 256 x 256 needs two rounds
// (280 blocks) and 128 x 128 three (1120 blocks on 512 slots).
// Epilogue of the 8-wave forms: lane (u, g) owns output row mrow0 + i*16 + u, 16 contiguous columns from ncol0 + g*16.
template <bool OUT_F32, bool ACCUM, int RT>
TR1_DEV void store_acc256(const f32x4_t (&acc)[RT][4], void* __restrict__ Cv, const bf16_t* __restrict__ bias, const bf16_t* __restrict__ residual,
                          int64_t M, int64_t N, int64_t ldc, int64_t ldr, int64_t mrow0, int64_t ncol0, int u, int g) {
    const int64_t nbase = ncol0 + g * 16;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int64_t m = mrow0 + i * 16 + u;
        if (m >= M) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t n = nbase + h * 8;
            if (n + 8 > N) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[i][h * 2 + (e >> 2)][e & 3];
            if (bias) {
                const u32x4_t bv = *reinterpret_cast<const u32x4_t*>(bias + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[2 * e] += bflo(bv[e]); v[2 * e + 1] += bfhi(bv[e]); }
            }
            if (residual) {
                const u32x4_t rv = *reinterpret_cast<const u32x4_t*>(residual + m * ldr + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[2 * e] += bflo(rv[e]); v[2 * e + 1] += bfhi(rv[e]); }
            }
            if (OUT_F32) {
                float* cp = reinterpret_cast<float*>(Cv) + m * ldc + n;
                f32x4_t o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                if (ACCUM) {
                    const f32x4_t p0 = *reinterpret_cast<const f32x4_t*>(cp), p1 = *reinterpret_cast<const f32x4_t*>(cp + 4);
                    o0 += p0; o1 += p1;
                }
                *reinterpret_cast<f32x4_t*>(cp) = o0;
                *reinterpret_cast<f32x4_t*>(cp + 4) = o1;
            } else {
                bf16_t* cp = reinterpret_cast<bf16_t*>(Cv) + m * ldc + n;
                u32x4_t o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = pack2bf(v[2 * e], v[2 * e + 1]);
                *reinterpret_cast<u32x4_t*>(cp) = o;
            }
        }
    }
}

// Epilogue of the phased 8-wave forms THROUGH LDS.  In the accumulator layout a store instruction of store_acc256 writes 64 separate
// 16-byte pieces (16 rows x 4 pieces at 64-byte stride for fp32): every 128-byte line is visited by 4-8 instructions, and the fp32
// read-modify-write of a weight gradient ran at 2.3 TB/s (904 against 1297 TFLOP/s for the same shape with bf16 output).  Here each wave
// parks CH of its 16-row accumulator tiles in its own slice of the (now dead) operand buffers - 256-byte rows, 16-byte chunks XOR-swizzled
// with the row so both directions are bank-conflict free - and reads them back row-contiguous: an instruction then covers 4 full fp32 rows
// (8 bf16 rows) of the wave's 64 columns, i.e. only whole lines.  Values and rounding are unchanged.
#ifndef TR1_EPI_LDS
#define TR1_EPI_LDS 1
#endif
#define TR1_EPI_STORE(ptr, val) (*(ptr) = (val))      // (non-temporal C stores measured: -0.55 us per round of tiles in a probe, nothing in the step)
// EPI = 1 ("lm_head -> log-prob / entropy"): nothing is stored to C.  The wave's 64 columns of a row are rounded to bf16 (the logits the
// reference materialises are bf16) and reduced to the online-softmax triple (max, sum e^(x-max), sum x e^(x-max)); lane c8 = 0 of a row writes it to
// part[row][ncol0 / 64] (Cv = float4 partials, ldc = column blocks per row, +1 slot per row for the target's logit), and the lane that holds
// column targets[row] writes that logit to slot ldc - 1.  `bias` carries the int32 targets.  The [R, V] logits never exist in HBM.
template <bool OUT_F32, bool ACCUM, int RT, int EPI = 0>
TR1_DEV void store_acc256_lds(const f32x4_t (&acc)[RT][4], char* __restrict__ wave_lds, void* __restrict__ Cv, const bf16_t* __restrict__ bias,
                              const bf16_t* __restrict__ residual, int64_t M, int64_t N, int64_t ldc, int64_t ldr, int64_t mrow0, int64_t ncol0,
                              int lane, float* __restrict__ sumsq_slot = nullptr, bf16_t* __restrict__ wire16 = nullptr, int64_t ldw16 = 0) {
    // wire16 (fp32 output only): the value stored is ALSO written, rounded to bf16, at the same (row, column) of a second matrix - the gradient exchange's
    // wire-format copy of a weight gradient, taken from the epilogue that produces the final fp32 value instead of a 6-byte-per-parameter staging pass
    constexpr int CH = (RT % 2 == 0) ? 4 : 3;                             // 16-row tiles per pass: 8 waves x CH x 4 KiB fit the operand buffers
    const int u = lane & 15, g = lane >> 4;
    float ssq = 0.f;                                                      // fp32 output only: sum of squares of what this wave stores (gradient norm)
#pragma unroll
    for (int i0 = 0; i0 < RT; i0 += CH) {
        const int cnt = RT - i0 < CH ? RT - i0 : CH;
#pragma unroll
        for (int ii = 0; ii < CH; ++ii) {
            if (ii < cnt) {
                const int row = ii * 16 + u;
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4_t*>(wave_lds + row * 256 + (((g * 4 + j) ^ u) << 4)) = acc[i0 + ii][j];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (OUT_F32) {
            const int c = lane & 15;
            const int64_t n = ncol0 + c * 4;
#pragma unroll
            for (int r4 = 0; r4 < CH * 4; ++r4) {
                if (r4 < cnt * 4) {
                    const int rr = r4 * 4 + (lane >> 4);
                    f32x4_t v = *reinterpret_cast<const f32x4_t*>(wave_lds + rr * 256 + ((c ^ (rr & 15)) << 4));
                    const int64_t m = mrow0 + i0 * 16 + rr;
                    if (m < M && n + 4 <= N) {
                        if (bias) { const u32x2_t bv = *reinterpret_cast<const u32x2_t*>(bias + n); v[0] += bflo(bv[0]); v[1] += bfhi(bv[0]); v[2] += bflo(bv[1]); v[3] += bfhi(bv[1]); }
                        if (residual) {
                            const u32x2_t rv = *reinterpret_cast<const u32x2_t*>(residual + m * ldr + n);
                            v[0] += bflo(rv[0]); v[1] += bfhi(rv[0]); v[2] += bflo(rv[1]); v[3] += bfhi(rv[1]);
                        }
                        float* cp = reinterpret_cast<float*>(Cv) + m * ldc + n;
                        if (ACCUM) v += *reinterpret_cast<const f32x4_t*>(cp);
                        TR1_EPI_STORE(reinterpret_cast<f32x4_t*>(cp), v);
                        if (wire16) *reinterpret_cast<u32x2_t*>(wire16 + m * ldw16 + n) = (u32x2_t){pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
                        ssq += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                    }
                }
            }
        } else {
            const int c8 = lane & 7;
            const int64_t n = ncol0 + c8 * 8;
            u32x4_t bv = {0, 0, 0, 0};
            if (EPI != 1 && bias && n + 8 <= N) bv = *reinterpret_cast<const u32x4_t*>(bias + n);      // (EPI 1: `bias` carries the int32 targets)
#pragma unroll
            for (int r8 = 0; r8 < CH * 2; ++r8) {
                if (r8 < cnt * 2) {
                    const int rr = r8 * 8 + (lane >> 3);
                    const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(wave_lds + rr * 256 + (((2 * c8) ^ (rr & 15)) << 4));
                    const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(wave_lds + rr * 256 + (((2 * c8 + 1) ^ (rr & 15)) << 4));
                    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    const int64_t m = mrow0 + i0 * 16 + rr;
                    if (EPI == 1) {
                        const bool ok = m < M && n + 8 <= N;
                        const int tg = m < M ? reinterpret_cast<const int*>(bias)[m] : -1;
                        float mx = -INFINITY;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { v[e] = ok ? bf2f(f2bf(v[e])) : -INFINITY; mx = fmaxf(mx, v[e]); }
                        mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64)); mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
                        const float ms = (mx == -INFINITY) ? 0.f : mx;
                        float se = 0.f, te = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float ex = __expf(v[e] - ms);               // exp(-inf) = 0 for masked columns
                            se += ex; te += (v[e] == -INFINITY) ? 0.f : v[e] * ex;
                        }
                        se += __shfl_xor(se, 1, 64); se += __shfl_xor(se, 2, 64); se += __shfl_xor(se, 4, 64);
                        te += __shfl_xor(te, 1, 64); te += __shfl_xor(te, 2, 64); te += __shfl_xor(te, 4, 64);
                        if (m < M) {
                            f32x4_t* prow = reinterpret_cast<f32x4_t*>(Cv) + m * ldc;
                            if (c8 == 0 && ncol0 < N) prow[ncol0 >> 6] = (f32x4_t){mx, se, te, 0.f};   // a wave's 64-column slice past N (N % 256 != 0) has no slot:
                            //                                                                            slot N/64 is the target logit, N/64 + 1 the next row
                            if (ok && tg >= n && tg < n + 8) reinterpret_cast<float*>(prow + (ldc - 1))[0] = v[tg - n];
                        }
                        continue;
                    }
                    if (m < M && n + 8 <= N) {
                        if (bias) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v[2 * e] += bflo(bv[e]); v[2 * e + 1] += bfhi(bv[e]); }
                        }
                        if (residual) {
                            const u32x4_t rv = *reinterpret_cast<const u32x4_t*>(residual + m * ldr + n);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v[2 * e] += bflo(rv[e]); v[2 * e + 1] += bfhi(rv[e]); }
                        }
                        if (EPI == 5) {      // QuickGELU (Qwen2-VL vision MLP, TF:300-301) on the bf16-rounded projection, as act_kernel<1> after the GEMM
#pragma unroll
                            for (int e = 0; e < 8; ++e) { const float x = bf2f(f2bf(v[e])); v[e] = x / (1.f + __expf(-1.702f * x)); }
                        }
                        u32x4_t o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = pack2bf(v[2 * e], v[2 * e + 1]);
                        TR1_EPI_STORE(reinterpret_cast<u32x4_t*>(reinterpret_cast<bf16_t*>(Cv) + m * ldc + n), o);
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // the next pass overwrites the slice
    }
    if (OUT_F32 && sumsq_slot) {      // the FINAL value of every gradient element passes through this epilogue in the window's last micro-step: its
#pragma unroll                        // squared norm costs a few FMAs here instead of a 30 GB read of the arena (sumsq_kernel) before AdamW
        for (int o = 32; o >= 1; o >>= 1) ssq += __shfl_xor(ssq, o, 64);
        if (lane == 0) *sumsq_slot = ssq;
    }
}

TR1_DEV float epi_silu(float x) { return x / (1.f + __expf(-x)); }

template <bool IS_B, int ROWS>
TR1_DEV void stage_tile2(const bf16_t* __restrict__ g, int64_t ld, int64_t row0, int64_t rows_valid, int64_t k0, char* lds_tile,
                         int wave, int lane) {
    constexpr int NINST = ROWS / 8;                    // instructions of 8 rows (1 KiB) each
    constexpr int PER_WAVE = (NINST + 7) / 8;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int inst = wave * PER_WAVE + i;
        if (inst < NINST) {                            // wave-uniform
            const int row = inst * 8 + (lane >> 3);
            const int phys = lane & 7;
            const int logical = phys ^ (IS_B ? keyB(row) : keyA(row));
            int64_t grow = row0 + row;
            if (grow >= rows_valid) grow = rows_valid - 1;
            const bf16_t* src = g + grow * ld + k0 + logical * 8;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds_tile + inst * 1024), 16, 0, 0);
        }
    }
}

template <bool OUT_F32, bool ACCUM, int RT>
__global__ __launch_bounds__(512) void gemm_nt256_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, void* __restrict__ Cv,
                                                         const bf16_t* __restrict__ bias, const bf16_t* __restrict__ residual,
                                                         int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                                                         int64_t ldr, int tiles_m, int tiles_n) {
    constexpr int BMX = RT * 32;                       // 2 waves along M
    constexpr int A_BYTES = BMX * BK * 2, BUF_BYTES = A_BYTES + TILE2_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem2[];  // [buf][A | B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int nwg = tiles_m * tiles_n;
    int wgid;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int GROUP_M = 4;
    const int group = wgid / (GROUP_M * tiles_n);
    const int first_m = group * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int in_group = wgid - group * GROUP_M * tiles_n;
    const int tm = first_m + in_group % gsz;
    const int tn = in_group / gsz;
    const int64_t m0 = (int64_t)tm * BMX, n0 = (int64_t)tn * BN2;

    f32x4_t acc[RT][4];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = (int)(K / BK);
    stage_tile2<false, BMX>(A, lda, m0, M, 0, smem2, wave, lane);
    stage_tile2<true, BN2>(B, ldb, n0, N, 0, smem2 + A_BYTES, wave, lane);
    const int u = lane & 15, g = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        char* curA = smem2 + (kt & 1) * BUF_BYTES;
        char* curB = curA + A_BYTES;
        if (kt + 1 < nk) {
            char* nxtA = smem2 + ((kt + 1) & 1) * BUF_BYTES;
            stage_tile2<false, BMX>(A, lda, m0, M, (int64_t)(kt + 1) * BK, nxtA, wave, lane);
            stage_tile2<true, BN2>(B, ldb, n0, N, (int64_t)(kt + 1) * BK, nxtA + A_BYTES, wave, lane);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t xa[RT], wb[4];
            const int chunk = ks * 4 + g;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wn * 64 + (u >> 2) * 16 + j * 4 + (u & 3);
                wb[j] = *reinterpret_cast<const bf16x8_t*>(curB + row * 128 + ((chunk ^ keyB(row)) << 4));
            }
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const int row = wm * (RT * 16) + i * 16 + u;
                xa[i] = *reinterpret_cast<const bf16x8_t*>(curA + row * 128 + ((chunk ^ keyA(row)) << 4));
            }
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
        }
    }
    store_acc256<OUT_F32, ACCUM, RT>(acc, Cv, bias, residual, M, N, ldc, ldr, m0 + wm * (RT * 16), n0 + wn * 64, u, g);
}

// ------------------------------------------------------------------------------------------------------------------
// Phased ("ping-pong") form of the 8-wave kernel.  Same block tile, LDS image, swizzle, fragment mapping and epilogue as
// gemm_nt256_kernel, but the K loop is cut into 4 phases per 64-wide K-tile and the two wave groups (wm = 0 / 1: the two waves
// that share a SIMD) run one barrier apart, so while one wave of a SIMD issues its MFMAs the other one reads its next fragments
// from LDS and issues the HBM->LDS DMA for the tiles ahead - the matrix pipe never waits for LDS.
//   phase q of tile t:  [ds_read the A fragments of M-quarter q (phase 0: also all B fragments);  issue this phase's DMA rounds;
//                        s_waitcnt lgkmcnt(0)]  s_barrier  [MFMA quarter q x all 4 N fragments x 2 k-steps]  s_barrier
//   DMA schedule (one "round" = 8 KiB = 64 rows, one global_load_lds per thread; every wave issues the same number of rounds so
//   the counted vmcnt below means the same thing in every wave):
//       phases 0, 1 of tile t: the A rounds of tile t+1   (that buffer's A region was last read in phase 3 of tile t-1)
//       phases 2, 3 of tile t: the 4 B rounds of tile t+2 (tile t's B region is only read in phase 0)
//   phase 3 waits vmcnt(4) (the B rounds of t+2 stay in flight) BEFORE its first barrier; tile t+1 is first read one phase later.
//   Every ds_read is retired (lgkmcnt(0)) before the barrier that ends its load section, so a region may be restaged from the next
//   barrier interval on.  A tiles whose row count is not a multiple of 64 send the surplus half round to a 4 KiB junk area.
// ------------------------------------------------------------------------------------------------------------------
// cache policy bits of the weight-stream DMA loads of the decode kernels (0 = default, 2 = nt: read-once weights leave L2 first)
#ifndef TR1_W_AUX
#define TR1_W_AUX 2
#endif
#define TR1_PIN() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define TR1_BARRIER() do { TR1_PIN(); __builtin_amdgcn_s_barrier(); TR1_PIN(); } while (0)

// K-major ("NN") B operand: B is [K, N] row-major (the weight itself in dX = dY * W).  A round stages 16 k-rows x 256 columns (512 bytes
// per row, two rows per wave instruction); row r keeps its logical 16-byte chunk c at position c ^ keyKM(r), and the MFMA fragments are read
// back TRANSPOSED with ds_read_b64_tr_b16 (a 16-lane group reads a 4 x 16 block, each lane supplying an 8-byte address and receiving one
// column), so no W^T copy is ever built.  With 512-byte rows eight k-rows of a half-wave share their banks; the key spreads them over the
// four translates a 16-byte-granular swizzle can reach (2-way conflicts on these reads, 8 of the 24 fragment reads of a tile).
typedef __attribute__((ext_vector_type(4))) short gemm_s16x4_t;
TR1_DEV u32x2_t gemm_lds_read_tr16(const char* p) {
    const gemm_s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gemm_s16x4_t*)(p));
    return __builtin_bit_cast(u32x2_t, v);
}
TR1_DEV int keyKM(int krow) { return (krow & 1) | ((krow & 2) << 2); }
TR1_DEV void stage_round_km(const bf16_t* __restrict__ g, int64_t ld, int64_t n0, int64_t n_valid, int64_t k0, int64_t k_valid, char* lds_region,
                            int round, int wave, int lane) {
    const int inst = round * 8 + wave;                                   // 2 k-rows (1 KiB) per wave-instruction
    const int krow = inst * 2 + (lane >> 5);
    const int logical = (lane & 31) ^ keyKM(krow);
    int64_t col = n0 + logical * 8;
    if (col + 8 > n_valid) col = n_valid - 8;                            // columns past N: any valid chunk (their outputs are never stored)
    int64_t kr = k0 + krow;
    if (kr >= k_valid) kr = k_valid - 1;                                 // k rows past the operand (weight gradient: the token count is not a multiple
    //                                                                      of 64): re-read the last row - the A side carries zeros there
    __builtin_amdgcn_global_load_lds((gptr_t)(g + kr * ld + col), (lptr_t)(lds_region + inst * 1024), 16, 0, 0);
}

// Fused-epilogue forms of the phased NT kernel (EPI template argument of gemm_nt8p_kernel); extra operands travel in GemmEpi.
//   EPI 2 "gate/up + SwiGLU" (training / prefill / reference forward; TF:459-466 act_fn(gate_proj(x)) * up_proj(x)): B = [2I, K] (gate rows then up
//         rows).  A block's 256 B-tile rows are 4 x (32 gate rows | the 32 up rows of the SAME intermediate columns), so every wave holds both
//         halves of 32 columns and the epilogue writes a = silu(g) * u [M, I] (C) and, when the backward needs it, gu (ep.p0) from the same tile.
//   EPI 3 "down-projection dgrad + SwiGLU backward" (K-major form): C = dgu [M, 2I] = [da * u * silu'(g) | da * silu(g)] with gu read in the epilogue.
//   EPI 4 "q/k/v projection + bias + M-RoPE" (TF:501-504, :212-222): rotating tiles (q and k heads of 128) hold per wave 32 columns d and their
//         rotate-half partners d + 64; q goes to C, k to ep.p0 (the K cache rows), v (plain tiles) to ep.p1.
// All three round exactly where the unfused path rounds (GEMM output to bf16 first), so results are bit-identical to GEMM + elementwise kernel.
struct GemmEpi {
    void* p0; void* p1; const float* f0; const float* f1;
    int64_t ld0, ld1;
    int i0, i1;
};

// B-tile row (0..255) -> row of the stored weight for the fused forms; -1 = past the operand (clamped by the caller)
template <int EPI>
TR1_DEV int64_t epi_brow(int row, int64_t n0, int64_t N, const GemmEpi& ep) {
    if (EPI == 2) {
        const int64_t col = (n0 >> 1) + (row >> 6) * 32 + (row & 31);        // intermediate column
        const int64_t I = ep.i0;
        return ((row >> 5) & 1) * I + (col < I ? col : I - 1);
    }
    if (EPI == 4) {
        if (n0 < (int64_t)ep.i0 + ep.i1) return n0 + (row >> 7) * 128 + ((row >> 6) & 1) * 32 + ((row >> 5) & 1) * 64 + (row & 31);
        const int64_t r = n0 + row;
        return r < N ? r : N - 1;
    }
    if (EPI == 7) {          // vision q|k|v: ep.i0 = pairs per section (n_heads * half), ep.i1 = half (40); a tile = 128 consecutive pairs of one section
        const int tps = ep.i0 >> 7, tile = (int)(n0 >> 8);
        const int sec = tile / tps, pair = (tile - sec * tps) * 128 + (row >> 6) * 32 + (row & 31);
        const int head = pair / ep.i1, d = pair - head * ep.i1;
        return (int64_t)sec * 2 * ep.i0 + head * 2 * ep.i1 + d + ((row >> 5) & 1) * ep.i1;
    }
    const int64_t r = n0 + row;
    return r < N ? r : N - 1;
}

template <bool IS_B, int REGION_ROWS, int EPI = 0>
TR1_DEV void stage_round(const bf16_t* __restrict__ g, int64_t ld, int64_t row0, int64_t rows_valid, int64_t k0, char* lds_region,
                         char* junk, int round, int wave, int lane, const GemmEpi* ep = nullptr) {
    const int inst = round * 8 + wave;                                   // 8 rows (1 KiB) per wave-instruction
    int row = inst * 8 + (lane >> 3);
    char* dst = lds_region + inst * 1024;
    if (REGION_ROWS % 64 != 0 && inst * 8 >= REGION_ROWS) { dst = junk + (wave & 3) * 1024; row = REGION_ROWS - 8 + (lane >> 3); }   // wave-uniform
    const int logical = (lane & 7) ^ (IS_B ? keyB(row) : keyA(row));
    int64_t grow;
    if (IS_B && (EPI == 2 || EPI == 4 || EPI == 7)) grow = epi_brow<EPI>(row, row0, rows_valid, *ep);
    else { grow = row0 + row; if (grow >= rows_valid) grow = rows_valid - 1; }
    __builtin_amdgcn_global_load_lds((gptr_t)(g + grow * ld + k0 + logical * 8), (lptr_t)dst, 16, 0, 0);
}


template <int RT, int EPI>
TR1_DEV void store_acc256_pairs(const f32x4_t (&acc)[RT][4], char* __restrict__ wave_lds, void* __restrict__ Cv, const bf16_t* __restrict__ bias,
                                int64_t M, int64_t N, int64_t ldc, int64_t mrow0, int64_t n0, int wn, int lane, const GemmEpi& ep) {
    constexpr int CH = (RT % 2 == 0) ? 4 : 3;
    const int u = lane & 15, g = lane >> 4;
    const int c4 = lane & 3, q = (lane >> 2) & 3;
    const int rr16 = (q & 1) + (q >> 1) * 8 + 2 * (lane >> 4);
    // column bookkeeping of this lane's 8 pair columns
    int64_t col_a;              // EPI 2: intermediate column; EPI 4: column inside q (or k) of the "a" half (d < 64)
    bf16_t *dst_a, *dst_b, *dst_ga = nullptr, *dst_gb = nullptr;
    int64_t ld_o, ld_g = 0;
    bool col_ok;
    u32x4_t ba = {0, 0, 0, 0}, bb = {0, 0, 0, 0};
    int dcs = 0;                // EPI 4: index of the lane's first column into a row of the cos / sin tables
    if (EPI == 2) {
        const int64_t I = ep.i0;
        col_a = (n0 >> 1) + wn * 32 + c4 * 8;
        col_ok = col_a + 8 <= I;
        if (bias && col_ok) { ba = *reinterpret_cast<const u32x4_t*>(bias + col_a); bb = *reinterpret_cast<const u32x4_t*>(bias + I + col_a); }
        dst_a = reinterpret_cast<bf16_t*>(Cv) + col_a; dst_b = nullptr; ld_o = ldc;
        if (ep.p0) { dst_ga = reinterpret_cast<bf16_t*>(ep.p0) + col_a; dst_gb = dst_ga + I; ld_g = ep.ld0; }
    } else if (EPI == 7) {
        // Qwen2-VL / 2.5-VL vision attention (TF:225-248 apply_rotary_pos_emb_vision, head dim 80): section = q | k | v, the lane's 8 pair columns d0..d0+7
        // (< 40) of one head and their partners d0 + 40.  Outputs go to 128-wide PADDED heads: d -> head*128 + d, d + 40 -> head*128 + 48 + d (64 + d when half > 48), so the head-dim-128
        // attention kernels (32x32x16 MFMA, K / V row-major) take the tower; the pad columns are zero-filled once by the caller.
        const int tps = ep.i0 >> 7, tile = (int)(n0 >> 8);
        const int sec = tile / tps, pair0 = (tile - sec * tps) * 128 + wn * 32 + c4 * 8;
        const int head = pair0 / ep.i1, d0 = pair0 - head * ep.i1;
        const int64_t ncol = (int64_t)sec * 2 * ep.i0 + head * 2 * ep.i1 + d0;
        col_ok = true;
        dcs = sec < 2 ? d0 : -1;                                        // -1: v, no rotation
        if (bias) { ba = *reinterpret_cast<const u32x4_t*>(bias + ncol); bb = *reinterpret_cast<const u32x4_t*>(bias + ncol + ep.i1); }
        if (sec == 0) { dst_a = reinterpret_cast<bf16_t*>(Cv); ld_o = ldc; }
        else if (sec == 1) { dst_a = reinterpret_cast<bf16_t*>(ep.p0); ld_o = ep.ld0; }
        else { dst_a = reinterpret_cast<bf16_t*>(ep.p1); ld_o = ep.ld1; }
        dst_a += head * 128 + d0;
        dst_b = dst_a + (ep.i1 <= 48 ? 48 : 64);       // round 6: halves of <= 48 features sit 48 apart - the head's live features then end at 96 (tr1_attn_fwd_rows_live96)
        col_a = ncol;
    } else {
        const int d = (wn & 1) * 32 + c4 * 8;
        const int64_t ncol = n0 + (wn >> 1) * 128 + d;                  // column of the fused q|k|v projection
        col_ok = true;
        dcs = d;
        if (bias) { ba = *reinterpret_cast<const u32x4_t*>(bias + ncol); bb = *reinterpret_cast<const u32x4_t*>(bias + ncol + 64); }
        if (n0 < ep.i0) { dst_a = reinterpret_cast<bf16_t*>(Cv) + ncol; ld_o = ldc; }
        else { dst_a = reinterpret_cast<bf16_t*>(ep.p0) + (ncol - ep.i0); ld_o = ep.ld0; }
        dst_b = dst_a + 64;
        col_a = ncol;
    }
#pragma unroll
    for (int i0 = 0; i0 < RT; i0 += CH) {
        const int cnt = RT - i0 < CH ? RT - i0 : CH;
#pragma unroll
        for (int ii = 0; ii < CH; ++ii) {
            if (ii < cnt) {
                const int row = ii * 16 + u;
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4_t*>(wave_lds + row * 256 + (((g * 4 + j) ^ u) << 4)) = acc[i0 + ii][j];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r16 = 0; r16 < CH; ++r16) {
            if (r16 < cnt) {
                const int rr = r16 * 16 + rr16;
                const char* rowp = wave_lds + rr * 256;
                const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(rowp + (((2 * c4) ^ rr16) << 4));
                const f32x4_t a1 = *reinterpret_cast<const f32x4_t*>(rowp + (((2 * c4 + 1) ^ rr16) << 4));
                const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(rowp + (((8 + 2 * c4) ^ rr16) << 4));
                const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(rowp + (((9 + 2 * c4) ^ rr16) << 4));
                float va[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                float vb[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                const int64_t m = mrow0 + i0 * 16 + rr;
                if (m < M && col_ok) {
                    u32x4_t oa, ob;
                    if (EPI == 2) {
                        u32x4_t pg, pu;
                        if (bias) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { va[2 * e] += bflo(ba[e]); va[2 * e + 1] += bfhi(ba[e]); vb[2 * e] += bflo(bb[e]); vb[2 * e + 1] += bfhi(bb[e]); }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) { pg[e] = pack2bf(va[2 * e], va[2 * e + 1]); pu[e] = pack2bf(vb[2 * e], vb[2 * e + 1]); }
                        if (dst_ga) {
                            *reinterpret_cast<u32x4_t*>(dst_ga + m * ld_g) = pg;
                            *reinterpret_cast<u32x4_t*>(dst_gb + m * ld_g) = pu;
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {        // as swiglu_fwd_kernel: on the bf16-rounded gate / up, silu rounded to bf16 before the product
                            const float x0 = bf2f(f2bf(epi_silu(bflo(pg[e])))) * bflo(pu[e]);
                            const float x1 = bf2f(f2bf(epi_silu(bfhi(pg[e])))) * bfhi(pu[e]);
                            oa[e] = pack2bf(x0, x1);
                        }
                        *reinterpret_cast<u32x4_t*>(dst_a + m * ld_o) = oa;
                    } else {
                        const int rs = EPI == 7 ? ep.i1 : 64, dc = dcs < 0 ? 0 : dcs;
                        const float* cp = ep.f0 + m * rs + dc;
                        const float* sp = ep.f1 + m * rs + dc;
                        f32x4_t c0 = *reinterpret_cast<const f32x4_t*>(cp), c1 = *reinterpret_cast<const f32x4_t*>(cp + 4);
                        f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(sp), s1 = *reinterpret_cast<const f32x4_t*>(sp + 4);
                        if (EPI == 7 && dcs < 0) { c0 = c1 = (f32x4_t){1.f, 1.f, 1.f, 1.f}; s0 = s1 = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }    // v: x * 1 - y * 0 = x exactly
                        const float cc[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                        const float ss[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            va[2 * e] += bflo(ba[e]); va[2 * e + 1] += bfhi(ba[e]); vb[2 * e] += bflo(bb[e]); vb[2 * e + 1] += bfhi(bb[e]);
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) { va[e] = bf2f(f2bf(va[e])); vb[e] = bf2f(f2bf(vb[e])); }      // the projection output as the unfused path stores it
#pragma unroll
                        for (int e = 0; e < 4; ++e) {        // as rope_apply_kernel (forward)
                            oa[e] = pack2bf(va[2 * e] * cc[2 * e] - vb[2 * e] * ss[2 * e], va[2 * e + 1] * cc[2 * e + 1] - vb[2 * e + 1] * ss[2 * e + 1]);
                            ob[e] = pack2bf(vb[2 * e] * cc[2 * e] + va[2 * e] * ss[2 * e], vb[2 * e + 1] * cc[2 * e + 1] + va[2 * e + 1] * ss[2 * e + 1]);
                        }
                        *reinterpret_cast<u32x4_t*>(dst_a + m * ld_o) = oa;
                        *reinterpret_cast<u32x4_t*>(dst_b + m * ld_o) = ob;
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// EPI 3: the wave's 64 columns of da (K-major dgrad of the down projection) -> dgu columns n (gate half) and I + n (up half), as swiglu_bwd_kernel
template <int RT>
TR1_DEV void store_acc256_glubwd(const f32x4_t (&acc)[RT][4], char* __restrict__ wave_lds, void* __restrict__ Cv, int64_t M, int64_t N, int64_t ldc,
                                 int64_t mrow0, int64_t ncol0, int lane, const GemmEpi& ep) {
    constexpr int CH = (RT % 2 == 0) ? 4 : 3;
    const int u = lane & 15, g = lane >> 4;
    const int c8 = lane & 7;
    const int64_t n = ncol0 + c8 * 8, I = ep.i0;
    const bf16_t* gu = reinterpret_cast<const bf16_t*>(ep.p0);
    bf16_t* dguT = reinterpret_cast<bf16_t*>(ep.p1);          // optional second output: dgu^T [2I, ep.ld1] (ep.ld1 = tokens rounded up to 64, columns >= M zero)
    if (dguT && mrow0 + RT * 16 >= M && mrow0 + RT * 16 < ep.ld1 && ((mrow0 + RT * 16) % (RT * 32)) == 0) {
        // the tile rows end (a multiple of 32) short of the padded width (a multiple of 64): the lower wave row of the last row tile zero-fills the 32 columns left
        const int64_t mz = mrow0 + RT * 16;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 64 + lane, col = idx >> 2, ch = idx & 3, part = col >> 6, ncol = col & 63;
            if (ncol0 + ncol < N && mz + ch * 8 < ep.ld1)
                *reinterpret_cast<u32x4_t*>(dguT + ((int64_t)part * I + ncol0 + ncol) * ep.ld1 + mz + ch * 8) = u32x4_t{0, 0, 0, 0};
        }
    }
#pragma unroll
    for (int i0 = 0; i0 < RT; i0 += CH) {
        const int cnt = RT - i0 < CH ? RT - i0 : CH;
#pragma unroll
        for (int ii = 0; ii < CH; ++ii) {
            if (ii < cnt) {
                const int row = ii * 16 + u;
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4_t*>(wave_lds + row * 256 + (((g * 4 + j) ^ u) << 4)) = acc[i0 + ii][j];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r8 = 0; r8 < CH * 2; ++r8) {
            if (r8 < cnt * 2) {
                const int rr = r8 * 8 + (lane >> 3);
                const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(wave_lds + rr * 256 + (((2 * c8) ^ (rr & 15)) << 4));
                const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(wave_lds + rr * 256 + (((2 * c8 + 1) ^ (rr & 15)) << 4));
                const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                if (dguT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (both reads of every lane have returned before the row is rewritten below)
                const int64_t m = mrow0 + i0 * 16 + rr;
                u32x4_t og = {0, 0, 0, 0}, ou = {0, 0, 0, 0};
                if (m < M && n + 8 <= N) {
                    const u32x4_t gg = *reinterpret_cast<const u32x4_t*>(gu + m * ep.ld0 + n);
                    const u32x4_t uu = *reinterpret_cast<const u32x4_t*>(gu + m * ep.ld0 + I + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gv[2] = {bflo(gg[e]), bfhi(gg[e])}, uv[2] = {bflo(uu[e]), bfhi(uu[e])};
                        const float dv[2] = {bf2f(f2bf(v[2 * e])), bf2f(f2bf(v[2 * e + 1]))};      // da as the unfused dgrad stores it
                        float rg[2], ru[2];
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const float sg = 1.f / (1.f + __expf(-gv[k]));
                            const float si = gv[k] * sg;
                            rg[k] = dv[k] * uv[k] * (sg * (1.f + gv[k] * (1.f - sg)));
                            ru[k] = dv[k] * si;
                        }
                        og[e] = pack2bf(rg[0], rg[1]); ou[e] = pack2bf(ru[0], ru[1]);
                    }
                    bf16_t* cp = reinterpret_cast<bf16_t*>(Cv) + m * ldc + n;
                    *reinterpret_cast<u32x4_t*>(cp) = og;
                    *reinterpret_cast<u32x4_t*>(cp + I) = ou;
                }
                if (dguT) {
                    // the row's fp32 values have been read (its 8 lanes cover all 16 chunks, LDS operations of a wave retire in order): its 256 bytes now
                    // hold the bf16 results (zeros for rows >= M: the transposed copy is zero-padded), gate chunk c8 at ((row & 1) << 3 | c8 ^ (row / 8 % 8)), up in
                    // the other half - conflict-free for these row writes (two rows per quarter wave use opposite halves) and for the column reads below
                    const int k3 = (rr >> 3) & 7, hb = rr & 1;
                    *reinterpret_cast<u32x4_t*>(wave_lds + rr * 256 + (((hb << 3) | (c8 ^ k3)) << 4)) = og;
                    *reinterpret_cast<u32x4_t*>(wave_lds + rr * 256 + ((((hb ^ 1) << 3) | (c8 ^ k3)) << 4)) = ou;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (dguT) {
            // transposed read-back: lane -> (column, 8-row chunk); the 8 lanes of a column write 128 contiguous bytes of dgu^T (8 row chunks = 64 rows)
            const int rc = lane & 7;
            const int64_t mchunk = mrow0 + i0 * 16 + rc * 8;
            if (rc < cnt * 2 && mchunk < ep.ld1) {
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const int part = it >> 3, ncol = (it & 7) * 8 + (lane >> 3);
                    if (ncol0 + ncol < N) {
                        unsigned short t[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int row = rc * 8 + e;
                            t[e] = *reinterpret_cast<const unsigned short*>(wave_lds + row * 256 + ((((part ^ (row & 1)) << 3) | ((ncol >> 3) ^ rc)) << 4) + (ncol & 7) * 2);
                        }
                        const u32x4_t w = {t[0] | ((unsigned)t[1] << 16), t[2] | ((unsigned)t[3] << 16), t[4] | ((unsigned)t[5] << 16), t[6] | ((unsigned)t[7] << 16)};
                        *reinterpret_cast<u32x4_t*>(dguT + ((int64_t)part * I + ncol0 + ncol) * ep.ld1 + mchunk) = w;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}

template <bool OUT_F32, bool ACCUM, int RT, bool BKM = false, int EPI = 0>
__global__ __launch_bounds__(512) void gemm_nt8p_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, void* __restrict__ Cv,
                                                        const bf16_t* __restrict__ bias, const bf16_t* __restrict__ residual,
                                                        int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                                                        int64_t ldr, int tiles_m, int tiles_n, const GemmEpi ep) {
    constexpr int BMX = RT * 32;
    constexpr int A_BYTES = BMX * BK * 2, BUF_BYTES = A_BYTES + TILE2_BYTES;
    constexpr int AR = (BMX + 63) / 64;                // A rounds per K-tile; phases 0 / 1 issue AR0 / AR1 of them
    constexpr int AR0 = (AR + 1) / 2;
    constexpr int Q0 = 0, Q1 = (RT + 3) / 4, Q2 = Q1 + (RT + 2) / 4, Q3 = Q2 + (RT + 1) / 4, Q4 = RT;   // M-quarters (m-tile ranges)
    constexpr int QMAX = Q1 - Q0;
    extern __shared__ __attribute__((aligned(16))) char smem2[];  // [buf][A | B] + junk
    char* const junk = smem2 + 2 * BUF_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int nwg = tiles_m * tiles_n;
    int wgid;
    {
        int b = blockIdx.x;
        if (EPI == 6) {                                  // S-way split-K (S = ep.i0): the grid holds every tile S times; copy kz reduces its share of the
            const int kz = b / nwg;                      // K tiles into its own fp32 plane of C (summed by splitk_reduce_kernel in a fixed order)
            b -= kz * nwg;
            const int nk_all = (int)(K / BK), q = nk_all / ep.i0, r = nk_all - q * ep.i0;
            const int64_t k0 = (int64_t)(kz * q + (kz < r ? kz : r)) * BK;
            K = (int64_t)(q + (kz < r ? 1 : 0)) * BK;
            A += k0;
            B += BKM ? k0 * ldb : k0;
            Cv = reinterpret_cast<float*>(Cv) + (int64_t)kz * ep.ld0;
        }
        const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int GROUP_M = 4;
    const int group = wgid / (GROUP_M * tiles_n);
    const int first_m = group * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int in_group = wgid - group * GROUP_M * tiles_n;
    const int tm = first_m + in_group % gsz;
    const int tn = in_group / gsz;
    const int64_t m0 = (int64_t)tm * BMX, n0 = (int64_t)tn * BN2;

    f32x4_t acc[RT][4];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = (int)(K / BK);
    const int u = lane & 15, g = lane >> 4;
    // per-lane LDS byte offsets of the fragment reads (k-step 1 = k-step 0 with chunk bit 2 flipped: offset ^ 64)
    const int a_off = (wm * (RT * 16) + u) * 128 + ((g ^ ((u >> 1) & 7)) << 4);
    const int b_off = (wn * 64 + (u >> 2) * 16 + (u & 3)) * 128 + ((g ^ (((u >> 2) << 1) | ((u >> 1) & 1))) << 4);
    // BKM fragment j of k step ks: lane L of a 16-lane group supplies row (L >> 2) of a 4-row block, 8 bytes at columns wn*64 + (L & 3)*16 + j*4.
    // Lane group g takes the blocks of k rows g*8 .. +3 and g*8 + 4 .. +7, i.e. the standard k order of the MFMA, so the A side is unchanged.
    const int bkm_row = g * 8 + (u >> 2);                                   // + ks*32 (+4 for the second half)
    const int bkm_key = keyKM(bkm_row);                                     // depends on the row's low two bits only
    const int64_t bkm_kvalid = (BKM && ldr > 0) ? ldr : K;                  // K-major B: `ldr` carries the number of valid k rows of B (no residual in this form)

#define STAGE_A(t, r) stage_round<false, BMX>(A, lda, m0, M, (int64_t)(t) * BK, smem2 + ((t) & 1) * BUF_BYTES, junk, (r), wave, lane)
#define STAGE_B(t, r) do { if (BKM) stage_round_km(B, ldb, n0, N, (int64_t)(t) * BK, bkm_kvalid, smem2 + ((t) & 1) * BUF_BYTES + A_BYTES, (r), wave, lane); \
                           else stage_round<true, BN2, EPI>(B, ldb, n0, N, (int64_t)(t) * BK, smem2 + ((t) & 1) * BUF_BYTES + A_BYTES, junk, (r), wave, lane, &ep); } while (0)
    // prologue: tile 0 complete, B of tile 1 in flight
#pragma unroll
    for (int r = 0; r < 4; ++r) STAGE_B(0, r);
#pragma unroll
    for (int r = 0; r < AR; ++r) STAGE_A(0, r);
    if (nk > 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) STAGE_B(1, r);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    TR1_BARRIER();
    if (wm == 1) TR1_BARRIER();                        // the second wave group runs one barrier interval behind the first

    bf16x8_t bf[4][2], af[QMAX][2];
#define LOAD_A(QA, QB) do {                                                                                    \
        _Pragma("unroll") for (int i = (QA); i < (QB); ++i) {                                                  \
            af[i - (QA)][0] = *reinterpret_cast<const bf16x8_t*>(curA + a_off + i * 2048);                     \
            af[i - (QA)][1] = *reinterpret_cast<const bf16x8_t*>(curA + (a_off ^ 64) + i * 2048);              \
        } } while (0)
#define COMPUTE(QA, QB) do {                                                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
        TR1_BARRIER();                                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                         \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                       \
        _Pragma("unroll") for (int i = (QA); i < (QB); ++i)                                                    \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                          \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][ks], af[i - (QA)][ks], acc[i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                         \
        TR1_BARRIER();                                                                                         \
    } while (0)

    for (int t = 0; t < nk; ++t) {
        const char* curA = smem2 + (t & 1) * BUF_BYTES;
        const char* curB = curA + A_BYTES;
        // ---- phase 0: all B fragments + A quarter 0; A rounds [0, AR0) of tile t+1
        if (BKM) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int lc = wn * 8 + (u & 3) * 2 + (j >> 1);                                    // logical 16-byte chunk of the piece
                    const char* pb = curB + (ks * 32 + bkm_row) * 512 + ((lc ^ bkm_key) << 4) + (j & 1) * 8;
                    const u32x2_t h0 = gemm_lds_read_tr16(pb), h1 = gemm_lds_read_tr16(pb + 4 * 512);    // rows +4: same key (low two bits unchanged)
                    u32x4_t w = {h0[0], h0[1], h1[0], h1[1]};
                    bf[j][ks] = __builtin_bit_cast(bf16x8_t, w);
                }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf[j][0] = *reinterpret_cast<const bf16x8_t*>(curB + b_off + j * 512);
                bf[j][1] = *reinterpret_cast<const bf16x8_t*>(curB + (b_off ^ 64) + j * 512);
            }
        }
        LOAD_A(Q0, Q1);
        if (t + 1 < nk) {
#pragma unroll
            for (int r = 0; r < AR0; ++r) STAGE_A(t + 1, r);
        }
        COMPUTE(Q0, Q1);
        // ---- phase 1: A quarter 1; the remaining A rounds of tile t+1
        LOAD_A(Q1, Q2);
        if (t + 1 < nk) {
#pragma unroll
            for (int r = AR0; r < AR; ++r) STAGE_A(t + 1, r);
        }
        COMPUTE(Q1, Q2);
        // ---- phase 2: A quarter 2; B rounds 0, 1 of tile t+2 (tile t's B region is free: read in phase 0 only)
        LOAD_A(Q2, Q3);
        if (t + 2 < nk) { STAGE_B(t + 2, 0); STAGE_B(t + 2, 1); }
        COMPUTE(Q2, Q3);
        // ---- phase 3: A quarter 3; B rounds 2, 3 of tile t+2; tile t+1 must have landed before the next barrier
        LOAD_A(Q3, Q4);
        if (t + 2 < nk) {
            STAGE_B(t + 2, 2); STAGE_B(t + 2, 3);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        COMPUTE(Q3, Q4);
    }
    if (wm == 0) TR1_BARRIER();
#undef LOAD_A
#undef COMPUTE
#undef STAGE_A
#undef STAGE_B
#ifdef TR1_PROBE_NO_EPI      // measurement build (tools/build_variant.py noepi -DTR1_PROBE_NO_EPI=1): the tile ends here, nothing is stored - what the epilogue costs
    if (K > 0) return;
#endif
#if TR1_EPI_LDS
    // every wave is past its last LDS read (the realignment barrier above): the operand buffers become 8 private staging slices
    if (EPI == 2 || EPI == 7 || (EPI == 4 && n0 < (int64_t)ep.i0 + ep.i1))
        store_acc256_pairs<RT, EPI>(acc, smem2 + wave * (((RT % 2 == 0) ? 4 : 3) * 4096), Cv, bias, M, N, ldc, m0 + wm * (RT * 16), n0, wn, lane, ep);
    else if (EPI == 4)       // V tile: bias only, into its own buffer
        store_acc256_lds<false, false, RT, 0>(acc, smem2 + wave * (((RT % 2 == 0) ? 4 : 3) * 4096), ep.p1, bias + ep.i0 + ep.i1, nullptr, M, N - ep.i0 - ep.i1,
                                              ep.ld1, 0, m0 + wm * (RT * 16), n0 - ep.i0 - ep.i1 + wn * 64, lane);
    else if (EPI == 6)
        store_acc256_lds<true, false, RT, 0>(acc, smem2 + wave * (((RT % 2 == 0) ? 4 : 3) * 4096), Cv, nullptr, nullptr, M, N, ldc, 0, m0 + wm * (RT * 16), n0 + wn * 64, lane);
    else if (EPI == 3)
        store_acc256_glubwd<RT>(acc, smem2 + wave * (((RT % 2 == 0) ? 4 : 3) * 4096), Cv, M, N, ldc, m0 + wm * (RT * 16), n0 + wn * 64, lane, ep);
    else
    store_acc256_lds<OUT_F32, ACCUM, RT, EPI>(acc, smem2 + wave * (((RT % 2 == 0) ? 4 : 3) * 4096), Cv, bias, residual, M, N, ldc, ldr,
                                              m0 + wm * (RT * 16), n0 + wn * 64, lane,
                                              (OUT_F32 && EPI == 0 && ep.p0) ? reinterpret_cast<float*>(ep.p0) + (int64_t)blockIdx.x * 8 + wave : nullptr,
                                              (OUT_F32 && EPI == 0 && ep.p0) ? reinterpret_cast<bf16_t*>(ep.p1) : nullptr, ep.ld1);
#else
    store_acc256<OUT_F32, ACCUM, RT>(acc, Cv, bias, residual, M, N, ldc, ldr, m0 + wm * (RT * 16), n0 + wn * 64, u, g);
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// Skinny GEMM for the decode regime (M <= 64 rows, one new token per rollout row): out[M,N] = x[M,K] * W[N,K]^T.
// HBM-bound weight streaming: every W element is read exactly once, straight from global memory into the MFMA A
// fragment (no LDS: the operand is not shared between waves).  A block owns 16 output columns; its 4 waves split K
// and the partial 16x16 tiles are reduced through LDS.  Each lane fetches 32 contiguous bytes of one W row per
// step, so a 16-lane group covers one full 128-byte line per row; the k-order inside the MFMA is permuted the same
// way for x (any k permutation is legal as long as A and B agree).
// ------------------------------------------------------------------------------------------------------------------
// Optional block-timeline probe (tools/probe_skinny.hip compiles this file with -DTR1_PROBE): wall_clock64() at block entry, after
// the k loop and at block exit, 4 slots per block.  Not compiled into the library.
#ifdef TR1_PROBE
__device__ unsigned long long* tr1_probe = nullptr;
#define TR1_PROBE_AT(slot) do { if (tr1_probe && threadIdx.x == 0) tr1_probe[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (slot)] = wall_clock64(); } while (0)
#else
#define TR1_PROBE_AT(slot) do { } while (0)
#endif

// NCOL: 16-column groups per wave.  The x (activation) fragment is loaded once per k-step and reused for NCOL weight fragments, so
// the L2 traffic for x drops from 1x to 1/NCOL of the weight stream (matters at M = 16, where x is as large as a block's W slab).
// MG: 16-row groups of x (M <= 16*MG): every weight fragment fetched from HBM feeds MG MFMAs, so batching more rollout rows into one
// decode step (G = 16, or several prompts of a gradient-accumulation window) keeps the single pass over the weights.
// XLDS (round 3; MG = 1, no cross-block split-K): x reaches the MFMA through ONE DMA copy into LDS per block instead of per-wave vector loads - see the
// note at norm_gemm_skinny_kernel (the L1 tag pipe looks up 64 pieces per 1 KiB load in the operand layout; at NCOL = 1 half of the loads were x).
template <int WAVES, int UNROLL, int NCOL, int MG, bool XLDS = false>
__global__ __launch_bounds__(WAVES * 64) void gemm_skinny_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W, bf16_t* __restrict__ C,
                                                                 float* __restrict__ Cf32, const bf16_t* __restrict__ bias,
                                                                 const bf16_t* __restrict__ residual, int M, int64_t N, int64_t K, int64_t ldx,
                                                                 int64_t ldw, int64_t ldc, int64_t ldr, float* __restrict__ fix_ws,
                                                                 int* __restrict__ fix_cnt) {
    __shared__ __attribute__((aligned(16))) float red[WAVES][NCOL][MG][16][17];
    __shared__ int s_ticket;
    extern __shared__ __attribute__((aligned(1024))) char sk_xs[];             // XLDS: [K/64 segments][16 rows][128 bytes]
    static_assert(!XLDS || MG == 1, "the LDS copy of x holds 16 rows");
    TR1_PROBE_AT(0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u = lane & 15, g = lane >> 4;
    const int64_t n0 = (int64_t)blockIdx.x * 16 * NCOL;
    // lane (u,g) takes k = g*8.. and 32+g*8.. of every 64-element step: a 16-lane group reads 64 contiguous bytes of a W row per load
    const bf16_t* wp[NCOL];
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
        int64_t wrow = n0 + c * 16 + u; if (wrow >= N) wrow = N - 1;
        wp[c] = W + wrow * ldw + g * 8;
    }
    // rows >= M of the MFMA B operand are padding: they re-read row M-1 (unconditional loads keep the k loop branch-free, so the
    // compiler can count the loads in flight instead of draining them with vmcnt(0)); their output columns are never stored
    const bf16_t* xp[MG];
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) xp[mg] = X + (int64_t)(mg * 16 + u < M ? mg * 16 + u : (M - 1)) * ldx + g * 8;
    // gridDim.y > 1: cross-block split-K - block (x, y) covers k-steps [kb, ke); the partial tiles are merged by the in-kernel fixup below
    const int64_t nsteps_all = K / 64;
    const int64_t per_split = (nsteps_all + gridDim.y - 1) / gridDim.y;
    const int64_t kb = (int64_t)blockIdx.y * per_split;
    int64_t ke = kb + per_split; if (ke > nsteps_all) ke = nsteps_all;
    const int64_t nsteps = ke > kb ? ke - kb : 0;
    const int64_t s_per = (nsteps + WAVES - 1) / WAVES;
    const int64_t s0 = kb + wave * s_per;
    int64_t s1 = s0 + s_per; if (s1 > ke) s1 = ke;
    f32x4_t acc[NCOL][MG][2];
#pragma unroll
    for (int c = 0; c < NCOL; ++c)
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) { acc[c][mg][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; acc[c][mg][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
    const unsigned xs_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)sk_xs;
    if (XLDS) {       // inline asm: after the builtin hipcc would drain the weight stream (vmcnt(0)) in front of every LDS read
        const int r8 = lane >> 3;
        const int n_inst = (int)(K >> 5);                                     // 1 KiB per instruction
        for (int i = __builtin_amdgcn_readfirstlane(wave); i < n_inst; i += WAVES) {
            const int r = (i & 1) * 8 + r8;
            const unsigned off = (unsigned)((r < M ? r : M - 1) * (int)ldx + (i >> 1) * 64 + (((lane & 7) ^ keyA(r)) << 3)) * 2u;
            const unsigned dst = xs_base + (unsigned)i * 1024u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(off), "s"(X) : "memory", "m0");
        }
    }
    const unsigned xs_lane = xs_base + (unsigned)(u * 128);
    const int xs_key = keyA(u);
    // Rotating software pipeline over UNROLL k-step buffers: a buffer is refilled (next k-step, UNROLL ahead) right after its MFMAs are
    // issued.  Measured neutral against the batch form (issue UNROLL steps, drain, repeat): hipcc still drains the queue once per trip
    // (s_waitcnt vmcnt(1)/vmcnt(0) at the loop header), and the N sweep of tools/probe_skinny.hip shows the kernel already at
    // t = 7 us + bytes / 5.4-5.8 TB/s, i.e. within ~15 % of what this access pattern streams at any size.
    bf16x8_t wa[UNROLL][NCOL][2], xa[UNROLL][MG][2];
#define SK_LOAD(q, st)                                                                                   \
    do {                                                                                                 \
        const int64_t k__ = (st) * 64;                                                                   \
        _Pragma("unroll") for (int c = 0; c < NCOL; ++c) {                                               \
            wa[q][c][0] = *reinterpret_cast<const bf16x8_t*>(wp[c] + k__);                               \
            wa[q][c][1] = *reinterpret_cast<const bf16x8_t*>(wp[c] + k__ + 32);                          \
        }                                                                                                \
        if (!XLDS) { _Pragma("unroll") for (int mg = 0; mg < MG; ++mg) {                                 \
            xa[q][mg][0] = *reinterpret_cast<const bf16x8_t*>(xp[mg] + k__);                             \
            xa[q][mg][1] = *reinterpret_cast<const bf16x8_t*>(xp[mg] + k__ + 32);                        \
        } }                                                                                              \
    } while (0)
#define SK_MFMA(q, st)                                                                                                        \
    do {                                                                                                                      \
        if (XLDS) {                                                                                                           \
            typedef const __attribute__((address_space(3))) bf16x8_t* xs_ptr_t;                                               \
            const unsigned xa__ = xs_lane + (unsigned)(st) * 2048u;                                                           \
            xa[q][0][0] = *(xs_ptr_t)(uintptr_t)(xa__ + (unsigned)(((0 + g) ^ xs_key) << 4));                                 \
            xa[q][0][1] = *(xs_ptr_t)(uintptr_t)(xa__ + (unsigned)(((4 + g) ^ xs_key) << 4));                                 \
        }                                                                                                                     \
        _Pragma("unroll") for (int c = 0; c < NCOL; ++c)                                                                      \
            _Pragma("unroll") for (int mg = 0; mg < MG; ++mg) {                                                               \
                acc[c][mg][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[q][c][0], xa[q][mg][0], acc[c][mg][0], 0, 0, 0);   \
                acc[c][mg][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[q][c][1], xa[q][mg][1], acc[c][mg][1], 0, 0, 0);   \
            }                                                                                                                 \
    } while (0)
    int64_t s = s0;
#pragma unroll
    for (int q = 0; q < UNROLL; ++q)
        if (s0 + q < s1) SK_LOAD(q, s0 + q);
    if (XLDS) {       // the x copy has landed for this wave when only the UNROLL * 2 NCOL younger register loads are still in flight
        static_assert(!XLDS || UNROLL * 2 * NCOL == 8, "vmcnt below is written for 8 register loads in the prologue");
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    for (; s + 2 * UNROLL <= s1; s += UNROLL) {          // steady state: branch-free
#pragma unroll
        for (int q = 0; q < UNROLL; ++q) { SK_MFMA(q, s + q); SK_LOAD(q, s + q + UNROLL); }
    }
#pragma unroll
    for (int q = 0; q < UNROLL; ++q)
        if (s + q < s1) { SK_MFMA(q, s + q); if (s + q + UNROLL < s1) SK_LOAD(q, s + q + UNROLL); }
    s += UNROLL;
#pragma unroll
    for (int q = 0; q < UNROLL; ++q)
        if (s + q < s1) SK_MFMA(q, s + q);
#undef SK_LOAD
#undef SK_MFMA
    // D[row = n index (g*4+r)][col = m (u)]
#pragma unroll
    for (int c = 0; c < NCOL; ++c)
#pragma unroll
        for (int mg = 0; mg < MG; ++mg)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][c][mg][u][g * 4 + r] = acc[c][mg][0][r] + acc[c][mg][1][r];
    TR1_PROBE_AT(1);
    __syncthreads();
    TR1_PROBE_AT(2);
    constexpr int TILE = NCOL * MG * 256;
    if (fix_cnt) {
        // Cross-block split-K with in-kernel fixup: every block of a column group parks its fp32 partial tile in L2-resident scratch and
        // takes a ticket; the block that draws the last ticket sums the gridDim.y tiles in slab order (deterministic), applies
        // bias / residual and writes C.  The tiles are written and read with device-scope (write-through / L2-bypassing) accesses, so
        // they are visible across XCDs (per-XCD L2) without flushing the caches.
        float* mine = fix_ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * TILE;
        for (int i = threadIdx.x; i < TILE; i += WAVES * 64) {
            const int c = i / (MG * 256), mg = (i >> 8) % MG, mm = (i >> 4) & 15, nn = i & 15;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) v += red[w][c][mg][mm][nn];
            __hip_atomic_store(mine + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // device-scope write-through: no L2 flush needed
        }
        // the tile stores above are complete (acknowledged at device scope) before the ticket is drawn; a full agent-scope fence
        // here would write back and invalidate the whole L2 and evict x for every other block (measured: 38 -> 55 us)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(&fix_cnt[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s_ticket != (int)gridDim.y - 1) return;
        for (int i = threadIdx.x; i < TILE; i += WAVES * 64) {
            const int c = i / (MG * 256), mg = (i >> 8) % MG, mm = (i >> 4) & 15, nn = i & 15;
            const int m = mg * 16 + mm;
            const int64_t n = n0 + c * 16 + nn;
            float v = 0.f;
            if (gridDim.y == 4) {       // unrolled: the four device-scope loads in flight together; same sum in the same order (see gemm_skinny_lds_fix_kernel)
                float t[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    t[ks] = __hip_atomic_load(fix_ws + ((int64_t)ks * gridDim.x + blockIdx.x) * TILE + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v = ((t[0] + t[1]) + t[2]) + t[3];
            } else if (gridDim.y <= 16) {   // up to 16 slabs: every load in flight before the first add, summed in slab order (what the loop below computes)
                float t[16];
#pragma unroll
                for (int ks = 0; ks < 16; ++ks)
                    t[ks] = ks < (int)gridDim.y ? __hip_atomic_load(fix_ws + ((int64_t)ks * gridDim.x + blockIdx.x) * TILE + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
                v = t[0];
#pragma unroll
                for (int ks = 1; ks < 16; ++ks) if (ks < (int)gridDim.y) v += t[ks];
            } else {
                for (int ks = 0; ks < (int)gridDim.y; ++ks)
                    v += __hip_atomic_load(fix_ws + ((int64_t)ks * gridDim.x + blockIdx.x) * TILE + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (m < M && n < N) {
                if (bias) v += bf2f(bias[n]);
                if (residual) v += bf2f(residual[(int64_t)m * ldr + n]);
                if (Cf32) Cf32[(int64_t)m * ldc + n] = v;
                else C[(int64_t)m * ldc + n] = f2bf(v);
            }
        }
        if (threadIdx.x == 0) fix_cnt[blockIdx.x] = 0;      // re-armed for the next launch (ordered by the kernel boundary)
        return;
    }
    for (int i = threadIdx.x; i < TILE; i += WAVES * 64) {   // (column group, row group, m, n)
        const int c = i / (MG * 256), mg = (i >> 8) % MG, mm = (i >> 4) & 15, nn = i & 15;
        const int m = mg * 16 + mm;
        const int64_t n = n0 + c * 16 + nn;
        if (m < M && n < N) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) v += red[w][c][mg][mm][nn];
            if (bias) v += bf2f(bias[n]);
            if (residual) v += bf2f(residual[(int64_t)m * ldr + n]);
            if (Cf32) Cf32[(int64_t)m * ldc + n] = v;
            else C[(int64_t)m * ldc + n] = f2bf(v);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Decode-step fusion: out = rmsnorm(x; lnw) @ W^T with the normalisation folded into the GEMM (every block streams all of x anyway):
//   prologue  x'[m,k] = bf16(x[m,k] * lnw[k]) feeds the MFMA while sum_k x^2 is accumulated per row; the epilogue scales row m by
//             rstd[m] = rsqrt(mean x^2 + eps)  (rmsnorm is linear after the row scale, so the GEMM commutes with it)
//   GLU       the block's two column groups are gate rows n0.. and up rows up_off+n0.. of W; the epilogue writes
//             silu(gate) * up (Qwen2MLP TF:459-466), so the [M, 2I] intermediate never reaches HBM.
// Saves the separate rmsnorm / swiglu launches of a decode layer (each ~6-7 us of pure launch + fill/drain at M <= 64 rows).
// ------------------------------------------------------------------------------------------------------------------
TR1_DEV float silu_f32(float x) { return x / (1.f + __expf(-x)); }
TR1_DEV bf16x8_t scale_frag_sumsq(bf16x8_t x, bf16x8_t w, float& ss) {
    const u32x4_t xu = __builtin_bit_cast(u32x4_t, x), wu = __builtin_bit_cast(u32x4_t, w);
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a = bflo(xu[e]), b = bfhi(xu[e]);
        ss = fmaf(a, a, fmaf(b, b, ss));
        o[e] = pack2bf(a * bflo(wu[e]), b * bfhi(wu[e]));
    }
    return __builtin_bit_cast(bf16x8_t, o);
}

#ifdef TR1_PROBE
// block timeline of the fused QKV launch (tools/bench_qkv32.py PROBE=1 against tools/_probe_lib.so): s_memtime at entry / first loads issued / stream
// consumed / partials reduced (after the barrier) / epilogue stored, for waves 0 and WAVES-1 of every block; written once, at the very end
__device__ unsigned long long* tr1_qkv_probe = nullptr;
extern "C" int probe_qkv_set_ptr(void* ptr) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(tr1_qkv_probe), &ptr, sizeof(ptr)); }
#define QKV_STAMPS unsigned long long qs_[6] = {0, 0, 0, 0, 0, 0}
#define QKV_STAMP(i) do { if (QKV) qs_[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define QKV_DUMP() do { if (QKV && tr1_qkv_probe && lane == 0 && (wave == 0 || wave == WAVES - 1)) { \
    _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) tr1_qkv_probe[((int64_t)blockIdx.x * 2 + (wave ? 1 : 0)) * 8 + i_] = qs_[i_]; } } while (0)
#else
#define QKV_STAMPS do { } while (0)
#define QKV_STAMP(i) do { } while (0)
#define QKV_DUMP() do { } while (0)
#endif
// QKV: the block's two column groups are columns (d, d + hd/2) of one head and the epilogue is qkv_epilogue_store (RoPE + cache append).
// XLDS (round 3, MG = 1): the activation rows do not travel through the vector-memory path at all.  In the MFMA operand layout a wave load touches 16
// rows x 64 bytes = 64 separate (line, 16-byte) pieces, and the L1 tag pipe looks them up one per cycle: a 1 KiB load instruction costs ~64 cycles
// (measured 38-54 GB/s per CU in this kernel, block timeline in DESIGN.md), and half of this kernel's load instructions were x and lnw.  With XLDS the
// block copies x ONCE into LDS by DMA (8 rows x 128 bytes per instruction = 8 lines; 16 rows x K, swizzled on the source address like every other
// tile here) and reads its fragments with ds_read_b128; only the weights stay on the register path.  Same values, same order of operations.
template <int WAVES, int UNROLL, int MG, bool GLU, int NCOL = 2, bool QKV = false, bool XLDS = false>
__global__ __launch_bounds__(WAVES * 64) void norm_gemm_skinny_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ lnw,
                                                                      const bf16_t* __restrict__ W, bf16_t* __restrict__ C,
                                                                      const bf16_t* __restrict__ bias, int M, int64_t N, int64_t K, int64_t ldx,
                                                                      int64_t ldw, int64_t ldc, float eps, int64_t up_off, QkvEpi qe = QkvEpi{}) {
    static_assert(!GLU || NCOL % 2 == 0, "GLU pairs NCOL/2 gate column groups with NCOL/2 up column groups");
    static_assert(!QKV || (NCOL == 2 && !GLU), "QKV pairs the two rotate-half column groups of a head");
    static_assert(!XLDS || MG == 1, "the LDS copy of x holds 16 rows");
    extern __shared__ __attribute__((aligned(1024))) char ng_xs[];             // XLDS: [K/64 segments][16 rows][128 bytes]
    __shared__ __attribute__((aligned(16))) float red[WAVES][NCOL][MG][16][17];
    __shared__ float ssred[WAVES][MG][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    QKV_STAMPS;
    QKV_STAMP(0);
    const int u = lane & 15, g = lane >> 4;
    const int qkv_gph = QKV ? qe.hd >> 5 : 1;                                   // 16-column group pairs per head
    const int qkv_h = QKV ? (int)blockIdx.x / qkv_gph : 0, qkv_j = QKV ? (int)blockIdx.x % qkv_gph : 0;
    constexpr int OG = GLU ? NCOL / 2 : (QKV ? 1 : NCOL);                       // output column groups per block
    const int64_t n0 = QKV ? (int64_t)qkv_h * qe.hd + qkv_j * 16 : (int64_t)blockIdx.x * (16 * OG);
    const bf16_t* wp[NCOL];
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
        int64_t wrow = QKV ? n0 + c * (qe.hd >> 1) + u : (GLU ? n0 + (c % OG) * 16 + u : n0 + c * 16 + u);
        if (wrow >= N) wrow = N - 1;
        if (GLU && c >= OG) wrow += up_off;
        wp[c] = W + wrow * ldw + g * 8;
    }
    const bf16_t* xp[MG];      // rows >= M re-read row M-1 (see gemm_skinny_kernel); their outputs are never stored
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) xp[mg] = X + (int64_t)(mg * 16 + u < M ? mg * 16 + u : (M - 1)) * ldx + g * 8;
    const bf16_t* lp = lnw + g * 8;
    const int64_t nsteps = K / 64;
    const int64_t s_per = (nsteps + WAVES - 1) / WAVES;
    const int64_t s0 = wave * s_per;
    int64_t s1 = s0 + s_per; if (s1 > nsteps) s1 = nsteps;
    constexpr int NA = (GLU && NCOL >= 4) ? 1 : 2;      // accumulators per tile: the wide GLU forms have enough independent tiles to hide the MFMA latency
    f32x4_t acc[NCOL][MG][NA];
    float ss[MG];
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) {
        ss[mg] = 0.f;
#pragma unroll
        for (int c = 0; c < NCOL; ++c) { acc[c][mg][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; acc[c][mg][NA - 1] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
    }
    const unsigned xs_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)ng_xs;
    if (XLDS) {       // issued from inline asm: after the builtin hipcc would put vmcnt(0) in front of every LDS read, draining the weight stream
        const int r8 = lane >> 3;
        const int n_inst = (int)(K >> 5);                                     // 1 KiB per instruction
        for (int i = __builtin_amdgcn_readfirstlane(wave); i < n_inst; i += WAVES) {
            const int r = (i & 1) * 8 + r8;
            const unsigned off = (unsigned)((r < M ? r : M - 1) * (int)ldx + (i >> 1) * 64 + (((lane & 7) ^ keyA(r)) << 3)) * 2u;
            const unsigned dst = xs_base + (unsigned)i * 1024u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(off), "s"(X) : "memory", "m0");
        }
    }
    const unsigned xs_lane = xs_base + (unsigned)(u * 128);
    const int xs_key = keyA(u);
    // rotating software pipeline over UNROLL k-step buffers (see gemm_skinny_kernel)
    bf16x8_t wa[UNROLL][NCOL][2], xa[UNROLL][MG][2], la[UNROLL][2];
#define NG_LOAD(q, st)                                                                                   \
    do {                                                                                                 \
        const int64_t k__ = (st) * 64;                                                                   \
        _Pragma("unroll") for (int c = 0; c < NCOL; ++c) {                                               \
            wa[q][c][0] = *reinterpret_cast<const bf16x8_t*>(wp[c] + k__);                               \
            wa[q][c][1] = *reinterpret_cast<const bf16x8_t*>(wp[c] + k__ + 32);                          \
        }                                                                                                \
        if (!XLDS) { _Pragma("unroll") for (int mg = 0; mg < MG; ++mg) {                                 \
            xa[q][mg][0] = *reinterpret_cast<const bf16x8_t*>(xp[mg] + k__);                             \
            xa[q][mg][1] = *reinterpret_cast<const bf16x8_t*>(xp[mg] + k__ + 32);                        \
        } }                                                                                              \
        la[q][0] = *reinterpret_cast<const bf16x8_t*>(lp + k__);                                         \
        la[q][1] = *reinterpret_cast<const bf16x8_t*>(lp + k__ + 32);                                    \
    } while (0)
#define NG_MFMA(q, st)                                                                                                        \
    do {                                                                                                                      \
        if (XLDS) {                                                                                                           \
            typedef const __attribute__((address_space(3))) bf16x8_t* xs_ptr_t;                                               \
            const unsigned xa__ = xs_lane + (unsigned)(st) * 2048u;                                                           \
            xa[q][0][0] = *(xs_ptr_t)(uintptr_t)(xa__ + (unsigned)(((0 + g) ^ xs_key) << 4));                                 \
            xa[q][0][1] = *(xs_ptr_t)(uintptr_t)(xa__ + (unsigned)(((4 + g) ^ xs_key) << 4));                                 \
        }                                                                                                                     \
        _Pragma("unroll") for (int mg = 0; mg < MG; ++mg) {                                                                   \
            const bf16x8_t x0__ = scale_frag_sumsq(xa[q][mg][0], la[q][0], ss[mg]);                                           \
            const bf16x8_t x1__ = scale_frag_sumsq(xa[q][mg][1], la[q][1], ss[mg]);                                           \
            _Pragma("unroll") for (int c = 0; c < NCOL; ++c) {                                                                \
                acc[c][mg][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[q][c][0], x0__, acc[c][mg][0], 0, 0, 0);           \
                acc[c][mg][NA - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[q][c][1], x1__, acc[c][mg][NA - 1], 0, 0, 0); \
            }                                                                                                                 \
        }                                                                                                                     \
    } while (0)
    int64_t s = s0;
#pragma unroll
    for (int q = 0; q < UNROLL; ++q)
        if (s0 + q < s1) NG_LOAD(q, s0 + q);
    QKV_STAMP(1);
    if (XLDS) {       // the x copy has landed for this wave when only the UNROLL * (2 NCOL + 2) younger register loads are still in flight
        static_assert(!XLDS || UNROLL * (2 * NCOL + 2) == 12, "vmcnt below is written for 12 register loads in the prologue");
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    for (; s + 2 * UNROLL <= s1; s += UNROLL) {
#pragma unroll
        for (int q = 0; q < UNROLL; ++q) { NG_MFMA(q, s + q); NG_LOAD(q, s + q + UNROLL); }
    }
#pragma unroll
    for (int q = 0; q < UNROLL; ++q)
        if (s + q < s1) { NG_MFMA(q, s + q); if (s + q + UNROLL < s1) NG_LOAD(q, s + q + UNROLL); }
    s += UNROLL;
#pragma unroll
    for (int q = 0; q < UNROLL; ++q)
        if (s + q < s1) NG_MFMA(q, s + q);
#undef NG_LOAD
#undef NG_MFMA
    QKV_STAMP(2);
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) {     // lanes u, u+16, u+32, u+48 hold disjoint k chunks of row u
        float v = ss[mg];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (g == 0) ssred[wave][mg][u] = v;
#pragma unroll
        for (int c = 0; c < NCOL; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][c][mg][u][g * 4 + r] = NA == 2 ? acc[c][mg][0][r] + acc[c][mg][1][r] : acc[c][mg][0][r];
    }
    __syncthreads();
    QKV_STAMP(3);
    const float inv_k = 1.f / (float)K;
    for (int i = threadIdx.x; i < OG * MG * 256; i += WAVES * 64) {   // (column group, row group, m, n)
        const int c = i / (MG * 256), mg = (i >> 8) % MG, mm = (i >> 4) & 15, nn = i & 15;
        const int m = mg * 16 + mm;
        const int64_t n = n0 + c * 16 + nn;
        if (m < M && n < N) {
            float sq = 0.f, v = 0.f, v2 = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) { sq += ssred[w][mg][mm]; v += red[w][c][mg][mm][nn]; if (GLU || QKV) v2 += red[w][GLU ? c + OG : 1][mg][mm][nn]; }
            const float rstd = rsqrtf(sq * inv_k + eps);
            v = __fmul_rn(v, rstd);          // explicitly rounded (no fma contraction): the fused-QKV and two-kernel paths agree bit for bit
            if (QKV) {      // same rounding points as projection -> bf16 qkv buffer -> decode_qkv_post
                const int64_t nb = n + (qe.hd >> 1);
                float vb = __fmul_rn(v2, rstd);
                if (bias) { v = __fadd_rn(v, bf2f(bias[n])); vb = __fadd_rn(vb, bf2f(bias[nb])); }
                qkv_epilogue_store(qe, m, qkv_h, qkv_j * 16 + nn, bf2f(f2bf(v)), bf2f(f2bf(vb)));
            } else if (GLU) {      // same rounding points as the unfused path: gate/up rounded to bf16, silu rounded, product rounded
                const float gt = bf2f(f2bf(v)), up = bf2f(f2bf(v2 * rstd));
                C[(int64_t)m * ldc + n] = f2bf(bf2f(f2bf(silu_f32(gt))) * up);
            } else {
                if (bias) v = __fadd_rn(v, bf2f(bias[n]));
                C[(int64_t)m * ldc + n] = f2bf(v);
            }
        }
    }
    QKV_STAMP(4);
    QKV_DUMP();
}

// ------------------------------------------------------------------------------------------------------------------
// LDS-streamed form of the fused rmsnorm + gate/up + SwiGLU decode GEMM (M <= 16 rows).
// The register-fragment stream of norm_gemm_skinny_kernel reads 64-byte pieces of 16 weight rows per wave instruction and tops out at
// ~4.8 TB/s (a pure load kernel with that shape: 5.5 TB/s; with row-contiguous requests: 6.1 TB/s, tools/probe_stream.hip).  Here the
// weights go HBM -> LDS with global_load_lds in full 128-byte row runs (8 rows per wave instruction), each wave keeps its OWN ring of
// stages in LDS and reads the MFMA fragments back with ds_read_b128, so nothing but the issuing wave's counted vmcnt orders a stage
// (no barrier in the stream).  Consequences used below:
//   * a block is PERSISTENT over a contiguous range of column-group pairs (16 gate rows + 16 up rows), so the activation fragments
//     x' = bf16(x * lnw) of the wave's k-slice (K/8 columns) are built ONCE and stay in registers, and sum x^2 is reduced once per block;
//   * after the last stage of a pair the eight waves drop their partial tiles in LDS, meet at ONE raw barrier (the DMA of the next pair
//     stays in flight) and 256 threads finish one output each, with the rounding points of norm_gemm_skinny_kernel's GLU epilogue.  The
//     epilogue's global stores share vmcnt with the DMA; that is safe for the counted waits: loads retire in order among themselves, so
//     "at most 4(R-1) operations outstanding" still implies that the stage being consumed has landed - outstanding stores can only make
//     a wait longer, never shorter.
// LDS image of a stage: [gate 16 rows | up 16 rows] x 128 bytes (64 k); row r keeps its logical 16-byte chunk c at position c ^ keyA(r)
// (applied on the SOURCE address of the DMA): the 16-row fragment reads are conflict-free.
// ------------------------------------------------------------------------------------------------------------------
// QKV (round 3): the same stream for the fused rmsnorm + q/k/v projection + M-RoPE + KV append at <= 16 rows.  One block per column-group pair
// (16 columns d of a head and their rotate-half partners d + hd/2: `up_off` = hd/2 weight rows), no persistence (144 pairs at 7B), epilogue and
// rounding points of norm_gemm_skinny_kernel's QKV form - same k-slices per wave, same two accumulators per tile, same wave-order reduction, so the
// result is bit-identical to it.  What changes is the path of the weights: full 128-byte row runs by DMA (8 tag look-ups per KiB) instead of
// 64-byte pieces of 16 rows per wave load (64 look-ups per KiB: the L1 tag pipe held the register-fragment kernel at ~40 GB/s per CU).
#ifdef TR1_PROBE
__device__ unsigned long long* tr1_glu_probe = nullptr;
extern "C" int probe_glu_set_ptr(void* ptr) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(tr1_glu_probe), &ptr, sizeof(ptr)); }
#define GLU_STAMPS unsigned long long gs_[6] = {0, 0, 0, 0, 0, 0}
#define GLU_STAMP(i) do { gs_[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define GLU_DUMP() do { if (tr1_glu_probe && lane == 0 && (wave == 0 || wave == 7)) { \
    _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) tr1_glu_probe[((int64_t)blockIdx.x * 2 + (wave ? 1 : 0)) * 8 + i_] = gs_[i_]; } } while (0)
#else
#define GLU_STAMPS do { } while (0)
#define GLU_STAMP(i) do { } while (0)
#define GLU_DUMP() do { } while (0)
#endif
// MODE 0: gate/up + SwiGLU.  MODE 1: fused QKV (above).  MODE 2 (round 3): plain projection C = rmsnorm(x) W^T for a wide N (the lm_head): the block's two
// row groups are output columns n and n + N/2 (`up_off` = N/2 weight rows apart), both stored as they are.
template <int NST, int R, int NRED = 2, int MG = 1, int MODE = 0>
__global__ __launch_bounds__(512) void norm_glu_lds_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ lnw, const bf16_t* __restrict__ W,
                                                           bf16_t* __restrict__ C, int M, int64_t N, int64_t K, int64_t ldx, int64_t ldw,
                                                           int64_t ldc, float eps, int64_t up_off, const bf16_t* __restrict__ bias = nullptr,
                                                           QkvEpi qe = QkvEpi{}, int c_frag = 0) {
    // c_frag (MODE 0, MG = 1; round 6): the SwiGLU output leaves FRAGMENT-MAJOR - element (m, n) at (n / 32) * 512 + m * 32 + n % 32, the layout
    // tr1_gemm_oproj_frag reads (one contiguous KiB per MFMA operand fragment) - for the all-stages-in-flight down projection of the 2B shapes
    constexpr bool QKV = MODE == 1, PLAIN = MODE == 2;
    constexpr int STAGE = 4096;                                            // bytes per stage: gate 2 KiB + up 2 KiB
    constexpr int REDW = MG * 2 * 16 * 17;                                 // floats of one wave's partial: MG row groups x (gate | up)
    // red[2][8][REDW] f32 | ssq[8][16] | [8 waves][R stages][4 KiB].  The rings come LAST: a DMA destination is passed as (slot - stage offset)
    // because the instruction's immediate offset is added to the LDS address too, and that pointer must not fall below the LDS base.
    extern __shared__ __attribute__((aligned(16))) char glu_lds[];
    float* red = reinterpret_cast<float*>(glu_lds);
    float* ssq = red + NRED * 8 * REDW;                                   // [8 waves][MG][16]
    char* rings = glu_lds + (NRED * 8 * REDW + 8 * MG * 16) * sizeof(float);
    static_assert((NRED * 8 * REDW + 8 * MG * 16) * sizeof(float) >= 6 * 128 && ((NRED * 8 * REDW + 8 * MG * 16) * sizeof(float)) % 16 == 0, "ring base");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, u = lane & 15, g = lane >> 4;
    GLU_STAMPS;
    GLU_STAMP(0);
    const int64_t NP = QKV ? (int64_t)gridDim.x : (PLAIN ? up_off / 16 : (N + 15) / 16);
    const int64_t p0 = NP * blockIdx.x / gridDim.x, p1 = NP * (blockIdx.x + 1) / gridDim.x;
    const int qkv_gph = QKV ? qe.hd >> 5 : 1;                               // column-group pairs per head
    const int qkv_h = QKV ? (int)blockIdx.x / qkv_gph : 0, qkv_j = QKV ? (int)blockIdx.x % qkv_gph : 0;
    const int64_t row_base = QKV ? (int64_t)qkv_h * qe.hd + qkv_j * 16 : p0 * 16;      // first weight row of the block's first pair
    const int npair = (int)(p1 - p0);
    const int64_t kb = (int64_t)wave * (K / 8);
    // ---- DMA lane map: instruction j covers rows 8j .. 8j+7; lane -> row 8j + (lane >> 3), physical chunk lane & 7, logical chunk ^ keyA(row).
    // Four per-lane source pointers (gate / up rows of the pair being ISSUED) advance by 16 rows per pair; the stage inside the pair is an
    // immediate offset of the DMA instruction, so issuing a stage costs no vector ALU work.
    char* ring = rings + wave * R * STAGE;
    const int total = npair * NST;
    const bf16_t* pg[2]; const bf16_t* pu[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 8 * j + (lane >> 3);
        pg[j] = W + (row_base + r) * ldw + kb + (((lane & 7) ^ keyA(r)) << 3);
        pu[j] = pg[j] + up_off * ldw;
    }
    const int64_t pair_step = 16 * ldw;
    int islot = 0;                                       // ring slot of the next item to issue (wave-uniform)
#define GLU_ISSUE(ST) do {                                                                                               \
        char* dst__ = ring + islot * STAGE - (ST) * 128;   /* the instruction offset is added to the LDS address as well */  \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                  \
            __builtin_amdgcn_global_load_lds((gptr_t)pg[j], (lptr_t)(dst__ + j * 1024), 16, (ST) * 128, TR1_W_AUX);              \
            __builtin_amdgcn_global_load_lds((gptr_t)pu[j], (lptr_t)(dst__ + 2048 + j * 1024), 16, (ST) * 128, TR1_W_AUX);       \
        }                                                                                                                \
        islot = (islot + 1 == R) ? 0 : islot + 1;                                                                        \
    } while (0)
#define GLU_ISSUE_ST(ST) do { switch (ST) { case 0: GLU_ISSUE(0); break; case 1: GLU_ISSUE(1); break; case 2: GLU_ISSUE(2); break; case 3: GLU_ISSUE(3); break; \
                                           case 4: GLU_ISSUE(4); break; case 5: GLU_ISSUE(5); break; default: GLU_ISSUE(6); break; } } while (0)
#define GLU_NEXT_PAIR() do { _Pragma("unroll") for (int j = 0; j < 2; ++j) { pg[j] += pair_step; pu[j] += pair_step; } } while (0)
    static_assert(NST <= 7, "stage offsets are enumerated up to 7");
    constexpr bool XDMA = (MG <= 2) && (2 * R + NRED >= NST);   // round 3: this wave's x slice goes through its own (still empty) ring first (MG = 2: one 16-row group after the other)
#define GLU_PROLOGUE(I0, I1) do {                                                                                \
        _Pragma("unroll") for (int i = (I0); i < (I1); ++i) {    /* prologue: items 0 .. R-2 */                  \
            if (i < total) {                                                                                     \
                if (i > 0 && i % NST == 0) GLU_NEXT_PAIR();                                                      \
                GLU_ISSUE_ST(i % NST);                                                                           \
            }                                                                                                    \
        } } while (0)
    // x staging that leaves ring slot 0 to the FIRST weight stage (requested together with the x copy): slots 1 .. R-1, the reduction slices and one
    // extra 2 KiB per wave behind the norm-weight area
    constexpr bool XSLOT0 = XDMA && MG == 1 && (2 * (R - 1) + NRED + 1 >= NST) && R >= 3;
    if (!XDMA) GLU_PROLOGUE(0, R - 1);
    GLU_STAMP(1);
    // ---- x' fragments of this wave's k-slice (once per block) and the row sums of squares - built AFTER the first weight stages were
    // issued, so the HBM stream starts at kernel entry instead of waiting for this L2 round trip
    bf16x8_t xr[MG][NST * 2];
    if (XDMA) {
        // The round-2 form loaded x and lnw with per-wave vector loads AFTER issuing the first weight stages: 28 loads per lane in four dependent
        // batches, each load touching 16 rows x 64 bytes (64 L1 tag look-ups per KiB) - 15 400 of the launch's 98 000 cycles, with only two weight stages
        // in flight meanwhile (block timeline).  Here the wave's x slice (16 rows x K/8 columns = NST stages of 2 KiB) is copied by DMA into
        // its own ring (and, past 2R stages, its reduction slices - all unused so far), read back as fragments, and only then does the weight
        // stream start: one L2 round trip instead of four, 8 tag look-ups per KiB.  Same fragments in the same order -> same sum of squares.
        // the wave's slice of the norm weight as well: ONE DMA instruction (1 KiB = 512 columns from kb on; lanes past the end of lnw re-read its
        // last 16 bytes) into a private KiB behind the rings instead of 2 NST vector loads per lane
        char* lnw_lds = rings + 8 * R * STAGE + wave * 1024;
        {
            int64_t col = kb + lane * 8;
            if (col + 8 > K) col = K - 8;
            __builtin_amdgcn_global_load_lds((gptr_t)(lnw + col), (lptr_t)lnw_lds, 16, 0, 0);
        }
        char* const x_extra = rings + 8 * R * STAGE + 8 * 1024 + wave * 2048;
        auto x_stage = [&](int st) -> char* {
            if (!XSLOT0) return st < 2 * R ? ring + st * 2048 : reinterpret_cast<char*>(red + ((st - 2 * R) * 8 + wave) * REDW);
            if (st < 2 * (R - 1)) return ring + STAGE + st * 2048;
            if (st - 2 * (R - 1) < NRED) return reinterpret_cast<char*>(red + ((st - 2 * (R - 1)) * 8 + wave) * REDW);
            return x_extra;
        };
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) {
        float ss = 0.f;
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            char* dst = x_stage(st);
#pragma unroll
            for (int jx = 0; jx < 2; ++jx) {
                const int r = mg * 16 + 8 * jx + (lane >> 3);
                const bf16_t* src = X + (int64_t)(r < M ? r : M - 1) * ldx + kb + st * 64 + (((lane & 7) ^ keyA(r & 15)) << 3);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + jx * 1024), 16, 0, 0);
            }
        }
        if (XSLOT0) {
            GLU_PROLOGUE(0, 1);                                           // the first weight stage travels while x' is built
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");              // everything but that stage's four DMA instructions has landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < NST * 2; ++i) {
            const int st = i >> 1, ks = i & 1;
            const char* sbx = x_stage(st);
            const bf16x8_t xv = *reinterpret_cast<const bf16x8_t*>(sbx + u * 128 + (((ks * 4 + g) ^ keyA(u)) << 4));
            const bf16x8_t lvi = *reinterpret_cast<const bf16x8_t*>(lnw_lds + (i * 32 + g * 8) * 2);
            u32x4_t f = __builtin_bit_cast(u32x4_t, scale_frag_sumsq(xv, lvi, ss));
            asm volatile("" : "+v"(f));
            xr[mg][i] = __builtin_bit_cast(bf16x8_t, f);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the ring is free again: next row group / start (continue) the weight stream
        float v = ss;
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (g == 0) ssq[(wave * MG + mg) * 16 + u] = v;
        }
        if (XSLOT0) GLU_PROLOGUE(1, R - 1); else GLU_PROLOGUE(0, R - 1);
    } else
    {
        float ss[MG];
        const bf16_t* xp[MG];
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) { ss[mg] = 0.f; xp[mg] = X + (int64_t)(mg * 16 + u < M ? mg * 16 + u : M - 1) * ldx + kb + g * 8; }
        const bf16_t* lp = lnw + kb + g * 8;
#pragma unroll
        for (int i0 = 0; i0 < NST * 2; i0 += 4) {        // four k-halves at a time: the loads of ALL fragments in flight at once would not fit the
#pragma unroll                                           // register file next to xr at MG = 2 (scheduling fence below)
            for (int i = i0; i < i0 + 4 && i < NST * 2; ++i) {
                const bf16x8_t lv = *reinterpret_cast<const bf16x8_t*>(lp + i * 32);
#pragma unroll
                for (int mg = 0; mg < MG; ++mg) {
                    const bf16x8_t xv = *reinterpret_cast<const bf16x8_t*>(xp[mg] + i * 32);
                    u32x4_t f = __builtin_bit_cast(u32x4_t, scale_frag_sumsq(xv, lv, ss[mg]));
                    asm volatile("" : "+v"(f));          // pin the PACKED fragment here: left alone, the compiler keeps the unpacked f32 products
                    xr[mg][i] = __builtin_bit_cast(bf16x8_t, f);     // live into the main loop (2x the registers) and spills at MG = 2
                }
            }
            if (MG > 1) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) {
            float v = ss[mg];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (g == 0) ssq[(wave * MG + mg) * 16 + u] = v;
        }
    }
    GLU_STAMP(2);
    TR1_BARRIER();                                       // the eight waves' sum-of-squares partials are in LDS (the prologue DMA is in flight)
    GLU_STAMP(3);
    float rstd;
    {
        float sq = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) sq += ssq[(w * MG + (MG > 1 ? (threadIdx.x >> 8) : 0)) * 16 + ((threadIdx.x >> 4) & 15)];
        rstd = rsqrtf(sq * (1.f / (float)K) + eps);
    }
    const int rd_off = u * 128;
    const int kA = keyA(u);
    int cslot = 0;                                       // ring slot of the item being consumed
    for (int pi = 0; pi < npair; ++pi) {
        f32x4_t ag[MG][2], au[MG][2];
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) ag[mg][0] = ag[mg][1] = au[mg][0] = au[mg][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const int item = pi * NST + st;
            if (item + R - 1 < total) {
                if ((st + R - 1) % NST == 0) GLU_NEXT_PAIR();           // the issue stream enters the next pair here (compile-time position)
                GLU_ISSUE_ST((st + R - 1) % NST);
            }
            const int rem = total - 1 - item;              // items issued after this one and still allowed in flight
            if (rem >= R - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (R - 1)) : "memory");
            else if (rem == 2 && R > 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (rem == 1 && R > 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const char* sb = ring + cslot * STAGE + rd_off;
            cslot = (cslot + 1 == R) ? 0 : cslot + 1;
            bf16x8_t wg[2], wu[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int off = ((ks * 4 + g) ^ kA) << 4;
                wg[ks] = *reinterpret_cast<const bf16x8_t*>(sb + off);
                wu[ks] = *reinterpret_cast<const bf16x8_t*>(sb + 2048 + off);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mg = 0; mg < MG; ++mg) {
                    ag[mg][ks] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wg[ks], xr[mg][st * 2 + ks], ag[mg][ks], 0, 0, 0);
                    au[mg][ks] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wu[ks], xr[mg][st * 2 + ks], au[mg][ks], 0, 0, 0);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this stage's reads have returned before its slot can be refilled
        }
        float* rw = red + ((NRED == 2 ? (pi & 1) : 0) * 8 + wave) * REDW;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mg = 0; mg < MG; ++mg) {
                rw[mg * 544 + u * 17 + g * 4 + r] = ag[mg][0][r] + ag[mg][1][r];
                rw[mg * 544 + 16 * 17 + u * 17 + g * 4 + r] = au[mg][0][r] + au[mg][1][r];
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TR1_BARRIER();                                                  // partial tiles are visible; the DMA of the next pair stays in flight
        if (threadIdx.x < 256 * MG) {
            const int mgi = MG > 1 ? (threadIdx.x >> 8) : 0, mm = (threadIdx.x >> 4) & 15, nn = threadIdx.x & 15;
            const float* rb = red + (NRED == 2 ? (pi & 1) : 0) * 8 * REDW + mgi * 544;
            float v = 0.f, v2 = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) { v += rb[w * REDW + mm * 17 + nn]; v2 += rb[w * REDW + 16 * 17 + mm * 17 + nn]; }
            v = __fmul_rn(v, rstd);
            if (QKV) {      // rounding points of norm_gemm_skinny_kernel's QKV epilogue (projection -> bf16 -> RoPE / cache append)
                const int64_t n = row_base + nn, nb = n + (qe.hd >> 1);
                float vb = __fmul_rn(v2, rstd);
                if (bias) { v = __fadd_rn(v, bf2f(bias[n])); vb = __fadd_rn(vb, bf2f(bias[nb])); }
                if (mgi * 16 + mm < M && n < N) qkv_epilogue_store(qe, mgi * 16 + mm, qkv_h, qkv_j * 16 + nn, bf2f(f2bf(v)), bf2f(f2bf(vb)));
            } else if (PLAIN) {
                const int64_t n = (p0 + pi) * 16 + nn;
                float vb = __fmul_rn(v2, rstd);
                if (bias) { v = __fadd_rn(v, bf2f(bias[n])); vb = __fadd_rn(vb, bf2f(bias[n + up_off])); }
                const int mrow = mgi * 16 + mm;
                if (mrow < M && n < up_off) { C[(int64_t)mrow * ldc + n] = f2bf(v); if (n + up_off < N) C[(int64_t)mrow * ldc + n + up_off] = f2bf(vb); }
            } else {
            const float gt = bf2f(f2bf(v)), up = bf2f(f2bf(v2 * rstd));
            const int64_t n = (p0 + pi) * 16 + nn;
            if (mgi * 16 + mm < M && n < N) C[c_frag ? (n >> 5) * 512 + mm * 32 + (n & 31) : (int64_t)(mgi * 16 + mm) * ldc + n] = f2bf(bf2f(f2bf(silu_f32(gt))) * up);
            }
        }
        if (NRED == 1) TR1_BARRIER();                                   // single reduction buffer: everybody has read it before the next pair writes
        if (pi == 0) GLU_STAMP(4);
    }
    GLU_STAMP(5);
    GLU_DUMP();
#undef GLU_ISSUE
#undef GLU_ISSUE_ST
#undef GLU_NEXT_PAIR
}

// TR1_NG32_CFG: A/B hook for the 17..32-row rmsnorm + projection kernels (plain and fused-QKV take the SAME form so they stay bit-identical).
// Default 1 = 8 waves x UNROLL 2 (fused QKV at 32 rows: 19.7 -> 18.4 us; 144 blocks for 256 CUs, so the extra waves are what adds loads in flight)
static int ng32_cfg() { static int c = -1; if (c < 0) { c = 1; } return c; }

// 1 when tr1_norm_gemm_skinny(..., glu = 2) can write the SwiGLU output fragment-major (the LDS-streamed <= 16-row form; N % 32: whole 32-column fragments)
extern "C" int tr1_norm_gemm_glu_frag_ok(int64_t M, int64_t N, int64_t K) {
    const int64_t nst = K / 512;
    return M >= 1 && M <= 16 && K % 512 == 0 && (nst == 7 || nst == 4 || nst == 3) && N % 32 == 0;
}

extern "C" int tr1_norm_gemm_skinny(const void* x, const void* lnw, const void* W, const void* bias, void* out, int64_t M, int64_t N, int64_t K,
                                    int64_t ldx, int64_t ldw, int64_t ldc, float eps, int glu, void* stream) {
    TR1_CHECK_ARG(K % BK == 0 && K >= BK, "norm_gemm_skinny: K must be a positive multiple of 64");
    TR1_CHECK_ARG(M >= 1 && M <= 64, "norm_gemm_skinny: 1 <= M <= 64 (decode rows)");
    TR1_CHECK_ARG(N % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0, "norm_gemm_skinny: N%8, ldx%8, ldw%8, ldc%8 required");
    TR1_CHECK_ARG(!glu || !bias, "norm_gemm_skinny: the GLU form takes no bias");
    hipStream_t s = (hipStream_t)stream;
    // N = output columns (glu: the intermediate size I; W then has 2*I rows, gate rows first)
#define NG(WV, UN, MGR, GL)                                                                                                          \
    hipLaunchKernelGGL((norm_gemm_skinny_kernel<WV, UN, MGR, GL>), dim3((unsigned)((N + (GL ? 16 : 32) - 1) / (GL ? 16 : 32))),      \
                       dim3(WV * 64), 0, s, (const bf16_t*)x, (const bf16_t*)lnw, (const bf16_t*)W, (bf16_t*)out, (const bf16_t*)bias, \
                       (int)M, N, K, ldx, ldw, ldc, eps, N)
    // gate/up + SwiGLU at <= 16 rows: UNROLL 2 keeps the kernel at 128 VGPRs = 4 blocks per CU (1024 slots for 1184 blocks); measured 59.2 vs 60.9 us
    static int glu_lds = -1;                         // TR1_GLU_LDS=0 selects the register-fragment form (A/B measurements)
    if (glu_lds < 0) { glu_lds = 1; }
    const int64_t nst = K / 512;                     // 64-wide stages per wave (8 waves split K)
    static int head_lds = -1;                        // TR1_HEAD_LDS=0: the lm_head stays on the register-fragment kernel (A/B measurements)
    if (head_lds < 0) { head_lds = 1; }
    if (!glu && M > 16 && M <= 32 && head_lds && glu_lds && N >= 65536 && N % 32 == 0 && K % 512 == 0 && (nst == 7 || nst == 4 || nst == 3)) {
        // 17 .. 32 rows (config 4): two row groups per wave against the same LDS stage, single reduction buffer
        constexpr int RING = 3;
        const size_t dyn = 8 * RING * 4096 + (1 * 8 * 2 * 2 * 16 * 17 + 8 * 2 * 16) * sizeof(float) + 8192;
        static int n_cu_h2 = 0;
        if (!n_cu_h2) {
            hipDeviceProp_t prop; int dev = 0;
            hipGetDevice(&dev); hipGetDeviceProperties(&prop, dev);
            n_cu_h2 = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
#define HL2_ATTR(NSTV) hipFuncSetAttribute(reinterpret_cast<const void*>(&norm_glu_lds_kernel<NSTV, RING, 1, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn)
            HL2_ATTR(7); HL2_ATTR(4); HL2_ATTR(3);
#undef HL2_ATTR
        }
        const int64_t NPh = N / 32;
        const unsigned gridh = (unsigned)(NPh < n_cu_h2 ? NPh : n_cu_h2);
#define HL2_LAUNCH(NSTV) hipLaunchKernelGGL((norm_glu_lds_kernel<NSTV, RING, 1, 2, 2>), dim3(gridh), dim3(512), dyn, s, (const bf16_t*)x, (const bf16_t*)lnw, (const bf16_t*)W, \
                                            (bf16_t*)out, (int)M, N, K, ldx, ldw, ldc, eps, N / 2, (const bf16_t*)bias, QkvEpi{})
        if (nst == 7) HL2_LAUNCH(7); else if (nst == 4) HL2_LAUNCH(4); else HL2_LAUNCH(3);
#undef HL2_LAUNCH
        TR1_LAUNCH_CHECK();
    }
    if (!glu && M <= 16 && head_lds && glu_lds && N >= 65536 && N % 32 == 0 && K % 512 == 0 && (nst == 7 || nst == 4 || nst == 3)) {
        // wide plain projection (the lm_head) through the LDS stream: 256 persistent blocks x 8 waves, column pairs (n, n + N/2)
        constexpr int RING = 3;
        const size_t dyn = 8 * RING * 4096 + (2 * 8 * 2 * 16 * 17 + 8 * 16) * sizeof(float) + 8192 + 16384;
        static int n_cu_h = 0;
        if (!n_cu_h) {
            hipDeviceProp_t prop; int dev = 0;
            hipGetDevice(&dev); hipGetDeviceProperties(&prop, dev);
            n_cu_h = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
#define HL_ATTR(NSTV) hipFuncSetAttribute(reinterpret_cast<const void*>(&norm_glu_lds_kernel<NSTV, RING, 2, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn)
            HL_ATTR(7); HL_ATTR(4); HL_ATTR(3);
#undef HL_ATTR
        }
        const int64_t NPh = N / 32;
        const unsigned gridh = (unsigned)(NPh < n_cu_h ? NPh : n_cu_h);
#define HL_LAUNCH(NSTV) hipLaunchKernelGGL((norm_glu_lds_kernel<NSTV, RING, 2, 1, 2>), dim3(gridh), dim3(512), dyn, s, (const bf16_t*)x, (const bf16_t*)lnw, (const bf16_t*)W, \
                                           (bf16_t*)out, (int)M, N, K, ldx, ldw, ldc, eps, N / 2, (const bf16_t*)bias, QkvEpi{})
        if (nst == 7) HL_LAUNCH(7); else if (nst == 4) HL_LAUNCH(4); else HL_LAUNCH(3);
#undef HL_LAUNCH
        TR1_LAUNCH_CHECK();
    }
    TR1_CHECK_ARG(glu != 2 || tr1_norm_gemm_glu_frag_ok(M, N, K), "norm_gemm_skinny: glu = 2 (fragment-major SwiGLU output) needs M <= 16 and the LDS-streamed form (tr1_norm_gemm_glu_frag_ok)");
    if (glu && M <= 32 && glu_lds && K % 512 == 0 && (nst == 7 || nst == 4 || nst == 3) && N % 16 == 0) {   // hidden 3584 / 2048 / 1536
        // <= 16 rows: ring of 3 + double reduction buffer (a ring of 4 with a single buffer and a second barrier per pair measured the same).
        // 17..32 rows (config 4 decodes 2 x 16 rollouts): two row groups per wave against the SAME LDS stage, ring of 3, single reduction
        // buffer (132 KB of LDS): 77.4 -> 52.8 us at 32 x 18944 x 3584 (5.1 TB/s of weights) over the register-fragment form.
        constexpr int RING = 3;
        const size_t dyn1 = 8 * RING * 4096 + (2 * 8 * 2 * 16 * 17 + 8 * 16) * sizeof(float) + 8192 + 16384;      // + the waves' norm-weight KiB + one x stage each
        const size_t dyn2 = 8 * RING * 4096 + (1 * 8 * 2 * 2 * 16 * 17 + 8 * 2 * 16) * sizeof(float) + 8192;
        static int n_cu = 0;
        if (!n_cu) {
            hipDeviceProp_t prop; int dev = 0;
            hipGetDevice(&dev); hipGetDeviceProperties(&prop, dev);
            n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
#define GLU_ATTR(NSTV)                                                                                                                          \
    hipFuncSetAttribute(reinterpret_cast<const void*>(&norm_glu_lds_kernel<NSTV, RING, 2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn1); \
    hipFuncSetAttribute(reinterpret_cast<const void*>(&norm_glu_lds_kernel<NSTV, RING, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn2)
            GLU_ATTR(7); GLU_ATTR(4); GLU_ATTR(3);
#undef GLU_ATTR
        }
        const int64_t NP = N / 16;
        const unsigned grid = (unsigned)(NP < n_cu ? NP : n_cu);
#define GLU_LAUNCH(NSTV)                                                                                                                        \
    do {                                                                                                                                        \
        if (M <= 16) hipLaunchKernelGGL((norm_glu_lds_kernel<NSTV, RING, 2, 1>), dim3(grid), dim3(512), dyn1, s, (const bf16_t*)x, (const bf16_t*)lnw, \
                                        (const bf16_t*)W, (bf16_t*)out, (int)M, N, K, ldx, ldw, ldc, eps, N, (const bf16_t*)nullptr, QkvEpi{}, glu == 2 ? 1 : 0); \
        else hipLaunchKernelGGL((norm_glu_lds_kernel<NSTV, RING, 1, 2>), dim3(grid), dim3(512), dyn2, s, (const bf16_t*)x, (const bf16_t*)lnw,         \
                                (const bf16_t*)W, (bf16_t*)out, (int)M, N, K, ldx, ldw, ldc, eps, N);                                           \
    } while (0)
        if (nst == 7) GLU_LAUNCH(7); else if (nst == 4) GLU_LAUNCH(4); else GLU_LAUNCH(3);
#undef GLU_LAUNCH
    }
    else if (glu) { if (M <= 16) NG(4, 2, 1, true); else if (M <= 32) NG(4, 2, 2, true); else NG(4, 2, 4, true); }
    else if (N >= 100000 && M <= 32) {      // lm_head: 4 column groups per block halve the re-reads of x (228 -> ~195 us at M = 16)
#define NG4(UN, MGR)                                                                                                                 \
    hipLaunchKernelGGL((norm_gemm_skinny_kernel<4, UN, MGR, false, 4>), dim3((unsigned)((N + 63) / 64)), dim3(256), 0, s, (const bf16_t*)x,    \
                       (const bf16_t*)lnw, (const bf16_t*)W, (bf16_t*)out, (const bf16_t*)bias, (int)M, N, K, ldx, ldw, ldc, eps, N)
        if (M <= 16) NG4(2, 1); else NG4(2, 2);
#undef NG4
    }
    else     { if (M <= 16) NG(8, 2, 1, false); else if (M <= 32) { const int c = ng32_cfg(); if (c == 1) NG(8, 2, 2, false); else if (c == 2) NG(8, 1, 2, false); else if (c == 3) NG(4, 1, 2, false); else NG(4, 2, 2, false); } else NG(4, 2, 4, false); }   // 8 waves: see tr1_norm_gemm_qkv
#undef NG
    TR1_LAUNCH_CHECK();
}

extern "C" int tr1_norm_gemm_qkv(const void* x, const void* lnw, const void* Wqkv, const void* bias, const void* cosb, const void* sinb, void* q_out,
                                 int64_t ld_q, void* kcache, int64_t k_ld, void* vtcache, int64_t vt_ld, const void* slots, int64_t M, int64_t n_heads,
                                 int64_t n_kv, int64_t head_dim, int64_t K, int64_t ldx, int64_t ldw, float eps, void* stream) {
    TR1_CHECK_ARG(K % BK == 0 && K >= BK, "norm_gemm_qkv: K must be a positive multiple of 64");
    TR1_CHECK_ARG(M >= 1 && M <= 64, "norm_gemm_qkv: 1 <= M <= 64 (decode rows)");
    TR1_CHECK_ARG(head_dim % 32 == 0 && ldx % 8 == 0 && ldw % 8 == 0, "norm_gemm_qkv: head_dim % 32, ldx % 8, ldw % 8 required");
    const int64_t heads = n_heads + 2 * n_kv, N = heads * head_dim;
    QkvEpi qe{(const float*)cosb, (const float*)sinb, (bf16_t*)q_out, ld_q, (bf16_t*)kcache, k_ld, (bf16_t*)vtcache, vt_ld, (const int*)slots,
              (int)n_heads, (int)n_kv, (int)head_dim};
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(heads * (head_dim / 32)));
    // M <= 16: only heads * hd/32 = 144 blocks (7B) for 256 CUs, so 8 waves per block split K and double the loads in flight per CU
    // (tools/microbench.py fused, TR1_NG_CFG: 15.8 -> 13.5 us)
#define NGQ(WV, UN, MGR)                                                                                                              \
    hipLaunchKernelGGL((norm_gemm_skinny_kernel<WV, UN, MGR, false, 2, true>), grid, dim3(WV * 64), 0, s, (const bf16_t*)x, (const bf16_t*)lnw, \
                       (const bf16_t*)Wqkv, (bf16_t*)nullptr, (const bf16_t*)bias, (int)M, N, K, ldx, ldw, (int64_t)0, eps, (int64_t)0, qe)
    // TR1_QKV_LDS (default 1): weights by DMA through per-wave LDS rings (norm_glu_lds_kernel, QKV mode; bit-identical).  Before the x slice was staged
    // by DMA as well the decode step measured 3 547 us with it against 3 472 us for the register-fragment form with x in LDS; with the staged x it is
    // 3 277 against 3 325 us.  0 selects the register-fragment kernel (x through LDS unless TR1_QKV_XLDS=0).
    static int qlds = -1;
    if (qlds < 0) { qlds = 1; }
    const int64_t nst = K / 512;
    if (M > 16 && M <= 32 && qlds && K % 512 == 0 && (nst == 7 || nst == 4 || nst == 3)) {      // 17 .. 32 rows (config 4): two row groups per wave, ring of 3
        constexpr int RING2 = 3;
        const size_t dyn = 8 * RING2 * 4096 + (1 * 8 * 2 * 2 * 16 * 17 + 8 * 2 * 16) * sizeof(float) + 8192;
        static bool attr_q2 = false;
        if (!attr_q2) {
#define QL2_ATTR(NSTV) hipFuncSetAttribute(reinterpret_cast<const void*>(&norm_glu_lds_kernel<NSTV, RING2, 1, 2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn)
            QL2_ATTR(7); QL2_ATTR(4); QL2_ATTR(3);
#undef QL2_ATTR
            attr_q2 = true;
        }
#define QL2_LAUNCH(NSTV) hipLaunchKernelGGL((norm_glu_lds_kernel<NSTV, RING2, 1, 2, 1>), grid, dim3(512), dyn, s, (const bf16_t*)x, (const bf16_t*)lnw, (const bf16_t*)Wqkv, \
                                            (bf16_t*)nullptr, (int)M, N, K, ldx, ldw, (int64_t)0, eps, (int64_t)(head_dim / 2), (const bf16_t*)bias, qe)
        if (nst == 7) QL2_LAUNCH(7); else if (nst == 4) QL2_LAUNCH(4); else QL2_LAUNCH(3);
#undef QL2_LAUNCH
        TR1_LAUNCH_CHECK();
    }
    if (M <= 16 && qlds && K % 512 == 0 && (nst == 7 || nst == 4 || nst == 3)) {
        constexpr int RING = 4;
        const size_t dyn = 8 * RING * 4096 + (1 * 8 * 2 * 16 * 17 + 8 * 16) * sizeof(float) + 8192;      // (ring of 4: the x staging fits without the extra stage area... see XSLOT0)
        static bool attr_q = false;
        if (!attr_q) {
#define QL_ATTR(NSTV) hipFuncSetAttribute(reinterpret_cast<const void*>(&norm_glu_lds_kernel<NSTV, RING, 1, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn)
            QL_ATTR(7); QL_ATTR(4); QL_ATTR(3);
#undef QL_ATTR
            attr_q = true;
        }
#define QL_LAUNCH(NSTV) hipLaunchKernelGGL((norm_glu_lds_kernel<NSTV, RING, 1, 1, 1>), grid, dim3(512), dyn, s, (const bf16_t*)x, (const bf16_t*)lnw, (const bf16_t*)Wqkv, \
                                           (bf16_t*)nullptr, (int)M, N, K, ldx, ldw, (int64_t)0, eps, (int64_t)(head_dim / 2), (const bf16_t*)bias, qe)
        if (nst == 7) QL_LAUNCH(7); else if (nst == 4) QL_LAUNCH(4); else QL_LAUNCH(3);
#undef QL_LAUNCH
        TR1_LAUNCH_CHECK();
    }
    static int xlds = -1;                            // TR1_QKV_XLDS=0: activation rows through the vector-memory path as well (A/B measurements)
    if (xlds < 0) { xlds = 1; }
    if (M <= 16 && xlds && K / 64 / 8 >= 2 && K * 32 <= 120 * 1024 && (int64_t)M * ldx * 2 < 0x7fffffffLL) {
        static bool attr_x = false;
        if (!attr_x) { hipFuncSetAttribute(reinterpret_cast<const void*>(&norm_gemm_skinny_kernel<8, 2, 1, false, 2, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024); attr_x = true; }
        hipLaunchKernelGGL((norm_gemm_skinny_kernel<8, 2, 1, false, 2, true, true>), grid, dim3(512), (size_t)(K * 32), s, (const bf16_t*)x, (const bf16_t*)lnw,
                           (const bf16_t*)Wqkv, (bf16_t*)nullptr, (const bf16_t*)bias, (int)M, N, K, ldx, ldw, (int64_t)0, eps, (int64_t)0, qe);
    }
    else if (M <= 16) NGQ(8, 2, 1); else if (M <= 32) { const int c = ng32_cfg(); if (c == 1) NGQ(8, 2, 2); else if (c == 2) NGQ(8, 1, 2); else if (c == 3) NGQ(4, 1, 2); else NGQ(4, 2, 2); } else NGQ(4, 2, 4);
#undef NGQ
    TR1_LAUNCH_CHECK();
}

// Launch of the decode-regime kernel (single pass over K; the split-K + fixup form is launched by tr1_gemm_skinny_fixup).
static void launch_skinny(const void* A, const void* B, void* C, const void* bias, const void* residual, int64_t M, int64_t N, int64_t K,
                          int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int out_f32, int ksplit, hipStream_t s) {
    static int force_ncol = -1;
    if (force_ncol < 0) { force_ncol = 0; }
#define SK(WV, UN, NC, MGR)                                                                                                          \
    hipLaunchKernelGGL((gemm_skinny_kernel<WV, UN, NC, MGR>), dim3((unsigned)((N + 16 * NC - 1) / (16 * NC)), (unsigned)ksplit),    \
                       dim3(WV * 64), 0, s, (const bf16_t*)A, (const bf16_t*)B, out_f32 ? nullptr : (bf16_t*)C,                     \
                       out_f32 ? (float*)C : nullptr, (const bf16_t*)bias, (const bf16_t*)residual, (int)M, N, K, lda, ldb, ldc, ldr,   \
                       (float*)nullptr, (int*)nullptr)
    // choices measured on MI355X with tools/microbench.py skinny (non-temporal loads hurt; 8-way in-block split-K pays for long K)
    // (A/B on MI355X, M = 16: gate_up 37888x3584 67.6 -> 56.8 us with NCOL 2; lm_head 152064x3584 247 -> 188 us with NCOL 4;
    //  the 3584x18944 down projection has too few column groups for NCOL > 1 and wants split-K instead)
    const int ncol = force_ncol > 0 ? force_ncol : (N >= 100000 ? 4 : (N >= 4096 && ksplit == 1 ? 2 : 1));
    const bool longk = K / ksplit >= 8192;
    if (M <= 16) {
        if (longk) { if (ncol >= 2 && N >= 16384) SK(8, 2, 2, 1); else SK(8, 4, 1, 1); }
        else if (ncol == 4) SK(4, 2, 4, 1);
        else if (ncol == 2) SK(4, 4, 2, 1);
        else {
            static int xlds = -1;                    // TR1_SKINNY_XLDS=0: activation rows through the vector-memory path (A/B measurements)
            if (xlds < 0) { xlds = 1; }
            if (xlds && ksplit == 1 && K / 64 / 4 >= 4 && K * 32 <= 120 * 1024 && (int64_t)M * lda * 2 < 0x7fffffffLL) {
                static bool attr_x = false;
                if (!attr_x) { hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_kernel<4, 4, 1, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024); attr_x = true; }
                hipLaunchKernelGGL((gemm_skinny_kernel<4, 4, 1, 1, true>), dim3((unsigned)((N + 15) / 16), 1u), dim3(256), (size_t)(K * 32), s, (const bf16_t*)A,
                                   (const bf16_t*)B, out_f32 ? nullptr : (bf16_t*)C, out_f32 ? (float*)C : nullptr, (const bf16_t*)bias, (const bf16_t*)residual,
                                   (int)M, N, K, lda, ldb, ldc, ldr, (float*)nullptr, (int*)nullptr);
            } else SK(4, 4, 1, 1);
        }
    } else if (M <= 32) {       // LDS reduce buffer: WAVES * NCOL * MG * 1088 B <= 64 KB
        if (longk) { if (ncol >= 2 && N >= 16384) SK(8, 2, 2, 2); else SK(8, 2, 1, 2); }
        else if (ncol == 4) SK(4, 2, 4, 2);
        else if (ncol == 2) SK(4, 2, 2, 2);
        else SK(4, 4, 1, 2);
    } else {
        if (longk) SK(8, 2, 1, 4);
        else if (ncol >= 2) SK(4, 2, 2, 4);
        else SK(4, 2, 1, 4);
    }
#undef SK
}

extern "C" int tr1_gemm_nt_bf16(const void* A, const void* B, void* C, const void* bias, const void* residual, int64_t M, int64_t N,
                                int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int out_f32, int accumulate, void* stream) {
    TR1_CHECK_ARG(K % BK == 0, "gemm_nt: K must be a multiple of 64 (pad the operands)");
    TR1_CHECK_ARG(N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, "gemm_nt: N%8, lda%8, ldb%8, ldc%4 required");
    TR1_CHECK_ARG(!accumulate || out_f32, "gemm_nt: accumulate requires fp32 output");
    TR1_CHECK_ARG(!residual || ldr % 8 == 0, "gemm_nt: ldr%8 required");
    TR1_CHECK_ARG(out_f32 || ldc % 8 == 0, "gemm_nt: ldc%8 required for bf16 output");
    if (M == 0 || N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (M <= 64 && !accumulate && K >= 256) {
        launch_skinny(A, B, C, bias, residual, M, N, K, lda, ldb, ldc, ldr, out_f32, 1, s);
        TR1_LAUNCH_CHECK();
    }
    {   // tile choice: CU-rounds x block area / relative efficiency of the structure (tile-count quantisation).
        // 128x128 runs 2 blocks per CU (512 slots), the phased 8-wave forms 1 block per CU at ~1.25x the MFMA rate per CU.
        static int force = -1;
        if (force < 0) { force = 0; }
        auto blocks = [&](int64_t bm, int64_t bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
        auto cost = [&](int64_t bm, int64_t bn, int64_t slots, double eff) {
            const int64_t t = blocks(bm, bn);
            return (double)((t + slots - 1) / slots) * (double)slots * (double)(bm * bn) / eff;
        };
        int rt = 0;
        if (force == 224) rt = 7; else if (force == 256) rt = 8; else if (force == 288) rt = 9; else if (force == 320) rt = 10; else if (force == 128) rt = 0;
        else if (M >= 512 && N >= 256) {
            double best = cost(BM, BN, 512, 0.80);
            // intrinsic efficiency of the phased 8-wave forms relative to 256 x 256 (more A-fragment reuse per B fragment with taller tiles),
            // measured on M = 37888, N = 3584, K = 5120 and 8192^3 after removing tile-count quantisation: 224: 0.94, 288: 1.025, 320: 1.03;
            // the 128 x 128 form reaches 0.80 of the 256 x 256 rate per CU (tools/microbench.py gemm with TR1_GEMM_TILE forced)
            static const double eff[4] = {0.94, 1.0, 1.025, 1.03};
            for (int r = 7; r <= 10; ++r) {
                const double c = cost(r * 32, BN2, 256, eff[r - 7]);
                if (c < best) { best = c; rt = r; }
            }
        }
        if (rt) {
            const int bmx = rt * 32;
            const int64_t t2m = (M + bmx - 1) / bmx, t2n = (N + BN2 - 1) / BN2;
            static int phased = -1;                    // TR1_GEMM_PHASED=0 selects the single-barrier form (A/B measurements)
            if (phased < 0) { phased = 1; }
            const size_t dyn = 2 * ((size_t)bmx * BK * 2 + TILE2_BYTES) + (phased ? 4096 : 0);
            static bool attr_set = false;
            if (!attr_set) {
                const int mx = (int)(2 * (320 * BK * 2 + TILE2_BYTES)) + 4096;
#define SETA(OF, AC, R) do { hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt256_kernel<OF, AC, R>), hipFuncAttributeMaxDynamicSharedMemorySize, mx); \
                             hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt8p_kernel<OF, AC, R>), hipFuncAttributeMaxDynamicSharedMemorySize, mx); } while (0)
#define SETR(R) do { SETA(false, false, R); SETA(true, false, R); SETA(true, true, R); } while (0)
                SETR(7); SETR(8); SETR(9); SETR(10);
#undef SETR
#undef SETA
                attr_set = true;
            }
            dim3 grid2((unsigned)(t2m * t2n));
#define LAUNCH2(OF, AC, R)                                                                                                            \
    do { if (phased) hipLaunchKernelGGL((gemm_nt8p_kernel<OF, AC, R>), grid2, dim3(512), dyn, s, (const bf16_t*)A, (const bf16_t*)B, C, (const bf16_t*)bias, \
                       (const bf16_t*)residual, M, N, K, lda, ldb, ldc, ldr, (int)t2m, (int)t2n, GemmEpi{});                          \
    else hipLaunchKernelGGL((gemm_nt256_kernel<OF, AC, R>), grid2, dim3(512), dyn, s, (const bf16_t*)A, (const bf16_t*)B, C, (const bf16_t*)bias, \
                       (const bf16_t*)residual, M, N, K, lda, ldb, ldc, ldr, (int)t2m, (int)t2n); } while (0)
#define LAUNCH2R(R) do { if (out_f32) { if (accumulate) LAUNCH2(true, true, R); else LAUNCH2(true, false, R); } else LAUNCH2(false, false, R); } while (0)
            if (rt == 7) LAUNCH2R(7); else if (rt == 9) LAUNCH2R(9); else if (rt == 10) LAUNCH2R(10); else LAUNCH2R(8);
#undef LAUNCH2R
#undef LAUNCH2
            TR1_LAUNCH_CHECK();
        }
    }
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (int)((N + BN - 1) / BN);
    dim3 grid((unsigned)(tiles_m * tiles_n));
#define LAUNCH(OF, AC)                                                                                                              \
    hipLaunchKernelGGL((gemm_nt_kernel<OF, AC>), grid, dim3(256), 0, s, (const bf16_t*)A, (const bf16_t*)B, C, (const bf16_t*)bias, \
                       (const bf16_t*)residual, M, N, K, lda, ldb, ldc, ldr, tiles_m, tiles_n)
    if (out_f32) { if (accumulate) LAUNCH(true, true); else LAUNCH(true, false); }
    else LAUNCH(false, false);
#undef LAUNCH
    TR1_LAUNCH_CHECK();
}

// C[M,N] = A[M,K] * B[K,N]  ("NN": B is K-major - the dgrad dX = dY * W reads the weight as stored, no W^T copy).  Phased 8-wave forms only
// (M >= 512, N >= 256): B tiles are staged as 16 k-rows x 256 columns per round and read back transposed (ds_read_b64_tr_b16).  bf16 output.
extern "C" int tr1_gemm_nn_bf16(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                                void* stream) {
    TR1_CHECK_ARG(K % BK == 0, "gemm_nn: K must be a multiple of 64");
    TR1_CHECK_ARG(N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0, "gemm_nn: N%8, lda%8, ldb%8, ldc%8 required");
    TR1_CHECK_ARG(M >= 512 && N >= 256, "gemm_nn: M >= 512 and N >= 256 required (smaller problems: transpose B and use gemm_nt)");
    hipStream_t s = (hipStream_t)stream;
    auto cost = [&](int64_t bm, double eff) {
        const int64_t t = ((M + bm - 1) / bm) * ((N + BN2 - 1) / BN2);
        return (double)((t + 255) / 256) * 256.0 * (double)(bm * BN2) / eff;
    };
    static const double eff[4] = {0.94, 1.0, 1.025, 1.03};
    int rt = 8; double best = cost(256, 1.0);
    for (int r = 7; r <= 10; ++r) { const double c = cost(r * 32, eff[r - 7]); if (c < best) { best = c; rt = r; } }
    const int bmx = rt * 32;
    const int64_t t2m = (M + bmx - 1) / bmx, t2n = (N + BN2 - 1) / BN2;
    const size_t dyn = 2 * ((size_t)bmx * BK * 2 + TILE2_BYTES) + 4096;
    static bool attr_set = false;
    if (!attr_set) {
        const int mx = (int)(2 * (320 * BK * 2 + TILE2_BYTES)) + 4096;
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt8p_kernel<false, false, 7, true>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt8p_kernel<false, false, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt8p_kernel<false, false, 9, true>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt8p_kernel<false, false, 10, true>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        attr_set = true;
    }
    dim3 grid2((unsigned)(t2m * t2n));
#define LAUNCHNN(R) hipLaunchKernelGGL((gemm_nt8p_kernel<false, false, R, true>), grid2, dim3(512), dyn, s, (const bf16_t*)A, (const bf16_t*)B, C, \
                                       (const bf16_t*)nullptr, (const bf16_t*)nullptr, M, N, K, lda, ldb, ldc, (int64_t)0, (int)t2m, (int)t2n, GemmEpi{})
    if (rt == 7) LAUNCHNN(7); else if (rt == 9) LAUNCHNN(9); else if (rt == 10) LAUNCHNN(10); else LAUNCHNN(8);
#undef LAUNCHNN
    TR1_LAUNCH_CHECK();
}

// C[M,N] (fp32) (+)= A[M,K] * B[K,N] with B K-major and only its first `b_rows` k rows valid: the WEIGHT GRADIENT dW[n,k] += sum_t dY^T[n,t] X[t,k]
// with A = dY^T (the transposed copy, zero-padded to a multiple of 64 tokens) and B = X AS STORED - no X^T copy (the 18944-column SwiGLU
// output of the down projection was the largest transpose of the backward).  Same kernel, tiles and accumulation order as tr1_gemm_nn_bf16.
extern "C" int tr1_gemm_nn_acc_f32(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                                   int accumulate, int64_t b_rows, void* stream) {
    TR1_CHECK_ARG(K % BK == 0, "gemm_nn_acc: K must be a multiple of 64");
    TR1_CHECK_ARG(N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, "gemm_nn_acc: N%8, lda%8, ldb%8, ldc%4 required");
    TR1_CHECK_ARG(M >= 512 && N >= 256, "gemm_nn_acc: M >= 512 and N >= 256 required");
    TR1_CHECK_ARG(b_rows >= 1 && b_rows <= K, "gemm_nn_acc: 1 <= b_rows <= K");
    hipStream_t s = (hipStream_t)stream;
    auto cost = [&](int64_t bm, double eff) {
        const int64_t t = ((M + bm - 1) / bm) * ((N + BN2 - 1) / BN2);
        return (double)((t + 255) / 256) * 256.0 * (double)(bm * BN2) / eff;
    };
    static const double eff[4] = {0.94, 1.0, 1.025, 1.03};
    int rt = 8; double best = cost(256, 1.0);
    for (int r = 7; r <= 10; ++r) { const double c = cost(r * 32, eff[r - 7]); if (c < best) { best = c; rt = r; } }
    const int bmx = rt * 32;
    const int64_t t2m = (M + bmx - 1) / bmx, t2n = (N + BN2 - 1) / BN2;
    const size_t dyn = 2 * ((size_t)bmx * BK * 2 + TILE2_BYTES) + 4096;
    static bool attr_set = false;
    if (!attr_set) {
        const int mx = (int)(2 * (320 * BK * 2 + TILE2_BYTES)) + 4096;
#define SETN(AC, R) hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt8p_kernel<true, AC, R, true>), hipFuncAttributeMaxDynamicSharedMemorySize, mx)
        SETN(false, 7); SETN(false, 8); SETN(false, 9); SETN(false, 10); SETN(true, 7); SETN(true, 8); SETN(true, 9); SETN(true, 10);
#undef SETN
        attr_set = true;
    }
    dim3 grid2((unsigned)(t2m * t2n));
#define LAUNCHNA(AC, R) hipLaunchKernelGGL((gemm_nt8p_kernel<true, AC, R, true>), grid2, dim3(512), dyn, s, (const bf16_t*)A, (const bf16_t*)B, C, \
                                           (const bf16_t*)nullptr, (const bf16_t*)nullptr, M, N, K, lda, ldb, ldc, b_rows, (int)t2m, (int)t2n, GemmEpi{})
#define LAUNCHNAR(R) do { if (accumulate) LAUNCHNA(true, R); else LAUNCHNA(false, R); } while (0)
    if (rt == 7) LAUNCHNAR(7); else if (rt == 9) LAUNCHNAR(9); else if (rt == 10) LAUNCHNAR(10); else LAUNCHNAR(8);
#undef LAUNCHNAR
#undef LAUNCHNA
    TR1_LAUNCH_CHECK();
}

static int epi_pick_rt(int64_t M, int64_t Ntiles);
// Weight gradient C[M, N] fp32 (+)= A B^T (b_kmajor = 0: B = X^T [N, K]) or A B (b_kmajor = 1: B = X as stored [K, N], its first b_rows rows valid), on the
// phased 8-wave kernel, which ALSO leaves the sum of squares of every value it stored in sumsq_partials (one float per wave: 8 x blocks; *n_partials receives
// the count) - in the last micro-step of an accumulation window that is the squared norm of the final gradient, so the optimizer's grad-norm pass does not
// have to read these matrices again.  Same kernel, k order and (for shapes the NT dispatch gives to the 8-wave tiles) tile choice as tr1_gemm_nt_bf16(out_f32) / tr1_gemm_nn_acc_f32: bit-identical C.
extern "C" int tr1_wgrad_f32_sumsq(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int accumulate,
                                   int b_kmajor, int64_t b_rows, void* sumsq_partials, int64_t partials_capacity, int64_t* n_partials, void* wire_bf16, int64_t ld_wire,
                                   void* stream) {
    TR1_CHECK_ARG(K % BK == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, "wgrad_f32_sumsq: K%64, N%8, lda%8, ldb%8, ldc%4 required");
    TR1_CHECK_ARG(M >= 512 && N >= 256, "wgrad_f32_sumsq: M >= 512 and N >= 256 required (smaller gradients: plain GEMM + tr1_sumsq_accum)");
    TR1_CHECK_ARG(!b_kmajor || (b_rows >= 1 && b_rows <= K), "wgrad_f32_sumsq: 1 <= b_rows <= K");
    TR1_CHECK_ARG(sumsq_partials && n_partials, "wgrad_f32_sumsq: partials buffer required");
#if !TR1_EPI_LDS
    // the sums of squares and the bf16 wire copy leave from the LDS-staged epilogue only (store_acc256_lds); a -DTR1_EPI_LDS=0 variant build must not
    // pretend to have written them (the caller would mark the range as exchanged-ready and all-reduce stale bytes)
    TR1_CHECK_ARG(false, "wgrad_f32_sumsq: built with TR1_EPI_LDS=0 - the sum-of-squares / wire-copy epilogue does not exist in this build");
#endif
    hipStream_t s = (hipStream_t)stream;
    const int64_t t2n = (N + BN2 - 1) / BN2;
    int rt;
    if (b_kmajor) rt = epi_pick_rt(M, t2n);
    else {      // the NT dispatch's own choice among the 8-wave tiles (tr1_gemm_nt_bf16), so C is bit-identical to that path
        auto cost = [&](int64_t bm, double eff) { const int64_t t = ((M + bm - 1) / bm) * t2n; return (double)((t + 255) / 256) * 256.0 * (double)(bm * BN2) / eff; };
        static const double eff[4] = {0.94, 1.0, 1.025, 1.03};
        rt = 7; double best = cost(224, eff[0]);
        for (int r = 8; r <= 10; ++r) { const double c = cost(r * 32, eff[r - 7]); if (c < best) { best = c; rt = r; } }
    }
    const int bmx = rt * 32;
    const int64_t t2m = (M + bmx - 1) / bmx, blocks = t2m * t2n;
    TR1_CHECK_ARG(blocks * 8 <= partials_capacity, "wgrad_f32_sumsq: partials buffer too small (8 floats per 256-column tile block)");
    const size_t dyn = 2 * ((size_t)bmx * BK * 2 + TILE2_BYTES) + 4096;
    {
        static bool set_ = false;
        if (!set_) {
            const int mx = (int)(2 * (320 * BK * 2 + TILE2_BYTES)) + 4096;
#define SETW(AC, R, KM) hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt8p_kernel<true, AC, R, KM>), hipFuncAttributeMaxDynamicSharedMemorySize, mx)
#define SETWR(R) do { SETW(false, R, false); SETW(true, R, false); SETW(false, R, true); SETW(true, R, true); } while (0)
            SETWR(7); SETWR(8); SETWR(9); SETWR(10);
#undef SETWR
#undef SETW
            set_ = true;
        }
    }
    TR1_CHECK_ARG(!wire_bf16 || ld_wire % 4 == 0, "wgrad_f32_sumsq: ld_wire % 4 required");
    GemmEpi ep{}; ep.p0 = sumsq_partials; ep.p1 = wire_bf16; ep.ld1 = ld_wire;
#define LW(AC, R, KM) hipLaunchKernelGGL((gemm_nt8p_kernel<true, AC, R, KM>), dim3((unsigned)blocks), dim3(512), dyn, s, (const bf16_t*)A, (const bf16_t*)B, C, \
                                         (const bf16_t*)nullptr, (const bf16_t*)nullptr, M, N, K, lda, ldb, ldc, (int64_t)(KM ? b_rows : 0), (int)t2m, (int)t2n, ep)
#define LWR(R) do { if (b_kmajor) { if (accumulate) LW(true, R, true); else LW(false, R, true); } else { if (accumulate) LW(true, R, false); else LW(false, R, false); } } while (0)
    if (rt == 7) LWR(7); else if (rt == 9) LWR(9); else if (rt == 10) LWR(10); else LWR(8);
#undef LWR
#undef LW
    *n_partials = blocks * 8;
    TR1_LAUNCH_CHECK();
}

// ---- fused-epilogue training GEMMs (EPI 2 / 3 / 4 of gemm_nt8p_kernel) ----------------------------------------------------------------
static int epi_pick_rt(int64_t M, int64_t Ntiles) {
    auto cost = [&](int64_t bm, double eff) {
        const int64_t t = ((M + bm - 1) / bm) * Ntiles;
        return (double)((t + 255) / 256) * 256.0 * (double)(bm * BN2) / eff;
    };
    static const double eff[4] = {0.94, 1.0, 1.025, 1.03};
    int rt = 8; double best = cost(256, 1.0);
    for (int r = 7; r <= 10; ++r) { const double c = cost(r * 32, eff[r - 7]); if (c < best) { best = c; rt = r; } }
    return rt;
}
#define EPI_LAUNCH(KERN_ARGS, RTV, ...)                                                                                                   \
    do { if (RTV == 7) { hipLaunchKernelGGL((gemm_nt8p_kernel<KERN_ARGS(7)>), __VA_ARGS__); }                                             \
         else if (RTV == 9) { hipLaunchKernelGGL((gemm_nt8p_kernel<KERN_ARGS(9)>), __VA_ARGS__); }                                        \
         else if (RTV == 10) { hipLaunchKernelGGL((gemm_nt8p_kernel<KERN_ARGS(10)>), __VA_ARGS__); }                                      \
         else { hipLaunchKernelGGL((gemm_nt8p_kernel<KERN_ARGS(8)>), __VA_ARGS__); } } while (0)
#define EPI_SETATTR(KERN_ARGS)                                                                                                             \
    do { static bool set_ = false; if (!set_) { const int mx = (int)(2 * (320 * BK * 2 + TILE2_BYTES)) + 4096;                             \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt8p_kernel<KERN_ARGS(7)>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);  \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt8p_kernel<KERN_ARGS(8)>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);  \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt8p_kernel<KERN_ARGS(9)>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);  \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt8p_kernel<KERN_ARGS(10)>), hipFuncAttributeMaxDynamicSharedMemorySize, mx); \
        set_ = true; } } while (0)

// a[M, I] = silu(x Wg^T) * (x Wu^T) with Wgu = [2I, K] (gate rows, then up rows); gu_out (optional) receives the projection itself [M, 2I] for the backward.
// Bit-identical to tr1_gemm_nt_bf16 + tr1_swiglu_fwd.
extern "C" int tr1_gemm_glu_bf16(const void* x, const void* Wgu, const void* bias, void* a_out, void* gu_out, int64_t M, int64_t I, int64_t K, int64_t ldx,
                                 int64_t ldw, int64_t lda, int64_t ldgu, void* stream) {
    TR1_CHECK_ARG(K % BK == 0 && I % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && lda % 8 == 0 && (!gu_out || ldgu % 8 == 0), "gemm_glu: K%64, I%8, ld%8 required");
    TR1_CHECK_ARG(I < (1 << 30), "gemm_glu: I too large");
    if (M == 0 || I == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int64_t t2n = (I + 127) / 128;
    const int rt = epi_pick_rt(M, t2n);
    const int bmx = rt * 32;
    const int64_t t2m = (M + bmx - 1) / bmx;
    const size_t dyn = 2 * ((size_t)bmx * BK * 2 + TILE2_BYTES) + 4096;
#define KA_GLU(R) false, false, R, false, 2
    EPI_SETATTR(KA_GLU);
    GemmEpi ep{}; ep.p0 = gu_out; ep.ld0 = ldgu; ep.i0 = (int)I;
    EPI_LAUNCH(KA_GLU, rt, dim3((unsigned)(t2m * t2n)), dim3(512), dyn, s, (const bf16_t*)x, (const bf16_t*)Wgu, a_out, (const bf16_t*)bias,
               (const bf16_t*)nullptr, M, 2 * I, K, ldx, ldw, lda, (int64_t)0, (int)t2m, (int)t2n, ep);
#undef KA_GLU
    TR1_LAUNCH_CHECK();
}

// y[M, N] = quick_gelu(x W^T + bias) (Qwen2-VL vision MLP fc1 + activation, TF:300-301).  Bit-identical to tr1_gemm_nt_bf16 (bias) + tr1_quickgelu_fwd.
extern "C" int tr1_gemm_bias_quickgelu_bf16(const void* x, const void* W, const void* bias, void* y, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw,
                                            int64_t ldy, void* stream) {
    TR1_CHECK_ARG(K % BK == 0 && N % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldy % 8 == 0, "gemm_bias_quickgelu: K%64, N%8, ld%8 required");
    if (M == 0 || N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int64_t t2n = (N + BN2 - 1) / BN2;
    const int rt = epi_pick_rt(M, t2n);
    const int bmx = rt * 32;
    const int64_t t2m = (M + bmx - 1) / bmx;
    const size_t dyn = 2 * ((size_t)bmx * BK * 2 + TILE2_BYTES) + 4096;
#define KA_QG(R) false, false, R, false, 5
    EPI_SETATTR(KA_QG);
    EPI_LAUNCH(KA_QG, rt, dim3((unsigned)(t2m * t2n)), dim3(512), dyn, s, (const bf16_t*)x, (const bf16_t*)W, y, (const bf16_t*)bias,
               (const bf16_t*)nullptr, M, N, K, ldx, ldw, ldy, (int64_t)0, (int)t2m, (int)t2n, GemmEpi{});
#undef KA_QG
    TR1_LAUNCH_CHECK();
}

// Fused q|k|v projection + bias + M-RoPE for head dim 128: q_out[M, n_heads*128] and k_out[M, n_kv*128] rotated (cos / sin fp32 [M, 64]), v_out[M, n_kv*128]
// plain.  Bit-identical to tr1_gemm_nt_bf16 (bias) + tr1_rope_apply on the q and k columns.  n_heads and n_kv must be even (whole 256-column tiles).
extern "C" int tr1_gemm_qkv_rope_bf16(const void* x, const void* Wqkv, const void* bias, const void* cosb, const void* sinb, void* q_out, int64_t ldq,
                                      void* k_out, int64_t ldk, void* v_out, int64_t ldv, int64_t M, int64_t n_heads, int64_t n_kv, int64_t head_dim,
                                      int64_t K, int64_t ldx, int64_t ldw, void* stream) {
    TR1_CHECK_ARG(head_dim == 128 && n_heads % 2 == 0 && n_kv % 2 == 0, "gemm_qkv_rope: head_dim 128 and even head counts required");
    TR1_CHECK_ARG(K % BK == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0, "gemm_qkv_rope: K%64, ld%8 required");
    if (M == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int64_t N = (n_heads + 2 * n_kv) * 128, t2n = N / 256;
    const int rt = epi_pick_rt(M, t2n);
    const int bmx = rt * 32;
    const int64_t t2m = (M + bmx - 1) / bmx;
    const size_t dyn = 2 * ((size_t)bmx * BK * 2 + TILE2_BYTES) + 4096;
#define KA_QKV(R) false, false, R, false, 4
    EPI_SETATTR(KA_QKV);
    GemmEpi ep{}; ep.p0 = k_out; ep.ld0 = ldk; ep.p1 = v_out; ep.ld1 = ldv; ep.f0 = (const float*)cosb; ep.f1 = (const float*)sinb;
    ep.i0 = (int)(n_heads * 128); ep.i1 = (int)(n_kv * 128);
    EPI_LAUNCH(KA_QKV, rt, dim3((unsigned)(t2m * t2n)), dim3(512), dyn, s, (const bf16_t*)x, (const bf16_t*)Wqkv, q_out, (const bf16_t*)bias,
               (const bf16_t*)nullptr, M, N, K, ldx, ldw, ldq, (int64_t)0, (int)t2m, (int)t2n, ep);
#undef KA_QKV
    TR1_LAUNCH_CHECK();
}

// Vision-tower attention input (Qwen2-VL / Qwen2.5-VL blocks, head dim 2 * half = 80): fused q|k|v projection + bias + 2-D rotary embedding (cos / sin fp32
// [M, half]), written as 128-wide zero-PADDED heads q128 / k128 / v128 [M, n_heads * 128] (feature d < half at d, d + half at 64 + d; the caller zero-fills
// the buffers once).  Values bit-identical to tr1_gemm_nt_bf16 (bias) + tr1_rope_apply on q and k.  n_heads * half must be a multiple of 128.
extern "C" int tr1_gemm_qkv_rope_vit_bf16(const void* x, const void* Wqkv, const void* bias, const void* cosb, const void* sinb, void* q128, int64_t ldq,
                                          void* k128, int64_t ldk, void* v128, int64_t ldv, int64_t M, int64_t n_heads, int64_t half, int64_t K,
                                          int64_t ldx, int64_t ldw, void* stream) {
    TR1_CHECK_ARG(half % 8 == 0 && half <= 64 && (n_heads * half) % 128 == 0, "gemm_qkv_rope_vit: half % 8 == 0, half <= 64, n_heads * half % 128 == 0 required");
    TR1_CHECK_ARG(K % BK == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0, "gemm_qkv_rope_vit: K%64, ld%8 required");
    if (M == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int64_t pairs = n_heads * half, N = 6 * pairs, t2n = N / 256;
    const int rt = epi_pick_rt(M, t2n);
    const int bmx = rt * 32;
    const int64_t t2m = (M + bmx - 1) / bmx;
    const size_t dyn = 2 * ((size_t)bmx * BK * 2 + TILE2_BYTES) + 4096;
#define KA_VQ(R) false, false, R, false, 7
    EPI_SETATTR(KA_VQ);
    GemmEpi ep{}; ep.p0 = k128; ep.ld0 = ldk; ep.p1 = v128; ep.ld1 = ldv; ep.f0 = (const float*)cosb; ep.f1 = (const float*)sinb;
    ep.i0 = (int)pairs; ep.i1 = (int)half;
    EPI_LAUNCH(KA_VQ, rt, dim3((unsigned)(t2m * t2n)), dim3(512), dyn, s, (const bf16_t*)x, (const bf16_t*)Wqkv, q128, (const bf16_t*)bias,
               (const bf16_t*)nullptr, M, N, K, ldx, ldw, ldq, (int64_t)0, (int)t2m, (int)t2n, ep);
#undef KA_VQ
    TR1_LAUNCH_CHECK();
}

// dgu[M, 2I] = SwiGLU backward of da = dh[M, H] * Wd[H, I] (Wd = the down projection as stored, K-major operand), with gu[M, 2I] the saved projection.
// Bit-identical to tr1_gemm_nn_bf16 + tr1_swiglu_bwd; da never exists in HBM.
extern "C" int tr1_gemm_nn_glubwd_bf16(const void* dh, const void* Wd, const void* gu, void* dgu, int64_t M, int64_t I, int64_t H, int64_t lda, int64_t ldb,
                                       int64_t ldgu, int64_t lddgu, void* dgu_t, int64_t ld_t, void* stream) {
    TR1_CHECK_ARG(!dgu_t || (ld_t % 64 == 0 && ld_t >= M && ld_t < M + 64), "gemm_nn_glubwd: dgu^T needs ld_t = tokens rounded up to 64");
    TR1_CHECK_ARG(H % BK == 0 && I % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldgu % 8 == 0 && lddgu % 8 == 0, "gemm_nn_glubwd: H%64, I%8, ld%8 required");
    TR1_CHECK_ARG(M >= 512 && I >= 256, "gemm_nn_glubwd: M >= 512 and I >= 256 required");
    hipStream_t s = (hipStream_t)stream;
    const int64_t t2n = (I + BN2 - 1) / BN2;
    const int rt = epi_pick_rt(M, t2n);
    const int bmx = rt * 32;
    const int64_t t2m = (M + bmx - 1) / bmx;
    const size_t dyn = 2 * ((size_t)bmx * BK * 2 + TILE2_BYTES) + 4096;
#define KA_GB(R) false, false, R, true, 3
    EPI_SETATTR(KA_GB);
    GemmEpi ep{}; ep.p0 = const_cast<void*>(gu); ep.ld0 = ldgu; ep.i0 = (int)I; ep.p1 = dgu_t; ep.ld1 = ld_t;
    EPI_LAUNCH(KA_GB, rt, dim3((unsigned)(t2m * t2n)), dim3(512), dyn, s, (const bf16_t*)dh, (const bf16_t*)Wd, dgu, (const bf16_t*)nullptr,
               (const bf16_t*)nullptr, M, I, H, lda, ldb, lddgu, (int64_t)0, (int)t2m, (int)t2n, ep);
#undef KA_GB
    TR1_LAUNCH_CHECK();
}

// C[M, N] (bf16) = A B^T or A B (+bias)(+residual) with a deterministic S-way split over K: thin outputs over a long K leave most CUs idle in the plain
// forms - the continuation forward's down projection (1600 x 3584 x 18944: 98 tiles of 256 x 256 for 256 CUs, 838 TFLOP/s), its o projection, and the
// lm_head's data gradient (1600 x 3584 over K = 152064, which used to pay a 2.2 GB transpose of the lm_head weight in front of a 128 x 128-tile GEMM).
// All S shares of the reduction run as blocks of ONE launch into fp32 planes of ws_f32, a second launch adds the planes in a fixed order.  (RT, S) by the
// cost model below: rounds of 256 blocks x tile rows x k tiles per share.  b_kmajor: B = [K, N] as stored (the weight itself in a dgrad).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ planes, int S, int64_t plane, const bf16_t* __restrict__ bias,
                                                            const bf16_t* __restrict__ residual, int64_t ldr, bf16_t* __restrict__ C, int64_t ldc, int64_t M, int64_t N) {
    const int64_t nch = N >> 3, total = M * nch;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / nch, n = (i - m * nch) * 8;
        const float* p = planes + m * N + n;
        f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(p), a1 = *reinterpret_cast<const f32x4_t*>(p + 4);
        for (int z = 1; z < S; ++z) {
            a0 += *reinterpret_cast<const f32x4_t*>(p + z * plane); a1 += *reinterpret_cast<const f32x4_t*>(p + z * plane + 4);
        }
        float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        if (bias) {
            const u32x4_t bv = *reinterpret_cast<const u32x4_t*>(bias + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] += bflo(bv[e]); v[2 * e + 1] += bfhi(bv[e]); }
        }
        if (residual) {
            const u32x4_t rv = *reinterpret_cast<const u32x4_t*>(residual + m * ldr + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] += bflo(rv[e]); v[2 * e + 1] += bfhi(rv[e]); }
        }
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack2bf(v[2 * e], v[2 * e + 1]);
        *reinterpret_cast<u32x4_t*>(C + m * ldc + n) = o;
    }
}
// (RT, S) for a split-K launch: minimise rounds(256 blocks) x tile rows x k tiles per share / efficiency, plus the partial planes' traffic
static void splitk_pick(int64_t M, int64_t N, int64_t K, int max_s, int* rt_out, int* s_out) {
    static const double eff[4] = {0.94, 1.0, 1.025, 1.03};
    const int64_t t2n = (N + BN2 - 1) / BN2, nk = K / BK;
    double best = 1e300; int brt = 8, bs = 1;
    for (int r = 7; r <= 10; ++r)
        for (int S = 1; S <= max_s && S <= nk; ++S) {
            const int64_t tiles = ((M + r * 32 - 1) / (r * 32)) * t2n, blocks = tiles * S;
            const double rounds = (double)((blocks + 255) / 256), kshare = (double)((nk + S - 1) / S);
            // block time ~ rows x k tiles; a plane costs one fp32 write + read of M x N per share, priced against the GEMM's per-CU rate (~25 k-tile rows per 4 KB)
            const double c = rounds * (r * 32) * kshare / eff[r - 7] + (S > 1 ? 0.02 * S * (double)(M * N) / 256.0 / 64.0 : 0.0);
            if (c < best) { best = c; brt = r; bs = S; }
        }
    *rt_out = brt; *s_out = bs;
}
extern "C" int64_t tr1_gemm_splitk_max_splits(void) { return 8; }
extern "C" int tr1_gemm_splitk_bf16(const void* A, const void* B, void* C, const void* bias, const void* residual, int64_t M, int64_t N, int64_t K,
                                    int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int b_kmajor, void* ws_f32, int64_t ws_floats, void* stream) {
    TR1_CHECK_ARG(K % BK == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && (!residual || ldr % 8 == 0), "gemm_splitk: K%64, N%8, ld%8 required");
    TR1_CHECK_ARG(ws_f32 && ws_floats >= 2 * M * N, "gemm_splitk: workspace of at least 2*M*N floats required (8*M*N for every split count)");
    if (M == 0 || N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    int rt, S;
    const int64_t max_s = ws_floats / (M * N);
    splitk_pick(M, N, K, (int)(max_s < 8 ? max_s : 8), &rt, &S);
    if (S < 2) S = 2;        // (the caller asked for the split form: thin outputs; S = 1 would be the plain GEMM)
    const int64_t t2n = (N + BN2 - 1) / BN2;
    const int bmx = rt * 32;
    const int64_t t2m = (M + bmx - 1) / bmx;
    const size_t dyn = 2 * ((size_t)bmx * BK * 2 + TILE2_BYTES) + 4096;
    GemmEpi ep{}; ep.ld0 = M * N; ep.i0 = S;
#define KA_SK(R) true, false, R, false, 6
#define KA_SKM(R) true, false, R, true, 6
    if (b_kmajor) {
        EPI_SETATTR(KA_SKM);
        EPI_LAUNCH(KA_SKM, rt, dim3((unsigned)(S * t2m * t2n)), dim3(512), dyn, s, (const bf16_t*)A, (const bf16_t*)B, ws_f32, (const bf16_t*)nullptr,
                   (const bf16_t*)nullptr, M, N, K, lda, ldb, N, (int64_t)0, (int)t2m, (int)t2n, ep);
    } else {
        EPI_SETATTR(KA_SK);
        EPI_LAUNCH(KA_SK, rt, dim3((unsigned)(S * t2m * t2n)), dim3(512), dyn, s, (const bf16_t*)A, (const bf16_t*)B, ws_f32, (const bf16_t*)nullptr,
                   (const bf16_t*)nullptr, M, N, K, lda, ldb, N, (int64_t)0, (int)t2m, (int)t2n, ep);
    }
#undef KA_SK
#undef KA_SKM
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(tr1_grid_1d(M * N / 8, 256, 2048)), dim3(256), 0, s, (const float*)ws_f32, S, M * N,
                       (const bf16_t*)bias, (const bf16_t*)residual, ldr, (bf16_t*)C, ldc, M, N);
    TR1_LAUNCH_CHECK();
}

// ---- lm_head -> (log-prob of the target, entropy, LSE) without materialised logits --------------------------------------------------
// ref: _get_per_token_logps (src/time_r1/rl/timer1_trainer.py:449-481) computes logits [G, L, V], log_softmax, gather and entropy; here the
// GEMM epilogue reduces every 64-column slice of a row to (max, sum e, sum x e) and this kernel merges the V / 64 slices of a row.
__global__ __launch_bounds__(256) void lse_combine_kernel(const f32x4_t* __restrict__ part, int64_t ncb, float* __restrict__ logp, float* __restrict__ ent,
                                                          float* __restrict__ lse_out) {
    __shared__ float red[16];
    const int64_t row = blockIdx.x;
    const f32x4_t* pr = part + row * (ncb + 1);
    float mx = -INFINITY;
    for (int64_t i = threadIdx.x; i < ncb; i += 256) mx = fmaxf(mx, pr[i][0]);
    mx = block_max(mx, red);
    __syncthreads();
    float se = 0.f, te = 0.f;
    for (int64_t i = threadIdx.x; i < ncb; i += 256) {
        const f32x4_t v = pr[i];
        const float w = (v[0] == -INFINITY) ? 0.f : __expf(v[0] - mx);
        se += v[1] * w; te += v[2] * w;
    }
    se = block_sum(se, red);
    __syncthreads();
    te = block_sum(te, red);
    if (threadIdx.x == 0) {
        const float l = mx + logf(se);
        const float xt = reinterpret_cast<const float*>(pr + ncb)[0];
        lse_out[row] = l; logp[row] = xt - l; ent[row] = l - te / se;
    }
}

extern "C" int64_t tr1_lmhead_lse_workspace_floats(int64_t M, int64_t N) { return M * ((N + 63) / 64 + 1) * 4; }

// hn [M, K] bf16 (final-norm output of the prediction rows), W [N, K] bf16 (lm_head), targets int32 [M]  ->  logp, entropy, lse fp32 [M].
extern "C" int tr1_lmhead_lse_fwd(const void* hn, const void* W, const void* targets, void* part_ws, int64_t ws_floats, void* logp, void* ent,
                                  void* lse, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, void* stream) {
    TR1_CHECK_ARG(K % BK == 0 && N % 64 == 0 && lda % 8 == 0 && ldb % 8 == 0, "lmhead_lse: K % 64, N % 64, lda % 8, ldb % 8 required");
    TR1_CHECK_ARG(part_ws && ws_floats >= tr1_lmhead_lse_workspace_floats(M, N), "lmhead_lse: workspace too small");
    if (M == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int64_t ncb = N / 64;
    auto cost = [&](int64_t bm, double eff) {
        const int64_t t = ((M + bm - 1) / bm) * ((N + BN2 - 1) / BN2);
        return (double)((t + 255) / 256) * 256.0 * (double)(bm * BN2) / eff;
    };
    static const double eff[4] = {0.94, 1.0, 1.025, 1.03};
    int rt = 8; double best = cost(256, 1.0);
    for (int r = 7; r <= 10; ++r) { const double c = cost(r * 32, eff[r - 7]); if (c < best) { best = c; rt = r; } }
    const int bmx = rt * 32;
    const int64_t t2m = (M + bmx - 1) / bmx, t2n = (N + BN2 - 1) / BN2;
    const size_t dyn = 2 * ((size_t)bmx * BK * 2 + TILE2_BYTES) + 4096;
    static bool attr_set = false;
    if (!attr_set) {
        const int mx = (int)(2 * (320 * BK * 2 + TILE2_BYTES)) + 4096;
#define SETL(R) hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt8p_kernel<false, false, R, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, mx)
        SETL(7); SETL(8); SETL(9); SETL(10);
#undef SETL
        attr_set = true;
    }
    dim3 grid2((unsigned)(t2m * t2n));
#define LAUNCHL(R) hipLaunchKernelGGL((gemm_nt8p_kernel<false, false, R, false, 1>), grid2, dim3(512), dyn, s, (const bf16_t*)hn, (const bf16_t*)W, part_ws, \
                                      (const bf16_t*)targets, (const bf16_t*)nullptr, M, N, K, lda, ldb, ncb + 1, (int64_t)0, (int)t2m, (int)t2n, GemmEpi{})
    if (rt == 7) LAUNCHL(7); else if (rt == 9) LAUNCHL(9); else if (rt == 10) LAUNCHL(10); else LAUNCHL(8);
#undef LAUNCHL
    hipLaunchKernelGGL(lse_combine_kernel, dim3((unsigned)M), dim3(256), 0, s, (const f32x4_t*)part_ws, ncb, (float*)logp, (float*)ent, (float*)lse);
    TR1_LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------------------------
// LDS-streamed form of the split-K + fixup decode projection (down_proj at M <= 16: N = 3584 columns, K = 18944).
// Same decomposition, workspace and ticket protocol as gemm_skinny_kernel<4, 2, 4, 1> launched by tr1_gemm_skinny_fixup (a block owns 64
// output columns and one of `gridDim.y` K-slabs; the last-arriving slab sums the tiles in slab order), but the 64 x 64 weight stages and
// the 16 x 64 activation stages travel HBM/L2 -> LDS as full 128-byte row runs (global_load_lds), each wave has its own two-slot ring and
// takes the slab's 64-wide stages round-robin, so the only ordering in the stream is the issuing wave's counted vmcnt.
// ------------------------------------------------------------------------------------------------------------------
// NWI = weight DMA instructions (8 rows each) per stage: 8 = 64-column blocks; 7 = 56-column blocks (round 3): 3584 columns are then 64 groups, and
// 64 x 4 K-slabs fill all 256 CUs (56 x 4 = 224 left 32 of them idle).  The MFMAs still run on four 16-row weight tiles - rows 56..63 of a stage are
// never written and only feed the eight output columns that are not stored - so every stored value is the same sum in the same order as with NWI = 8.
#ifdef TR1_PROBE
__device__ unsigned long long* tr1_down_probe = nullptr;
extern "C" int probe_down_set_ptr(void* ptr) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(tr1_down_probe), &ptr, sizeof(ptr)); }
#define DOWN_STAMPS unsigned long long ds_[6] = {0, 0, 0, 0, 0, 0}
#define DOWN_STAMP(i) do { ds_[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define DOWN_DUMP() do { if (tr1_down_probe && threadIdx.x == 0) { \
    _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) tr1_down_probe[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + i_] = ds_[i_]; } } while (0)
#else
#define DOWN_STAMPS do { } while (0)
#define DOWN_STAMP(i) do { } while (0)
#define DOWN_DUMP() do { } while (0)
#endif
template <int WAVES, int MG = 1, int NWI = 8>
__global__ __launch_bounds__(WAVES * 64) void gemm_skinny_lds_fix_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W, bf16_t* __restrict__ C,
                                                                         const bf16_t* __restrict__ bias, const bf16_t* __restrict__ residual, int M,
                                                                         int64_t N, int64_t K, int64_t ldx, int64_t ldw, int64_t ldc, int64_t ldr,
                                                                         float* __restrict__ fix_ws, int* __restrict__ fix_cnt) {
    constexpr int NC = 4, XOFF = NWI * 1024, STAGE = XOFF + MG * 2048;      // NWI * 8 weight rows + 16*MG activation rows, 128 bytes each (56-column stages: the MFMA's
                                                                            // weight rows 56..63 read into the activation area - they feed output columns that are never stored)
    constexpr int TILE = NC * MG * 256;
    extern __shared__ __attribute__((aligned(16))) char sk_lds[];           // [WAVES][2][STAGE]; afterwards red[WAVES][NC][MG][16][17] f32; ticket at the end
    int* s_ticket = reinterpret_cast<int*>(sk_lds + WAVES * 2 * STAGE);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, u = lane & 15, g = lane >> 4;
    DOWN_STAMPS;
    DOWN_STAMP(0);
    constexpr int COLS = NWI * 8;                                           // output columns a block owns
    const int64_t n0 = (int64_t)blockIdx.x * COLS;
    const int64_t kslab = K / gridDim.y, k0 = (int64_t)blockIdx.y * kslab;
    const int nst = (int)(kslab / 64);
    const int n_my = wave < nst ? (nst - wave + WAVES - 1) / WAVES : 0;       // stages wave, wave + WAVES, ...
    char* ring = sk_lds + wave * 2 * STAGE;
    // DMA lane map (8 rows x 128 bytes per instruction): lane -> row 8j + (lane >> 3), physical chunk lane & 7, logical chunk ^ keyA(row)
    const bf16_t* pw[NWI]; const bf16_t* px[2 * MG];
#pragma unroll
    for (int j = 0; j < NWI; ++j) {
        const int r = 8 * j + (lane >> 3);
        int64_t row = n0 + r; if (row >= N) row = N - 1;
        pw[j] = W + row * ldw + k0 + (int64_t)wave * 64 + (((lane & 7) ^ keyA(r)) << 3);
    }
#pragma unroll
    for (int j = 0; j < 2 * MG; ++j) {
        const int r = 8 * j + (lane >> 3);
        px[j] = X + (int64_t)(r < M ? r : M - 1) * ldx + k0 + (int64_t)wave * 64 + (((lane & 7) ^ keyA(r)) << 3);
    }
#define SKL_ISSUE(SLOT) do {                                                                                              \
        char* dst__ = ring + (SLOT) * STAGE;                                                                              \
        _Pragma("unroll") for (int j = 0; j < NWI; ++j) { __builtin_amdgcn_global_load_lds((gptr_t)pw[j], (lptr_t)(dst__ + j * 1024), 16, 0, TR1_W_AUX); pw[j] += WAVES * 64; } \
        _Pragma("unroll") for (int j = 0; j < 2 * MG; ++j) { __builtin_amdgcn_global_load_lds((gptr_t)px[j], (lptr_t)(dst__ + XOFF + j * 1024), 16, 0, 0); px[j] += WAVES * 64; } \
    } while (0)
    f32x4_t acc[NC][MG][2];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) acc[c][mg][0] = acc[c][mg][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int rd_w = u * 128, kA = keyA(u);
#define SKL_CONSUME(SLOT) do {                                                                                            \
        const char* sb__ = ring + (SLOT) * STAGE;                                                                         \
        bf16x8_t xf__[MG][2], wf__[NC][2];                                                                                \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                                \
            const int off__ = ((ks * 4 + g) ^ kA) << 4;                                                                   \
            _Pragma("unroll") for (int mg = 0; mg < MG; ++mg) xf__[mg][ks] = *reinterpret_cast<const bf16x8_t*>(sb__ + XOFF + mg * 2048 + rd_w + off__); \
            _Pragma("unroll") for (int c = 0; c < NC; ++c) wf__[c][ks] = *reinterpret_cast<const bf16x8_t*>(sb__ + c * 2048 + rd_w + off__); \
        }                                                                                                                 \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                  \
        _Pragma("unroll") for (int c = 0; c < NC; ++c)                                                                    \
        _Pragma("unroll") for (int mg = 0; mg < MG; ++mg)                                                                 \
            acc[c][mg][ks] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf__[c][ks], xf__[mg][ks], acc[c][mg][ks], 0, 0, 0); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                \
    } while (0)
    if (n_my > 0) SKL_ISSUE(0);
    for (int i = 0; i < n_my; i += 2) {
        if (i + 1 < n_my) { SKL_ISSUE(1); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWI + 2 * MG) : "memory"); } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SKL_CONSUME(0);
        if (i + 1 < n_my) {
            if (i + 2 < n_my) { SKL_ISSUE(0); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWI + 2 * MG) : "memory"); } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            SKL_CONSUME(1);
        }
    }
#undef SKL_ISSUE
#undef SKL_CONSUME
    DOWN_STAMP(1);
    TR1_BARRIER();
    DOWN_STAMP(2);                                                          // every wave is done with its ring: the space becomes the reduction buffer
    float* red = reinterpret_cast<float*>(sk_lds);                          // [WAVES][NC][MG][16][17]
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int mg = 0; mg < MG; ++mg)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(((wave * NC + c) * MG + mg) * 16 + u) * 17 + g * 4 + r] = acc[c][mg][0][r] + acc[c][mg][1][r];
    __syncthreads();
    // ---- cross-block fixup (see gemm_skinny_kernel): park the tile with device-scope stores, draw a ticket, the last slab sums in slab order
    float* mine = fix_ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * TILE;
    for (int i = threadIdx.x; i < TILE; i += WAVES * 64) {
        const int c = i / (MG * 256), mg = (i >> 8) % MG, mm = (i >> 4) & 15, nn = i & 15;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) v += red[(((w * NC + c) * MG + mg) * 16 + mm) * 17 + nn];
        __hip_atomic_store(mine + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    DOWN_STAMP(3);
    if (threadIdx.x == 0) *s_ticket = __hip_atomic_fetch_add(&fix_cnt[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    DOWN_STAMP(4);
    if (*s_ticket != (int)gridDim.y - 1) { DOWN_DUMP(); return; }
    for (int i = threadIdx.x; i < TILE; i += WAVES * 64) {
        const int c = i / (MG * 256), mg = (i >> 8) % MG, mm = mg * 16 + ((i >> 4) & 15), nn = i & 15;
        const int64_t n = n0 + c * 16 + nn;
        float v = 0.f;
        if (gridDim.y == 4) {       // the usual case, unrolled: all four device-scope loads in flight together (as a loop each one waited for the one
            float t[4];             // before it: 8 300 cycles for the last-arriving block, block timeline in DESIGN.md); same sum in the same order
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                t[ks] = __hip_atomic_load(fix_ws + ((int64_t)ks * gridDim.x + blockIdx.x) * TILE + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = ((t[0] + t[1]) + t[2]) + t[3];
        } else if (gridDim.y <= 16) {       // up to 16 slabs: every load in flight before the first add, summed in slab order (what the loop below computes)
            float t[16];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
                t[ks] = ks < (int)gridDim.y ? __hip_atomic_load(fix_ws + ((int64_t)ks * gridDim.x + blockIdx.x) * TILE + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
            v = t[0];
#pragma unroll
            for (int ks = 1; ks < 16; ++ks) if (ks < (int)gridDim.y) v += t[ks];
        } else {
            for (int ks = 0; ks < (int)gridDim.y; ++ks)
                v += __hip_atomic_load(fix_ws + ((int64_t)ks * gridDim.x + blockIdx.x) * TILE + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (mm < M && n < N && c * 16 + nn < COLS) {
            if (bias) v += bf2f(bias[n]);
            if (residual) v += bf2f(residual[(int64_t)mm * ldr + n]);
            C[(int64_t)mm * ldc + n] = f2bf(v);
        }
    }
    if (threadIdx.x == 0) fix_cnt[blockIdx.x] = 0;
    DOWN_STAMP(5);
    DOWN_DUMP();
}

// ---- narrow-N decode projections (o_proj, down_proj): cross-block split-K with in-kernel fixup -----------------------------------
// N/16 column groups cannot fill 256 CUs and every block re-reads all of x from L2; 4 k-slabs x wider column groups (NCOL 2-4) cut
// the x traffic and put ~4x more blocks in flight.  Measured at M = 16 (tools/microbench.py splitk): down 3584x18944 38.5 -> 29 us.
static int skinny_fix_cfg(int64_t M, int64_t N, int64_t K, int* ncol, int* mg) {
    *mg = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
    *ncol = (K >= 8192 && *mg <= 2) ? 4 : 2;
    int ks = K >= 2048 ? 4 : 1;
    // Few column groups (a narrow output over a long K - the Qwen2-VL-2B down projection: 1536 / 64 = 24 groups x 4 slabs = 96 blocks on 256 CUs, 2.1 TB/s): as
    // many K slabs (<= 16, whole 64-k stages each) as still give at most one block per CU - 24 x 10 = 240 blocks there.  TR1_DOWN_KS=4 keeps the four slabs.
    static int ks_max = -1;
    if (ks_max < 0) { ks_max = 16; }
    const int64_t groups = (N + 16 * *ncol - 1) / (16 * *ncol);
    if (ks == 4 && *mg == 1 && groups * 4 < 192)
        for (int cand = 5; cand <= ks_max && cand <= 16; ++cand)
            if (K % ((int64_t)cand * 64) == 0 && groups * cand <= 256) ks = cand;
    return ks;
}

// 56-column blocks for the LDS-streamed <= 16-row form when that is what fills the chip: N % 56 == 0 and N/56 x ks <= 256 < more blocks than N/64 x ks
// (7B down projection: 64 x 4 = 256 blocks instead of 56 x 4 = 224).  TR1_DOWN_COLS=64 keeps the 64-column blocks (A/B measurements).
static bool skinny_fix_cols56(int64_t N, int ks, int ncol, int mg) {
    static int cols = -1;
    if (cols < 0) { cols = 56; }
    return cols == 56 && mg <= 2 && ncol == 4 && ks > 1 && N % 56 == 0 && (N / 56) * ks <= 256 && (N / 56) > (N + 63) / 64;
}

extern "C" int64_t tr1_gemm_skinny_fixup_workspace_floats(int64_t M, int64_t N, int64_t K) {
    // fp32 tiles [ksplit][column groups][NCOL*MG*256] followed by one int32 ticket counter per column group (zero-initialised ONCE by
    // the caller; the kernel re-arms them)
    int ncol, mg;
    const int ks = skinny_fix_cfg(M, N, K, &ncol, &mg);
    int64_t groups = (N + 16 * ncol - 1) / (16 * ncol);
    if (skinny_fix_cols56(N, ks, ncol, mg)) groups = N / 56;
    return ks * groups * ncol * mg * 256 + groups;
}

extern "C" int tr1_gemm_skinny_fixup(const void* A, const void* B, void* C, const void* bias, const void* residual, int64_t M, int64_t N, int64_t K,
                                     int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, void* ws_f32, int64_t ws_floats, void* stream) {
    TR1_CHECK_ARG(K % BK == 0 && K >= 256, "gemm_skinny_fixup: K must be a multiple of 64 and >= 256");
    TR1_CHECK_ARG(M >= 1 && M <= 64, "gemm_skinny_fixup: 1 <= M <= 64 (decode rows)");
    TR1_CHECK_ARG(N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && (!residual || ldr % 8 == 0), "gemm_skinny_fixup: N%8, ld%8 required");
    int ncol, mg;
    const int ks = skinny_fix_cfg(M, N, K, &ncol, &mg);
    static int down_lds = -1;                        // TR1_DOWN_LDS=0 selects the register-fragment form (A/B measurements); 6 / 7 = waves per block
    if (down_lds < 0) { down_lds = 7; }
    const bool c56 = down_lds && (K / ks) % 64 == 0 && skinny_fix_cols56(N, ks, ncol, mg);
    const int64_t groups = c56 ? N / 56 : (N + 16 * ncol - 1) / (16 * ncol);
    TR1_CHECK_ARG(ws_f32 && ws_floats >= ks * groups * ncol * mg * 256 + groups, "gemm_skinny_fixup: workspace too small");
    float* tiles = (float*)ws_f32;
    int* cnt = (int*)(tiles + ks * groups * ncol * mg * 256);
    hipStream_t s = (hipStream_t)stream;
#define SKF(WV, UN, NC, MGR)                                                                                                         \
    hipLaunchKernelGGL((gemm_skinny_kernel<WV, UN, NC, MGR>), dim3((unsigned)groups, (unsigned)ks), dim3(WV * 64), 0, s,            \
                       (const bf16_t*)A, (const bf16_t*)B, (bf16_t*)C, (float*)nullptr, (const bf16_t*)bias, (const bf16_t*)residual, \
                       (int)M, N, K, lda, ldb, ldc, ldr, tiles, ks > 1 ? cnt : (int*)nullptr)
    if (c56 && mg == 2) {       // round 5, 17 .. 32 rows: 56-column blocks as well (256 blocks), whose 11 KiB stages leave room for a seventh wave
        static bool attr562 = false;
        constexpr int DYN = 7 * 2 * (7 * 1024 + 2 * 2048) + 16;
        if (!attr562) { hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_lds_fix_kernel<7, 2, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, DYN); attr562 = true; }
        hipLaunchKernelGGL((gemm_skinny_lds_fix_kernel<7, 2, 7>), dim3((unsigned)groups, (unsigned)ks), dim3(448), DYN, s, (const bf16_t*)A, (const bf16_t*)B,
                           (bf16_t*)C, (const bf16_t*)bias, (const bf16_t*)residual, (int)M, N, K, lda, ldb, ldc, ldr, tiles, cnt);
    }
    else if (mg == 2 && ncol == 4 && ks > 1 && down_lds && (K / ks) % 64 == 0 && N % 64 == 0) {       // 17 .. 32 rows: 12 KiB stages, 6 waves
        static bool attr2 = false;
        if (!attr2) { hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_lds_fix_kernel<6, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 2 * 12288 + 16); attr2 = true; }
        hipLaunchKernelGGL((gemm_skinny_lds_fix_kernel<6, 2>), dim3((unsigned)groups, (unsigned)ks), dim3(384), 6 * 2 * 12288 + 16, s, (const bf16_t*)A, (const bf16_t*)B,
                           (bf16_t*)C, (const bf16_t*)bias, (const bf16_t*)residual, (int)M, N, K, lda, ldb, ldc, ldr, tiles, cnt);
    }
    else if (c56) {
        static bool attr56 = false;
        constexpr int DYN = 7 * 2 * (7 * 1024 + 2048) + 16;
        if (!attr56) { hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_lds_fix_kernel<7, 1, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, DYN); attr56 = true; }
        hipLaunchKernelGGL((gemm_skinny_lds_fix_kernel<7, 1, 7>), dim3((unsigned)groups, (unsigned)ks), dim3(448), DYN, s, (const bf16_t*)A, (const bf16_t*)B,
                           (bf16_t*)C, (const bf16_t*)bias, (const bf16_t*)residual, (int)M, N, K, lda, ldb, ldc, ldr, tiles, cnt);
    }
    else if (mg == 1 && ncol == 4 && ks > 1 && down_lds && (K / ks) % 64 == 0 && N % 64 == 0) {
        static bool attr_set = false;
        if (!attr_set) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_lds_fix_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 2 * 10240 + 16);
            hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_lds_fix_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize, 7 * 2 * 10240 + 16);
            attr_set = true;
        }
        if (down_lds == 6)
            hipLaunchKernelGGL((gemm_skinny_lds_fix_kernel<6>), dim3((unsigned)groups, (unsigned)ks), dim3(384), 6 * 2 * 10240 + 16, s, (const bf16_t*)A, (const bf16_t*)B,
                               (bf16_t*)C, (const bf16_t*)bias, (const bf16_t*)residual, (int)M, N, K, lda, ldb, ldc, ldr, tiles, cnt);
        else
            hipLaunchKernelGGL((gemm_skinny_lds_fix_kernel<7>), dim3((unsigned)groups, (unsigned)ks), dim3(448), 7 * 2 * 10240 + 16, s, (const bf16_t*)A, (const bf16_t*)B,
                               (bf16_t*)C, (const bf16_t*)bias, (const bf16_t*)residual, (int)M, N, K, lda, ldb, ldc, ldr, tiles, cnt);
    }
    else if (mg == 1) { if (ncol == 4) SKF(4, 2, 4, 1); else SKF(4, 4, 2, 1); }
    else if (mg == 2) { if (ncol == 4) SKF(4, 2, 4, 2); else SKF(4, 2, 2, 2); }
    else SKF(4, 2, 2, 4);
#undef SKF
    TR1_LAUNCH_CHECK();
}
