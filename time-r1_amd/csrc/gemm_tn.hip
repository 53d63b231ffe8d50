// Weight-gradient GEMM with BOTH operands read as stored (round 3):   C[N, K] (fp32) (+)= dY[T, N]^T * X[T, K]
//
// ref: the autograd of every nn.Linear of the policy (torch: grad_weight = grad_output^T @ input; loss.backward() at
// src/time_r1/rl/timer1_trainer.py:709 through transformers' Qwen2-VL decoder layers).  The contraction runs over the TOKEN index, which is the slow
// (row) index of both operands as the backward pass holds them.  Rounds 1-2 built dY^T (and, except for the down projection, X^T) with a
// transpose kernel per operand and ran the NT GEMM on the copies: 1.2 GB of extra HBM traffic per decoder layer, 25 ms of the 154 ms backward
// at config 3 (measured by skipping the copies).  Here the 32-token x 256-column operand tiles are streamed row-major into LDS by DMA
// (global_load_lds, swizzled on the source address) and the MFMA fragments come out of LDS TRANSPOSED with ds_read_b64_tr_b16: a 16-lane group
// reads a 4-token x 16-column block and every lane receives one column's four tokens.  Both operands are read with the same address pattern,
// so the order in which the 16 tokens of a k-step land in the 16 k-slots of v_mfma_f32_32x32x16_bf16 is the same on both sides and drops out
// of the sum.  Tile images are unpadded 256-byte rows with the skey() XOR swizzle of the attention kernels (attn_common.h): conflict-free for
// the DMA writes and for the transposing reads.
//
// Block tile 256 (N) x 256 (K) x 32 tokens per stage, 4-stage LDS ring (128 KB).  NBT = 2: 8 waves, wave tile 128 x 64 (two waves per SIMD cover
// each other's LDS latency); NBT = 4: 4 waves, wave tile 128 x 128 (256 accumulator registers, 2/3 of the LDS traffic per MFMA).
// The last, partial token tile re-reads row T-1 for the missing rows and zeroes them in the dY image before use.
#include "attn_common.h"
#include <stdlib.h>

#define TN_STAGE 32768          // [dY img0 | dY img1 | X img0 | X img1], each 32 rows x 256 B
#define TN_NB 4

template <int NBT>
__global__ __launch_bounds__(NBT == 2 ? 512 : 256) void gemm_tn32_kernel(const bf16_t* __restrict__ P, const bf16_t* __restrict__ Q, float* __restrict__ C,
                                                                         int T, unsigned ldp_b, unsigned ldq_b, int64_t ldc, int tiles_m, int tiles_n,
                                                                         int accumulate) {
    constexpr int WAVES = NBT == 2 ? 8 : 4, PER = 32 / WAVES;       // DMA wave-instructions (1 KiB = 4 rows of one image) per wave and stage
    extern __shared__ __attribute__((aligned(256))) char dyn_lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int h = lane >> 5;
    constexpr int WN = 8 / NBT;                                      // waves along K
    const int wm = wave / WN, wn = wave % WN;
    // block -> tile: consecutive workgroups go to the 8 XCDs round-robin; each XCD walks a contiguous range of the tile list, which itself runs
    // down groups of 4 N-tiles x all K-tiles, so the blocks resident on one XCD share their dY / X column tiles in that XCD's L2
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

    const unsigned lds_base = (unsigned)(uintptr_t)(att_lptr_t)dyn_lds;
    const char* pbase = reinterpret_cast<const char*>(P) + (int64_t)tm * 512;
    const char* qbase = reinterpret_cast<const char*>(Q) + (int64_t)tn * 512;
    const int n_tiles = (T + 31) >> 5;
#define DMA16(voff, sbase, m0v) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff), "s"(sbase) : "memory", "m0")
    // waves 0 .. WAVES/2-1 stage the dY images, the others the X images (wave-uniform: one operand base / leading dimension per wave)
    const bool is_x = wave >= WAVES / 2;
    const char* const obase = is_x ? qbase : pbase;
    const unsigned old_b = is_x ? ldq_b : ldp_b;
    const int q0 = wave * PER;                                        // 0..31: operand (q >> 4), image (q >> 3) & 1, row group q & 7
    auto issue_tile = [&](int tile, int slot) {
        const unsigned buf = lds_base + slot * TN_STAGE;
        int ln = lane;
        asm volatile("" : "+v"(ln));                                  // keep the lane constants out of loop-invariant hoisting (they would spill)
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int q = q0 + j;
            const unsigned row = 4u * (unsigned)(q & 7) + ((unsigned)ln >> 4);
            unsigned t = (unsigned)tile * 32u + row; t = t < (unsigned)T ? t : (unsigned)T - 1u;
            const unsigned ch = (unsigned)(((ln & 15) ^ skey(row & 15)) << 4) + (unsigned)((q >> 3) & 1) * 256u;
            DMA16(t * old_b + ch, obase, buf + q * 1024);
        }
    };
#undef DMA16
#pragma unroll
    for (int j = 0; j < TN_NB - 1; ++j)
        if (j < n_tiles) issue_tile(j, j);

    f32x16_t acc[4][NBT];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NBT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

#define LDS_TR16(addr) __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(uintptr_t)(addr)))
    // transposing read: lane (ti, tgrp, h) supplies token row 4h + (ti >> 2) of a 16-token chunk, 8 bytes at column 16 tgrp + 4 (ti & 3) of a 32-column
    // tile, and receives column 16 tgrp + ti; the second read of a fragment takes rows +8 (the swizzle key's low bits become h ^ 2: address ^ 32)
    const int ti = lane & 15, tgrp = (lane >> 4) & 1;
    const unsigned t_lane = (unsigned)((4 * h + (ti >> 2)) * 256 + (ti & 1) * 8 + (((tgrp * 2 + ((ti & 3) >> 1)) ^ (((ti >> 2) << 2) | h)) << 4));
    // per-lane image offsets of the two reads of every fragment (16 lane constants; + stage base, + 4096 for the second 16-token step)
    unsigned pa0[4], pa1[4], pb0[NBT], pb1[NBT];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        pa0[a] = ((unsigned)wm * 8192u + t_lane) ^ (unsigned)(a * 64);                      // dY image wm: the wave's 128 N rows = its 4 column tiles
        pa1[a] = (pa0[a] ^ 32u) + 2048u;
    }
#pragma unroll
    for (int b = 0; b < NBT; ++b) {
        const int ct = wn * NBT + b;                                                         // 32-column tile of the block's 256 X columns
        pb0[b] = 16384u + (unsigned)(ct >> 2) * 8192u + (t_lane ^ (unsigned)((ct & 3) * 64));
        pb1[b] = (pb0[b] ^ 32u) + 2048u;
    }
    bf16x8_t af[2][4], bf[2][NBT];
#define LOAD_A(set, base) do { _Pragma("unroll") for (int a = 0; a < 4; ++a) af[set][a] = make_frag(LDS_TR16((base) + pa0[a]), LDS_TR16((base) + pa1[a])); } while (0)
#define LOAD_B(set, base) do { _Pragma("unroll") for (int b = 0; b < NBT; ++b) bf[set][b] = make_frag(LDS_TR16((base) + pb0[b]), LDS_TR16((base) + pb1[b])); } while (0)
#define MFMA_ROWS(set, A0, A1)                                                                           \
    do {                                                                                                 \
        _Pragma("unroll") for (int a = (A0); a < (A1); ++a)                                              \
        _Pragma("unroll") for (int b = 0; b < NBT; ++b)                                                  \
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[set][a], bf[set][b], acc[a][b], 0, 0, 0); \
    } while (0)
#define SB() __builtin_amdgcn_sched_barrier(0)
    // tile `next` has landed for this wave's own DMA (at most `later` younger stages may still be in flight)
#define WAIT_STAGE(later)                                                                                \
    do {                                                                                                 \
        if (PER == 4) { if ((later) >= 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } \
        else { if ((later) >= 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }           \
    } while (0)
    auto zero_tail = [&](unsigned sbz) {                               // partial last tile: zero the dY rows past T (the X rows then do not matter)
        const int rem = T & 31;
        typedef __attribute__((address_space(3))) u32x4_t* lds_w128_t;
        for (int i = threadIdx.x; i < (32 - rem) * 32; i += WAVES * 64) {
            const int img = i / ((32 - rem) * 16), j = i - img * ((32 - rem) * 16);
            *(lds_w128_t)(uintptr_t)(sbz + (unsigned)img * 8192u + (unsigned)rem * 256u + (unsigned)j * 16u) = (u32x4_t){0u, 0u, 0u, 0u};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // The barrier that publishes tile it+1 sits inside the second 16-token step of tile it: the first fragments of the next tile are
    // requested before that step's last three MFMA rows, so a wave that has its SIMD to itself (NBT = 4) never waits for LDS at a tile boundary.
    //   slot reuse: the DMA issued after that barrier overwrites tile it-1, whose last fragment reads were consumed by MFMAs every wave issued
    //   before arriving.  Lead: a stage is requested two tiles before the barrier that needs it.
    if (n_tiles >= 3) { if (PER == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }   // tiles 1, 2 in flight
    else WAIT_STAGE(n_tiles - 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (n_tiles == 1 && (T & 31)) zero_tail(lds_base);
    // Order inside a 16-token step: first MFMA row (everything still pending on the LDS queue is exactly what it needs, so the compiler's wait is
    // exact even where it cannot count across the loop edge), then ALL fragment reads of the next step, then the other three MFMA rows, which
    // cover the latency of those reads.
    LOAD_A(0, lds_base); LOAD_B(0, lds_base);
    for (int it = 0; it < n_tiles; ++it) {
        const unsigned sb = lds_base + (unsigned)(it % TN_NB) * TN_STAGE;
        MFMA_ROWS(0, 0, 1); SB();
        LOAD_A(1, sb + 4096u); LOAD_B(1, sb + 4096u); SB();
        MFMA_ROWS(0, 1, 4); SB();
        MFMA_ROWS(1, 0, 1); SB();
        // (no MFMA inside a conditional: two code paths through the accumulators would cost hipcc 256 register copies at the join)
        const unsigned sn = lds_base + (unsigned)((it + 1 < n_tiles ? it + 1 : it) % TN_NB) * TN_STAGE;   // after the last tile: harmless re-reads
        if (it + 1 < n_tiles) {
            WAIT_STAGE(it + 2 < n_tiles ? 1 : 0);
            __builtin_amdgcn_s_barrier();                             // tile it+1 has landed for everybody; nobody reads tile it-1 any more
            asm volatile("" ::: "memory");
            if (it + TN_NB - 1 < n_tiles) issue_tile(it + TN_NB - 1, (it + TN_NB - 1) % TN_NB);
            if (it + 2 == n_tiles && (T & 31)) zero_tail(sn);
        }
        SB();
        LOAD_A(0, sn); LOAD_B(0, sn); SB();
        MFMA_ROWS(1, 1, 4); SB();
    }
#undef LOAD_A
#undef LOAD_B
#undef MFMA_ROWS
#undef SB
#undef WAIT_STAGE
#undef LDS_TR16
    // ---- epilogue: each wave transposes its tile through its own LDS slice (the ring is free) so that a lane moves 16 bytes of ONE row of C:
    //      32 rows x (NBT*32 columns) per pass, row stride NBT*128 + 32 bytes (conflict-free for the b32 writes in the accumulator layout: the
    //      h = 1 half lands 4 rows = 128 bytes (mod 256) further), read back as b128 with 8*NBT lanes per row.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                      // every wave is past its last fragment read
    asm volatile("" ::: "memory");
    {
        constexpr int RS = NBT * 128 + 32, LPR = NBT * 8, RPI = 64 / LPR;       // lanes per row, rows per wave-instruction
        int tid2 = threadIdx.x;
        asm volatile("" : "+v"(tid2));
        const int e_c = tid2 & 31, e_h = (tid2 >> 5) & 1, e_l = tid2 & 63;
        const unsigned wbase = lds_base + (unsigned)wave * (unsigned)(32 * RS);
        const unsigned w_addr = wbase + (unsigned)(4 * e_h) * RS + (unsigned)e_c * 4u;
        const unsigned r_addr = wbase + (unsigned)(e_l / LPR) * RS + (unsigned)(e_l % LPR) * 16u;
        float* cb = C + ((int64_t)tm * 256 + wm * 128 + e_l / LPR) * ldc + (int64_t)tn * 256 + wn * (NBT * 32) + (e_l % LPR) * 4;
        typedef __attribute__((address_space(3))) float* lds_f32_t;
        typedef __attribute__((address_space(3))) f32x4_t* lds_f128_t;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
            for (int b = 0; b < NBT; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    *(lds_f32_t)(uintptr_t)(w_addr + (unsigned)((r & 3) + 8 * (r >> 2)) * RS + (unsigned)b * 128u) = acc[a][b][r];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int q = 0; q < 32 / RPI; ++q) {
                f32x4_t v = *(lds_f128_t)(uintptr_t)(r_addr + (unsigned)(q * RPI) * RS);
                float* ptr = cb + (int64_t)(a * 32 + q * RPI) * ldc;
                if (accumulate) {
                    const f32x4_t o = *reinterpret_cast<const f32x4_t*>(ptr);
                    v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
                }
                *reinterpret_cast<f32x4_t*>(ptr) = v;
            }
            asm volatile("" ::: "memory");
        }
    }
}

// C[N, K] fp32 (+)= dY[T, N]^T X[T, K]; dY / X row-major bf16 with leading dimensions ldp / ldq (elements).  N and K multiples of 256; every
// operand byte offset must fit 32 bits (T * ld * 2 < 4 GiB).  accumulate = 0 overwrites C.
extern "C" int tr1_gemm_tn_acc_f32(const void* dY, const void* X, void* C, int64_t T, int64_t N, int64_t K, int64_t ldp, int64_t ldq, int64_t ldc,
                                   int accumulate, void* stream) {
    TR1_CHECK_ARG(T >= 1 && N >= 256 && K >= 256 && N % 256 == 0 && K % 256 == 0, "gemm_tn_acc_f32: N and K must be positive multiples of 256");
    TR1_CHECK_ARG(ldp % 8 == 0 && ldq % 8 == 0 && ldp >= N && ldq >= K && ldc >= K, "gemm_tn_acc_f32: leading dimensions (multiples of 8, >= the row width)");
    TR1_CHECK_ARG(T * ldp * 2 < (int64_t)0xffffffffLL && T * ldq * 2 < (int64_t)0xffffffffLL, "gemm_tn_acc_f32: operand larger than 4 GiB");
    TR1_CHECK_ARG((((uintptr_t)dY | (uintptr_t)X) & 15) == 0 && (((uintptr_t)C) & 3) == 0, "gemm_tn_acc_f32: operands must be 16-byte aligned");
    const int tiles_m = (int)(N / 256), tiles_n = (int)(K / 256);
    static int nbt = -1;                                              // TR1_TN_NBT=2: the 8-wave form (A/B measurements)
    if (nbt < 0) { const char* e = getenv("TR1_TN_NBT"); nbt = e ? atoi(e) : 4; }
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn32_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, TN_NB * TN_STAGE);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn32_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, TN_NB * TN_STAGE);
        attr_set = true;
    }
    const unsigned grid = (unsigned)(tiles_m * tiles_n);
    if (nbt == 2)
        hipLaunchKernelGGL((gemm_tn32_kernel<2>), dim3(grid), dim3(512), TN_NB * TN_STAGE, (hipStream_t)stream, (const bf16_t*)dY, (const bf16_t*)X, (float*)C,
                           (int)T, (unsigned)(ldp * 2), (unsigned)(ldq * 2), ldc, tiles_m, tiles_n, accumulate);
    else
        hipLaunchKernelGGL((gemm_tn32_kernel<4>), dim3(grid), dim3(256), TN_NB * TN_STAGE, (hipStream_t)stream, (const bf16_t*)dY, (const bf16_t*)X, (float*)C,
                           (int)T, (unsigned)(ldp * 2), (unsigned)(ldq * 2), ldc, tiles_m, tiles_n, accumulate);
    TR1_LAUNCH_CHECK();
}
