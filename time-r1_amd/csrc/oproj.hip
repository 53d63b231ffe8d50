// Decode step, round 5: the attention output projection at <= 32 rows with the WHOLE weight slice of a block in flight at once and no cross-block fixup.
// reference: o_proj of Qwen2VLAttention.forward (TF:553-556) inside model.generate (src/time_r1/rl/timer1_trainer.py:568-578).
//
// gemm_skinny_kernel<4, 4, 1, 1> (the form this replaces) reads its weights in the MFMA operand layout - 16 rows x 64 bytes per wave load = 64 tag look-ups
// per KiB on the L1 tag pipe (DESIGN "what actually bounds the small decode GEMMs") - 9.6 us for 25.7 MB.  Here a block owns `cols` (<= 16) output columns
// over the whole K:
//   * its 8 waves request ALL K/64 weight stages (16 rows x 128 B each, 100-112 KB at K = 3584) HBM -> LDS by DMA at kernel entry (full 128-byte row runs:
//     8 look-ups per KiB, non-temporal), so the stream is one memory round trip deep, not a ring of them;
//   * the activation comes in FRAGMENT-MAJOR layout - [k / 32][16 rows][32 k], written that way by the split-KV merge kernel (attn_combine128_kernel,
//     AttnParams::o_frag): the 16 rows x 64 bytes of an MFMA B fragment are then ONE contiguous KiB (8 look-ups instead of 64) and go straight to registers;
//   * cols is the smallest divisor of N that still gives at most one block per CU (7B: 14 -> 256 blocks; 2B: 6 -> 256), the waves' partial tiles meet in
//     LDS in wave order, the residual is added, done: no ticket, no partial tiles in HBM.
#include "tr1_common.h"

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define OPJ_WAVES 8
#define OPJ_MAXS 7        // weight stages per wave of the o-projection forms: K <= 8 * 7 * 64 = 3584
#define OPJ_MAXS_LONG 18  // round 6, the long-K form (<= 8 columns per block, 1 KiB stages, 16 rows): K <= 8 * 18 * 64 = 9216 (Qwen2-VL-2B down projection: 8960)

namespace {
TR1_DEV int opj_key(int row) { return (row >> 1) & 7; }      // chunk swizzle of a 128-byte stage row (keyA of gemm.hip)
}

// MAXS = weight stages per wave; NJ = 8-row DMA instructions (KiB) per stage: 2 -> up to 16 columns per block; 1 -> up to 8 columns (the MFMA's weight rows 8..15 then
// read the following KiB - the next stage or the reduction area - and only feed output columns that are never stored, cf. the 56-column down projection)
template <int MG, int MAXS = OPJ_MAXS, int NJ = 2>
__global__ __launch_bounds__(OPJ_WAVES * 64) void oproj_frag_kernel(const bf16_t* __restrict__ Xf, const bf16_t* __restrict__ W, const bf16_t* __restrict__ residual,
                                                                    bf16_t* __restrict__ C, int M, int64_t N, int64_t K, int64_t ldw, int64_t ldr, int64_t ldc, int cols) {
    constexpr int SB = NJ * 1024;                                    // bytes per stage
    extern __shared__ __attribute__((aligned(16))) char opj_lds[];   // [K/64][8 NJ rows x 128 B] | red [8][MG][16][17] f32
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), u = lane & 15, g = lane >> 4;
    const int nst = (int)(K >> 6);
    float* red = reinterpret_cast<float*>(opj_lds + (size_t)nst * SB);
    const int64_t n0 = (int64_t)blockIdx.x * cols;
    // ---- weights: every stage of this wave in flight
    {
        const int r0 = lane >> 3;
        const int nj = (NJ == 2 && cols > 8) ? 2 : 1;
#pragma unroll
        for (int i = 0; i < MAXS; ++i) {
            const int s = wave + i * OPJ_WAVES;
            if (s < nst) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if (j < nj) {
                        const int r = 8 * j + r0;
                        int64_t row = n0 + (r < cols ? r : cols - 1); if (row >= N) row = N - 1;
                        const bf16_t* src = W + row * ldw + (int64_t)s * 64 + (((lane & 7) ^ opj_key(r)) << 3);
                        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(opj_lds + s * SB + j * 1024), 16, 0, 2);
                    }
            }
        }
    }
    // ---- x fragments of this wave's stages (fragment-major: one contiguous KiB per wave load)
    const int64_t kq = K >> 5;
    u32x4_t xf[MAXS][MG][2];
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
        const int s = wave + i * OPJ_WAVES;
#pragma unroll
        for (int mg = 0; mg < MG; ++mg)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                xf[i][mg][ks] = *reinterpret_cast<const u32x4_t*>(Xf + ((mg * kq + (int64_t)(s < nst ? 2 * s + ks : 0)) * 16) * 32 + u * 32 + g * 8);      // row u, 16-byte piece g of the 64-byte chunk
    }
    // the residual of this thread's output element, requested under the stream
    const int oi = (int)threadIdx.x;
    const int omg = oi >> 8, omm = (oi >> 4) & 15, onn = oi & 15, om = omg * 16 + omm;
    const int64_t on = n0 + onn;
    const bool oact = oi < MG * 256 && om < M && onn < cols && on < N;
    const float rv = (oact && residual) ? bf2f(residual[(int64_t)om * ldr + on]) : 0.f;
    f32x4_t acc[MG][2];
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) acc[mg][0] = acc[mg][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int kA = opj_key(u);
    // this wave's weight stages have landed (the compiler does not order the LDS reads below behind the DMA that fills them: without the explicit wait a
    // cold first launch read its stages before they arrived)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
        const int s = wave + i * OPJ_WAVES;
        if (s < nst) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8_t wf = *reinterpret_cast<const bf16x8_t*>(opj_lds + s * SB + u * 128 + (((ks * 4 + g) ^ kA) << 4));
#pragma unroll
                for (int mg = 0; mg < MG; ++mg)
                    acc[mg][ks] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8_t, xf[i][mg][ks]), acc[mg][ks], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int mg = 0; mg < MG; ++mg)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * MG + mg) * 16 + u) * 17 + g * 4 + r] = acc[mg][0][r] + acc[mg][1][r];
    __syncthreads();
    if (oact) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < OPJ_WAVES; ++w) v += red[((w * MG + omg) * 16 + omm) * 17 + onn];
        if (residual) v += rv;
        C[(int64_t)om * ldc + on] = f2bf(v);
    }
}

// per-device caches (a process normally drives ONE GPU, but tools / tests may touch more: the CU count and the raised dynamic-LDS limit belong to a device)
static int opj_dev() { int dev = 0; return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16) ? dev : 0; }
static int opj_cols(int64_t N) {
    static int cus_of[16] = {0};
    const int dev = opj_dev();
    int cus = cus_of[dev];
    if (!cus) {
        hipDeviceProp_t pr;
        cus = (hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256;
        if (cus <= 0) cus = 256;
        cus_of[dev] = cus;
    }
    for (int c = (int)((N + cus - 1) / cus); c <= 16; ++c)
        if (c >= 1 && N % c == 0) return c;
    return 0;
}

// the long-K form (round 6): <= 8 columns per block in 1 KiB stages over the WHOLE K, 16 rows - K / 64 KiB of weights + the reduction area must fit one CU's LDS
static bool opj_long_ok(int64_t M, int64_t N, int64_t K) {
    const int cols = opj_cols(N);
    return M <= 16 && cols >= 1 && cols <= 8 && K <= (int64_t)OPJ_WAVES * OPJ_MAXS_LONG * 64 &&
           (size_t)(K / 64) * 1024 + (size_t)OPJ_WAVES * 16 * 17 * 4 + 16 <= (size_t)(160 * 1024 - 512);
}

// 1 when tr1_gemm_oproj_frag covers the shape: the decode driver then asks the split-KV merge (o projection) / the gate-up epilogue (down projection) for the
// fragment-major activation (tr1_attn_fwd_planned_frag, tr1_norm_gemm_skinny glu = 2)
extern "C" int tr1_gemm_oproj_frag_ok(int64_t M, int64_t N, int64_t K) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("TR1_O_FRAG"); on = e ? atoi(e) : 1; }
    if (!(on && M >= 1 && M <= 32 && K % 64 == 0 && K >= 512 && N % 2 == 0 && opj_cols(N) > 0)) return 0;
    return K <= OPJ_WAVES * OPJ_MAXS * 64 || opj_long_ok(M, N, K);
}

// C[M, N] = X @ W[N, K]^T (+ residual), M <= 32 decode rows, X in fragment-major layout: element (m, k) at ((m / 16) * (K / 32) + k / 32) * 512 + (m % 16) * 32 + k % 32
// (bf16; ceil(M / 16) * 16 * K elements; rows of a 16-group beyond M may hold anything).
extern "C" int tr1_gemm_oproj_frag(const void* Xfrag, const void* W, const void* residual, void* C, int64_t M, int64_t N, int64_t K, int64_t ldw, int64_t ldr,
                                   int64_t ldc, void* stream) {
    TR1_CHECK_ARG(tr1_gemm_oproj_frag_ok(M, N, K), "gemm_oproj_frag: shape not covered (tr1_gemm_oproj_frag_ok)");
    TR1_CHECK_ARG(Xfrag && W && C && ldw % 8 == 0, "gemm_oproj_frag: null argument / ldw % 8");
    const int cols = opj_cols(N), mg = M <= 16 ? 1 : 2;
    const bool lng = K > OPJ_WAVES * OPJ_MAXS * 64;
    const size_t dyn = (size_t)(K / 64) * (lng ? 1024 : 2048) + (size_t)OPJ_WAVES * mg * 16 * 17 * 4 + 16;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(N / cols));
#define OPJ_LAUNCH(KERNEL, SLOT)                                                                                                                      \
    do {                                                                                                                                              \
        static bool attr_[3][16] = {{false}};                                                                                                         \
        if (!attr_[SLOT][opj_dev()]) { hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512); attr_[SLOT][opj_dev()] = true; } \
        hipLaunchKernelGGL((KERNEL), grid, dim3(OPJ_WAVES * 64), dyn, s, (const bf16_t*)Xfrag, (const bf16_t*)W, (const bf16_t*)residual, (bf16_t*)C, (int)M, N, K, \
                           ldw, ldr, ldc, cols);                                                                                                      \
    } while (0)
    if (lng) OPJ_LAUNCH((oproj_frag_kernel<1, OPJ_MAXS_LONG, 1>), 2);
    else if (mg == 1) OPJ_LAUNCH((oproj_frag_kernel<1, OPJ_MAXS, 2>), 0);
    else OPJ_LAUNCH((oproj_frag_kernel<2, OPJ_MAXS, 2>), 1);
#undef OPJ_LAUNCH
    TR1_LAUNCH_CHECK();
}
