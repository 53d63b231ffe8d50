// Decode step, round 5: the merge of the split-KV attention partials FUSED into the attention output projection (one launch instead of
// attn_combine_kernel + o_proj).  reference: Qwen2VLAttention.forward, softmax(QK^T)V -> o_proj (TF:521-556) inside model.generate
// (src/time_r1/rl/timer1_trainer.py:568-578).
//
// Why one launch: both kernels are latency chains (merge: 6.4 us for 7 MB of L2-resident partials; o_proj: 9.6 us for 25.7 MB of weights, of which
// ~4 us are a dependent ramp before the first weight byte lands).  Here a block of the projection owns 16 output columns over the WHOLE K and
//   1. its WS streaming waves request ALL of the block's weight rows (K/64 stages of 16 rows x 128 B, 112 KB at K = 3584) HBM -> LDS at kernel entry
//      - the weights do not depend on the attention;
//   2. meanwhile its extra wave merges TWO packed rows of the attention partials (one of the n_batch * n_kv * T * group / 2 tasks, spread over the grid)
//      with attn_combine_kernel's operations in attn_combine_kernel's order (att_merge_stats, the four split-group fma chains: BIT-IDENTICAL rows),
//      publishes them write-through (sc1 stores), drains, and adds its task count to an arrival counter;
//   3. every block waits until all tasks have arrived (one lane polls), then reads the merged rows [M, K] straight into MFMA operand registers with
//      sc1 loads (L1-bypassing: the rows were written by other CUs during this launch), multiplies against the LDS-resident weights, reduces the
//      WS waves' partial tiles through LDS in wave order and adds the residual.
// The all-to-all edge (~3 us) runs UNDER the weight stream (~4.5 us), and nothing is split across blocks, so there is no ticket fixup at the end.
// Placement-independent protocol (MI355X guide, Guideline 16 R1): sc1 payload stores -> every storing wave drains vmcnt -> one agent-scope counter
// add; consumer: relaxed agent-scope poll -> block barrier -> sc1 loads.  The grid must be co-resident (launcher: blocks <= CUs); the wait is bounded (10 ms) and raises a flag the host reads with tr1_grid_sync_error().
#include "attn_common.h"

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

namespace {
TR1_DEV int okey(int row) { return (row >> 1) & 7; }      // chunk swizzle of a 128-byte stage row (the keyA of gemm.hip)

struct OprojFused {
    const float* Opart; const float* mpart; const float* lpart;      // split-KV partials [nsplit][n_batch * n_kv][nRpad][128], [..][nRpad] x 2
    int T, group, n_kv, n_batch, nsplit; unsigned group_magic; int64_t nRpad;
    bf16_t* O; int64_t o_ld;                                          // merged attention output [n_batch * T, n_heads * 128]
    const bf16_t* W; int64_t ldw;                                     // [N, K = n_heads * 128]
    const bf16_t* residual; int64_t ldr;
    bf16_t* C; int64_t ldc;
    int M; int64_t N, K;
    int* flags;                                                       // one word per merge task: == epoch once the task's rows are published (zeroed once)
    int* err;
    int n_tasks, epoch;                                               // epoch: process-wide launch counter (never 0, never repeated)
};

int* g_sync_err = nullptr;                                            // library-owned error word (device)
}  // namespace

#ifdef TR1_PROBE
__device__ unsigned long long* tr1_opf_probe = nullptr;          // [blocks][2 waves: 0 = streaming wave 0, 1 = merge wave][8 stamps]  (tools/bench_oproj_fused.py PROBE=1)
extern "C" int probe_opf_set_ptr(void* ptr) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(tr1_opf_probe), &ptr, sizeof(ptr)); }
#define OPF_STAMPS unsigned long long ost_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define OPF_STAMP(i) do { ost_[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define OPF_FLUSH(WSV) do { if (tr1_opf_probe && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == (WSV))) { \
    _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) tr1_opf_probe[((size_t)blockIdx.x * 2 + ((threadIdx.x >> 6) ? 1 : 0)) * 8 + s_] = ost_[s_]; } } while (0)
#else
#define OPF_STAMPS do { } while (0)
#define OPF_STAMP(i) do { } while (0)
#define OPF_FLUSH(WSV) do { } while (0)
#endif
#define OPF_MAXS 8        // weight stages per streaming wave (K <= WS * 8 * 64)
#define OPF_NSP 32        // split partials a merge lane keeps in flight (nsplit <= 32)

template <int WS, int MG>
__global__ __launch_bounds__((WS + 1) * 64) void oproj_combine_kernel(OprojFused p) {
    extern __shared__ __attribute__((aligned(16))) char opf_lds[];   // [K/64][16 rows x 128 B] | red [WS][MG][16][17] f32 | sm_m, sm_l [2][64]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), u = lane & 15, g = lane >> 4;
    const int nst = (int)(p.K >> 6);
    char* wl = opf_lds;
    float* red = reinterpret_cast<float*>(opf_lds + (size_t)nst * 2048);
    float* sm_m = red + WS * MG * 16 * 17;
    float* sm_l = sm_m + 128;
    const int64_t n0 = (int64_t)blockIdx.x * 16;
    OPF_STAMPS;
    OPF_STAMP(0);
    // Order of the block's memory requests: the merge wave's partial loads go FIRST (the CU's vector-memory queue is a FIFO: behind the 112 KB weight
    // burst they returned after 10 us), the barrier below releases the streaming waves' DMA behind them
    if (wave == WS) {
        // ---- 1. merge tasks: two packed rows per task, lanes 0..31 row 0, 32..63 row 1, four features per lane
        const int nR = p.T * p.group, tpg = (nR + 1) >> 1, nby = p.n_kv * p.n_batch;
        const int rr = lane >> 5, c = lane & 31;
        bool first = true;
        for (int task = blockIdx.x; task < p.n_tasks; task += gridDim.x) {
            const int by = task / tpg, rp = task - by * tpg;
            const int R = 2 * rp + rr;
            const bool act = R < nR;
            const int Rc = act ? R : nR - 1;
            f32x4_t ov[OPF_NSP];
#pragma unroll
            for (int sp = 0; sp < OPF_NSP; ++sp)
                ov[sp] = *reinterpret_cast<const f32x4_t*>(p.Opart + (((int64_t)(sp < p.nsplit ? sp : 0) * nby + by) * p.nRpad + Rc) * 128 + c * 4);
            const int spc = c < p.nsplit ? c : 0;
            const int64_t slot = ((int64_t)spc * nby + by) * p.nRpad + Rc;
            const float mv = p.mpart[slot], lv = p.lpart[slot];
            sm_m[rr * 64 + c] = (act && c < p.nsplit) ? mv : NEG_INF;
            sm_l[rr * 64 + c] = (act && c < p.nsplit) ? lv : 0.f;
#ifdef OPF_RELEASE_EARLY
            if (first) { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); first = false; }
#else
            // the partial rows have LANDED (the statistics were requested last and loads return in order): only now release the block's weight DMA - every CU
            // of the grid does the same, so the merge inputs are not queued behind 25 MB of weight requests in the memory system either
            if (first) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); first = false; }
#endif
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float Mx, Ms, L;
            att_merge_stats(sm_m + rr * 64, sm_l + rr * 64, p.nsplit, Mx, Ms, L);
            const float inv = L > 0.f ? 1.f / L : 0.f;
            f32x4_t part[4];
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) {                          // attn_combine_kernel's four split groups, each an fma chain over splits sg, sg + 4, ...
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < OPF_NSP / 4; ++i) {
                    const int sp = sg + 4 * i;
                    if (sp < p.nsplit) {
                        const float w = exp2f(sm_m[rr * 64 + sp] - Ms);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(w, ov[sp][j], acc[j]);
                    }
                }
                part[sg] = acc;
            }
            const f32x4_t acc = (part[0] + part[1]) + (part[2] + part[3]);
            OPF_STAMP(1);
            __builtin_amdgcn_wave_barrier();                          // (the statistics rows are rewritten by the next task)
            if (act) {
                const unsigned tu = p.group == 1 ? (unsigned)R : __umulhi((unsigned)R, p.group_magic);
                const int t = (int)tu, hq = R - t * p.group;
                const int b = by / p.n_kv, kvh = by - b * p.n_kv;
                bf16_t* orow = p.O + ((int64_t)b * p.T + t) * p.o_ld + (int64_t)(kvh * p.group + hq) * 128 + c * 4;
                const unsigned long long w = (unsigned long long)pack2bf(acc[0] * inv, acc[1] * inv) | ((unsigned long long)pack2bf(acc[2] * inv, acc[3] * inv) << 32);
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(orow), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // 8-byte sc1 (write-through) store
            }
            OPF_STAMP(2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's merged rows have reached the device-coherent level ...
            if (lane == 0) __hip_atomic_store(p.flags + task, p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ... before the task's flag can be seen
        }
        if (first) { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }      // (a block without a task)
        OPF_STAMP(3);
        // every task of the launch published?  One sweep = the flag words spread over the lanes (no shared counter: 224 arrivals + 224 pollers on ONE word cost 5 us)
        const unsigned long long t0 = wall_clock64();
        for (;;) {
            bool ok = true;
            for (int j = lane; j < p.n_tasks; j += 64) ok &= __hip_atomic_load(p.flags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p.epoch;
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > 1000000ull) { if (lane == 0) __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }      // 10 ms of the 100 MHz clock
        }
    } else {
        asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
        // ---- 2. the block's weight rows, all stages in flight (non-temporal: read once per decode step)
        const int r0 = lane >> 3;
#pragma unroll
        for (int i = 0; i < OPF_MAXS; ++i) {
            const int s = wave + i * WS;
            if (s < nst) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int r = 8 * j + r0;
                    int64_t row = n0 + r; if (row >= p.N) row = p.N - 1;
                    const bf16_t* src = p.W + row * p.ldw + (int64_t)s * 64 + (((lane & 7) ^ okey(r)) << 3);
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(wl + s * 2048 + j * 1024), 16, 0, 2);
                }
            }
        }
        OPF_STAMP(1);
    }
    OPF_STAMP(4);
    __syncthreads();                                                  // every merged row of the launch is visible at the device-coherent level
    OPF_STAMP(5);
    if (wave < WS) {
        // ---- 3. x fragments of this wave's stages: row u (+16 mg), 16-byte chunks (ks * 4 + g) of the stage's 64 k - sc1 loads, all in flight
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.O, 0, (int)(((int64_t)(p.M - 1) * p.o_ld + p.K) * 2), 0x00020000);
        u32x4_t xf[OPF_MAXS][MG][2];
#pragma unroll
        for (int i = 0; i < OPF_MAXS; ++i) {
            const int s = wave + i * WS;
#pragma unroll
            for (int mg = 0; mg < MG; ++mg) {
                const int row = mg * 16 + u < p.M ? mg * 16 + u : p.M - 1;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int off = (int)(((int64_t)row * p.o_ld + (int64_t)(s < nst ? s : 0) * 64 + (ks * 4 + g) * 8) * 2);
                    xf[i][mg][ks] = __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 16);
                }
            }
        }
        OPF_STAMP(6);
        f32x4_t acc[MG][2];
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) acc[mg][0] = acc[mg][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const int kA = okey(u);
#pragma unroll
        for (int i = 0; i < OPF_MAXS; ++i) {
            const int s = wave + i * WS;
            if (s < nst) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8_t wf = *reinterpret_cast<const bf16x8_t*>(wl + s * 2048 + u * 128 + (((ks * 4 + g) ^ kA) << 4));
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
    }
    // (MG * 256 <= 512 outputs: one per thread; the residual is requested before the barrier so that its latency hides behind the waves' MFMAs)
    const int oi = (int)threadIdx.x;
    const int omg = oi >> 8, omm = (oi >> 4) & 15, onn = oi & 15, om = omg * 16 + omm;
    const int64_t on = n0 + onn;
    const bool oact = oi < MG * 256 && om < p.M && on < p.N;
    const float rv = (oact && p.residual) ? bf2f(p.residual[(int64_t)om * p.ldr + on]) : 0.f;
    __syncthreads();
    if (oact) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WS; ++w) v += red[((w * MG + omg) * 16 + omm) * 17 + onn];
        if (p.residual) v += rv;
        p.C[(int64_t)om * p.ldc + on] = f2bf(v);
    }
    OPF_STAMP(7);
    OPF_FLUSH(WS);
}

static int* grid_sync_err_word() {
    if (!g_sync_err) {
        if (hipMalloc(reinterpret_cast<void**>(&g_sync_err), 64) != hipSuccess) { g_sync_err = nullptr; return nullptr; }
        hipMemset(g_sync_err, 0, 64);
    }
    return g_sync_err;
}

// -> 1 if a fused launch gave up waiting for its partner blocks since the last call (its output is then wrong: the caller must raise), and clears the
// flag; 0 otherwise.  Synchronous (4-byte device-to-host copy): call it where the host already waits for the rollout's tokens.
extern "C" int tr1_grid_sync_error(void) {
    int* w = grid_sync_err_word();
    if (!w) return 0;
    int v = 0;
    if (hipMemcpy(&v, w, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (v) hipMemset(w, 0, 64);
    return v;
}

// 1 when tr1_attn_combine_oproj covers the shape (the decode driver then launches the attention without its merge kernel), else 0
extern "C" int tr1_attn_combine_oproj_ok(int64_t M, int64_t n_heads, int64_t n_kv, int64_t head_dim, int64_t nsplit, int64_t N) {
    static int on = -1, cus = 0;
    if (on < 0) {
        const char* e = getenv("TR1_O_FUSED"); on = e ? atoi(e) : 1;
        int dev = 0; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount;
    }
    const int64_t K = n_heads * head_dim;
    // every block resident at once (blocks <= CUs): K * 32 B of weight stages per block, N / 16 blocks
    return on && head_dim == 128 && n_kv > 0 && n_heads % n_kv == 0 && nsplit >= 2 && nsplit <= OPF_NSP && M >= 1 && M <= 32 && K <= 7 * OPF_MAXS * 64 &&
           K % 64 == 0 && N % 16 == 0 && N / 16 <= cus;
}

// Split-KV partials (workspace of tr1_attn_fwd_partials: same layout as tr1_attn_fwd's) -> O [n_batch * T, n_heads * 128] (merged attention rows, also an
// output) and C[M, N] = O @ W[N, K]^T (+ residual), M = n_batch * T.  sync_i32: 1024 ints (one flag per merge task), zeroed ONCE by the caller.
extern "C" int tr1_attn_combine_oproj(const void* ws_f32, int64_t ws_floats, int64_t T, int64_t n_heads, int64_t n_kv, int64_t head_dim, int64_t nsplit,
                                      int64_t n_batch, void* O, int64_t o_ld, const void* W, int64_t ldw, const void* residual, int64_t ldr, void* C,
                                      int64_t ldc, int64_t N, void* sync_i32, void* stream) {
    const int64_t M = n_batch * T;
    TR1_CHECK_ARG(tr1_attn_combine_oproj_ok(M, n_heads, n_kv, head_dim, nsplit, N), "attn_combine_oproj: shape not covered (tr1_attn_combine_oproj_ok)");
    TR1_CHECK_ARG(ws_f32 && O && W && C && sync_i32, "attn_combine_oproj: null argument");
    TR1_CHECK_ARG(o_ld % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && (!residual || ldr % 8 == 0), "attn_combine_oproj: leading dims must be multiples of 8");
    OprojFused p; memset(&p, 0, sizeof(p));
    AttnParams ap; memset(&ap, 0, sizeof(ap));
    const int group = (int)(n_heads / n_kv);
    TR1_CHECK_ARG(att_set_group(ap, T * n_batch, group), "attn_combine_oproj: T * group^2 must stay below 2^32");
    const int64_t nR = T * group, nRpad = (nR + 63) / 64 * 64, nby = n_kv * n_batch;
    TR1_CHECK_ARG(ws_floats >= nsplit * nby * nRpad * (128 + 2), "attn_combine_oproj: partials workspace too small");
    p.Opart = (const float*)ws_f32; p.mpart = p.Opart + nsplit * nby * nRpad * 128; p.lpart = p.mpart + nsplit * nby * nRpad;
    p.T = (int)T; p.group = group; p.n_kv = (int)n_kv; p.n_batch = (int)n_batch; p.nsplit = (int)nsplit; p.group_magic = ap.group_magic; p.nRpad = nRpad;
    p.O = (bf16_t*)O; p.o_ld = o_ld; p.W = (const bf16_t*)W; p.ldw = ldw; p.residual = (const bf16_t*)residual; p.ldr = ldr; p.C = (bf16_t*)C; p.ldc = ldc;
    p.M = (int)M; p.N = N; p.K = n_heads * head_dim;
    p.flags = (int*)sync_i32; p.err = grid_sync_err_word();
    TR1_CHECK_ARG(p.err, "attn_combine_oproj: could not allocate the error word");
    p.n_tasks = (int)(nby * ((nR + 1) / 2));
    TR1_CHECK_ARG(p.n_tasks <= 1024, "attn_combine_oproj: more than 1024 merge tasks");
    static int g_epoch = 0;
    if (++g_epoch <= 0) g_epoch = 1;
    p.epoch = g_epoch;
    const int mg = M <= 16 ? 1 : 2;
    const size_t dyn = (size_t)(p.K / 64) * 2048 + (size_t)7 * mg * 16 * 17 * 4 + 1024 + 16;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(N / 16));
    if (mg == 1) {
        static bool a1 = false;
        if (!a1) { hipFuncSetAttribute(reinterpret_cast<const void*>(&oproj_combine_kernel<7, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512); a1 = true; }
        hipLaunchKernelGGL((oproj_combine_kernel<7, 1>), grid, dim3(512), dyn, s, p);
    } else {
        static bool a2 = false;
        if (!a2) { hipFuncSetAttribute(reinterpret_cast<const void*>(&oproj_combine_kernel<7, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512); a2 = true; }
        hipLaunchKernelGGL((oproj_combine_kernel<7, 2>), grid, dim3(512), dyn, s, p);
    }
    TR1_LAUNCH_CHECK();
}
