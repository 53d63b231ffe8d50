// What could a persistent decode engine buy on MI355X?  Upper-bound probe: the six phases of a 7B decode layer reduced to what a
// launch-per-op chain and a persistent kernel differ in - (1) a dependent read of the previous phase's output, (2) the phase's weight bytes
// streamed HBM -> LDS by per-wave DMA rings (the structure of norm_glu_lds_kernel), (3) a cross-wave reduction and a small output - with
//   A  one kernel launch per phase (back to back on one stream),
//   B  one persistent kernel, hierarchical grid barrier between phases,
//   C  as B, and the first ring stages of the NEXT phase's weights are requested before the barrier.
// No MFMA work, no fixups, no attention arithmetic: everything a real engine adds makes B / C slower, not faster.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_persistent_layer.hip -o /tmp/ppl && /tmp/ppl [2b]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
#define NPH 6
#define NLAYER 28
#define WAVES 8
#define RING 3
#define STAGE 4096                       // bytes per wave stage: 4 DMA instructions of 1 KiB
#define SPIN_MAX (1 << 22)

struct Phases { const char* w[NPH]; long long stages[NPH]; };   // stages = 4 KiB wave-stages of the whole phase (all blocks)

__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// wave-level stream of this wave's stages [s0, s0 + n) of a phase through its LDS ring; `pre` stages were already requested (prefetch)
__device__ __forceinline__ float stream_phase(const char* w, long long s0, int n, int pre, char* ring, int lane) {
    float acc = 0.f;
    int issued = pre;
#define ISSUE(i) do { const char* src__ = w + (s0 + (i)) * STAGE + lane * 16; char* dst__ = ring + ((i) % RING) * STAGE;                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds((gptr_t)(src__ + j * 1024), (lptr_t)(dst__ + j * 1024), 16, 0, 2); } while (0)
    for (; issued < RING - 1 && issued < n; ++issued) ISSUE(issued);
    for (int i = 0; i < n; ++i) {
        if (issued < n) { ISSUE(issued); ++issued; }
        const int ahead = issued - i - 1;               // stages requested after stage i
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += *reinterpret_cast<const float*>(ring + (i % RING) * STAGE + lane * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#undef ISSUE
    return acc;
}

__device__ __forceinline__ void phase_body(const Phases& ph, int p, int pre, float* act_in, float* act_out, char* lds, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nb = gridDim.x, b = blockIdx.x;
    // (1) dependent read of the previous phase's output (another block's values, device scope)
    const float dep = __hip_atomic_load(act_in + (size_t)((b + 1) % nb) * 512 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (2) this wave's share of the phase
    const long long total = ph.stages[p], per_wave = total / ((long long)nb * WAVES);
    const long long s0 = ((long long)b * WAVES + wave) * per_wave;
    float acc = stream_phase(ph.w[p], s0, (int)per_wave, pre, lds + wave * RING * STAGE, lane);
    // (3) cross-wave reduction + small output
    red[threadIdx.x] = acc + dep;
    __syncthreads();
    float v = 0.f;
    for (int w = 0; w < WAVES; ++w) v += red[w * 64 + lane];
    __hip_atomic_store(act_out + (size_t)b * 512 + threadIdx.x, v * 1e-30f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
}

__global__ __launch_bounds__(512) void phase_kernel(Phases ph, int p, float* act_in, float* act_out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ float red[512];
    phase_body(ph, p, 0, act_in, act_out, lds, red);
}

template <bool PREFETCH>
__global__ __launch_bounds__(512) void layer_kernel(const Phases* layers, int n_layers, float* act0, float* act1, int* cnt, int* flag, int* fail) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ float red[512];
    const int nb = gridDim.x, b = blockIdx.x, g = b & 7, per = nb / 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int e = 0, pre = 0;
    for (int L = 0; L < n_layers; ++L) {
        const Phases ph = layers[L];
        for (int p = 0; p < NPH; ++p, ++e) {
            phase_body(ph, p, pre, (e & 1) ? act1 : act0, (e & 1) ? act0 : act1, lds, red);
            pre = 0;
            if (PREFETCH) {                 // first RING-1 stages of the next phase, requested before the hand-off
                const Phases* nx = (p + 1 < NPH) ? &layers[L] : (L + 1 < n_layers ? &layers[L + 1] : nullptr);
                if (nx) {
                    const int np = (p + 1) % NPH;
                    const long long per_wave = nx->stages[np] / ((long long)nb * WAVES);
                    const long long s0 = ((long long)b * WAVES + wave) * per_wave;
                    char* ring = lds + wave * RING * STAGE;
                    for (int i = 0; i < RING - 1 && i < per_wave; ++i) {
                        const char* src = nx->w[np] + (s0 + i) * STAGE + lane * 16;
                        for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds((gptr_t)(src + j * 1024), (lptr_t)(ring + i * STAGE + j * 1024), 16, 0, 2);
                        ++pre;
                    }
                }
            }
            // hierarchical grid barrier (8 group counters -> top counter -> 8 release flags), bounded spin
            if (threadIdx.x == 0) {
                const int old = __hip_atomic_fetch_add(&cnt[16 * g], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == per * (e + 1) - 1) {
                    const int old2 = __hip_atomic_fetch_add(&cnt[16 * 8], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (old2 == 8 * (e + 1) - 1)
                        for (int q = 0; q < 8; ++q) __hip_atomic_store(&flag[16 * q], e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                int it = 0;
                while (ld_agent(&flag[16 * g]) < e + 1 && ++it < SPIN_MAX) __builtin_amdgcn_s_sleep(1);
                if (it >= SPIN_MAX) atomicAdd(fail, 1);
            }
            __syncthreads();
        }
    }
}

int main(int argc, char** argv) {
    // bytes per phase of a decode layer at 16 rows: qkv, attention KV, split-KV partials, o, gate/up, down.  Default: Qwen2-VL-7B (config 3);
    // `2b`: Qwen2-VL-2B at config 2 (hidden 1536, 12 / 2 heads of 128, intermediate 8960, 16 frames: P ~ 2 520 prompt tokens, 20 splits)
    const double mb7[NPH] = {33.0, 17.5, 6.4, 25.7, 271.6, 135.8}, mb2[NPH] = {6.3, 5.2, 2.0, 4.7, 55.0, 27.5};
    const bool two_b = argc > 1 && argv[1][0] == '2';
    const double* mb = two_b ? mb2 : mb7;
    printf("model: %s\n", two_b ? "Qwen2-VL-2B (config 2)" : "Qwen2-VL-7B (config 3)");
    const int nb = 256;
    std::vector<Phases> h(NLAYER);
    for (int L = 0; L < NLAYER; ++L)
        for (int p = 0; p < NPH; ++p) {
            long long per_wave = (long long)(mb[p] * 1e6 / STAGE / (nb * WAVES));
            if (per_wave < 1) per_wave = 1;
            h[L].stages[p] = per_wave * nb * WAVES;
            void* w; (void)hipMalloc(&w, (size_t)h[L].stages[p] * STAGE); (void)hipMemset(w, 0, (size_t)h[L].stages[p] * STAGE);
            h[L].w[p] = (const char*)w;
        }
    Phases* d; (void)hipMalloc(&d, sizeof(Phases) * NLAYER); (void)hipMemcpy(d, h.data(), sizeof(Phases) * NLAYER, hipMemcpyHostToDevice);
    float *act0, *act1; (void)hipMalloc(&act0, nb * 512 * 4); (void)hipMalloc(&act1, nb * 512 * 4); (void)hipMemset(act0, 0, nb * 512 * 4); (void)hipMemset(act1, 0, nb * 512 * 4);
    int *cnt, *flag, *fail; (void)hipMalloc(&cnt, 4096); (void)hipMalloc(&flag, 4096); (void)hipMalloc(&fail, 4);
    const size_t dyn = WAVES * RING * STAGE;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&phase_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&layer_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&layer_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int reps = 5;
    double bytes = 0; for (int p = 0; p < NPH; ++p) bytes += (double)h[0].stages[p] * STAGE;
    auto launches = [&]() {
        int e = 0;
        for (int L = 0; L < NLAYER; ++L)
            for (int p = 0; p < NPH; ++p, ++e)
                hipLaunchKernelGGL(phase_kernel, dim3(nb), dim3(512), dyn, 0, h[L], p, (e & 1) ? act1 : act0, (e & 1) ? act0 : act1);
    };
    launches(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0); for (int r = 0; r < reps; ++r) launches(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double usA = ms * 1000.0 / reps / NLAYER;
    printf("A  launch per phase                         %7.2f us per layer  (%5.0f GB/s of weights)\n", usA, bytes / usA / 1e3);
    {   // per-phase cost in the launch-per-phase chain: the same phase of all layers back to back
        const char* names[NPH] = {"qkv", "attention KV", "split-KV partials", "o", "gate/up", "down"};
        for (int p = 0; p < NPH; ++p) {
            (void)hipEventRecord(e0, 0);
            for (int r = 0; r < reps; ++r)
                for (int L = 0; L < NLAYER; ++L)
                    hipLaunchKernelGGL(phase_kernel, dim3(nb), dim3(512), dyn, 0, h[L], p, (L & 1) ? act1 : act0, (L & 1) ? act0 : act1);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float m; (void)hipEventElapsedTime(&m, e0, e1);
            const double us = m * 1000.0 / reps / NLAYER, by = (double)h[0].stages[p] * STAGE;
            printf("   phase %-18s %6.1f MB  %6.2f us  (%5.0f GB/s)\n", names[p], by / 1e6, us, by / us / 1e3);
        }
    }
    auto persistent = [&](const char* name, void (*k)(const Phases*, int, float*, float*, int*, int*, int*)) {
        double us = 0;
        for (int r = 0; r < reps + 1; ++r) {
            (void)hipMemset(cnt, 0, 4096); (void)hipMemset(flag, 0, 4096); (void)hipMemset(fail, 0, 4);
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(nb), dim3(512), dyn, 0, d, NLAYER, act0, act1, cnt, flag, fail);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float m; (void)hipEventElapsedTime(&m, e0, e1);
            if (r > 0) us += m * 1000.0 / NLAYER;
        }
        us /= reps;
        int f; (void)hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
        printf("%-43s %7.2f us per layer  (%5.0f GB/s of weights)  = %.3f x A   (spin give-ups: %d)\n", name, us, bytes / us / 1e3, us / usA, f);
    };
    persistent("B  persistent, grid barrier between phases", layer_kernel<false>);
    persistent("C  persistent + next phase prefetched", layer_kernel<true>);
    return 0;
}
