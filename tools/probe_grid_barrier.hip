// How expensive is an in-kernel grid barrier on MI355X (256 blocks x 512 threads, one per CU) compared with a kernel boundary?
//   hipcc --offload-arch=gfx950 -O2 tools/probe_grid_barrier.hip -o /tmp/pgb && /tmp/pgb
// Variants: flat (one counter, one flag), hierarchical (8 group counters -> top counter -> 8 flags), each with and without the
// __threadfence() pair that makes plain stores of the previous phase visible.  Spins are bounded (no hang on a lost block).
#include <hip/hip_runtime.h>
#include <stdio.h>

#define SPIN_MAX (1 << 22)
__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <bool HIER, bool FENCE>
__global__ __launch_bounds__(512) void bar_kernel(int* cnt, int* flag, float* data, int iters, int* fail) {
    const int nb = gridDim.x, b = blockIdx.x, g = b & 7, per = nb / 8;
    float acc = 0.f;
    for (int e = 0; e < iters; ++e) {
        // "phase": every thread writes one float, then (after the barrier) reads a neighbour block's value
        data[(size_t)b * 512 + threadIdx.x] = (float)(e + b);
        if (FENCE) __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            if (HIER) {
                const int old = __hip_atomic_fetch_add(&cnt[16 * g], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == per * (e + 1) - 1) {
                    const int old2 = __hip_atomic_fetch_add(&cnt[16 * 8], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (old2 == 8 * (e + 1) - 1)
                        for (int q = 0; q < 8; ++q) __hip_atomic_store(&flag[16 * q], e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                int it = 0;
                while (ld_agent(&flag[16 * g]) < e + 1 && ++it < SPIN_MAX) __builtin_amdgcn_s_sleep(1);
                if (it >= SPIN_MAX) atomicAdd(fail, 1);
            } else {
                const int old = __hip_atomic_fetch_add(&cnt[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == nb * (e + 1) - 1) __hip_atomic_store(&flag[0], e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int it = 0;
                while (ld_agent(&flag[0]) < e + 1 && ++it < SPIN_MAX) __builtin_amdgcn_s_sleep(1);
                if (it >= SPIN_MAX) atomicAdd(fail, 1);
            }
        }
        __syncthreads();
        if (FENCE) __threadfence();
        acc += data[(size_t)((b + 1) % nb) * 512 + threadIdx.x];
    }
    if (acc == -1.f) data[0] = acc;
}

__global__ __launch_bounds__(512) void phase_kernel(float* data, int e) {
    const int nb = gridDim.x, b = blockIdx.x;
    const float v = data[(size_t)((b + 1) % nb) * 512 + threadIdx.x];
    data[(size_t)b * 512 + threadIdx.x] = v * 0.f + (float)(e + b);
}

int main() {
    int *cnt, *flag, *fail; float* data;
    (void)hipMalloc(&cnt, 4096); (void)hipMalloc(&flag, 4096); (void)hipMalloc(&fail, 4); (void)hipMalloc(&data, 256 * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    auto run = [&](const char* name, void (*k)(int*, int*, float*, int, int*)) {
        (void)hipMemset(cnt, 0, 4096); (void)hipMemset(flag, 0, 4096); (void)hipMemset(fail, 0, 4);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, cnt, flag, data, 10, fail);
        (void)hipDeviceSynchronize();
        (void)hipMemset(cnt, 0, 4096); (void)hipMemset(flag, 0, 4096);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, cnt, flag, data, iters, fail);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        int f; (void)hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
        printf("%-44s %.2f us per phase+barrier  (spin give-ups: %d)\n", name, ms * 1000.f / iters, f);
    };
    run("flat barrier, no fence", bar_kernel<false, false>);
    run("flat barrier, __threadfence pair", bar_kernel<false, true>);
    run("hierarchical barrier, no fence", bar_kernel<true, false>);
    run("hierarchical barrier, __threadfence pair", bar_kernel<true, true>);
    (void)hipEventRecord(e0, 0);
    for (int e = 0; e < iters; ++e) hipLaunchKernelGGL(phase_kernel, dim3(256), dim3(512), 0, 0, data, e);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.2f us per launch\n", "the same phase as back-to-back launches", ms * 1000.f / iters);
    return 0;
}
