// LDS read throughput per CU by instruction type, with the conflict-free address patterns of the attention kernels (round 3).
// One block per CU (256 blocks), W waves per block; every wave issues `iters` x 16 reads of one type over a 32 KiB swizzled tile image and
// XORs the results (so nothing is dropped); cycles from s_memtime around the loop of wave 0.
//   type 0: ds_read_b128, lane (row = lane & 31, half h = lane >> 5) reads chunk (2 ks + h) ^ skey(row) of a 256-byte row   (K as the A operand)
//   type 1: ds_read_b64_tr_b16 with the t_lane pattern of attn_fwd32 / gemm_tn                                                 (V^T from V rows)
//   type 2: ds_read_b64, same addresses as type 1
//   hipcc --offload-arch=gfx950 -O3 tools/probe_lds_rate.hip -o tools/_probe_lds.bin && tools/_probe_lds.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ int skey(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <int TYPE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(int iters, unsigned* sink, long long* cyc) {
    extern __shared__ __attribute__((aligned(256))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768 / 4; i += WAVES * 64) reinterpret_cast<unsigned*>(lds)[i] = i * 2654435761u;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
    const int c32 = lane & 31, h = lane >> 5, ti = lane & 15, tgrp = (lane >> 4) & 1;
    const unsigned a_lane = (unsigned)(c32 * 256 + ((h ^ skey(c32 & 15)) << 4));
    const unsigned t_lane = (unsigned)((4 * h + (ti >> 2)) * 256 + (ti & 1) * 8 + (((tgrp * 2 + ((ti & 3) >> 1)) ^ (((ti >> 2) << 2) | h)) << 4));
    u32x4 acc = {0, 0, 0, 0};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const unsigned off = (unsigned)((it + wave) & 1) * 16384u;
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            if (TYPE == 0) {
                const unsigned a = base + off + (a_lane ^ ((n & 7) * 32)) + (n >> 3) * 8192;
                const u32x4 v = *(const __attribute__((address_space(3))) u32x4*)(uintptr_t)a;
                acc ^= v;
            } else if (TYPE == 1) {
                const unsigned a = base + off + (t_lane ^ ((n & 3) * 64)) + (n >> 2) * 4096;
                const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)a);
                const u32x2 w = __builtin_bit_cast(u32x2, v);
                acc[0] ^= w[0]; acc[1] ^= w[1];
            } else {
                const unsigned a = base + off + (t_lane ^ ((n & 3) * 64)) + (n >> 2) * 4096;
                const u32x2 w = *(const __attribute__((address_space(3))) u32x2*)(uintptr_t)a;
                acc[0] ^= w[0]; acc[1] ^= w[1];
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int TYPE, int WAVES>
void run(const char* name, int bytes_per_lane) {
    unsigned* sink; long long* cyc;
    hipMalloc(&sink, 4); hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<TYPE, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    k<TYPE, WAVES><<<256, WAVES * 64, 65536>>>(100, sink, cyc);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<TYPE, WAVES><<<256, WAVES * 64, 65536>>>(iters, sink, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double bytes = (double)iters * 16 * 64 * bytes_per_lane * WAVES;      // per CU
    printf("%-22s %d waves/CU: %7.1f B per shader clock per CU (s_memtime-domain cycles %lld), %6.1f GB/s per CU, %.3f ms\n", name, WAVES, bytes / (double)c, c,
           bytes / (ms * 1e-3) / 1e9, ms);
    hipFree(sink); hipFree(cyc);
}

int main() {
    run<0, 8>("ds_read_b128 rows", 16); run<0, 4>("ds_read_b128 rows", 16);
    run<1, 8>("ds_read_b64_tr_b16", 8); run<1, 4>("ds_read_b64_tr_b16", 8);
    run<2, 8>("ds_read_b64 (same addr)", 8); run<2, 4>("ds_read_b64 (same addr)", 8);
    return 0;
}
