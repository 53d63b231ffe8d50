// Does the decode GEMM's weight access pattern (16 rows x 64 B per wave instruction, dictated by the MFMA A-fragment layout) cost bandwidth
// against a row-contiguous stream (1 row x 1 KB per instruction, what an LDS-staged layout would issue)?  Pure load kernels, same bytes, same
// loads in flight, no MFMA.   hipcc --offload-arch=gfx950 -O3 tools/probe_stream.hip -o /tmp/pb/probe_stream && /tmp/pb/probe_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

template <int PATTERN, int UNROLL>
__global__ __launch_bounds__(256) void stream_kernel(const unsigned short* __restrict__ W, unsigned* __restrict__ out, long K, long ldw) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long n0 = (long)blockIdx.x * 32;                   // 32 rows per block, the 4 waves split K
    const long kq = K / 4, k_begin = wave * kq;
    u32x4_t acc = {0, 0, 0, 0};
    if (PATTERN == 0) {                                      // fragment pattern: lane (u, g) -> row u / 16 + u, 16 B at k + g*8 and k + 32 + g*8
        const int u = lane & 15, g = lane >> 4;
        const unsigned short* p0 = W + (n0 + u) * ldw + k_begin + g * 8;
        const unsigned short* p1 = W + (n0 + 16 + u) * ldw + k_begin + g * 8;
        for (long k = 0; k < kq; k += 64 * UNROLL) {
            u32x4_t v[UNROLL][4];
#pragma unroll
            for (int q = 0; q < UNROLL; ++q) {
                v[q][0] = *reinterpret_cast<const u32x4_t*>(p0 + k + q * 64); v[q][1] = *reinterpret_cast<const u32x4_t*>(p0 + k + q * 64 + 32);
                v[q][2] = *reinterpret_cast<const u32x4_t*>(p1 + k + q * 64); v[q][3] = *reinterpret_cast<const u32x4_t*>(p1 + k + q * 64 + 32);
            }
#pragma unroll
            for (int q = 0; q < UNROLL; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc ^= v[q][j];
        }
    } else if (PATTERN == 1) {                               // 4 rows x 256 B per instruction
        const int r = lane >> 4, c = lane & 15;
        for (long k = 0; k < kq; k += 128 * UNROLL / 2) {     // UNROLL*4 loads in flight, like pattern 0
#pragma unroll 1
            for (int rb = 0; rb < 32; rb += 4 * 4) {
                u32x4_t v[UNROLL / 2][4 * 2];
#pragma unroll
                for (int q = 0; q < UNROLL / 2; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned short* p = W + (n0 + rb + j * 4 + r) * ldw + k_begin + k + q * 128 + c * 8;
                        v[q][2 * j] = *reinterpret_cast<const u32x4_t*>(p);
                        v[q][2 * j + 1] = *reinterpret_cast<const u32x4_t*>(p + 16 * ldw);
                    }
#pragma unroll
                for (int q = 0; q < UNROLL / 2; ++q)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc ^= v[q][j];
            }
        }
    } else {                                                  // 1 row x 1 KB per instruction (64 lanes x 16 B contiguous)
        for (long k = 0; k < kq; k += 512) {
#pragma unroll 1
            for (int rb = 0; rb < 32; rb += 4 * UNROLL) {
                u32x4_t v[4 * UNROLL];
#pragma unroll
                for (int j = 0; j < 4 * UNROLL; ++j) {
                    long kk = k + lane * 8; if (kk >= kq) kk = kq - 8;
                    v[j] = *reinterpret_cast<const u32x4_t*>(W + (n0 + rb + j) * ldw + k_begin + kk);
                }
#pragma unroll
                for (int j = 0; j < 4 * UNROLL; ++j) acc ^= v[j];
            }
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = 1;
}

int main() {
    const long N = 37888, K = 3584, ncopy = 6;
    std::vector<unsigned short*> Ws(ncopy);
    for (auto& w : Ws) { hipMalloc(&w, N * K * 2); hipMemset(w, 0x11, N * K * 2); }
    unsigned* out; hipMalloc(&out, N * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto bench = [&](const char* name, auto launch) {
        for (int i = 0; i < 10; ++i) launch(Ws[i % ncopy]);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 60; ++i) launch(Ws[i % ncopy]);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %7.2f us  %5.2f TB/s\n", name, ms * 1000 / 60, N * K * 2.0 / (ms / 60 * 1e-3) / 1e12);
    };
    const dim3 grid(N / 32), blk(256);
    bench("fragment pattern 16 rows x 64 B, unroll 4", [&](unsigned short* w) { hipLaunchKernelGGL((stream_kernel<0, 4>), grid, blk, 0, 0, w, out, K, K); });
    bench("fragment pattern 16 rows x 64 B, unroll 8", [&](unsigned short* w) { hipLaunchKernelGGL((stream_kernel<0, 8>), grid, blk, 0, 0, w, out, K, K); });
    bench("4 rows x 256 B, 16 loads in flight", [&](unsigned short* w) { hipLaunchKernelGGL((stream_kernel<1, 4>), grid, blk, 0, 0, w, out, K, K); });
    bench("1 row x 1 KB, 16 loads in flight", [&](unsigned short* w) { hipLaunchKernelGGL((stream_kernel<2, 4>), grid, blk, 0, 0, w, out, K, K); });
    bench("1 row x 1 KB, 32 loads in flight", [&](unsigned short* w) { hipLaunchKernelGGL((stream_kernel<2, 8>), grid, blk, 0, 0, w, out, K, K); });
    return 0;
}
