// Round 5: does the MFMA instruction SHAPE change the power-limited sustained rate of the matrix cores?  Register operands only (no LDS, no HBM): 8 waves per CU,
// every wave issues back-to-back MFMAs on N(0,1) bf16 operands into independent accumulators - 16x16x32 (what gemm_nt8p's k loop issues) against 32x32x16
// (half the register-file operand reads per FLOP) - for ~20 ms each so that the power controller settles.  Prints sustained TFLOP/s of each.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_shape.hip -o /tmp/pms && /tmp/pms
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int SHAPE>
__global__ __launch_bounds__(512) void k(const bf16x8_t* __restrict__ src, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x;
    bf16x8_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = src[(blockIdx.x * 512 + lane) * 8 + i]; b[i] = src[(blockIdx.x * 512 + lane) * 8 + 4 + i]; }
    if (SHAPE == 16) {
        f32x4_t acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        out[blockIdx.x * 512 + lane] = s;
    } else {
        f32x16_t acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i + 2 * rep], b[j + 2 * rep], acc[i * 2 + j], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][r];
        out[blockIdx.x * 512 + lane] = s;
    }
}

int main() {
    const int blocks = 256 * 1;
    const size_t n = (size_t)blocks * 512 * 8;
    bf16x8_t* src; float* out;
    hipMalloc(&src, n * sizeof(bf16x8_t)); hipMalloc(&out, blocks * 512 * 4);
    unsigned short* h = (unsigned short*)malloc(n * 16);
    srand(1);
    for (size_t i = 0; i < n * 8; ++i) {      // ~N(0,1) bf16 bit patterns (sum of 4 uniforms), random signs: realistic toggling
        float v = ((rand() % 2001) + (rand() % 2001) + (rand() % 2001) + (rand() % 2001) - 4000) / 1155.0f;
        unsigned u; memcpy(&u, &v, 4); h[i] = (unsigned short)(u >> 16);
    }
    hipMemcpy(src, h, n * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
        for (int shape = 16; shape <= 32; shape += 16) {
            const int iters = 200000;
            // FLOP per iteration per wave: SHAPE 16: 16 MFMAs x 16*16*32*2; SHAPE 32: 8 MFMAs x 32*32*16*2  (both 262144)
            const double flop = (double)blocks * 8 * iters * 262144.0;
            hipEventRecord(e0);
            if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(512), 0, 0, src, out, iters);
            else hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(512), 0, 0, src, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("mfma %dx%dx%d bf16, 256 CUs x 8 waves, register operands, N(0,1) data: %.1f ms -> %.0f TFLOP/s\n", shape, shape, shape == 16 ? 32 : 16, ms, flop / (ms * 1e-3) / 1e12);
        }
    return 0;
}
