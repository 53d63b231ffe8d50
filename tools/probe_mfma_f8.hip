// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950: operand layout and block-scale semantics (what csrc/gemm_w8.hip's W8A8 kernel relies on).
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mfma_f8.hip -o /tmp/probe_f8 && /tmp/probe_f8
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void k(const unsigned char* a, const unsigned char* b, float* c, const int* sa, const int* sb) {
    i32x8 A, B;
    const int* ap = reinterpret_cast<const int*>(a) + threadIdx.x * 8;
    const int* bp = reinterpret_cast<const int*>(b) + threadIdx.x * 8;
    for (int i = 0; i < 8; ++i) { A[i] = ap[i]; B[i] = bp[i]; }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc, 0, 0, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
    for (int i = 0; i < 4; ++i) c[threadIdx.x * 4 + i] = acc[i];
}

static float e4m3(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
    return s ? -x : x;
}

int main() {
    unsigned char ha[64 * 32], hb[64 * 32];
    int hsa[64], hsb[64];
    srand(1);
    for (int i = 0; i < 64 * 32; ++i) { ha[i] = (rand() % 0x48) | ((rand() & 1) << 7); hb[i] = (rand() % 0x48) | ((rand() & 1) << 7); }   // |x| <= 3.5
    for (int l = 0; l < 64; ++l) { hsa[l] = 127; hsb[l] = 127; }
    unsigned char *da, *db; float* dc; int *dsa, *dsb;
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dc, 64 * 4 * 4); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
    float hc[256];
    auto run = [&]() {
        hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
        hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb);
        hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
    };
    run();
    // hypothesis H1: lane l byte t  <->  row/col l % 16, k = (l / 16) * 32 + t;   D[lane][r] = D[row (l/16)*4 + r][col l % 16]
    // hypothesis H2: lane l byte t  <->  k = (t / 16) * 64 + (l / 16) * 16 + t % 16
    for (int hyp = 1; hyp <= 2; ++hyp) {
        double maxerr = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const int i = (l / 16) * 4 + r, j = l % 16;
                double ref = 0;
                for (int g = 0; g < 4; ++g)
                    for (int t = 0; t < 32; ++t) {
                        (void)hyp;
                        ref += (double)e4m3(ha[(g * 16 + i) * 32 + t]) * e4m3(hb[(g * 16 + j) * 32 + t]);
                    }
                if (hyp == 1) maxerr = fmax(maxerr, fabs(ref - hc[l * 4 + r]));
            }
        if (hyp == 1) printf("H1 (row = lane%%16, k = (lane/16)*32 + byte; both operands) max |err| = %g\n", maxerr);
    }
    // scales: A lanes of k-block 2 get 2^1 (E8M0 128), B lanes of k-block 1 get 2^-2 (125): expected D = sum_g sA[g] sB[g] partial_g
    for (int l = 0; l < 64; ++l) { hsa[l] = (l / 16 == 2) ? 128 : 127; hsb[l] = (l / 16 == 1) ? 125 : 127; }
    run();
    double maxerr = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int i = (l / 16) * 4 + r, j = l % 16;
            double ref = 0;
            for (int g = 0; g < 4; ++g) {
                double part = 0;
                for (int t = 0; t < 32; ++t) part += (double)e4m3(ha[(g * 16 + i) * 32 + t]) * e4m3(hb[(g * 16 + j) * 32 + t]);
                ref += part * (g == 2 ? 2.0 : 1.0) * (g == 1 ? 0.25 : 1.0);
            }
            maxerr = fmax(maxerr, fabs(ref - hc[l * 4 + r]));
        }
    printf("per-lane block scales (byte 0 of the scale VGPR = E8M0 of the lane's own 32-k block) max |err| = %g\n", maxerr);
    // scale byte selection: put the scale in byte 1 with opsel 0 -> should be ignored if only byte 0 is read
    for (int l = 0; l < 64; ++l) { hsa[l] = 127 | (130 << 8); hsb[l] = 127; }
    run();
    printf("D[0][0] with scale VGPR = 127 | 130<<8 : %g\n", hc[0]);
    for (int l = 0; l < 64; ++l) { hsa[l] = 127; }
    run();
    printf("D[0][0] with scale VGPR = 127          : %g\n", hc[0]);
    return 0;
}
