// Which (row, 32-k block) does lane l's scale VGPR (byte 0, opsel 0) of v_mfma_scale_f32_16x16x128_f8f6f4 apply to?  One lane at a time gets 2^1.
#include <hip/hip_runtime.h>
#include <stdio.h>
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
int main() {
    unsigned char ha[2048], hb[2048]; int hsa[64], hsb[64]; float hc[256];
    const unsigned char val[4] = {0x30, 0x38, 0x40, 0x48};      // 0.5, 1, 2, 4 in e4m3
    unsigned char *da, *db; float* dc; int *dsa, *dsb;
    (void)hipMalloc(&da, 2048); (void)hipMalloc(&db, 2048); (void)hipMalloc(&dc, 1024); (void)hipMalloc(&dsa, 256); (void)hipMalloc(&dsb, 256);
    for (int which = 0; which < 2; ++which) {
        for (int l = 0; l < 64; ++l) for (int t = 0; t < 32; ++t) { ha[l * 32 + t] = which == 0 ? val[l / 16] : 0x38; hb[l * 32 + t] = which == 1 ? val[l / 16] : 0x38; }
        printf("%s scale: lane -> (row/col, k-block)\n", which == 0 ? "A" : "B");
        for (int l0 = 0; l0 < 64; ++l0) {
            for (int l = 0; l < 64; ++l) { hsa[l] = 127; hsb[l] = 127; }
            (which == 0 ? hsa : hsb)[l0] = 128;
            (void)hipMemcpy(da, ha, 2048, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 2048, hipMemcpyHostToDevice);
            (void)hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb);
            (void)hipMemcpy(hc, dc, 1024, hipMemcpyDeviceToHost);
            // D[i][j] at lane (i/4)*16 + j, reg i%4
            float dd = 0.f;
            int hit_r = -1, hit_g = -1, nhit = 0;
            for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
                const float d = hc[((i / 4) * 16 + j) * 4 + (i % 4)] - 240.f;
                if (d != 0.f) {
                    dd = d;
                    const int r = which == 0 ? i : j;
                    int g = -1; for (int q = 0; q < 4; ++q) if (d == 32.f * (0.5f * (1 << q))) g = q;
                    if (r != hit_r || g != hit_g) { hit_r = r; hit_g = g; ++nhit; }
                }
            }
            printf(" %2d->(%d,%d,%g)%s", l0, hit_r, hit_g, dd, nhit > 1 ? "*" : "");
            if (l0 % 8 == 7) printf("\n");
        }
    }
    return 0;
}
