// Fused GPU video preprocessing: uint8 frames -> bicubic antialiased resize -> round/clamp to uint8 levels
// -> rescale 1/255 -> CLIP mean/std normalise -> Qwen2-VL patch layout, bf16, K padded for the patch-embed GEMM.
//
// Replaces, for pre-decoded uint8 frames, the host-side chain the reference runs inside compute_loss:
//   torchvision resize(BICUBIC, antialias=True).float()          reference src/utils/vision_process.py:467-472
//   Qwen2VLVideoProcessor rescale / normalize / patchify          transformers/models/qwen2_vl/video_processing_qwen2_vl.py:236-274
// Filter tables (first tap index + normalised cubic weights per output row / column, a = -0.5, support scaled by the
// down-sampling factor) are built on the host exactly like ATen's _compute_indices_weights_aa; the kernel evaluates the separable
// filter as a direct 2-D sum per output pixel (<= 7x7 taps at the reference's sizes) and scatters into the patch layout.
#include "tr1_common.h"

__global__ __launch_bounds__(256) void video_preprocess_kernel(const unsigned char* __restrict__ frames, bf16_t* __restrict__ out, int64_t ld_out,
                                                               const int* __restrict__ ymin, const float* __restrict__ wy, int ty,
                                                               const int* __restrict__ xmin, const float* __restrict__ wx, int tx, int T_in, int T_out,
                                                               int H, int W, int Ho, int Wo, float m0, float m1, float m2, float s0, float s1,
                                                               float s2, int patch, int tpatch, int merge) {
    const int64_t total = (int64_t)T_out * 3 * Ho * Wo;
    const int GH = Ho / patch, GW = Wo / patch;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % Wo);
        const int y = (int)((idx / Wo) % Ho);
        const int c = (int)((idx / ((int64_t)Wo * Ho)) % 3);
        const int t = (int)(idx / ((int64_t)Wo * Ho * 3));
        const int ts = t < T_in ? t : T_in - 1;                      // odd frame counts are padded by repeating the last frame
        const unsigned char* src = frames + ((int64_t)ts * 3 + c) * H * W;
        const int y0 = ymin[y], x0 = xmin[x];
        const float* wyr = wy + (int64_t)y * ty;
        const float* wxr = wx + (int64_t)x * tx;
        float acc = 0.f;
        for (int j = 0; j < ty; ++j) {
            const float wj = wyr[j];
            if (wj == 0.f) continue;
            const unsigned char* rowp = src + (int64_t)min(y0 + j, H - 1) * W;
            float racc = 0.f;
            for (int i = 0; i < tx; ++i) racc += wxr[i] * (float)rowp[min(x0 + i, W - 1)];
            acc += wj * racc;
        }
        float v = fminf(fmaxf(rintf(acc), 0.f), 255.f);              // torchvision: uint8 in -> uint8 levels out, then .float()
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), stdv = c == 0 ? s0 : (c == 1 ? s1 : s2);
        v = (v * (1.0f / 255.0f) - mean) / stdv;
        const int gt = t / tpatch, tp = t % tpatch, gh = y / patch, ph = y % patch, gw = x / patch, pw = x % patch;
        const int bh = gh / merge, ih = gh % merge, bw = gw / merge, iw = gw % merge;
        const int64_t n = (((int64_t)gt * (GH / merge) + bh) * (GW / merge) + bw) * (merge * merge) + ih * merge + iw;
        const int f = ((c * tpatch + tp) * patch + ph) * patch + pw;
        out[n * ld_out + f] = f2bf(v);
    }
}

extern "C" int tr1_video_preprocess(const void* frames_u8, void* out_bf16, int64_t ld_out, const void* ymin, const void* wy, int64_t taps_y,
                                    const void* xmin, const void* wx, int64_t taps_x, int64_t T_in, int64_t T_out, int64_t H, int64_t W, int64_t Ho,
                                    int64_t Wo, float mean0, float mean1, float mean2, float std0, float std1, float std2, int64_t patch,
                                    int64_t temporal_patch, int64_t merge, void* stream) {
    TR1_CHECK_ARG(Ho % (patch * merge) == 0 && Wo % (patch * merge) == 0 && T_out % temporal_patch == 0,
                  "video_preprocess: output size must be a multiple of patch*merge and T_out of the temporal patch");
    TR1_CHECK_ARG(ld_out >= 3 * temporal_patch * patch * patch && T_in >= 1 && T_out >= T_in, "video_preprocess: bad ld_out / frame counts");
    const int64_t total = T_out * 3 * Ho * Wo;
    hipLaunchKernelGGL(video_preprocess_kernel, dim3(tr1_grid_1d(total, 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)frames_u8,
                       (bf16_t*)out_bf16, ld_out, (const int*)ymin, (const float*)wy, (int)taps_y, (const int*)xmin, (const float*)wx, (int)taps_x,
                       (int)T_in, (int)T_out, (int)H, (int)W, (int)Ho, (int)Wo, mean0, mean1, mean2, std0, std1, std2, (int)patch, (int)temporal_patch,
                       (int)merge);
    TR1_LAUNCH_CHECK();
}
