// Native decode-step driver: one C call enqueues every kernel of a rollout decode step (28 layers x 7 launches + head) on the stream.
//
// Why: at M <= 64 rows a decode layer is ~150 us of GPU work in 7 launches; driven op by op from the host language each launch costs
// ~20 us of interpreter + binding time, so the host cannot stay ahead of the GPU and every kernel starts after an idle gap
// (measured with tools/rocpd_gaps.py: 2-4 us in front of each GEMM, 16 us per layer).  Launched back to back from native code the
// gaps are < 0.5 us and the host spends ~0.5 ms per step instead of ~4 ms.
//
// Reference: the per-token body of `model.generate` (transformers GenerationMixin sampling loop, called from
// src/time_r1/rl/timer1_trainer.py:568-573): Qwen2VLDecoderLayer.forward TF:559-624 x n_layers, final norm TF:839, lm_head TF:1323.
#include "tr1_common.h"
#include "../../include/timer1_hip.h"

#include <vector>

namespace {
// ---- optional per-launch timing of the decode GEMM families (bench.py roofline): HIP events recorded by THIS driver around each GEMM launch,
// i.e. with the launches back to back as in the timed region (events recorded from the host language bracket ~20 us of interpreter time per
// launch, during which the stream idles - the round-2 bench line was conservative for that reason).  Off unless tr1_decode_profile_begin().
struct DecProf {
    bool on = false;
    std::vector<hipEvent_t> ev[5];        // families: 0 qkv, 1 o, 2 gate/up, 3 down, 4 lm_head; events in (start, stop) pairs
};
DecProf g_prof;
struct ProfScope {
    int fam; hipStream_t s; bool on;
    ProfScope(int f, void* stream) : fam(f), s((hipStream_t)stream), on(g_prof.on) {
        if (on) { hipEvent_t e; hipEventCreate(&e); hipEventRecord(e, s); g_prof.ev[fam].push_back(e); }
    }
    ~ProfScope() {
        if (on) { hipEvent_t e; hipEventCreate(&e); hipEventRecord(e, s); g_prof.ev[fam].push_back(e); }
    }
};
struct Carve {
    char* p; size_t left; bool ok = true;
    void* take(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (bytes > left) { ok = false; return nullptr; }
        void* r = p; p += bytes; left -= bytes; return r;
    }
};
// D_QMASK (fp8 steps only): which matrices of the SAMPLING policy are fp8 - bit 0 qkv, 1 o, 2 gate/up, 3 down, 4 lm_head; a clear bit means that slot of the
// layer table (and the lm_head argument) holds the bf16 weight and the bf16 kernel runs (mixed-precision sampling policies: config 5's drift study)
enum { D_LAYERS, D_HIDDEN, D_HEADS, D_KV, D_HEAD_DIM, D_INTER, D_VOCAB, D_ROWS, D_BATCH, D_SCAP, D_NSPLIT, D_QMASK, D_N };
enum { QM_QKV = 1, QM_O = 2, QM_GU = 4, QM_DOWN = 8, QM_LM = 16, QM_ALL = 31 };
}  // namespace

static int64_t decode_ws_bytes(const int64_t* d) {
    const int64_t R = d[D_ROWS], hid = d[D_HIDDEN], qd = d[D_HEADS] * d[D_HEAD_DIM], kvd = d[D_KV] * d[D_HEAD_DIM];
    const int64_t T = R / d[D_BATCH];
    const int64_t att = d[D_BATCH] * tr1_attn_fwd_workspace_floats(T, d[D_HEADS], d[D_KV], d[D_HEAD_DIM], d[D_NSPLIT]);
    auto al = [](int64_t b) { return (b + 255) & ~(int64_t)255; };
    const int64_t fix = tr1_gemm_skinny_fixup_workspace_floats(R, hid, d[D_INTER]);
    const int64_t plan = tr1_attn_plan_ints(T, d[D_HEADS], d[D_KV], d[D_BATCH]);
    return al(R * hid * 2) * 2 + al(R * (qd + 2 * kvd) * 2) + al(R * qd * 2) + al(((R + 15) / 16 * 16) * qd * 2) + al(((R + 15) / 16 * 16) * d[D_INTER] * 2) + al(att * 4) + al(fix * 4) + al(plan * 4) + 4096;
}

extern "C" int64_t tr1_decode_step_workspace_bytes(const int64_t* dims) { return decode_ws_bytes(dims); }

// One decode step.  w8 = false: bf16 weights, 9 pointers per layer.  w8 = true: the four matrices are fp8 e4m3 with fp32 row scales
// (csrc/gemm_w8.hip), 13 pointers per layer {ln1, qkv.q, qkv.b, o.q, ln2, gu.q, down.q, K cache, V^T cache, qkv.s, o.s, gu.s, down.s}.
static int decode_step_impl(int w8, const void* layer_ptrs, const int64_t* dims, const void* embed, const void* final_norm, const void* lm_head,
                            const void* lm_head_scale, const void* ids, const void* cosb, const void* sinb, const void* slots, const void* pre,
                            const void* lo, const void* hi, void* work, int64_t work_bytes, void* logits, float eps, float scale, void* stream) {
    const int64_t L = dims[D_LAYERS], hid = dims[D_HIDDEN], nh = dims[D_HEADS], nkv = dims[D_KV], hd = dims[D_HEAD_DIM], inter = dims[D_INTER];
    const int64_t V = dims[D_VOCAB], R = dims[D_ROWS], B = dims[D_BATCH], scap = dims[D_SCAP], nsplit = dims[D_NSPLIT];
    TR1_CHECK_ARG(R >= 1 && R <= 64 && B >= 1 && R % B == 0, "decode_step: 1 <= rows <= 64, rows % n_batch == 0");
    TR1_CHECK_ARG(layer_ptrs && dims && work && logits, "decode_step: null argument");
    TR1_CHECK_ARG(!w8 || !(dims[D_QMASK] & QM_LM) || lm_head_scale, "decode_step_w8: lm_head scale missing");
    const int64_t qd = nh * hd, kvd = nkv * hd, qkvd = qd + 2 * kvd, T = R / B;
    const int64_t att_floats = B * tr1_attn_fwd_workspace_floats(T, nh, nkv, hd, nsplit);
    Carve c{(char*)work, (size_t)work_bytes};
    void* hA = c.take(R * hid * 2); void* hB = c.take(R * hid * 2);
    void* qkv = c.take(R * qkvd * 2); void* q = c.take(R * qd * 2); void* o = c.take(((R + 15) / 16 * 16) * qd * 2);
    void* a = c.take(((R + 15) / 16 * 16) * inter * 2); void* att = c.take(att_floats * 4);
    // down_proj (N = hidden, K = intermediate): split-K with in-kernel fixup pays from 16 rows up (tools/microbench.py fixup:
    // 37.9 -> 34.9 us at M = 16, 54 -> 45 us at M = 32); its ticket counters live in `work`, which the caller zero-fills ONCE
    const int64_t fix_floats = tr1_gemm_skinny_fixup_workspace_floats(R, hid, inter);
    void* fix = c.take(fix_floats * 4);
    // relevant-tile lists of this step's split-KV attention: written by layer 0's launch, read by the others
    void* plan = c.take(tr1_attn_plan_ints(T, nh, nkv, B) * 4);
    const int qm = w8 ? (int)(dims[D_QMASK] & QM_ALL) : 0;                     // fp8 matrices of this step
    const bool planned = nsplit > 1 && L > 1;
    const bool down_fixup = !(qm & QM_DOWN) && R >= 16 && inter >= 8192;
    const bool down_fixup8 = w8 == 2 && (qm & QM_DOWN) && R <= 16 && inter >= 8192 && inter % 512 == 0 && hid % 64 == 0;      // fp8 MFMA: LDS-streamed split-K form
    // round 6 (2B shapes): down projection with its whole weight slice in flight (csrc/oproj.hip long-K form) on the fragment-major SwiGLU output of the gate/up launch
    static int down_frag_on = -1;
    if (down_frag_on < 0) { const char* e = getenv("TR1_DOWN_FRAG"); down_frag_on = e ? atoi(e) : 1; }
    const bool down_frag = down_frag_on && !(qm & (QM_DOWN | QM_GU)) && R <= 16 && inter > 3584 && tr1_gemm_oproj_frag_ok(R, hid, inter) &&
                           tr1_norm_gemm_glu_frag_ok(R, inter, hid);
    TR1_CHECK_ARG(c.ok, "decode_step: workspace too small (tr1_decode_step_workspace_bytes)");
    const void* const* lp = (const void* const*)layer_ptrs;
    const int stride = w8 ? 13 : 9;
#define CK(call) do { int e__ = (call); if (e__) return e__; } while (0)
    // w8 == 2: the fp8-MFMA (W8A8) projections; w8 == 1: W8A16 (fp8 codes converted to bf16 in registers)
    auto gemm8 = w8 == 2 ? tr1_gemm_skinny_w8a8 : tr1_gemm_skinny_w8;
    CK(tr1_gather_rows(embed, ids, hA, R, hid, stream));
    void* h = hA; void* h2 = hB;
    for (int64_t i = 0; i < L; ++i) {
        const void* const* w = lp + i * stride;
        if (qm & QM_QKV) {
            CK(gemm8(h, w[0], w[1], w[9], w[2], nullptr, qkv, R, qkvd, hid, hid, hid, qkvd, 0, eps, 0, stream));
            CK(tr1_decode_qkv_post(qkv, qkvd, cosb, sinb, q, qd, (void*)w[7], kvd, (void*)w[8], B * scap, slots, R, nh, nkv, hd, stream));
        } else if (hd % 32 == 0) {      // norm + q/k/v projection + M-RoPE + KV append in one launch
            ProfScope ps(0, stream);
            CK(tr1_norm_gemm_qkv(h, w[0], w[1], w[2], cosb, sinb, q, qd, (void*)w[7], kvd, (void*)w[8], B * scap, slots, R, nh, nkv, hd, hid, hid, hid,
                                 eps, stream));
        } else {
            CK(tr1_norm_gemm_skinny(h, w[0], w[1], w[2], qkv, R, qkvd, hid, hid, hid, qkvd, eps, 0, stream));
            CK(tr1_decode_qkv_post(qkv, qkvd, cosb, sinb, q, qd, (void*)w[7], kvd, (void*)w[8], B * scap, slots, R, nh, nkv, hd, stream));
        }
        // bf16 o projection on the all-stages-in-flight kernel (csrc/oproj.hip) where the shape is covered: the split-KV merge then writes its rows fragment-major
        const bool o_frag = !(qm & QM_O) && nsplit > 1 && hd == 128 && tr1_gemm_oproj_frag_ok(R, hid, qd);
        if (o_frag)
            CK(tr1_attn_fwd_planned_frag(q, qd, w[7], kvd, w[8], B * scap, o, pre, lo, hi, T, nh, nkv, scap, hd, scale, nsplit, att, att_floats, B, scap,
                                         planned ? plan : nullptr, planned ? (i == 0 ? 1 : 2) : 0, stream));
        else
            CK(tr1_attn_fwd_planned(q, qd, w[7], kvd, w[8], B * scap, o, qd, nullptr, pre, lo, hi, T, nh, nkv, scap, hd, scale, nsplit, att, att_floats, B,
                                    scap, planned ? plan : nullptr, planned ? (i == 0 ? 1 : 2) : 0, stream));
        {
            ProfScope ps(1, stream);
            if (o_frag) CK(tr1_gemm_oproj_frag(o, w[3], h, h2, R, hid, qd, qd, hid, hid, stream));
            else if (qm & QM_O) CK(gemm8(o, nullptr, w[3], w[10], nullptr, h, h2, R, hid, qd, qd, qd, hid, hid, eps, 0, stream));
            else CK(tr1_gemm_nt_bf16(o, w[3], h2, nullptr, h, R, hid, qd, qd, qd, hid, hid, 0, 0, stream));          // h2 = o Wo^T + h
        }
        {
            ProfScope ps(2, stream);
            if (qm & QM_GU) CK(gemm8(h2, w[4], w[5], w[11], nullptr, nullptr, a, R, inter, hid, hid, hid, inter, 0, eps, 1, stream));
            else CK(tr1_norm_gemm_skinny(h2, w[4], w[5], nullptr, a, R, inter, hid, hid, hid, inter, eps, down_frag ? 2 : 1, stream));
        }
        ProfScope ps3(3, stream);
        if (w8 == 2 && down_fixup8) CK(tr1_gemm_skinny_fixup_w8a8(a, w[6], w[12], h, nullptr, h2, R, hid, inter, inter, inter, hid, hid, fix, fix_floats, stream));
        else if (qm & QM_DOWN) CK(gemm8(a, nullptr, w[6], w[12], nullptr, h2, h, R, hid, inter, inter, inter, hid, hid, eps, 0, stream));
        else if (down_frag) CK(tr1_gemm_oproj_frag(a, w[6], h2, h, R, hid, inter, inter, hid, hid, stream));         // h = a Wd^T + h2
        else if (down_fixup) CK(tr1_gemm_skinny_fixup(a, w[6], h, nullptr, h2, R, hid, inter, inter, inter, hid, hid, fix, fix_floats, stream));
        else CK(tr1_gemm_nt_bf16(a, w[6], h, nullptr, h2, R, hid, inter, inter, inter, hid, hid, 0, 0, stream));  // h = a Wd^T + h2
    }
    {
        ProfScope ps(4, stream);
        if (qm & QM_LM) CK(gemm8(h, final_norm, lm_head, lm_head_scale, nullptr, nullptr, logits, R, V, hid, hid, hid, V, 0, eps, 0, stream));
        else CK(tr1_norm_gemm_skinny(h, final_norm, lm_head, nullptr, logits, R, V, hid, hid, hid, V, eps, 0, stream));
    }
#undef CK
    return 0;
}

// Measurement helpers (bench.py `roofline`): between begin and end every decode step records HIP events around its GEMM launches.
// end() synchronises, writes per family (qkv, o, gate/up, down, lm_head) the summed and minimum milliseconds and the launch count, frees the events.
extern "C" int tr1_decode_profile_begin(void) {
    for (auto& v : g_prof.ev) { for (auto e : v) hipEventDestroy(e); v.clear(); }
    g_prof.on = true;
    return 0;
}
extern "C" int tr1_decode_profile_end(double* ms_by_family, double* min_ms_by_family, int64_t* launches_by_family) {
    g_prof.on = false;
    for (int f = 0; f < 5; ++f) {
        double ms = 0.0, mn = 0.0;
        auto& v = g_prof.ev[f];
        for (size_t i = 0; i + 1 < v.size(); i += 2) {
            hipEventSynchronize(v[i + 1]);
            float t = 0.f;
            hipEventElapsedTime(&t, v[i], v[i + 1]);
            ms += t;
            if (i == 0 || t < mn) mn = t;
        }
        if (ms_by_family) ms_by_family[f] = ms;
        if (min_ms_by_family) min_ms_by_family[f] = mn;
        if (launches_by_family) launches_by_family[f] = (int64_t)(v.size() / 2);
        for (auto e : v) hipEventDestroy(e);
        v.clear();
    }
    return 0;
}

extern "C" int tr1_decode_step(const void* layer_ptrs, const int64_t* dims, const void* embed, const void* final_norm, const void* lm_head,
                               const void* ids, const void* cosb, const void* sinb, const void* slots, const void* pre, const void* lo,
                               const void* hi, void* work, int64_t work_bytes, void* logits, float eps, float scale, void* stream) {
    return decode_step_impl(0, layer_ptrs, dims, embed, final_norm, lm_head, nullptr, ids, cosb, sinb, slots, pre, lo, hi, work, work_bytes, logits,
                            eps, scale, stream);
}

extern "C" int tr1_decode_step_w8(const void* layer_ptrs, const int64_t* dims, const void* embed, const void* final_norm, const void* lm_head_fp8,
                                  const void* lm_head_scale, const void* ids, const void* cosb, const void* sinb, const void* slots, const void* pre,
                                  const void* lo, const void* hi, void* work, int64_t work_bytes, void* logits, float eps, float scale, void* stream) {
    return decode_step_impl(1, layer_ptrs, dims, embed, final_norm, lm_head_fp8, lm_head_scale, ids, cosb, sinb, slots, pre, lo, hi, work, work_bytes,
                            logits, eps, scale, stream);
}

extern "C" int tr1_decode_step_w8a8(const void* layer_ptrs, const int64_t* dims, const void* embed, const void* final_norm, const void* lm_head_fp8,
                                    const void* lm_head_scale, const void* ids, const void* cosb, const void* sinb, const void* slots, const void* pre,
                                    const void* lo, const void* hi, void* work, int64_t work_bytes, void* logits, float eps, float scale, void* stream) {
    return decode_step_impl(2, layer_ptrs, dims, embed, final_norm, lm_head_fp8, lm_head_scale, ids, cosb, sinb, slots, pre, lo, hi, work, work_bytes,
                            logits, eps, scale, stream);
}
