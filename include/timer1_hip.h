/*
 * timer1_hip.h - C ABI of libtimer1_hip.so: the MI355X (gfx950) kernels behind the Time-R1 GRPO rollout-and-update path.
 *
 * The reference (xiaomi-research/time-r1) has no native code of its own; its hot path reaches native kernels only through
 * third-party wheels (torch/cuBLAS/flash-attn/DeepSpeed).  Every entry point below names the reference call site whose native
 * work it replaces ("ref:" = path under the reference repo, "TF:" = transformers/models/qwen2_vl/modeling_qwen2_vl.py).
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless stated; bf16 = raw uint16 bfloat16 bits;
 *   - `stream` is a hipStream_t (pass 0 for the null stream); every call is asynchronous on that stream;
 *   - return value 0 = success, otherwise a hipError_t or 1000 (argument check failed); tr1_last_error() returns the message;
 *   - the library never allocates device memory: workspaces are caller-provided.
 *
 * This header is parsed by time-r1_amd/hip.py to build the ctypes signatures, so keep one declaration per statement and
 * only the types: void*, const void*, char*, int64_t*, const int64_t*, int64_t, uint64_t, int, float.
 */
#ifndef TIMER1_HIP_H
#define TIMER1_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- plumbing ------------------------------------------------------------------------------------------------------ */
int tr1_version(void);
const char* tr1_last_error(void);
int tr1_device_info(int device, char* arch, int64_t arch_len, int64_t* n_cu, int64_t* hbm_bytes);
/* Measurement helper (bench.py `peak_probe`, SURVEY 8d roofline): reads `bytes` of device memory once with row-contiguous 1 KiB wave requests;
 * timed by the caller, it gives the practical HBM read ceiling of the box that the decode GEMMs' `roofline.frac_of_measured` is read against. */
int tr1_probe_hbm_read(const void* buf, int64_t bytes, void* sink_u32, void* stream);

/* ---- GEMM: C[M,N] = A[M,K] * B[N,K]^T (+bias[N]) (+residual[M,N]); bf16 in, fp32 accumulate on MFMA ------------------- */
/* ref: every nn.Linear on the path - TF:501-504 (q/k/v/o), TF:459-466 (MLP), TF:251-274 (patch embed), TF:277-290 (merger),
 * TF:1323 (lm_head); called from src/time_r1/rl/timer1_trainer.py:452-457 (logprob forward) and :569-573 (generate).
 * K % 64 == 0 (pad), N % 8 == 0.  out_f32: C is fp32; accumulate (fp32 only): C += result (weight-gradient accumulation).
 * M <= 16 dispatches the HBM-streaming skinny kernel used by rollout decode. */
int tr1_gemm_nt_bf16(const void* A, const void* B, void* C, const void* bias, const void* residual, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int out_f32, int accumulate, void* stream);
/* The same product (b_kmajor = 0: B = [N, K]; 1: B = [K, N] as stored, e.g. the weight itself in a dgrad) for THIN outputs over a long K - the continuation
 * forward's down / o projections (1600 x 3584 x 18944 / 3584), the lm_head's data gradient (1600 x 3584 over K = 152064): the K tiles are dealt to S <= 8 shares
 * that run as blocks of ONE launch into fp32 planes of ws_f32 (>= 2*M*N floats; 8*M*N lets the cost model pick any S), a second launch adds the planes in a
 * fixed order (+bias, +residual): deterministic split-K.  K % 64 == 0; bf16 output. */
int tr1_gemm_splitk_bf16(const void* A, const void* B, void* C, const void* bias, const void* residual, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int b_kmajor, void* ws_f32, int64_t ws_floats, void* stream);
int64_t tr1_gemm_splitk_max_splits(void);
/* C[M,N] = A[M,K] * B[K,N], B K-major ("NN").  The dgrad of a Linear (dX = dY * W; reference: autograd of F.linear under accelerator.backward,
 * TF trainer.py:1952-1961) reads the weight as stored instead of a transposed copy.  Needs M >= 512, N >= 256, K % 64 == 0; bf16 in / out. */
int tr1_gemm_nn_bf16(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, void* stream);
/* Fused-epilogue training GEMMs ("fuse SiLU*up", "M-RoPE fused into QKV epilogue"): each is bit-identical to the GEMM followed by the elementwise
 * kernel it absorbs (the GEMM output is rounded to bf16 exactly where the unfused path stores it).
 * tr1_gemm_glu_bf16: a[M, I] = silu(x Wg^T) * (x Wu^T), Wgu = [2I, K] gate rows then up rows (TF:459-466 Qwen2MLP, reached from
 *   src/time_r1/rl/timer1_trainer.py:452-457); gu_out (NULL = not needed) receives the projection [M, 2I] the backward reads. */
int tr1_gemm_glu_bf16(const void* x, const void* Wgu, const void* bias, void* a_out, void* gu_out, int64_t M, int64_t I, int64_t K, int64_t ldx, int64_t ldw, int64_t lda, int64_t ldgu, void* stream);
/* y = quick_gelu(x W^T + bias): the Qwen2-VL vision MLP's fc1 + activation in one launch (TF:300-301 VisionMlp); bias (optional) of tr1_gemm_glu_bf16 is the
 * [2I] gate|up bias of the Qwen2.5-VL vision MLP (modeling_qwen2_5_vl.py:85-96). */
int tr1_gemm_bias_quickgelu_bf16(const void* x, const void* W, const void* bias, void* y, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw, int64_t ldy, void* stream);
/* tr1_gemm_qkv_rope_bf16: fused q|k|v projection + bias + multimodal rotary embedding for head dim 128 (TF:501-504 q/k/v_proj, TF:212-222
 *   apply_multimodal_rotary_pos_emb): q_out / k_out rotated with cos / sin fp32 [M, 64], v_out plain; k_out may point into the KV cache rows. */
int tr1_gemm_qkv_rope_bf16(const void* x, const void* Wqkv, const void* bias, const void* cosb, const void* sinb, void* q_out, int64_t ldq, void* k_out, int64_t ldk, void* v_out, int64_t ldv, int64_t M, int64_t n_heads, int64_t n_kv, int64_t head_dim, int64_t K, int64_t ldx, int64_t ldw, void* stream);
/* tr1_gemm_qkv_rope_vit_bf16: the vision blocks' q|k|v projection + bias + 2-D rotary embedding (TF:225-248 apply_rotary_pos_emb_vision inside VisionAttention
 *   TF:322-360; Qwen2.5-VL modeling_qwen2_5_vl.py:160-230) for head dim 2 * half = 80, written as 128-wide zero-padded heads (d < half at column d, d + half at
 *   48 + d when half <= 48 - 64 + d otherwise; the caller zero-fills q128 / k128 / v128 once) so that the head-dim-128 attention kernel runs the tower
 *   (tr1_attn_fwd_rows_live96 when half <= 48: every head's live features end at 96).  cos / sin fp32 [M, half]. */
int tr1_gemm_qkv_rope_vit_bf16(const void* x, const void* Wqkv, const void* bias, const void* cosb, const void* sinb, void* q128, int64_t ldq, void* k128, int64_t ldk, void* v128, int64_t ldv, int64_t M, int64_t n_heads, int64_t half, int64_t K, int64_t ldx, int64_t ldw, void* stream);
/* tr1_gemm_nn_glubwd_bf16: dgu[M, 2I] = SwiGLU backward of da = dh[M, H] Wd[H, I] (the down projection as stored) with the saved gu[M, 2I]; da is never
 *   written (autograd of TF:459-466 under accelerator.backward, src/time_r1/rl/timer1_trainer.py:709-737).  dgu_t (optional): the same values once more as
 *   dgu^T [2I, ld_t] (ld_t = M rounded up to 64, columns >= M zero) - the operand of the gate/up weight gradient, written from the epilogue's LDS staging
 *   instead of by a separate transpose pass over the 384 MB of dgu. */
int tr1_gemm_nn_glubwd_bf16(const void* dh, const void* Wd, const void* gu, void* dgu, int64_t M, int64_t I, int64_t H, int64_t lda, int64_t ldb, int64_t ldgu, int64_t lddgu, void* dgu_t, int64_t ld_t, void* stream);
/* Weight gradient without the X^T copy: C[M,N] fp32 (+)= A[M,K] B[K,N], B K-major with only its first b_rows rows valid (A = dY^T zero-padded to
 * K = tokens rounded up to 64, B = the saved activation as stored).  ref: autograd of nn.Linear inside HF Trainer.training_step (TF trainer.py:1892-1961). */
int tr1_gemm_nn_acc_f32(const void* A, const void* B, void* C_f32, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int accumulate, int64_t b_rows, void* stream);
/* Decode-step fusion (M <= 64 rows): out = rmsnorm(x; lnw, eps) @ W[N,K]^T (+ bias), the norm folded into the GEMM's operand load
 * (ref: input_layernorm -> q/k/v_proj TF:559-580 and post_attention_layernorm -> gate/up_proj TF:600-610 inside generate).
 * glu != 0: W is [2N, K] (gate rows, then up rows) and out[M, N] = silu(gate) * up (Qwen2MLP TF:459-466) - no [M, 2N] intermediate.
 * glu == 2 (M <= 16, tr1_norm_gemm_glu_frag_ok): the same values leave FRAGMENT-MAJOR - element (m, n) at (n / 32) * 512 + m * 32 + n % 32 of `out`
 * (16 x N bf16) - the operand layout of tr1_gemm_oproj_frag, which then runs the down projection with its whole weight slice in flight (2B shapes). */
int tr1_norm_gemm_glu_frag_ok(int64_t M, int64_t N, int64_t K);
int tr1_norm_gemm_skinny(const void* x, const void* lnw, const void* W, const void* bias, void* out, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw, int64_t ldc, float eps, int glu, void* stream);
/* Decode rows: input_layernorm -> fused q/k/v projection -> M-RoPE -> KV-cache append in ONE launch (tr1_norm_gemm_skinny + tr1_decode_qkv_post):
 * roped q -> q_out[M, n_heads*hd]; roped k -> kcache[slots[m], :]; v -> vtcache[:, slots[m]].  Wqkv: [(n_heads + 2 n_kv)*hd, K] (q | k | v rows).
 * ref: Qwen2VLAttention.forward TF:521-556 + DynamicCache.update inside generate (timer1_trainer.py:568-573).  head_dim % 32 == 0. */
int tr1_norm_gemm_qkv(const void* x, const void* lnw, const void* Wqkv, const void* bias, const void* cosb, const void* sinb, void* q_out, int64_t ld_q, void* kcache, int64_t k_ld, void* vtcache, int64_t vt_ld, const void* slots, int64_t M, int64_t n_heads, int64_t n_kv, int64_t head_dim, int64_t K, int64_t ldx, int64_t ldw, float eps, void* stream);
/* Narrow decode projections (o_proj / down_proj at M <= 64 rows): C = A B^T (+bias)(+residual) with cross-block split-K and an in-kernel
 * fixup (the last block of a column group sums the fp32 partial tiles).  ws_f32: tr1_gemm_skinny_fixup_workspace_floats() floats whose
 * trailing ticket counters must be ZERO before the first call (the kernel re-arms them).  Same call sites as tr1_gemm_nt_bf16 in generate. */
int tr1_gemm_skinny_fixup(const void* A, const void* B, void* C, const void* bias, const void* residual, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, void* ws_f32, int64_t ws_floats, void* stream);
int64_t tr1_gemm_skinny_fixup_workspace_floats(int64_t M, int64_t N, int64_t K);
/* out[c, r] = in[r, c]; columns [R, ld_out) of out are zero-filled (feeds the NT GEMM for dgrad / wgrad). */
int tr1_transpose_bf16(const void* in, int64_t ld_in, void* out, int64_t ld_out, int64_t R, int64_t C, void* stream);
/* The same transpose with colsum_f32[c] += sum_r in[r, c]: the bias gradient of a Linear (reference: autograd of F.linear, q/k/v_proj bias TF:501-504) taken
 * from the pass that builds dY^T for its weight gradient. */
int tr1_transpose_colsum_bf16(const void* in, int64_t ld_in, void* out, int64_t ld_out, int64_t R, int64_t C, void* colsum_f32, void* stream);

/* ---- normalisation ------------------------------------------------------------------------------------------------- */
/* ref: Qwen2RMSNorm TF:96-110 (fp32 math, cast to bf16 BEFORE the weight multiply).  If residual != NULL the kernel first forms
 * xsum = bf16(x + residual) (the decoder's residual add, TF:559-624), writes it, and normalises xsum.  rstd (fp32[rows]) optional. */
int tr1_rmsnorm_fwd(const void* x, const void* residual, const void* w, void* y, void* xsum, void* rstd, int64_t rows, int64_t cols, float eps, void* stream);
/* dx = rmsnorm'(dy) (+ dres if given);  dw_f32[c] += sum_r dy*xhat   (autograd of the above; ref: accelerator.backward, TF trainer.py:1952-1961).
 * dw_f32 may be NULL (no weight gradient); otherwise ws_f32 holds tr1_rmsnorm_bwd_workspace_floats() floats of per-block partial rows that a
 * second kernel adds in a fixed order (no atomics: the gradient is reproducible). */
int tr1_rmsnorm_bwd(const void* dy, const void* x, const void* w, const void* rstd, const void* dres, void* dx, void* dw_f32, void* ws_f32, int64_t ws_floats, int64_t rows, int64_t cols, void* stream);
int64_t tr1_rmsnorm_bwd_workspace_floats(int64_t rows, int64_t cols);
/* ref: nn.LayerNorm(eps=1e-6) in VisionBlock TF:425-449 and PatchMerger.ln_q TF:277-290 */
int tr1_layernorm_fwd(const void* x, const void* w, const void* b, void* y, void* mean, void* rstd, int64_t rows, int64_t cols, float eps, void* stream);
int tr1_layernorm_bwd(const void* dy, const void* x, const void* w, const void* mean, const void* rstd, void* dx, void* dw_f32, void* db_f32, int64_t rows, int64_t cols, void* stream);

/* ---- activations / elementwise ---------------------------------------------------------------------------------------- */
/* ref: Qwen2MLP TF:459-466: out = silu(gate) * up with gu = [gate | up] along the feature axis */
int tr1_swiglu_fwd(const void* gu, void* out, int64_t rows, int64_t inter, void* stream);
int tr1_swiglu_bwd(const void* dout, const void* gu, void* dgu, int64_t rows, int64_t inter, void* stream);
/* ref: PatchMerger GELU(erf) TF:277-290; VisionMlp quick_gelu TF:293-301 */
int tr1_gelu_fwd(const void* x, void* y, int64_t n, void* stream);
int tr1_gelu_bwd(const void* x, const void* dy, void* dx, int64_t n, void* stream);
int tr1_quickgelu_fwd(const void* x, void* y, int64_t n, void* stream);
int tr1_add_bf16(const void* a, const void* b, void* y, int64_t n, void* stream);
int tr1_cast_f32_to_bf16(const void* x, void* y, int64_t n, void* stream);
int tr1_cast_bf16_to_f32(const void* x, void* y, int64_t n, void* stream);
int tr1_colsum_accum(const void* dy, void* dbias_f32, int64_t rows, int64_t cols, void* stream);

/* ---- rotary embeddings ------------------------------------------------------------------------------------------------ */
/* ref: Qwen2VLRotaryEmbedding TF:117-169 + apply_multimodal_rotary_pos_emb TF:180-222.  pos3: int32 [3, T] (t,h,w) from
 * get_rope_index TF:914-1016; cos/sin: fp32 [T, head_dim/2]; sections (sec_t, sec_h, sec_w) = mrope_section. */
int tr1_mrope_table(const void* pos3, void* cosb, void* sinb, int64_t T, int64_t head_dim, int64_t sec_t, int64_t sec_h, int64_t sec_w, float theta, int round_bf16, void* stream);
/* ref: VisionRotaryEmbedding + apply_rotary_pos_emb_vision TF:225-248; hw: int32 [N, 2] patch (h, w) ids in merge-block order */
int tr1_vision_rope_table(const void* hw, void* cosb, void* sinb, int64_t N, int64_t head_dim, float theta, void* stream);
/* rotate-half RoPE on n_heads heads stored inside rows of `in` (row stride ld_in); backward != 0 applies the adjoint rotation */
int tr1_rope_apply(const void* in, int64_t ld_in, void* out, int64_t ld_out, const void* cosb, const void* sinb, int64_t T, int64_t n_heads, int64_t head_dim, int backward, void* stream);

/* ---- embedding / scatter ---------------------------------------------------------------------------------------------- */
/* ref: embed_tokens TF:1160 and the masked_scatter of video embeddings TF:1170-1176 */
int tr1_gather_rows(const void* table, const void* ids, void* out, int64_t T, int64_t cols, void* stream);
int tr1_scatter_rows(const void* src, const void* idx, void* dst, int64_t T, int64_t cols, void* stream);
int tr1_embed_bwd(const void* dout, const void* ids, void* dtable_f32, int64_t T, int64_t cols, void* stream);

/* ---- attention -------------------------------------------------------------------------------------------------------- */
/* Two-interval masked flash attention: key kv is visible to query token t iff kv < pre[t] or lo[t] <= kv <= hi[t].
 * ref: flash_attn_varlen_func / SDPA via TF:379-396 (ViT, cu_seqlens segments) and TF:521-556 (LLM causal GQA); the shared-prefix
 * form replaces the G-times replicated prompt of timer1_trainer.py:594-599.  Q/O: [T, n_heads*head_dim]; K: [slots, n_kv*head_dim];
 * VT: [n_kv*head_dim, vt_ld] (slot-contiguous).  lse (optional): fp32 [n_heads, T].  nsplit > 1 = split-KV (decode) with a
 * caller workspace of n_batch * tr1_attn_fwd_workspace_floats() floats.  n_batch > 1 runs n_batch independent problems in one launch:
 * problem b uses Q/O/mask rows [b*T, (b+1)*T) and cache slots starting at b*kv_batch_slots (decode over several prompts' caches). */
int tr1_attn_fwd(const void* Q, int64_t q_ld, const void* K, int64_t k_ld, const void* VT, int64_t vt_ld, void* O, int64_t o_ld, void* lse, const void* pre, const void* lo, const void* hi, int64_t T, int64_t n_heads, int64_t n_kv, int64_t n_slots, int64_t head_dim, float scale, int64_t nsplit, void* ws_f32, int64_t ws_floats, int64_t n_batch, int64_t kv_batch_slots, void* stream);
int64_t tr1_attn_fwd_workspace_floats(int64_t T, int64_t n_heads, int64_t n_kv, int64_t head_dim, int64_t nsplit);
/* The same attention (nsplit = 1, one problem, head_dim = 128) with K AND V row-major ([n_slots, n_kv*128], any leading dims % 8 == 0): the
 * 32x32x16-MFMA kernel of round 3 transposes V in its LDS reads, so the LLM's training / prefill / reference-policy forwards (TF:521-556 via
 * timer1_trainer.py:452-457, attn_implementation=flash_attention_2 in scripts/posttrain/train_rl.sh:33) need no V^T copy.  Same results as
 * tr1_attn_fwd up to the order of the fp32 accumulation.  From 3 072 key slots on the launch takes the 64-rows-per-wave kernel of round 6
 * (csrc/attn_fwd64.hip: one wave per SIMD, the softmax of tile t in the MFMA gaps of tiles t-1 / t+1) - bit-identical O and LSE; the
 * environment variable TR1_FWD64 = 1 / 0, read per call, forces / forbids it. */
int tr1_attn_fwd_rows(const void* Q, int64_t q_ld, const void* K, int64_t k_ld, const void* V, int64_t v_ld, void* O, int64_t o_ld, void* lse, const void* pre, const void* lo, const void* hi, int64_t T, int64_t n_heads, int64_t n_kv, int64_t n_slots, int64_t head_dim, float scale, void* stream);
/* tr1_attn_fwd_rows for 128-wide heads whose features 96..127 are ZERO in Q, K and V (the vision towers' head dim 80 in the padded layout of
 * tr1_gemm_qkv_rope_vit_bf16): 12 + 12 instead of 16 + 16 MFMAs per wave and key tile; O columns 96..127 of every head are not written (the caller keeps them
 * zero).  ref: VisionAttention.forward TF:379-396 / Qwen2.5-VL's windowed form, frozen towers (src/time_r1/rl/timer1_trainer.py:264-269). */
int tr1_attn_fwd_rows_live96(const void* Q, int64_t q_ld, const void* K, int64_t k_ld, const void* V, int64_t v_ld, void* O, int64_t o_ld, void* lse, const void* pre, const void* lo, const void* hi, int64_t T, int64_t n_heads, int64_t n_kv, int64_t n_slots, int64_t head_dim, float scale, void* stream);
/* Split-KV decode over the layers of ONE decode step (same pre/lo/hi in every layer of model.generate's step, timer1_trainer.py:568-573): the
 * launch with plan_mode 1 publishes each query tile's relevant-tile list in `plan` (int32[tr1_attn_plan_ints()]), launches with plan_mode 2 start
 * from it instead of reducing the masks again; plan_mode 0 (plan may be NULL) is tr1_attn_fwd.  Results are bit-identical in all three modes. */
int tr1_attn_fwd_planned(const void* Q, int64_t q_ld, const void* K, int64_t k_ld, const void* VT, int64_t vt_ld, void* O, int64_t o_ld, void* lse, const void* pre, const void* lo, const void* hi, int64_t T, int64_t n_heads, int64_t n_kv, int64_t n_slots, int64_t head_dim, float scale, int64_t nsplit, void* ws_f32, int64_t ws_floats, int64_t n_batch, int64_t kv_batch_slots, void* plan, int plan_mode, void* stream);
/* tr1_attn_fwd_planned whose merged rows leave FRAGMENT-MAJOR for tr1_gemm_oproj_frag: Ofrag holds ceil(n_batch * T / 16) * 16 x n_heads * 128 bf16, element (row m,
 * feature k) at ((m / 16) * (n_heads * 4) + k / 32) * 512 + (m % 16) * 32 + k % 32 (the 16 rows x 64 bytes of an MFMA operand fragment are one contiguous KiB).
 * Split-KV launches (nsplit > 1) at head dim 128.  ref: attention half of Qwen2VLAttention.forward inside model.generate (TF:521-552). */
int tr1_attn_fwd_planned_frag(const void* Q, int64_t q_ld, const void* K, int64_t k_ld, const void* VT, int64_t vt_ld, void* Ofrag, const void* pre, const void* lo, const void* hi, int64_t T, int64_t n_heads, int64_t n_kv, int64_t n_slots, int64_t head_dim, float scale, int64_t nsplit, void* ws_f32, int64_t ws_floats, int64_t n_batch, int64_t kv_batch_slots, void* plan, int plan_mode, void* stream);
/* Decode rows (M <= 32): C[M, N] = X @ W[N, K]^T (+ residual) with X fragment-major (layout above, K = n_heads * 128 <= 3584): every block keeps its whole
 * weight slice in flight (HBM -> LDS DMA) and owns its columns over the whole K - no split-K fixup (csrc/oproj.hip).  tr1_gemm_oproj_frag_ok: shape covered?
 * Round 6: also K <= 9216 at M <= 16 when a block's columns (N / 256 rounded up to a divisor of N, <= 8) fit 152 KB of LDS over the whole K - the
 * Qwen2-VL-2B down projection (1536 x 8960: 6 columns x 256 blocks), fed by tr1_norm_gemm_skinny(glu = 2).
 * ref: o_proj of Qwen2VLAttention.forward (TF:553-556) and down_proj of Qwen2MLP.forward (TF:459-466) inside model.generate
 * (src/time_r1/rl/timer1_trainer.py:568-578). */
int tr1_gemm_oproj_frag(const void* Xfrag, const void* W, const void* residual, void* C, int64_t M, int64_t N, int64_t K, int64_t ldw, int64_t ldr, int64_t ldc, void* stream);
int tr1_gemm_oproj_frag_ok(int64_t M, int64_t N, int64_t K);
int64_t tr1_attn_plan_ints(int64_t T, int64_t n_heads, int64_t n_kv, int64_t n_batch);
/* Backward of the above (recompute based): needs K, V row-major.  KT / kt_ld are kept for ABI stability and ignored (may be NULL / 0): the dQ
 * kernel reads its K^T fragments from the K rows with ds_read_b64_tr_b16.  QT / dOT (tr1_pack_transpose copies) only for head dims padded to
 * 32 or 96 (may be NULL for 64 / 128: the 8-wave dK/dV kernel transposes in its LDS reads the same way).
 * delta: fp32 [2*n_heads, T] scratch (delta, then the log2-scaled LSE); qmeta_ws: int32 [8*ceil(T*group/64)] scratch; ws_f32: tr1_attn_bwd_workspace_floats() floats (fp32 dK/dV
 * partials of the query-split dK/dV kernel).  Writes dQ [T, n_heads*hd], dK, dV [slots, n_kv*hd]. */
int tr1_attn_bwd(const void* Q, int64_t q_ld, const void* K, int64_t k_ld, const void* V, int64_t v_ld, const void* KT, int64_t kt_ld, const void* QT, int64_t qt_ld, const void* dOT, int64_t dot_ld, const void* O, int64_t o_ld, const void* dO, int64_t do_ld, const void* lse, void* delta, void* dQ, int64_t dq_ld, void* dK, int64_t dk_ld, void* dV, int64_t dv_ld, const void* pre, const void* lo, const void* hi, void* qmeta_ws, void* ws_f32, int64_t ws_floats, int64_t T, int64_t n_heads, int64_t n_kv, int64_t n_slots, int64_t head_dim, float scale, void* stream);
/* The same backward with the M-RoPE backward (TF:212-222 transposed) folded in: dQ / dK come back with respect to the UN-rotated projections, rotated in
 * the dQ kernel's epilogue and in the dK / dV partial-sum kernel (head dim 128; other shapes run tr1_rope_apply in place).  cos / sin fp32 [T, d/2], n_slots == T. */
int tr1_attn_bwd_rope(const void* Q, int64_t q_ld, const void* K, int64_t k_ld, const void* V, int64_t v_ld, const void* KT, int64_t kt_ld, const void* QT, int64_t qt_ld, const void* dOT, int64_t dot_ld, const void* O, int64_t o_ld, const void* dO, int64_t do_ld, const void* lse, void* delta, void* dQ, int64_t dq_ld, void* dK, int64_t dk_ld, void* dV, int64_t dv_ld, const void* pre, const void* lo, const void* hi, void* qmeta_ws, void* ws_f32, int64_t ws_floats, int64_t T, int64_t n_heads, int64_t n_kv, int64_t n_slots, int64_t head_dim, float scale, const void* rope_cos, const void* rope_sin, void* stream);
int64_t tr1_attn_bwd_workspace_floats(int64_t T, int64_t n_heads, int64_t n_kv, int64_t n_slots, int64_t head_dim);
/* out[(kvh*hd + d) * ld_out + col] = in[t*ld_in + (kvh*group + hq)*hd + d], col = t*group + hq (or slots[t] when slots != NULL, group 1);
 * without slots, columns [T*group, zero_cols) are zero-filled */
int tr1_pack_transpose(const void* in, int64_t ld_in, void* out, int64_t ld_out, const void* slots, int64_t T, int64_t n_heads, int64_t n_kv, int64_t head_dim, int64_t zero_cols, void* stream);
/* Decode-step post-projection in one launch: M-RoPE on q and k of the new tokens, k -> kcache[slots[r]], v -> vtcache[:, slots[r]]
 * (ref: Qwen2VLAttention.forward TF:521-556 with a DynamicCache update inside generate, timer1_trainer.py:568-573) */
int tr1_decode_qkv_post(const void* qkv, int64_t ld, const void* cosb, const void* sinb, void* q_out, int64_t ld_q, void* kcache, int64_t k_ld, void* vtcache, int64_t vt_ld, const void* slots, int64_t R, int64_t n_heads, int64_t n_kv, int64_t head_dim, void* stream);
/* KV-cache append: dst[slots[t], :] = src[t, :] */
int tr1_scatter_slots(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, const void* slots, int64_t T, int64_t cols, void* stream);

/* ---- fp8 weight storage for the rollout (BASELINE config "fp8 weights") --------------------------------------------------------------- */
/* Per-row symmetric quantisation to OCP fp8 e4m3: scale[n] = amax_n / 448 (1 for a zero row), q = rne(w * 448 / amax_n).  Run once per
 * optimizer step on the decoder matrices; only the SAMPLING policy of generate (timer1_trainer.py:568-573) reads the fp8 copy. */
int tr1_quantize_fp8_rows(const void* w_bf16, int64_t ldw, void* q_fp8, int64_t ldq, void* scale_f32, int64_t N, int64_t K, void* stream);
/* Decode rows (M <= 64) against fp8 weights, dequantised to bf16 in registers (W8A16): out = act(x) W^T * scale (+bias)(+residual).
 * lnw != NULL folds rmsnorm(x; lnw, eps) into the operand load; glu != 0: W_fp8 is [2N, K] (gate rows then up rows), out = silu(gate)*up.
 * Same call sites as tr1_gemm_nt_bf16 / tr1_norm_gemm_skinny inside generate.  K % 128 == 0. */
int tr1_gemm_skinny_w8(const void* x, const void* lnw, const void* W_fp8, const void* wscale, const void* bias, const void* residual, void* out, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw, int64_t ldc, int64_t ldr, float eps, int glu, void* stream);
/* W8A8, the "CDNA4 fp8 MFMA" form (BASELINE.json configs[4]): same arguments and call sites, but the fp8 weight codes feed
 * v_mfma_scale_f32_16x16x128_f8f6f4 directly and the activations (after the optional rmsnorm weight) are quantised in the operand load to
 * e4m3 with one power-of-two (E8M0) scale per row and per 32 consecutive k (OCP microscaling); weight row scale in the epilogue. */
int tr1_gemm_skinny_w8a8(const void* x, const void* lnw, const void* W_fp8, const void* wscale, const void* bias, const void* residual, void* out, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw, int64_t ldc, int64_t ldr, float eps, int glu, void* stream);
/* W8A8 split-K + in-kernel fixup form (decode down projection, M <= 16, K % 512 == 0, N % 64 == 0); workspace as tr1_gemm_skinny_fixup. */
int tr1_gemm_skinny_fixup_w8a8(const void* x, const void* W_fp8, const void* wscale, void* out, const void* bias, const void* residual, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw, int64_t ldc, int64_t ldr, void* ws_f32, int64_t ws_floats, void* stream);

/* ---- native decode-step driver ------------------------------------------------------------------------------------------------ */
/* One call enqueues a whole rollout decode step for R <= 64 rows: embed gather -> n_layers x {norm+qkv, rope + KV append, split-KV attention,
 * o_proj + residual, norm + gate/up + SwiGLU, down_proj + residual} -> final norm + lm_head -> logits[R, vocab] (bf16).
 * ref: per-token body of model.generate (timer1_trainer.py:568-573; Qwen2VLDecoderLayer TF:559-624, norm TF:839, lm_head TF:1323).
 * layer_ptrs: HOST array of 9 * n_layers DEVICE pointers {ln1, qkv.w, qkv.b, o.w, ln2, gu.w, down.w, K cache [B*s_cap, kv_dim],
 * V^T cache [kv_dim, B*s_cap]} per layer; dims: HOST int64[12] {n_layers, hidden, n_heads, n_kv, head_dim, intermediate, vocab, rows,
 * n_batch, s_cap, nsplit, fp8_mask (fp8 steps only, ignored here)}; ids int32[R]; cosb/sinb fp32 [R, head_dim/2]; slots int32[R] (absolute cache slot of each row's new token);
 * pre/lo/hi int32[R] (two-interval mask, cache-local per batch entry); work: device scratch of tr1_decode_step_workspace_bytes(dims),
 * ZERO-FILLED once by the caller before the first step (it holds self re-arming split-K ticket counters). */
int tr1_decode_step(const void* layer_ptrs, const int64_t* dims, const void* embed, const void* final_norm, const void* lm_head, const void* ids, const void* cosb, const void* sinb, const void* slots, const void* pre, const void* lo, const void* hi, void* work, int64_t work_bytes, void* logits, float eps, float scale, void* stream);
int64_t tr1_decode_step_workspace_bytes(const int64_t* dims);
/* Same step with fp8 weights (tr1_quantize_fp8_rows): 13 pointers per layer {ln1, qkv.q, qkv.b, o.q, ln2, gu.q, down.q, K cache, V^T cache,
 * qkv.scale, o.scale, gu.scale, down.scale}; lm_head_fp8 / lm_head_scale likewise.  Embedding, norms, biases, KV cache stay bf16.
 * dims[11] = fp8_mask: bit 0 qkv, 1 o, 2 gate/up, 3 down, 4 lm_head (31 = all).  A CLEAR bit keeps that matrix of the sampling policy in bf16: its slot
 * (and `lm_head_fp8`) then holds the bf16 weight and the bf16 decode kernel runs - the mixed-precision policies of the config-5 drift study. */
int tr1_decode_step_w8(const void* layer_ptrs, const int64_t* dims, const void* embed, const void* final_norm, const void* lm_head_fp8, const void* lm_head_scale, const void* ids, const void* cosb, const void* sinb, const void* slots, const void* pre, const void* lo, const void* hi, void* work, int64_t work_bytes, void* logits, float eps, float scale, void* stream);
/* Same step, every projection through tr1_gemm_skinny_w8a8 (fp8 MFMA). */
int tr1_decode_step_w8a8(const void* layer_ptrs, const int64_t* dims, const void* embed, const void* final_norm, const void* lm_head_fp8, const void* lm_head_scale, const void* ids, const void* cosb, const void* sinb, const void* slots, const void* pre, const void* lo, const void* hi, void* work, int64_t work_bytes, void* logits, float eps, float scale, void* stream);
/* Measurement helpers (bench.py `roofline`): between begin and end every tr1_decode_step* call records a pair of HIP events on its
 * stream around each projection GEMM it launches (families 0 qkv, 1 o, 2 gate/up, 3 down, 4 lm_head).  end() synchronises those events and
 * writes, per family, the summed and the minimum launch duration in ms and the number of launches (host arrays of 5).  Not thread-safe. */
int tr1_decode_profile_begin(void);
int tr1_decode_profile_end(double* ms_by_family, double* min_ms_by_family, int64_t* launches_by_family);

/* ---- video preprocessing (SURVEY 8f "next" row 1) ------------------------------------------------------------------------ */
/* ref: torchvision resize(BICUBIC, antialias) at src/utils/vision_process.py:467-472 + Qwen2VLVideoProcessor rescale/normalize/patchify
 * (transformers video_processing_qwen2_vl.py:236-274).  frames: uint8 [T_in,3,H,W] on the device; out: bf16 [N_v, ld_out] patch rows (caller
 * zero-fills the K padding); ymin/wy, xmin/wx: per-output first tap + normalised weights [Ho,taps_y], [Wo,taps_x] (vision_process.aa_filter). */
int tr1_video_preprocess(const void* frames_u8, void* out_bf16, int64_t ld_out, const void* ymin, const void* wy, int64_t taps_y, const void* xmin, const void* wx, int64_t taps_x, int64_t T_in, int64_t T_out, int64_t H, int64_t W, int64_t Ho, int64_t Wo, float mean0, float mean1, float mean2, float std0, float std1, float std2, int64_t patch, int64_t temporal_patch, int64_t merge, void* stream);

/* ---- vocabulary side -------------------------------------------------------------------------------------------------- */
/* ref: src/time_r1/rl/timer1_trainer.py:458-481 (_get_per_token_logps: log_softmax, gather, entropy) */
int tr1_logp_entropy_fwd(const void* logits, int64_t ld, const void* targets, void* logp, void* entropy, void* lse, int64_t R, int64_t V, void* stream);
/* lm_head fused with the log-softmax statistics: hn [M,K] x W [N,K]^T is reduced in the GEMM epilogue to per-64-column (max, sum e, sum x e) and merged
 * per row - the [M, N] logits of _get_per_token_logps (timer1_trainer.py:449-481) never reach HBM.  part_ws: tr1_lmhead_lse_workspace_floats(M, N) floats. */
int tr1_lmhead_lse_fwd(const void* hn, const void* W, const void* targets, void* part_ws, int64_t ws_floats, void* logp, void* ent, void* lse, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, void* stream);
int64_t tr1_lmhead_lse_workspace_floats(int64_t M, int64_t N);
int tr1_logp_bwd(const void* logits, int64_t ld, const void* targets, const void* lse, const void* dlogp, void* dlogits, int64_t ld_out, int64_t R, int64_t V, void* stream);
/* ref: timer1_trainer.py:635-639 (k3 KL), :713-737 (both loss branches).  out3 = {loss, mean masked kl, sum mask}. */
int tr1_grpo_loss(const void* logp, const void* ref_logp, const void* mask, const void* adv, void* dlogp, void* out3, void* row_len, void* row_kl, int64_t G, int64_t C, float beta, int use_grpo, float grad_scale, void* stream);
/* ref: model.generate(do_sample=True, temperature, top_k) at timer1_trainer.py:568-573.  tokens[row*tok_ld + *step_ptr] = draw.
 * group_rows > 0: rows [b*group_rows, (b+1)*group_rows) belong to prompt b and draw from the stream (seed + b*seed_stride, row % group_rows, step),
 * so several prompts sampled in one launch get exactly the tokens of one launch per prompt. */
int tr1_sample_tokens(const void* logits, int64_t ld, int64_t rows, int64_t V, float temperature, int64_t top_k, uint64_t seed, int64_t group_rows, uint64_t seed_stride, const void* step_ptr, void* tokens, int64_t tok_ld, void* finished, int64_t eos_id, int64_t pad_id, int stop_at_eos, void* u_out, void* ws_u32, int64_t ws_words, void* stream);
/* The decode loop's form (model.generate's per-token sampling, timer1_trainer.py:568-573): next_ids[row] (optional) receives the drawn token as well - the
 * buffer the next step's embedding gather reads - and ws_zeroed != 0 promises a workspace zero-filled ONCE and only used through this entry point (the pick
 * kernel re-zeroes it), so neither a copy kernel nor a memset runs between two decode steps. */
int tr1_sample_tokens_step(const void* logits, int64_t ld, int64_t rows, int64_t V, float temperature, int64_t top_k, uint64_t seed, int64_t group_rows, uint64_t seed_stride, const void* step_ptr, void* tokens, int64_t tok_ld, void* finished, int64_t eos_id, int64_t pad_id, int stop_at_eos, void* u_out, void* ws_u32, int64_t ws_words, void* next_ids, int ws_zeroed, void* stream);
int64_t tr1_sample_workspace_words(int64_t rows);

/* ---- optimizer -------------------------------------------------------------------------------------------------------- */
/* ref: DeepSpeed FusedAdam / DeepSpeedCPUAdam selected by scripts/zero3.json:13-21 and zero3_offload.json:24-31 (AdamW, clip 1.0) */
int tr1_sumsq_accum(const void* g, int64_t n, void* out_scalar, void* stream);
int tr1_adamw_step(void* p_f32, void* m_f32, void* v_f32, void* g_f32, void* p_bf16, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, const void* sumsq_scalar, float max_norm, float grad_mult, int zero_grad, void* stream);
/* Data-parallel forms: the all-reduced gradient is consumed from its bf16 wire buffer (no copy back into the fp32 accumulator, which is only
 * zeroed).  ref: DeepSpeed's bf16 gradient all-reduce + FusedAdam (scripts/zero3.json:13-33). */
int tr1_adamw_step_g16(void* p_f32, void* m_f32, void* v_f32, void* g_f32, const void* g_bf16, void* p_bf16, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, const void* sumsq_scalar, float max_norm, float grad_mult, int zero_grad, void* stream);
/* Gradient norm without re-reading the large gradient matrices (ref: torch.nn.utils.clip_grad_norm_ inside HF Trainer.training_step, TF trainer.py:1785 /
 * DeepSpeed gradient_clipping, scripts/zero3.json:35).  tr1_wgrad_f32_sumsq is the weight-gradient GEMM (NT or K-major B form, bit-identical C) whose epilogue
 * also leaves one sum of squares per wave of what it stored - in the LAST micro-step of an accumulation window that is the final gradient; *n_partials (HOST)
 * receives the number of floats written.  tr1_sumsq_partials_accum adds them to the norm scalar in a fixed order (two levels; the buffer must have room for 256 more floats behind its n entries); tr1_sumsq_ranges_periodic adds the small
 * per-layer tensors (same range description as tr1_zero_ranges_periodic).  wire_bf16 (optional, [M, N] bf16, leading dimension ld_wire): the stored value rounded to bf16 as well - the data-parallel gradient exchange's wire copy (dist.GradSync / ShardSync) without its staging pass. */
int tr1_wgrad_f32_sumsq(const void* A, const void* B, void* C_f32, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int accumulate, int b_kmajor, int64_t b_rows, void* sumsq_partials, int64_t partials_capacity, int64_t* n_partials, void* wire_bf16, int64_t ld_wire, void* stream);
int tr1_sumsq_partials_accum(const void* partials_f32, int64_t n, void* out_scalar, void* stream);
int tr1_sumsq_ranges_periodic(const void* g_f32, int64_t base, int64_t stride, int64_t count, const int64_t* rel_ranges, int64_t n_ranges, void* out_scalar, void* stream);
/* g[base + l*stride + r] = 0 for l < count, r in the <= 8 half-open ranges (rel_ranges = HOST array of 2*n_ranges offsets inside one period): clears the small
 * per-layer gradient tensors when the optimizer leaves the large matrices un-zeroed (their weight-gradient GEMMs overwrite them on the first micro-step of the
 * next accumulation window).  ref: optimizer.zero_grad() in HF Trainer.training_step / DeepSpeed engine.step (scripts/zero3.json). */
int tr1_zero_ranges_periodic(void* g_f32, int64_t base, int64_t stride, int64_t count, const int64_t* rel_ranges, int64_t n_ranges, void* stream);
int tr1_sumsq_accum_bf16(const void* g_bf16, int64_t n, void* out_scalar, void* stream);

/* ---- Collectives (rccl_{init, allreduce, reduce_scatter, allgather}) ------------------------------------------------------------------
 * There is ONE exchange path: time-r1_amd/dist.py drives RCCL through torch.distributed's "nccl" backend (which IS librccl on ROCm): per-segment
 * all-reduce (GradSync) or reduce-scatter + all-gather (ShardSync) of the bf16 wire arena, overlapped with the backward.  The round-4/5
 * tr1_rccl_* entry points duplicated that binding without a caller in the product and were removed in round 6; a native host links librccl directly
 * (ncclCommInitRank / ncclAllReduce / ncclReduceScatter / ncclAllGather on the stream it hands to the tr1_* kernels) - see INTEGRATION.md 2c. */

#ifdef __cplusplus
}
#endif
#endif
