/*
 * mi355_vlm.h -- C ABI of libmi355vlm.so: hand-written gfx950 (MI355X / CDNA4) HIP kernels for the
 * LLM-quest VLM forward/backward hot path.
 *
 * The reference (casinca/LLM-quest) has NO native/FFI layer: every op is a stock torch call inside an
 * nn.Module.forward.  Each entry point below therefore cites the reference *call site*
 * it replaces (file:line relative to the reference checkout); the binding a maintainer adds on the reference
 * side is a ctypes stub, shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers are DEVICE pointers (HBM) unless stated; sizes are element counts unless "_bytes"
 *   - bf16 tensors are passed as `const void*` / `void*` (2 bytes per element, round-to-nearest-even)
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); all calls are asynchronous on it
 *   - no allocation, no synchronisation, no global state inside any entry point (graph-capture safe)
 *   - return 0 on success; non-zero = error, text via mi355_last_error() (thread-local)
 *   - row-major everywhere; "ld" = leading dimension in elements
 */
#ifndef MI355_VLM_H
#define MI355_VLM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_DT_BF16 0
#define MI355_DT_F32 1
/* an fp32 value as three bf16 values [hi | lo | hi] in three column blocks of the row (hi = bf16(x), lo = bf16(x - hi)): the activation operand of an fp32-grade
 * product on bf16 MFMA (mi355_split3_bf16); an OUTPUT dtype of mi355_layernorm_fwd and of mi355_gemm_bf16 (NT, plain / GELU epilogue, no residual): y / C is
 * bf16 [rows, 3 * width], and the producing kernel saves the fp32 round trip of a separate split pass (reference vlm_engine.py:99-104: the fp32 vision tower) */
#define MI355_DT_SPLIT3 2

/* GEMM operand forms (all row-major storage):
 *   NT: C[m,n] = sum_k A[m,k] * B[n,k]   (A: MxK, B: NxK)  -- y = x W^T, nn.Linear forward
 *   NN: C[m,n] = sum_k A[m,k] * B[k,n]   (A: MxK, B: KxN)  -- dx = dy W,   nn.Linear dgrad
 *   TN: C[m,n] = sum_k A[k,m] * B[k,n]   (A: KxM, B: KxN)  -- dW = dy^T x, nn.Linear wgrad            */
#define MI355_GEMM_NT 0
#define MI355_GEMM_NN 1
#define MI355_GEMM_TN 2

/* epilogue flags */
#define MI355_EPI_NONE 0
#define MI355_EPI_GELU_ERF 1 /* out = gelu(acc + bias) (applied before the residual add) */
/* SwiGLU backward fused into the dgrad of lin2 (qwen3_transformer_block.py:48-53): acc = d(act) [M, N]; `residual` = the forward's
 * gate-up output [u | g] [M, 2N] (ldr >= 2N); C = d(gate-up) [M, 2N] (ldc >= 2N) = [acc g sig(g) | acc u sig(g) (1 + g (1 - sig(g)))].
 * Replaces mi355_swiglu_bwd and the d(act) round trip through HBM; bit-identical to the two-kernel form. */
#define MI355_EPI_SWIGLU_BWD 2
/* SwiGLU forward fused into the gate-up projection (NT form; B = the fused weight [lin1 | lin_gate], N = 2F rows): C = the gate-up
 * output [M, 2F] exactly as without the epilogue (the backward needs it), and `residual` is an OUTPUT here: a = lin1(x) * silu(lin_gate(x))
 * [M, F] (ldr >= F).  The kernel fetches the weight rows of a tile as [32 lin1 | 32 lin_gate] groups so u and g of a hidden unit meet
 * in one wave's epilogue; replaces mi355_swiglu_fwd, bit-identical. */
#define MI355_EPI_SWIGLU_FWD 3
/* Linear -> GELU of a training FFN in one launch (vit_transformer_block.py:59-67, vit_engine.py:43-53, qwen3_5_vision_model.py:112-125):
 * C = acc + bias (the pre-activation, kept for the backward), `residual` is an OUTPUT: gelu(C) (erf: nn.GELU(); tanh: approximate="tanh"). */
#define MI355_EPI_GELU_DUAL_ERF 4
#define MI355_EPI_GELU_DUAL_TANH 5
/* GELU backward fused into the dgrad of the following Linear: acc = d(act), `residual` = the forward's pre-activation, C = acc * gelu'(residual).
 * Both pairs are bit-identical to GEMM + mi355_gelu_fwd / mi355_gelu_bwd. */
#define MI355_EPI_GELU_BWD_ERF 6
#define MI355_EPI_GELU_BWD_TANH 7
/* internal to mi355_gemm_bf16_attn_delta (not accepted by mi355_gemm_bf16) */
#define MI355_EPI_ATTN_DELTA 8

const char* mi355_last_error(void);
int mi355_abi_version(void);

/* bf16 x bf16 -> fp32-accumulate MFMA GEMM with fused epilogue:
 *   C = epi(acc + bias[n]) + residual[m,n]
 * Replaces F.linear / nn.Linear (+ bias, + GELU, + residual add) at qwen3_attention.py:91-93,148,
 * qwen3_transformer_block.py:48-53, vit_attention.py:59-61,88, vit_transformer_block.py:59-67,
 * vit_engine.py:43-53, qwen3_model.py:92 and their autograd backward.
 *   out_dtype    MI355_DT_BF16 | MI355_DT_F32 (dtype of C and of `residual`) | MI355_DT_SPLIT3 (NT form, epilogue NONE / GELU_ERF, no residual: C is bf16 [M, 3N], ldc >= 3N)
 *   bias         fp32 [N] or NULL;  residual  [M,N] with ldr, or NULL (may alias C: accumulate)
 *   workspace    optional fp32 scratch (16-byte aligned) of workspace_bytes: lets problems with few output tiles
 *                and a long K (weight gradients) split K over several workgroups (slabs + reduce); NULL = never split.
 *   tile_hint    0 = choose by shape; 1 = 128x128 tile (4 waves), 2 = 256x256 tile (8 waves), 3 = 256x256, alternating wave groups,
 *                4 = 3 with one barrier per phase, 5 = 256x256 as four waves of 128x128 (one per SIMD, accumulators in AGPRs),
 *                7 = 2 as ONE workgroup per CU that walks its share of the tiles (K-tile stream across tile boundaries, packed-bf16 write-out beside the
 *                    stages; same bits as 2): NT form, bf16 output, K % 64 == 0, K >= 128, N % 8 == 0, no bias; plain / residual / SwiGLU-forward /
 *                    SwiGLU-backward epilogues; other calls fall back to 2.  0 takes it by itself from 512 tiles upward (environment variable
 *                    MI355_GEMM_PERSIST_MIN_TILES, read per call, moves that threshold: its 256 workgroups assume they all start together, so a caller that runs
 *                    collectives or another stream's kernels beside its GEMMs raises it for that time -- llm_quest_amd/ddp.py does)
 * Requirements: K-contiguous dims multiple of 8 elements (16-byte rows); see DESIGN.md.            */
int mi355_gemm_bf16(int form, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                    int64_t ldb, void* C, int64_t ldc, int out_dtype, const float* bias, const void* residual,
                    int64_t ldr, int epilogue, void* workspace, int64_t workspace_bytes, int tile_hint, void* stream);

/* Several independent GEMMs of ONE operand form in one launch, no bias / epilogue / K split.  Replaces the weight-gradient
 * half of autograd's backward for the Linear layers of one transformer block (qwen3_transformer_block.py:48-53,84-99,
 * qwen3_attention.py:91-93,148 and the ViT equivalents): each dW = dY^T X alone has fewer output tiles than the chip has
 * CUs, the block's four together fill it.  `residual` may alias C (gradient accumulation).  count <= 8.
 *   tile_hint  0 = choose by total tile count; 1 = 128x128; 3 = 256x256                                              */
typedef struct mi355_gemm_problem {
    int64_t M, N, K;
    const void* A; int64_t lda;
    const void* B; int64_t ldb;
    void* C; int64_t ldc;
    const void* residual; int64_t ldr;
} mi355_gemm_problem;
int mi355_gemm_bf16_grouped(int form, int count, const mi355_gemm_problem* problems, int out_dtype, int tile_hint, void* stream);

/* The out-projection's dgrad of an attention block with the attention backward's row constants as its epilogue (replaces the GEMM + the delta pass
 * of mi355_attn_bwd*; reference: autograd of ctx @ out_proj.weight^T followed by the softmax backward's row sums, qwen3_attention.py:123-146):
 *   C[M, N] = A[M, K] B[N, K]^T          (NT: A = d(out), B = the out-projection's weight TRANSPOSED, i.e. [Hq*D, d_model]; C = d(ctx), bf16)
 *   delta[b, h, s]         = sum_d C[b*S + s, h*D + d] * ctx[b*S + s, h*D + d]     (on the ROUNDED bf16 C, as the stand-alone delta pass reads it)
 *   neg_delta[b, h, s]     = -delta
 *   neg_lse_log2e[b, h, s] = -lse[b, h, s] * log2(e)
 * D = 128, N = Hq * D, M = B * S >= 256; ctx bf16 [M, N] pitch ldctx; lse / delta / neg_* fp32 [B, Hq, S].  neg_lse_log2e and neg_delta are the two
 * arrays inside the attention-backward workspace at mi355_attn_bwd_workspace_rowconst_offset(B, S, Hq, D, 0 / 1) bytes; pass
 * `causal | MI355_ATTN_DELTA_READY` to mi355_attn_bwd_ws / mi355_attn_bwd_qnorm afterwards and they skip their delta pass. */
int mi355_gemm_bf16_attn_delta(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                               const void* ctx, int64_t ldctx, int S, int Hq, int D, const float* lse, float* delta, float* neg_lse_log2e,
                               float* neg_delta, void* stream);

/* column sums: out[n] (+)= sum_m X[m,n]  (bias / cls-token / pos-embedding gradients).  X bf16 or fp32 [M,N] ld=ldx,
 * out fp32 [N]. */
int mi355_colsum(int64_t M, int64_t N, const void* X, int x_dtype, int64_t ldx, float* out, int accumulate, void* stream);

/* RMSNorm, fp32 math, rows of `width` (PytorchRMSNorm, qwen3_attention.py:19-29): y = x*rsqrt(mean(x^2)+eps)*w.
 * x,y,w bf16; rstd fp32 [rows] saved for backward. */
int mi355_rmsnorm_fwd(int64_t rows, int width, const void* x, const void* w, void* y, float* rstd, float eps,
                      void* stream);
/* dx = rstd*(w*dy - xhat*mean(w*dy*xhat)) (+ dres if given); dw_partial fp32 [parts, width] (sum over parts
 * done by mi355_reduce_rows_f32). */
int mi355_rmsnorm_bwd(int64_t rows, int width, const void* x, const void* w, const float* rstd, const void* dy,
                      const void* dres, void* dx, float* dw_partial, int parts, void* stream);
/* out_bf16[n] (+)= sum_p partial[p,n] */
int mi355_reduce_rows_f32(int parts, int64_t n, const float* partial, void* out, int out_dtype, int accumulate,
                          void* stream);

/* Fused per-head QK-RMSNorm + RoPE on the token-major QKV projection (qwen3_attention.py:99-115,
 * common/rope.py:180-243).  qkv bf16 [tokens, (Hq+2Hkv)*D] (q heads, then k heads, then v heads);
 * writes q_out [tokens,Hq*D], k_out [tokens,Hkv*D] bf16 and rstd fp32 [tokens, Hq+Hkv].
 * cos/sin fp32 [ctx, D] (cast to bf16 before use, as the reference does); pos int32 [tokens] = row of cos/sin.
 * qw == kw == NULL: RoPE only, no normalisation (Qwen3.5 vision attention with 2-D axial tables, rope.py:485-500). */
int mi355_qknorm_rope_fwd(int64_t tokens, int Hq, int Hkv, int D, const void* qkv, const void* qw, const void* kw,
                          const float* cos, const float* sin, const int32_t* pos, void* q_out, void* k_out,
                          float* rstd, float eps, void* stream);
/* backward: dq,dk (post-RoPE grads) -> d(qkv)[:, :Hq*D + Hkv*D] written into dqkv (v part untouched);
 * weight-grad partials fp32 [parts, 2*D] (q then k).  dq == NULL: the key heads only (their rows of dqkv, the k half of the partials; the q half is
 * written as zeros) -- the query heads' share then runs inside mi355_attn_bwd_qnorm. */
int mi355_qknorm_rope_bwd(int64_t tokens, int Hq, int Hkv, int D, const void* qkv, const void* qw, const void* kw,
                          const float* cos, const float* sin, const int32_t* pos, const float* rstd,
                          const void* dq, const void* dk, void* dqkv, float* dw_partial, int parts, void* stream);

/* SwiGLU (qwen3_transformer_block.py:48-53): gu bf16 [tokens, 2*F] = [lin1 | lin_gate]; a = lin1*silu(gate). */
int mi355_swiglu_fwd(int64_t tokens, int F, const void* gu, void* a, void* stream);
int mi355_swiglu_bwd(int64_t tokens, int F, const void* gu, const void* da, void* dgu, void* stream);

/* GELU on a bf16 tensor and its backward, n % 8 == 0.  kind 0: exact erf (nn.GELU() in ViTAdapter, vit_engine.py:50;
 * ViTMergeAdapter, qwen3_5_vision_model.py:408); kind 1: tanh approximation (Qwen3_5VisionFFN, qwen3_5_vision_model.py:122). */
int mi355_gelu_fwd(int64_t n, const void* x, void* y, int kind, void* stream);
int mi355_gelu_bwd(int64_t n, const void* x, const void* dy, void* dx, int kind, void* stream);

/* Flash-style attention, token-major operands, never materialising SxS (replaces qwen3_attention.py:121-146,
 * vit_attention.py:74-86).  q [B*S, Hq*D] ld=ldq, k/v [B*S, Hkv*D] ld=ldk/ldv, o [B*S, Hq*D] ld=ldo, bf16;
 * lse fp32 [B,Hq,S] (natural-log-sum-exp of scaled, masked scores).  D in {64,128}.
 * causal: key j visible to query i iff j<=i.  key_mask uint8 [B,S] (1 = real token) or NULL.
 * Masked scores take the reference's finite fill value (finfo(bf16).min/2), not -inf. */
int mi355_attn_fwd(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk,
                   const void* v, int64_t ldv, void* o, int64_t ldo, float* lse, const uint8_t* key_mask,
                   int causal, float scale, void* stream);
/* delta fp32 [B,Hq,S] workspace; dq/dk/dv token-major like q/k/v. */
int mi355_attn_bwd(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk,
                   const void* v, int64_t ldv, const void* o, int64_t ldo, const void* d_o, int64_t lddo,
                   const float* lse, float* delta, void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv,
                   int64_t lddv, const uint8_t* key_mask, int causal, float scale, void* stream);
/* The same backward with a caller-owned scratch matrix (16-byte aligned, mi355_attn_bwd_workspace_bytes(B,S,Hq,D) bytes; 0 = this
 * shape has no such form and workspace may be NULL): the dK/dV pass leaves dS there in bf16 and dQ = scale * dS K is one product over
 * it, instead of a pass that recomputes the scores.  Same results contract as mi355_attn_bwd; the scratch holds nothing afterwards. */
int64_t mi355_attn_bwd_workspace_bytes(int B, int S, int Hq, int D);
/* byte offset, inside that workspace, of the fp32 [B, Hq, S] array -lse * log2(e) (which = 0) or -delta (which = 1); -1 = no workspace form */
int64_t mi355_attn_bwd_workspace_rowconst_offset(int B, int S, int Hq, int D, int which);
/* OR into `causal` of mi355_attn_bwd_ws / mi355_attn_bwd_qnorm: delta and both row-constant arrays are already filled (mi355_gemm_bf16_attn_delta) */
#define MI355_ATTN_DELTA_READY 0x1000000
int mi355_attn_bwd_ws(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk,
                      const void* v, int64_t ldv, const void* o, int64_t ldo, const void* d_o, int64_t lddo,
                      const float* lse, float* delta, void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv,
                      int64_t lddv, const uint8_t* key_mask, int causal, float scale, void* workspace, int64_t workspace_bytes,
                      void* stream);
/* The same with the backward of the projection's QK-norm + RoPE (mi355_qknorm_rope_bwd, query heads) as the write-out of the dQ pass: dQ never
 * exists as a matrix.  head_dim 128 and the workspace form only.  qkv: the PRE-norm projections (query head h at column h*D of a row of pitch
 * ldqkv), q_weight bf16 [D], cos / sin fp32 [positions, D], pos int32 [B*S], rstd fp32 [B*S, rstd_heads] as mi355_qknorm_rope_fwd left them;
 * d(qkv) rows of the query heads go to dqkv (pitch lddqkv); dqw_partial fp32 [mi355_attn_bwd_qnorm_partials(B,S,Hq), D]: one row per
 * workgroup, summed by the caller (mi355_reduce_rows_f32) into the norm-weight gradient.  The key heads: mi355_qknorm_rope_bwd with dq = NULL.
 * rope_cs16 (optional, NULL = read cos / sin): bf16 [positions][cos[:, :D/2] | sin[:, :D/2]], 16-byte aligned -- pass it ONLY when cos[:, D/2:] ==
 * cos[:, :D/2] and sin[:, D/2:] == sin[:, :D/2] (plain RoPE tables, common/rope.py:38-96); the coefficients are rounded to bf16 before use in
 * either form (as the forward applies them), so the results are identical bits from a quarter of the coefficient loads.
 * Reference: GroupedQueryAttention.forward, llm_quest/qwen/qwen3/qwen3_attention.py:103-160 (q_norm, RoPE, SDPA) under autograd. */
int64_t mi355_attn_bwd_qnorm_partials(int B, int S, int Hq);
int mi355_attn_bwd_qnorm(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                         const void* o, int64_t ldo, const void* d_o, int64_t lddo, const float* lse, float* delta, void* dk, int64_t lddk,
                         void* dv, int64_t lddv, const uint8_t* key_mask, int causal, float scale, void* workspace, int64_t workspace_bytes,
                         const void* qkv, int64_t ldqkv, const void* q_weight, const float* cos, const float* sin, const int32_t* pos,
                         const float* rstd, int rstd_heads, void* dqkv, int64_t lddqkv, float* dqw_partial, const void* rope_cs16, void* stream);

/* Row-wise cross entropy on bf16 logits with ignore_index=-100 (engine.py:45,60; vlm_engine.py:39).
 * logits [rows, V] ld=ldl.  loss_rows fp32 [rows] (0 for ignored).  If dlogits != NULL writes
 * (softmax - onehot) * (*grad_scale) (0 for ignored rows) as bf16 (may alias logits: in place). */
int mi355_cross_entropy(int64_t rows, int64_t V, const void* logits, int64_t ldl, const int64_t* targets,
                        float* loss_rows, void* dlogits, const float* grad_scale, void* stream);
/* out[0] = sum(loss_rows)/count, out[1] = count, out[2] = 1/count  (count = #targets != -100) */
int mi355_ce_finalize(int64_t rows, const float* loss_rows, const int64_t* targets, float* out3, void* stream);

/* Embedding gather (qwen3_model.py:69): out[t,:] = table[ids[t],:]; bit-exact copy. */
int mi355_embedding_fwd(int64_t tokens, int width, int64_t vocab, const int64_t* ids, const void* table, void* out,
                        int64_t ldo, void* stream);
/* scatter-add of token grads into an fp32 accumulator [vocab,width] */
int mi355_embedding_bwd(int64_t tokens, int width, int64_t vocab, const int64_t* ids, const void* dout, int64_t ldd,
                        float* dtable_f32, void* stream);

/* The same gradient without atomics, bit-reproducible for any ids (torch's scatter-add into nn.Embedding.weight.grad is not; the reference step
 * qwen3_model.py:69 -> its autograd backward).  sorted_ids: the ids sorted STABLY, perm: the permutation that sorted them (sorted_ids[j] =
 * ids[perm[j]]); dout bf16 [tokens, width] indexed by perm; table bf16 [vocab, width], row pitch ldt: every vocabulary row that occurs becomes
 * (accumulate ? row : 0) + scale * sum of its tokens' rows (summed in token order, fp32); rows that do not occur are left untouched.
 * Ids outside [0, vocab) are skipped.  width, ldd, ldt multiples of 8.  The sum is a fixed two-level tree over blocks of 32 sorted positions (a run
 * of thousands of equal ids -- padding, placeholder tokens -- does not serialise on one wave): workspace fp32, 16-byte aligned,
 * mi355_embedding_bwd_sorted_workspace_bytes(tokens, width) bytes, holds the per-block parts between the two passes. */
int64_t mi355_embedding_bwd_sorted_workspace_bytes(int64_t tokens, int width);
int mi355_embedding_bwd_sorted(int64_t tokens, int width, int64_t vocab, const int64_t* sorted_ids, const int64_t* perm, const void* dout,
                               int64_t ldd, float scale, void* table, int64_t ldt, int accumulate, float* workspace, int64_t workspace_bytes,
                               void* stream);

/* y[c][r] = x[r][c], bf16, pitches in elements.  rows, cols, ldx, ldy multiples of 8; pointers 16-byte aligned.  The backward of a Linear
 * (reference: every nn.Linear on the path, e.g. llm_quest/qwen/qwen3/qwen3_transformer_block.py:7-53) uses it once per weight so that the
 * dgrad GEMM dX = dY W reads W^T in the K-contiguous NT form.                                                                              */
int mi355_transpose_bf16(int64_t rows, int64_t cols, const void* x, int64_t ldx, void* y, int64_t ldy, void* stream);

/* Strided 2-D copy (early-fusion concat, vlm_engine.py:114; logits/hidden row slicing): bit-exact.
 * dst[r, 0:width] = src[r, 0:width] for r in [0, rows), element size elem_bytes. */
int mi355_copy2d(int64_t rows, int64_t width_bytes, const void* src, int64_t src_pitch_bytes, void* dst,
                 int64_t dst_pitch_bytes, void* stream);

/* im2row patch gather for Conv2d(k=s=P) (vit_model.py:50-57,77-84): img fp32 NCHW -> rows bf16 or fp32
 * [B*gh*gw, C*P*P] with K ordered (c,i,j); patches row-major over (ph,pw). */
int mi355_patchify(int B, int C, int H, int W, int P, const float* img, void* rows, int out_dtype, void* stream);

/* 3-D patch gather for Conv3d(k = s = (TP,P,P)) (qwen3_5_vision_model.py:79-107): img fp32 (B,C,T,H,W) -> rows
 * [B*(T/TP)*gh*gw, C*TP*P*P], tokens ordered (t',ph,pw), K ordered (c,dt,i,j); bit-exact index map. */
int mi355_patchify3d(int B, int C, int T, int H, int W, int P, int TP, const float* img, void* rows, int out_dtype, void* stream);
/* ViTMergeAdapter m x m spatial merge (qwen3_5_vision_model.py:421-424), a bit-exact row permutation:
 * merged[(f,bh,bw)][(i,j,:)] = x[(f, bh*m+i, bw*m+j)][:]; inverse != 0 applies the inverse map (backward). */
int mi355_merge_patches(int64_t frames, int gh, int gw, int m, int64_t row_bytes, const void* src, void* dst, int inverse, void* stream);
/* masked_scatter of vision rows into the embedded token sequence, row-major fill (qwen3_5_vlm_model.py:206-211).
 * forward  (backward=0): out_a[t] = mask[t] ? b[slot[t]] : a[t]            (a = token embeddings, b = vision rows)
 * backward (backward=1): out_a[t] = mask[t] ? 0 : a[t];  out_b[slot[t]] = a[t] where mask[t]   (a = grad of the fused rows)
 * slot[t] = number of set mask entries before t (int32, computed by the caller). */
int mi355_scatter_rows(int64_t tokens, int64_t row_bytes, const uint8_t* mask, const int32_t* slot, const void* a, const void* b,
                       void* out_a, void* out_b, int backward, void* stream);

/* LayerNorm on fp32 rows.  mode 0: the reference's ViT/GPT LayerNorm, eps ADDED TO sigma (vit_transformer_block.py:12-31);
 * mode 1: nn.LayerNorm, eps inside the square root (Qwen3.5 vision blocks, qwen3_5_vision_model.py:213-214,406).
 * y bf16, fp32 or MI355_DT_SPLIT3 (bf16 [rows, 3 * width]); saves mean / 1/(sigma-term) fp32 [rows] if non-NULL. */
int mi355_layernorm_fwd(int64_t rows, int width, const float* x, const float* scale, const float* shift, void* y,
                        int y_dtype, float* mean, float* rsig, float eps, int mode, void* stream);

/* backward of the same LayerNorm: dy bf16 or fp32; dx fp32 (+ dres, the residual-stream gradient); per-block partials
 * [parts][2*width] = (dscale | dshift), summed by mi355_reduce_rows_f32. */
int mi355_layernorm_bwd(int64_t rows, int width, const float* x, const float* scale, const float* mean, const float* rsig,
                        const void* dy, int dy_dtype, const float* dres, float* dx, float* dparam_partial, int parts,
                        float eps, int mode, void* stream);

/* dtype conversion / elementwise helpers */
int mi355_cast(int64_t n, const void* src, int src_dtype, void* dst, int dst_dtype, void* stream);
/* y[b, s, :] = x[b, s, :] + pos[s, :]  with row 0 of each batch = cls + pos[0] (vit_model.py:86-87,145) */
int mi355_vit_embed_assemble(int B, int S, int width, const float* patch_proj, const float* cls, const float* pos,
                             float* out, void* stream);
/* The frozen vision tower at the reference's precision (multimodal/vlm_engine.py:99-104 calls the ViT without autocast: fp32 tensors end to end).
 * mi355_split3_bf16: x fp32 [rows, K] (row pitch ldx) -> y bf16 [rows, 3K] dense, hi = bf16(x), lo = bf16(x - hi);  weight_order 0: [hi | lo | hi]
 * (activations), 1: [hi | hi | lo] (weights), so that ONE mi355_gemm_bf16(NT) with K' = 3K accumulates a_hi w_hi + a_lo w_hi + a_hi w_lo in fp32:
 * the nn.Linear calls of vit_attention.py:58-60,88, vit_transformer_block.py:59-63 and the patch projection vit_model.py:83 at fp32 grade. */
int mi355_split3_bf16(int64_t rows, int K, const float* x, int64_t ldx, void* y, int weight_order, void* stream);
/* softmax(Q K^T * scale) V on fp32 tensors, all keys visible (vit_attention.py:73-82), exact-fp32 MFMA.  q/k/v/o token-major [B*S, H*D] views with
 * row pitches in floats; D == 64, S <= 288. */
int mi355_attn_f32_fwd(int B, int S, int H, int D, const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                       float* o, int64_t ldo, float scale, void* stream);
/* sum of squares of a bf16/fp32 vector into out[0] (+=) : global grad-norm for clip_grad_norm_ (engine.py:445).  partials: scratch of
 * MI355_SUMSQ_PARTS floats owned by the caller -- one per stream (per-block partial sums, added in a fixed order: no float atomics). */
#define MI355_SUMSQ_PARTS 4096
int mi355_sumsq(int64_t n, const void* x, int dtype, float* out, float* partials, void* stream);
/* x *= min(1, max_norm / (sqrt(*sumsq) + 1e-6))  (torch.nn.utils.clip_grad_norm_ semantics) */
int mi355_clip_scale(int64_t n, void* x, int dtype, const float* sumsq, float max_norm, void* stream);
/* y = x * (*scale), bf16, scale is a DEVICE fp32 scalar (autograd's incoming grad_output, no host sync) */
int mi355_scale_bf16(int64_t n, const void* x, const float* scale, void* y, void* stream);
/* One AdamW step (torch.optim.AdamW semantics: decoupled weight decay, bias correction by `step` >= 1) over a flat buffer
 * of n elements -- one launch per parameter arena instead of one foreach chain per tensor (engine.py:444-450 optimizer.step()).
 * param / grad bf16 or fp32, moments fp32.  sumsq != NULL: the gradient is scaled by min(1, max_norm / (sqrt(*sumsq) + 1e-6))
 * on the fly (the global-norm clip of engine.py:441 fused in; *sumsq from mi355_sumsq over all gradients). */
int mi355_adamw(int64_t n, void* param, int p_dtype, const void* grad, int g_dtype, float* exp_avg, float* exp_avg_sq, float lr,
                float beta1, float beta2, float eps, float weight_decay, int step, const float* sumsq, float max_norm, void* stream);
/* fp32 -> bf16 with add: dst_bf16 = bf16(a_f32 + (b_bf16 or 0)) */
int mi355_add_f32_to_bf16(int64_t n, const float* a, const void* b_bf16, void* dst_bf16, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * Qwen3.5 hybrid text stack (BASELINE config 5): csrc/qwen35.hip, csrc/attention_generic.hip.
 * "proj" below is the token-major output of ONE fused projection GEMM; column blocks of it are passed as pointer + ld.
 * ------------------------------------------------------------------------------------------------------------------- */

/* w_eff = bf16(1 + scale): the (1.0 + self.scale) factor of ZeroCenteredRMSNorm (qwen3_next_attention.py:38-46), rounded to
 * bf16 where the reference rounds it.  Feeds mi355_rmsnorm_fwd/bwd and mi355_headnorm_rope_*; d(scale) = d(w_eff). */
int mi355_zc_weight(int64_t n, const void* scale, void* w_eff, void* stream);

/* Per-token interleaved MRoPE coefficient rows (RoPE.apply_mrope / interleave_mrope_coeffs, common/rope.py:246-343):
 * cos_t[t, j] = cos_t[t, j + R/2] = cos[position_ids[axis(j)][t], j], axis(j) = H for j = 1,4,7.. < 3*sec_h, W for
 * j = 2,5,8.. < 3*sec_w, else T.  cos/sin fp32 [ctx, R]; position_ids int64 [3, tokens]; outputs fp32 [tokens, R]. */
int mi355_mrope_table(int64_t tokens, int R, int64_t ctx, const float* cos, const float* sin, const int64_t* position_ids, int sec_h,
                      int sec_w, float* cos_t, float* sin_t, void* stream);

/* y = x * mask[row]  (the in-place `x *= attn_mask` of FusedGatedDeltaNet, qwen3_5_text_model.py:109-110, and its backward). */
int mi355_rowmask(int64_t rows, int width, const void* x, const uint8_t* mask, void* y, void* stream);

/* Per-head ZeroCenteredRMSNorm + partial rotary embedding on heads that sit STRIDED inside a fused projection
 * (GatedAttention / MRoPEGatedAttention, qwen3_next_attention.py:224-236, qwen3_5_text_model.py:227-233):
 * head h of token t = src[t*ld + h*head_stride .. +D].  w = mi355_zc_weight(scale) bf16 [D].  The first R (<= 64) features
 * rotate with rows pos[t] of cos_t/sin_t fp32 [*, R] (R = 0: no rotation).  out bf16 [tokens, H*D], rstd fp32 [tokens, H]. */
int mi355_headnorm_rope_fwd(int64_t tokens, int H, int D, int R, const void* src, int64_t ld, int64_t head_stride, const void* w,
                            const float* cos_t, const float* sin_t, const int32_t* pos, void* out, float* rstd, float eps, void* stream);
/* dout bf16 [tokens, H*D] -> dsrc (strided like src, own ld / head stride); dw_partial fp32 [parts, D] (sum: mi355_reduce_rows_f32). */
int mi355_headnorm_rope_bwd(int64_t tokens, int H, int D, int R, const void* src, int64_t ld, int64_t head_stride, const void* w,
                            const float* cos_t, const float* sin_t, const int32_t* pos, const float* rstd, const void* dout, void* dsrc,
                            int64_t ldd, int64_t dhead_stride, float* dw_partial, int parts, void* stream);

/* out = ctx * sigmoid(gate)  (qwen3_next_attention.py:221,257); gate head h of token t at gate[t*ldg + h*gate_head_stride .. +D]. */
int mi355_sigmoid_gate_fwd(int64_t tokens, int H, int D, const void* ctx, const void* gate, int64_t ldg, int64_t gate_head_stride, void* out,
                           void* stream);
int mi355_sigmoid_gate_bwd(int64_t tokens, int H, int D, const void* ctx, const void* gate, int64_t ldg, int64_t gate_head_stride,
                           const void* dout, void* dctx, void* dgate, int64_t lddg, int64_t dgate_head_stride, void* stream);

/* Attention with F.scaled_dot_product_attention's boolean-mask semantics as the Qwen3-Next / Qwen3.5 layers call it
 * (qwen3_next_attention.py:238-254, qwen3_5_text_model.py:244-259): causal, -inf fill, and -- as upstream -- padded keys
 * (key_mask == 0) visible to every query.  D in {32, 64, 128, 256}.  Operand layout as mi355_attn_fwd/bwd. */
int mi355_attn_generic_fwd(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                           int64_t ldv, void* o, int64_t ldo, float* lse, const uint8_t* key_mask, float scale, void* stream);
int mi355_attn_generic_bwd(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                           int64_t ldv, const void* o, int64_t ldo, const void* d_o, int64_t lddo, const float* lse, float* delta, void* dq,
                           int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, const uint8_t* key_mask, float scale, void* stream);

/* GDN gates (qwen3_5_text_model.py:115-117, compute_alpha_factor qwen3_next_attention.py:71-100): b_lin / a_lin = the w_beta /
 * w_alpha columns of proj (bf16, row pitch ld).  beta = sigmoid(b_lin) (bf16-rounded), alpha = exp(-exp(log_A) *
 * softplus(a_lin + dt_bias)); both fp32 [tokens, Hv]. */
int mi355_gdn_gates_fwd(int64_t tokens, int Hv, const void* b_lin, const void* a_lin, int64_t ld, const float* log_A, const void* dt_bias,
                        float* beta, float* alpha, void* stream);
/* dparam_partial fp32 [parts, 2*Hv] = (dlog_A | ddt_bias) per block. */
int mi355_gdn_gates_bwd(int64_t tokens, int Hv, const void* b_lin, const void* a_lin, int64_t ld, const float* log_A, const void* dt_bias,
                        const float* dbeta, const float* dalpha, void* db_lin, void* da_lin, int64_t ldd, float* dparam_partial, int parts,
                        void* stream);

/* Depthwise causal Conv1d(kernel 4, left padding 3, cropped to S) + SiLU over the sequence axis of the token-major fused QKV
 * projection (qwen3_5_text_model.py:81-90,138-140).  x bf16 [B*S, C] row pitch ldx; w bf16 [C, 4]; y bf16 [B*S, C]. */
int mi355_causal_conv_silu_fwd(int B, int S, int C, int ksize, const void* x, int64_t ldx, const void* w, void* y, void* stream);
/* dc_ws bf16 [B*S, C] scratch; dx row pitch lddx; dw_partial fp32 [B*ceil(S/token_chunk), C*4]. */
int mi355_causal_conv_silu_bwd(int B, int S, int C, int ksize, const void* x, int64_t ldx, const void* w, const void* dy, void* dc_ws, void* dx,
                               int64_t lddx, float* dw_partial, int token_chunk, void* stream);

/* One decoded token of the same conv (_causal_conv1d_update, qwen3_5_text_model.py:425-456, + SiLU): conv_state bf16 [B, 4, C] TOKEN-MAJOR
 * (the last 4 pre-conv inputs; upstream keeps (b, C, 4)) is shifted by one and takes x_new [B, C] (row pitch ldx); y bf16 [B, C]. */
int mi355_causal_conv_silu_step(int B, int C, int ksize, const void* x_new, int64_t ldx, void* conv_state, const void* w, void* y, void* stream);

/* l2_norm of q / k heads (qwen3_next_attention.py:51-60): x heads at x[t*ldx + h*D]; y bf16 [tokens, H*D]. */
int mi355_l2norm_fwd(int64_t tokens, int H, int D, const void* x, int64_t ldx, void* y, void* stream);
int mi355_l2norm_bwd(int64_t tokens, int H, int D, const void* x, int64_t ldx, const void* dy, void* dx, int64_t lddx, void* stream);

/* gated_delta_rule (qwen3_next_attention.py:103-159), fp32 state, q/k bf16 [B*S, Hqk*Dk] (already l2-normalised), v bf16
 * [B*S, Hv*Dv] row pitch ldv, beta/alpha fp32 [B*S, Hv]; value head h uses q/k head h / (Hv/Hqk) (repeat_interleave).
 * o bf16 [B*S, Hv*Dv].  checkpoints (training): fp32 [B, Hv, ceil(S/chunk), Dv, Dk], chunk = mi355_gated_delta_rule_chunk().
 * initial_state / final_state (optional): fp32 [B, Hv, Dv, Dk], the recurrent state of Qwen3_5Cache (utils.py:535-624); they may be the same
 * buffer (decode: S = 1, _gated_delta_rule_step of qwen3_5_text_model.py:459-507).  Dk in {16, 128}; Dv % 16 == 0. */
int mi355_gated_delta_rule_chunk(void);
int mi355_gated_delta_rule_fwd(int B, int S, int Hqk, int Hv, int Dk, int Dv, const void* q, const void* k, const void* v, int64_t ldv,
                               const float* beta, const float* alpha, void* o, float* checkpoints, const float* initial_state, float* final_state,
                               void* stream);
int64_t mi355_gated_delta_rule_bwd_workspace_bytes(int B, int S, int Hv, int Dk, int Dv);
int mi355_gated_delta_rule_bwd(int B, int S, int Hqk, int Hv, int Dk, int Dv, const void* q, const void* k, const void* v, int64_t ldv,
                               const float* beta, const float* alpha, const float* checkpoints, const void* d_o, void* dq, void* dk, void* dv,
                               int64_t lddv, float* dbeta, float* dalpha, void* workspace, int64_t workspace_bytes, const float* d_final_state,
                               float* d_initial_state, void* stream);
/* d_final_state / d_initial_state (optional, fp32 [B, Hv, Dv, Dk], may be the same buffer): the gradient arriving at the forward's final_state
 * (the state gradient's starting value) and where d(loss) / d(initial_state) is left -- training through a carried recurrent state
 * (gated_delta_rule(..., prev_state), qwen3_next_attention.py:103-159).  The forward must have run with that initial_state and checkpoints. */

/* out = bf16(silu(float(gate)) * RMSNorm_fp32(float(o)))  (post_norm + output gate, qwen3_5_text_model.py:181-187): o bf16
 * [tokens, H*D], w fp32 [D], gate bf16 at gate[t*ldg + h*D]. */
int mi355_gated_rmsnorm_fwd(int64_t tokens, int H, int D, const void* o, const float* w, const void* gate, int64_t ldg, void* out, float* rstd,
                            float eps, void* stream);
int mi355_gated_rmsnorm_bwd(int64_t tokens, int H, int D, const void* o, const float* w, const void* gate, int64_t ldg, const float* rstd,
                            const void* dout, void* d_o, void* dgate, int64_t lddg, float* dw_partial, int parts, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * Input pipeline on the step's left edge: MultimodalDataset's per-sample work, dataset.py:295-383.
 * ------------------------------------------------------------------------------------------------------------------- */

/* transforms.Resize((s, s)) on a PIL image = Pillow's fixed-point two-pass bilinear resampler (Resample.c), bit-exact.  bounds
 * int32 [out, 2] = (first input index, count) and kk int32 [out, ksize] = weights * 2^22 are the tables of precompute_coeffs /
 * normalize_coeffs_8bpc (built on the host, llm_quest_amd/dataset.py::resize_tables).  Horizontal pass: src uint8 (H, W_in, C)
 * rows `src_pitch` bytes apart -> dst uint8 (H, W_out, C).  C <= 4. */
int mi355_resize_h_u8(int H, int W_in, int W_out, int C, const uint8_t* src, int64_t src_pitch, const int32_t* bounds, const int32_t* kk, int ksize,
                      uint8_t* dst, void* stream);
/* Vertical pass fused with transforms.ToTensor and transforms.Normalize (dataset.py:343-349): src uint8 (H_in, W, C) contiguous ->
 * dst fp32 (C, H_out, W) = ((u8 / 255) - mean[c]) / std[c]; mean == std == NULL: ToTensor only (standardize=False). */
int mi355_resize_v_normalize(int H_in, int H_out, int W, int C, const uint8_t* src, const int32_t* bounds, const int32_t* kk, int ksize,
                             const float* mean, const float* stdv, float* dst, void* stream);
/* tokenizer(..., truncation=True, max_length=L, padding="max_length") with pad = eos (dataset.py:337,367-373) on already
 * tokenised captions: flat_ids int64 = all captions back to back, offsets int64 [B+1]; ids_out int64 [B, L], mask_out uint8 [B, L]. */
int mi355_pad_tokens(int B, int L, const int64_t* flat_ids, const int64_t* offsets, int64_t pad_id, int64_t* ids_out, uint8_t* mask_out,
                     void* stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * KV-cache decoding on the step's right edge: generate_loop_kv_cache (generate.py:97-151) with
 * utils.KVCache (utils.py:409-531).  csrc/decode.hip.
 * ------------------------------------------------------------------------------------------------------------------- */

/* y[m, n] = sum_k x[m, k] W[n, k] (+ residual[m, n]) for M <= 8 rows: nn.Linear on one new token per sequence -- a weight
 * stream, one wave per output column.  bf16 in / out, fp32 accumulate. */
int mi355_gemv_bf16(int M, int64_t N, int K, const void* x, int64_t ldx, const void* W, int64_t ldw, void* y, int64_t ldy, const void* residual,
                    int64_t ldr, void* stream);
/* The same weight stream with the row operation that feeds it folded in -- a one-token step is bound by its launch count (12 -> 6 per
 * block): prologue 1 = x rows are RMS-normalised with weight norm_w / eps first (PytorchRMSNorm, qwen3_attention.py:19-29, in front of the QKV,
 * gate-up and head projections); prologue 2 = x holds the fused lin1 | lin_gate rows [M, 2K] and the operand is lin1 * silu(lin_gate)
 * (FFN.forward, qwen3_transformer_block.py:48-53).  Same arithmetic and rounding as mi355_rmsnorm_fwd / mi355_swiglu_fwd followed by
 * mi355_gemv_bf16 (bit-identical); M * K * 2 bytes <= 64 KiB. */
int mi355_gemv_bf16_pro(int M, int64_t N, int K, const void* x, int64_t ldx, int prologue, const void* norm_w, float eps, const void* W, int64_t ldw,
                        void* y, int64_t ldy, const void* residual, int64_t ldr, void* stream);
/* One query row per (batch, head) against `len` cached keys / values (qwen3_attention.py:117-146 with a KV cache and q_seq_len 1):
 * q, o bf16 [B, Hq*D]; caches bf16 token-major, sequence b at k_cache + b*batch_stride, key j at + j*ld, kv head g at + g*D.
 * key_mask uint8 [B, >= len] row pitch ldm (1 = real token) or NULL; masked keys take the reference's finite fill. D in {64,128,256}.
 * len_dev != NULL: the number of valid keys is min(len, *len_dev) read on the device (hipGraph replay), len is then the capacity bound.
 * splits > 1 (long caches: one workgroup streams about 256 keys per memory round trip): the keys of a (sequence, head) are dealt to `splits`
 * workgroups, whose partial results (workspace: B * Hq * splits * (D + 2) floats) a second launch folds in split order. */
int mi355_attn_decode(int B, int Hq, int Hkv, int D, const void* q, const void* k_cache, const void* v_cache, int64_t batch_stride, int64_t ld,
                      int len, const int32_t* len_dev, const uint8_t* key_mask, int64_t ldm, void* o, float scale, int splits, float* workspace, void* stream);
/* mi355_attn_decode started from the fused QKV stream's raw rows qkv bf16 [B, (Hq + 2 Hkv) * D] of ONE new token per sequence
 * (GroupedQueryAttention.forward with a KV cache, qwen3_attention.py:100-146): QK-RMSNorm + RoPE of each head's query and of its kv head's
 * new key inside the launch (arithmetic of mi355_qknorm_rope_fwd), the new key and value heads written to cache row *write_pos, attention
 * over min(capacity, *len_dev) keys -- mi355_qknorm_rope_fwd + mi355_kv_append + mi355_attn_decode in one launch, bit-identical.
 * pos int32 [B] rotary positions; write_pos, len_dev int32 [1] on the device. D in {64,128}. */
int mi355_attn_decode_qkv(int B, int Hq, int Hkv, int D, const void* qkv, int64_t ldqkv, const void* q_norm_w, const void* k_norm_w, const float* cos,
                          const float* sin, const int32_t* pos, void* k_cache, void* v_cache, int64_t batch_stride, int64_t ld, int capacity,
                          const int32_t* write_pos, const int32_t* len_dev, const uint8_t* key_mask, int64_t ldm, void* o, float scale, float eps, int splits,
                          float* workspace, void* stream);
/* The tail of a greedy step under hipGraph replay: tok[b] = next_ids[b], rope_pos[b] += 1, *write_pos += 1, *length += 1 (generate.py:139-148's
 * bookkeeping, kept on the device). */
int mi355_decode_advance(int B, const int64_t* next_ids, int64_t* tok, int32_t* rope_pos, int32_t* write_pos, int32_t* length, void* stream);
/* KVCache append of ONE decoded token per sequence with the write position on the device (so a captured hipGraph of the decode
 * step can be replayed): cache[b, *pos, :] = rows[b, :] for keys and values; width = kv_heads * head_dim. */
int mi355_kv_append(int B, int width, const void* k_rows, int64_t ldk, const void* v_rows, int64_t ldv, void* k_cache, void* v_cache,
                    int64_t batch_stride, int64_t ld, int capacity, const int32_t* pos, void* stream);
/* Greedy sampling (generate.py:472-476, temp == 0): out[r] = index of the row maximum of bf16 logits [rows, V] (first on ties).
 * workspace: rows * 64 * 12 bytes (+8), 8-byte aligned: 64 column ranges per row are scanned in parallel, then merged. */
int mi355_argmax_rows(int64_t rows, int64_t V, const void* logits, int64_t ld, int64_t* out, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * Measurement aid (csrc/probe.hip), no reference counterpart: `blocks` workgroups of four waves run reps x 16 v_mfma_f32_32x32x16_bf16 on random
 * bf16 operands held in registers (2 * 32*32*16 * 16 * reps * 4 * blocks FLOP, no memory traffic) -- the rate the board's power cap leaves the
 * matrix pipe, which bench.py reports beside the dense peak.  out: blocks * 256 floats (a checksum sink).  reps < 0: the shape the GEMMs issue instead,
 * -reps x 32 v_mfma_f32_16x16x32_bf16 on a 128 x 64 wave tile's twelve fragments (2 * 16*16*32 * 32 * -reps * 4 * blocks FLOP).
 * ------------------------------------------------------------------------------------------------------------------- */
int mi355_mfma_pipe_probe(int blocks, int reps, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * Stand-alone rotary embedding and dropout (csrc/rope_dropout.hip, csrc/attention_generic.hip).
 * ------------------------------------------------------------------------------------------------------------------- */

/* RoPE.apply / RoPE.apply_mrope / VisionRoPE.apply on a device tensor (common/rope.py:180-243, 297-358, 484-500):
 *   out = cos * x + sin * cat(-x2, x1) on the first R features of every head, features [R, D) copied (partial rotation),
 * evaluated with the reference's rounding points (bf16: coefficients, both products and the sum each rounded to bf16).
 * x / out: (B, H, S, D) views given by element strides (sb, sh, ss / ob, oh, os), unit inner stride; dtype bf16 or fp32.
 * cos_t / sin_t fp32 [table_rows, R]; coefficient row of (b, s) = idx ? idx[b * S + s] : s.  transpose != 0 applies the
 * adjoint map (the backward: dx from dy). */
int mi355_rope_apply(int B, int H, int S, int D, int R, const void* x, int dtype, int64_t sb, int64_t sh, int64_t ss, const float* cos_t,
                     const float* sin_t, int64_t table_rows, const int32_t* idx, void* out, int64_t ob, int64_t oh, int64_t os, int transpose,
                     void* stream);

/* nn.Dropout(p) in train mode (vit_model.py:146, vit_transformer_block.py:117,124, vit_engine.py:51):
 *   y[i] = (residual ? residual[i] : 0) + (keep(i) ? x[i] / (1 - p) : 0),
 *   keep(i) = Philox4x32-10(key = seed; counter = (i / 4, offset))[i % 4] >= round(p * 2^32).
 * Nothing is stored: the backward is the same call on dy with the same (seed, offset).  x bf16 / fp32, y (and residual, which
 * has y's dtype) bf16 / fp32 -- so the residual add of a block and the bf16 cast of its backward are fused into the pass. */
int mi355_dropout(int64_t n, const void* x, int x_dtype, const void* residual, void* y, int y_dtype, float p, uint64_t seed, uint64_t offset,
                  void* stream);

/* ViTMultiHeadAttention in train mode (vit_attention.py:74-81): softmax over every key (causal = 0) or the keys <= query
 * (causal = 1), dropout with probability p on the normalised weights, then the weighted sum of V.  Operand layout as
 * mi355_attn_fwd/bwd; D in {32, 64, 128, 256}.  Weight (b, h, query, key) keeps iff
 * Philox4x32-10(seed; (key / 4, (b * Hq + h) * S + query, offset))[key % 4] >= round(p * 2^32); p = 0 skips the generator. */
int mi355_attn_dropout_fwd(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                           int64_t ldv, void* o, int64_t ldo, float* lse, int causal, float scale, float p, uint64_t seed, uint64_t offset,
                           void* stream);
int mi355_attn_dropout_bwd(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                           int64_t ldv, const void* o, int64_t ldo, const void* d_o, int64_t lddo, const float* lse, float* delta, void* dq,
                           int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, int causal, float scale, float p, uint64_t seed,
                           uint64_t offset, void* stream);
/* The SDPA call of GatedAttention / MRoPEGatedAttention WITH dropout_p and a padding mask (qwen3_next_attention.py:240-253, qwen3_5_text_model.py:244-259): the mask
 * semantics of mi355_attn_generic_fwd (causal OR padded key, upstream's quirk included) with the Philox dropout of mi355_attn_dropout_fwd on the normalised weights.
 * key_mask uint8 [B, S] (1 = real token) or NULL. */
int mi355_attn_generic_dropout_fwd(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                                   int64_t ldv, void* o, int64_t ldo, float* lse, const uint8_t* key_mask, float scale, float p, uint64_t seed,
                                   uint64_t offset, void* stream);
int mi355_attn_generic_dropout_bwd(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                                   int64_t ldv, const void* o, int64_t ldo, const void* d_o, int64_t lddo, const float* lse, float* delta,
                                   void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, const uint8_t* key_mask, float scale,
                                   float p, uint64_t seed, uint64_t offset, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355_VLM_H */
