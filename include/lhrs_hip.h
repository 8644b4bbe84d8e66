/* lhrs_hip.h - C ABI of the MI355X (gfx950) engine for the LHRS-Bot hot path
 *
 *   CLIP ViT-L/14  ->  AttnPooler projector  ->  LLaMA2-7B (+LoRA)   forward / backward / optimizer step
 *
 * The reference (NJU-LHRS/LHRS-Bot) is 100 % Python and has no FFI of its own; every entry point below replaces
 * the torch / transformers / deepspeed / timm operator that the cited reference line reaches.  The shared
 * library built from lhrs_bot_amd/csrc (liblhrs_hip.so) exports exactly these symbols.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless stated; the caller owns all buffers including workspaces,
 *     the library never allocates device memory;
 *   - lhrs_bf16_t is raw bfloat16 bits; matrices are row-major with an explicit leading dimension in ELEMENTS;
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued, never synchronised;
 *   - return value 0 = enqueued, -1 = rejected (message via lhrs_last_error(), thread-local); no exceptions;
 *   - one host thread per GPU/rank (the reference runs one process per GPU: Script/train_stage1.sh:6-17).
 */
#ifndef LHRS_HIP_H
#define LHRS_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t lhrs_bf16_t;

/* ---- library ---------------------------------------------------------------------------------------- */
const char* lhrs_last_error(void);
int lhrs_abi_version(void);
const char* lhrs_target_arch(void); /* "gfx950" */

/* ---- GEMM (bf16 MFMA) ------------------------------------------------------------------------------- *
 * C[M,N] = act(alpha * A[M,K] . B[N,K]^T + bias[N]) + residual[M,N]
 * Replaces every nn.Linear on the path: HF CLIPAttention/CLIPMLP and LlamaAttention/LlamaMLP/lm_head reached
 * from lhrs/models/rgb_vision_modal.py:166-172 and lhrs/models/text_modal.py:281-292; nn.MultiheadAttention
 * in/out projections and the c_fc/c_proj MLP of lhrs/models/common_arch.py:276-295,302-333; AttnPooler.out_proj
 * (common_arch.py:132,171); and their backward products (dX = dY.W, dW = dY^T.X, both expressed as NT).
 * act: 0 none, 1 quick_gelu (CLIP), 2 gelu-erf (pooler), 3 silu.  out_f32: C is float (for weight gradients);
 * accumulate (f32 only): C += result.  K % 64 == 0, N % 4 == 0, lda/ldb % 8 == 0. */
int lhrs_gemm_bf16_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                      const void* bias, const void* residual, int ldr, int act, int out_f32, int accumulate,
                      float alpha, void* stream);

/* fused LoRA: C = alpha * (A.B^T + A2.B2^T) + bias + residual, A2 [M,K2] = (alpha_lora/r) * X.A_lora^T, B2 [N,K2] = B_lora
 * (peft lora.Linear.forward reached from lhrs/models/text_modal.py:133-151); K2 % 64 == 0 (zero padded). */
int lhrs_gemm_bf16_nt_lora(const void* A, int lda, const void* B, int ldb, const void* A2, int lda2, const void* B2, int ldb2,
                           int K2, void* C, int ldc, int M, int N, int K, const void* bias, const void* residual, int ldr,
                           int out_f32, int accumulate, float alpha, void* stream);
/* 8-bit frozen base weights (the reference trains stages 2/3 with `bits: 8`, lhrs/models/text_modal.py:91-131 -> bitsandbytes LLM.int8;
 * here OCP e4m3 with per-row fp32 scales on the 2x-rate block-scaled MFMA, unit block scales): C[M, N] bf16 =
 * sa[m] * sb[n] * (A8[M, K] . B8[N, K]^T) (+ residual).  A8 / B8 from lhrs_quant_fp8_rows; lda / ldb in bytes; K % 128 == 0. */
int lhrs_gemm_fp8_nt(const void* A8, long lda, const float* sa, const void* B8, long ldb, const float* sb, void* C, int ldc,
                     int M, int N, int K, const void* residual, int ldr, float alpha, void* stream);
/* ... + alpha * A2[M, K2] . B2[N, K2]^T in bf16 on the same accumulators: the LoRA update of an 8-bit base linear (peft lora.Linear over
 * a bitsandbytes base, text_modal.py:91-151); lda2 / ldb2 in elements, K2 % 64 == 0 */
int lhrs_gemm_fp8_nt_lora(const void* A8, long lda, const float* sa, const void* B8, long ldb, const float* sb, const void* A2,
                          int lda2, const void* B2, int ldb2, int K2, void* C, int ldc, int M, int N, int K,
                          const void* residual, int ldr, float alpha, void* stream);
/* lhrs_rmsnorm_fwd / lhrs_rmsnorm_bwd that also emit the per-row e4m3 copy of their result (bytes [rows, cols] + fp32 scale [rows]),
 * bit-identical to lhrs_quant_fp8_rows of the bf16 result; y may be NULL in the forward */
int lhrs_rmsnorm_fwd_q(const void* x, long ldx, const void* w, void* y, long ldy, void* y8, float* y8scale, int rows,
                       int cols, float eps, void* stream);
int lhrs_rmsnorm_bwd_q(const void* dy, const void* x, const void* w, const float* rstd, const void* add, void* dx,
                       void* dx8, float* dx8scale, int rows, int cols, float eps, void* stream);
/* SwiGLU forward / backward (HF LlamaMLP) emitting the per-row e4m3 operand of the next 8-bit-base GEMM directly; the bf16 result is
 * written only when the pointer is non-NULL (an adapter needs it).  act8 [rows, F] / dgu8 [rows, 2F] bytes, scale [rows] fp32. */
int lhrs_swiglu_fwd_q(const void* gate_up, void* act, void* act8, float* scale, long rows, int F, void* stream);
int lhrs_swiglu_bwd_q(const void* dact, const void* gate_up, void* dgu, void* dgu8, float* scale, long rows, int F, void* stream);
/* skinny-N product C[M, N <= 384] = alpha * A[M, K] . B[N, K]^T - the LoRA down-projections s * x.A^T and s * dy.B of peft lora.Linear
 * (text_modal.py:133-151) - with K split across blocks; workspace: lhrs_gemm_skinny_splits(K, N) * M * N floats, caller-owned. */
int lhrs_gemm_skinny_splits(int K, int N);
int lhrs_gemm_bf16_nt_skinny(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, float alpha,
                             float* workspace, void* stream);
/* Long-K product with few output tiles and f32 output - the projector weight gradients dW[out, in] = dY^T X accumulated over all
 * tokens (what autograd computes for nn.Linear / nn.MultiheadAttention in common_arch.py:93-173,302-333): K split across blocks into
 * f32 slabs of `workspace` (lhrs_gemm_splitk_splits(M, N, K) * M * N floats; may be NULL when that is 1), summed in a fixed order. */
int lhrs_gemm_splitk_splits(int M, int N, int K);
/* The same weight gradients straight from the TOKEN-major operands: C[Mo, No] f32 = P[T, Mo]^T . Q[T, No] (P = dY, Q = X as they lie in
 * HBM; both MFMA operands are formed by transposing LDS reads, no transposed copies are written).  Mo, No multiples of 128, ldp / ldq
 * multiples of 8; the token range is split lhrs_gemm_tn_splits() ways into f32 slabs of `workspace` (splits * Mo * No floats; may be
 * NULL when that is 1) summed in a fixed order. */
int lhrs_gemm_tn_splits(int T, int Mo, int No);
int lhrs_gemm_tn_f32(const void* P, long ldp, const void* Q, long ldq, float* C, long ldc, int T, int Mo, int No, float* workspace,
                     void* stream);
int lhrs_gemm_bf16_nt_splitk_f32(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K,
                                 float* workspace, void* stream);
/* LLaMA MLP with SwiGLU fused into the GEMM epilogues (HF LlamaMLP.forward: down(silu(gate(x)) * up(x)); text_modal.py:258-294).
 * fwd: gu[M, 2*ff] = X.Wgu^T (+ A2.B2^T), act[M, ff] = silu(gate) * up.  bwd: dgu[M, 2*ff] = swiglu'(gu) * (dY.WdT^T (+ A2.B2^T)),
 * dgu may alias gu; dact_scratch [M, ff] is only touched by the unfused fallback (may be NULL when lhrs_gemm_swiglu_fusable() == 1).
 * Bit-identical to GEMM + lhrs_swiglu_fwd / lhrs_swiglu_bwd. */
int lhrs_gemm_swiglu_fusable(int M, int ff, int K_fwd, int K_bwd, int K2);
int lhrs_gemm_swiglu_fwd(const void* X, int ldx, const void* Wgu, int ldw, const void* A2, int lda2, const void* B2, int ldb2,
                         int K2, void* gu, int ld_gu, void* act, int ld_act, int M, int ff, int K, void* stream);
int lhrs_gemm_swiglu_bwd(const void* dY, int ldy, const void* WdT, int ldw, const void* A2, int lda2, const void* B2, int ldb2,
                         int K2, const void* gu, void* dgu, int ld_gu, void* dact_scratch, int M, int ff, int K, void* stream);
/* qkv projection with the RoPE of its q / k heads in the GEMM epilogue (HF LlamaAttention.forward: apply_rotary_pos_emb on
 * q_proj / k_proj outputs, rotate_half convention; text_modal.py:258-294): C[M, N] = X.W^T (+ A2.B2^T); the heads of width
 * head_dim in columns [0, rope_cols) are rotated with position m %% pos_mod + pos0 of their row (cos / sin fp32 [pos][head_dim/2]),
 * the other columns are stored as computed.  Bit-identical to lhrs_gemm_bf16_nt(_lora) + lhrs_rope, which is also the fallback. */
int lhrs_gemm_rope_fwd(const void* X, int ldx, const void* W, int ldw, const void* A2, int lda2, const void* B2, int ldb2, int K2,
                       void* C, int ldc, int M, int N, int K, const float* cos_t, const float* sin_t, int pos_mod, int pos0,
                       int rope_cols, int head_dim, void* stream);
/* peft lora_dropout (lora.Linear.forward: lora_B(lora_A(dropout(x))); p = 0.05 in Config/multi_modal_stage2.yaml, train mode only).
 * out = dropout(x) with a counter-based mask over the element index (regenerated identically by the backward), and the masked product
 * C = mask * (alpha * A.B^T) / (1 - p) + residual the dX path needs (dx = dy.W + mask * (s dy B A) / (1 - p)). */
int lhrs_dropout_bf16(const void* x, long ldx, void* out, long ldo, long rows, int cols, float p, unsigned seed, void* stream);
int lhrs_gemm_bf16_nt_dropmask(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                               const void* residual, int ldr, float alpha, float p, unsigned seed, void* stream);
/* persistent big-tile kernels (tuning / A-B tests): 0 never (small tiles only: neither the 16-wave / 144-row kernels nor the four-wave gemm_u4_kernel), non-zero = default */
int lhrs_gemm_set_policy(int allow_256);
/* kernel A/B tests only: 0 disables the tail-row rule (a product whose last round of 256x256 tiles would be nearly empty is cut into
 * whole tile rows for the 16-wave kernel + the remaining rows for the small-tile kernel); default 1 */
int lhrs_gemm_set_tail_split(int on);
/* kernel A/B tests only: 1 (default) = the 16-wave 256x256 kernel runs one persistent workgroup per CU walking its tiles (the next tile's
 * first operands are fetched under the current tile's epilogue); 0 = one workgroup per tile */
int lhrs_gemm_set_persistent(int on);

/* tile height of the persistent GEMM kernels: 0 = always 256 rows (gemm_nt_256s_kernel), 1 (default) = 144 rows (gemm_nt_144s_kernel, 12 waves)
 * whenever ceil(tiles / CUs) rounds of the smaller tile finish first - M = 2184, the reference's micro-batch 8 (Script/train_stage1.sh:11),
 * is 16 x 16 = 256 tiles of 144 x 256 instead of 144 tiles of 256 x 256 -, 2 = 144 rows wherever the kernel applies (A/B tests).  Results
 * are bit-identical between the two tile heights. */
int lhrs_gemm_set_bm144(int mode);
/* ---- the GEMM workspace (no reference counterpart: the reference's GEMMs are torch calls reached from lhrs/models/text_modal.py:258-294) ----
 * CALLER-OWNED device memory of lhrs_gemm_workspace_bytes() bytes, 256-B aligned, registered per device; every GEMM launch that may use it must
 * be ordered on one stream.  ws = NULL unregisters.  One user, on while it is registered: the tail rows of a row-split product (M = 8736:
 * 8192 rows on 256-row tiles + 544 rows) with K >= 8192 are computed split-K (f32 slabs per K slice, summed in a fixed order together with
 * the residual) instead of by 160 small tiles walking all of K. */
long lhrs_gemm_workspace_bytes(void);
int lhrs_gemm_set_workspace(void* ws, long bytes);
/* ---- LLM.int8() base (the reference's `bits: 8`: lhrs/models/text_modal.py:91-131 -> bitsandbytes MatMul8bitLt, 0.41 series) --------------
 * weights once: lhrs_quant_int8_rows -> int8 rows + factor absmax / 127 per row; lhrs_dequant_int8_rows -> the 16-bit weight CB * factor that
 * the backward (dx = dy . dequant(W)) and generate() use.  Per product: lhrs_int8_prepare scans x for outlier columns (any |x| >= thr = 6.0),
 * compacts them (meta[0] = count n, meta[1] = n rounded up to 64; no cap: idx / A2 / B2 are sized for K columns), quantises x per row with
 * those columns zeroed and gathers x[:, O] and dequant(W)[:, O] into the bf16 pair; lhrs_gemm_int8_nt then computes
 * sa[m] * sb[n] * (XQ . WQ^T in int32) + A2 . B2^T (+ residual) in one launch, reading the outlier column count from k2_dev (= meta + 1) on the
 * device.  Workspaces (flags int[K] - the outlier BIT mask lives in its first K / 8 bytes -, idx int[K rounded up to 64], meta int[2]) are the caller's. */
int lhrs_quant_int8_rows(const void* W, long ldw, void* Q, long ldq, float* scale, int N, int K, void* stream);
int lhrs_dequant_int8_rows(const void* Q, long ldq, const float* scale, void* W, long ldw, int N, int K, void* stream);
int lhrs_int8_prepare(const void* X, long ldx, int M, int K, float thr, const void* WQ, long ldwq, const float* wscale, int N, void* XQ,
                      long ldq, float* sx, int* flags, int* idx, int* meta, void* A2, long lda2, void* B2, long ldb2, void* stream);
int lhrs_gemm_int8_nt(const void* A8, long lda, const float* sa, const void* B8, long ldb, const float* sb, const void* A2, int lda2,
                      const void* B2, int ldb2, int K2, const int* k2_dev, void* C, int ldc, int M, int N, int K, const void* residual, int ldr,
                      float alpha, void* stream);
/* ---- 4-bit base storage (the reference's `bits: 4`, `quant_type: nf4 | fp4`, `double_quant`: lhrs/models/text_modal.py:91-107 ->
 * bitsandbytes quantize_4bit / dequantize_4bit, 0.41 series).  Blocks of 64 consecutive elements: absmax + 16-level codes, two per byte (first
 * element in the high nibble); lhrs_dequant4_blocks expands table[code] * absmax back into the bf16 weight that every product reads (bitsandbytes
 * computes y = x . dequant(W)^T in 16 bit, forward and backward).  double_quant: the caller subtracts the mean of absmax and stores the rest through
 * lhrs_quant8_dynamic (blocks of 256 values, the package's sorted 256-entry "dynamic" table passed as code256, nearest entry) /
 * lhrs_dequant8_dynamic (code256[q] * absmax2 + offset). */
int lhrs_quant4_blocks(const void* W, long n, int fp4, void* packed, float* absmax, void* stream);
int lhrs_dequant4_blocks(const void* packed, const float* absmax, long n, int fp4, void* W, void* stream);
int lhrs_quant8_dynamic(const float* x, long n, const float* code256, void* q, float* absmax2, void* stream);
int lhrs_dequant8_dynamic(const void* q, const float* absmax2, long n, const float* code256, float offset, float* out, void* stream);
/* kernel A/B tests only: fewest 64x128 tiles for which the small-tile GEMM takes 64x128 tiles instead of 64x64 (default 256) */
int lhrs_gemm_set_small_thresh(int n);
/* kernel A/B tests only: fewest 256x256 tiles for which lhrs_gemm_bf16_nt picks the big-tile kernel (default 128) */
int lhrs_gemm_set_min_tiles(int n);
/* live HIP-event timing of the persistent GEMM launches, on their launch stream, for bench.py's roofline leg (gemm.hip):
 * enable(n) arms n event pairs (0 = off); read() -> {launches of the plain-epilogue 16-wave kernels (256-row + 144-row tiles), their ms, their
 * flops, all GEMM launches, all GEMM flops}; read_kinds() -> [10][3] = {launches, ms, flops} per kernel instantiation: 0 gemm_nt_256s_kernel plain, 1 its SwiGLU fwd,
 * 2 SwiGLU bwd, 3 RoPE epilogue, 4 gemm_nt_144s_kernel plain, 5 gemm_u4_kernel<0, true> (plain + residual), 6 gemm_u4_kernel<0, false> (plain), 7 gemm_u4_kernel<1, false>
 * (SwiGLU fwd), 8 <2, false> (SwiGLU bwd), 9 <3, false> (RoPE) */
int lhrs_gemm_profile_enable(int max_samples);
/* bracket only every n-th launch of each epilogue variant (default 1): the event records themselves cost stream time (1-2 % of a step) */
int lhrs_gemm_profile_stride(int n);
int lhrs_gemm_profile_read(double* out5_host);
int lhrs_gemm_profile_read_kinds(double* out30_host);
/* gemm_u4_kernel (csrc/gemm_u4.hip): the hand-written four-wave kernel for the plain long-k products (the nn.Linear calls of HF LlamaDecoderLayer / lm_head with
 * no fused epilogue, reached from lhrs/models/text_modal.py:133-151, 258-294: no bias, no activation, bf16 out, alpha 1, K >= 4096, M and N >= 1024) -
 * 256x256x64 tile, 128x128 per wave, AGPR accumulators, paced LDS-DMA, persistent; bit-identical to gemm_nt_256s_kernel without a residual (a residual joins
 * the fp32 sum before the one rounding) and 5-18 % faster on these shapes.  lhrs_gemm_bf16_nt takes it by a SHAPE rule (its 256x256 tiles fill >= 80 % of one
 * round of the CUs): u4_takes() is that rule, a pure function of its arguments - no timing, no cache: runs are bit-reproducible and every data-parallel rank
 * runs the same kernels.  set_u4(0) / LHRS_GEMM_U4=0: the 16-wave kernels everywhere (kernel A/B tests).  u4_nt is the raw launch: 0 launched, 1 not its
 * problem (K % 64, K < 128, alignment), -1 error. */
int lhrs_gemm_set_u4(int on);
int lhrs_gemm_u4_takes(int M, int N, int K, int lda, int ldb, int ldc, int ldr, int has_bias, int act, int out_f32, int accumulate, float alpha);
int lhrs_gemm_u4_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const void* residual, int ldr,
                    void* stream);
/* ... + A2 . B2^T (the fused LoRA update of lhrs_gemm_bf16_nt_lora, K2 a multiple of 64) as K2 / 64 more stages of the same k-loop; lhrs_gemm_bf16_nt_lora takes
 * it by the same shape rule (and the same row cut; the cut rows keep the pair on the small-tile kernels). */
int lhrs_gemm_u4_nt_lora(const void* A, int lda, const void* B, int ldb, const void* A2, int lda2, const void* B2, int ldb2, int K2, void* C, int ldc,
                         int M, int N, int K, const void* residual, int ldr, void* stream);
/* the four-wave kernel with the decoder layer's fused epilogues (lhrs_gemm_rope_fwd / lhrs_gemm_swiglu_fwd / lhrs_gemm_swiglu_bwd semantics without a LoRA pair;
 * bit-identical to them): raw launches, 0 launched, 1 not its problem.  The three operator entry points above take it by the shape rule u4_fused_takes(kind, M, tile
 * columns, K, K2): a LoRA pair only with RoPE (u4_rope_lora: stage 3's q|k|v adapters; K2 a multiple of 64), K >= 4096, M >= 1024, tiles that fill >= 80 % of a round of the CUs with <= 15 % of the last round idle (kind 0 RoPE: tile columns
 * N / 256; kind 1 SwiGLU forward: ff / 128) or <= 5 % (kind 2 SwiGLU backward: ff / 256 - its write-out is VALU-bound on four waves and only wins on a whole-round walk).  u4_main_rows(M, N): rows of a plain product that go to the four-wave kernel when its last
 * round would be mostly empty (the rest: small tiles / split-K over the workspace); = M when nothing is cut. */
int lhrs_gemm_u4_fused_takes(int kind, int M, int tiles_n, int K, int K2);
int lhrs_gemm_u4_main_rows(int M, int N);
int lhrs_gemm_u4_rope(const void* X, int ldx, const void* W, int ldw, void* C, int ldc, int M, int N, int K, const float* cos_t,
                      const float* sin_t, int pos_mod, int pos0, int rope_cols, void* stream);
int lhrs_gemm_u4_rope_lora(const void* X, int ldx, const void* W, int ldw, const void* A2, int lda2, const void* B2, int ldb2, int K2, void* C, int ldc,
                           int M, int N, int K, const float* cos_t, const float* sin_t, int pos_mod, int pos0, int rope_cols, void* stream);
int lhrs_gemm_u4_swiglu_fwd(const void* X, int ldx, const void* Wgu, int ldw, void* gu, int ld_gu, void* act, int ld_act, int M, int ff, int K,
                            void* stream);
int lhrs_gemm_u4_swiglu_bwd(const void* dY, int ldy, const void* WdT, int ldw, const void* gu, void* dgu, int ld_gu, int M, int ff, int K,
                            void* stream);

/* ---- LoRA gradients (peft lora.Linear backward; lhrs/models/text_modal.py:133-151) ------------------------- *
 * C[KP,N] (+)= P[M,KP]^T . Q[M,N]: dA = (s dy B)^T x and dB^T = (s x A^T)^T dy straight from token-major operands.  */
int lhrs_tn_skinny_splits(int M, int N); /* host helper: workspace = splits * KP * N floats */
int lhrs_gemm_tn_skinny(const void* P, long ldp, const void* Q, long ldq, float* C, long ldc, float* partial, int M, int N,
                        int KP, int accumulate, void* stream);
int lhrs_blockdiag_mask(float* g, long ld, int rows, int cols, int r, int w, int active_mask, void* stream);

/* ---- normalisation ---------------------------------------------------------------------------------- *
 * LayerNorm: lhrs/models/common_arch.py:253-259 (+ HF CLIP pre_layrnorm / layer_norm1/2); eps 1e-5.
 * RMSNorm : HF LlamaRMSNorm inside CustomLlamaForCausalLM (lhrs/models/text_modal.py:30-60).            */
int lhrs_layernorm_fwd(const void* x, long ldx, const void* gamma, const void* beta, void* y, long ldy, float* mean,
                       float* rstd, int rows, int cols, float eps, void* stream);
int lhrs_layernorm_bwd_nblk(int rows); /* host helper: workspace rows */
int lhrs_layernorm_bwd(const void* dy, long ld_dy, const void* x, long ldx, const void* gamma, const float* mean,
                       const float* rstd, const void* add, void* dx, long ld_dx, float* dgamma, float* dbeta,
                       float* partial, int accumulate, int rows, int cols, void* stream);
int lhrs_rmsnorm_fwd(const void* x, long ldx, const void* w, void* y, long ldy, float* rstd, int rows, int cols,
                     float eps, void* stream);
int lhrs_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* add, void* dx,
                     int rows, int cols, float eps, void* stream);

/* ---- attention -------------------------------------------------------------------------------------- *
 * desc: int32 [nseq][8] = {q_off, q_len, kv_off, kv_len, kv_rows, causal_off, 0, 0} (token offsets/lengths).
 * Replaces HF CLIPAttention, nn.MultiheadAttention (common_arch.py:302-313) and HF LlamaAttention
 * (causal + key-padding from `attention_mask`, text_modal.py:281-292).  D = 64 or 128.
 * lse / delta are fp32 [nseq][H][LTq] (LTq = max_q rounded up to 64).  Token-transposed operands are formed in LDS
 * by ds_read_b64_tr_b16; no transposed copies are needed.                                                  */
int lhrs_seq_transpose(const void* in, long ld_in, void* out, int cols, int LT, const int* desc, int nseq, int use_kv,
                       void* stream); /* utility; the attention kernels no longer need transposed copies */
int lhrs_attn_fwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo, float* lse,
                  const int* desc, int nseq, int H, int D, int max_q, int max_kv, int LTq, int causal, float scale,
                  void* stream);
/* lhrs_attn_fwd with an HF attention_mask over the keys (batched evaluation with LEFT-padded prompts: DataCollatorForVGSupervisedDataset,
 * lhrs/Dataset/cap_dataset.py:813-854 -> main_vqa.py:205-214 -> HF _prepare_4d_causal_attention_mask): key j of sequence s is
 * visible iff key_mask[s * ld_mask + j] != 0, AND the causal rule, AND j < kv_len. */
int lhrs_attn_fwd_kmask(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo,
                        float* lse, const int* desc, int nseq, int H, int D, int max_q, int max_kv, int LTq,
                        int causal, float scale, const unsigned char* key_mask, long ld_mask, void* stream);
int lhrs_attn_delta(const void* o, long ldo, const void* dout, long ld_do, float* delta, const int* desc, int nseq, int H,
                    int D, int max_q, int LTq, void* stream);
int lhrs_attn_bwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* dout,
                  long ld_do, const float* lse, const float* delta, void* dq, long ld_dq, void* dk, long ld_dk, void* dv,
                  long ld_dv, const int* desc, int nseq, int H, int D, int max_q, int max_kv, int LTq, int causal,
                  float scale, void* stream);
/* ... of ROTATED q / k (HF LlamaAttention.forward: apply_rotary_pos_emb precedes the scores; text_modal.py:281-292): dq / dk leave the
 * kernel as gradients of the un-rotated projections - the inverse rotation (token row m at position m % pos_mod + pos0; fp32 cos / sin
 * tables [pos][D / 2]) rides in the stores.  Bit-identical to lhrs_attn_bwd + lhrs_rope(inverse) over rows [0, rows) of dq and dk. */
int lhrs_attn_bwd_rope(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* dout,
                       long ld_do, const float* lse, const float* delta, void* dq, long ld_dq, void* dk, long ld_dk, void* dv,
                       long ld_dv, const int* desc, int nseq, int H, int D, int max_q, int max_kv, int LTq, int causal,
                       float scale, const float* cos_t, const float* sin_t, int pos_mod, int pos0, long rows, void* stream);
/* lhrs_attn_delta + lhrs_attn_bwd[_rope] in one call (same reference lines): o = forward output rows, delta is WRITTEN.  With LDS-resident
 * operands (the training path) the dQ kernel computes delta = rowsum(dO * O) itself - one launch and one pass over O / dO less per layer;
 * longer sequences launch the delta kernel internally.  cos_t == NULL: no rotation. */
int lhrs_attn_bwd_o(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* dout, long ld_do,
                    const void* o, long ldo, const float* lse, float* delta, void* dq, long ld_dq, void* dk, long ld_dk, void* dv,
                    long ld_dv, const int* desc, int nseq, int H, int D, int max_q, int max_kv, int LTq, int causal, float scale,
                    const float* cos_t, const float* sin_t, int pos_mod, int pos0, long rows, void* stream);

/* ---- element-wise / layout -------------------------------------------------------------------------- *
 * patchify/assemble: HF CLIPVisionEmbeddings (rgb_vision_modal.py:166-172); rope: HF apply_rotary_pos_emb;
 * swiglu: HF LlamaMLP; map: nn.GELU of common_arch.py:286-292 (+ its derivative), residual add.          */
int lhrs_patchify(const float* rgb, void* out, int B, int img, int P, int KP, void* stream);
int lhrs_vit_assemble(const void* patch, const void* cls, const void* pos, void* out, int B, int NP, int dim,
                      void* stream);
int lhrs_rope(void* x, long ld, int rows, int nheads, int D, const float* cos_t, const float* sin_t, const int* pos_ids,
              int pos_mod, int pos0, int inverse, void* stream);
int lhrs_swiglu_fwd(const void* gate_up, void* act, long rows, int F, void* stream);
int lhrs_swiglu_bwd(const void* dact, const void* gate_up, void* dgate_up, long rows, int F, void* stream);
int lhrs_map(int op, const void* a, const void* b, void* out, long n, void* stream);
int lhrs_colsum_nsplit(int rows); /* host helper */
int lhrs_colsum(const void* x, long ld, float* out, float* partial, int rows, int cols, int accumulate, void* stream);
int lhrs_cast_f32_to_bf16(const float* in, void* out, long n, void* stream);
int lhrs_cast_bf16_to_f32(const void* in, float* out, long n, void* stream);
int lhrs_transpose(const void* in, long ld_in, void* out, long ld_out, int rows, int cols, int rows_pad, void* stream);
/* n transposes in one launch; desc = device int64 [n][7] {src, dst, ld_in, ld_out, rows, cols, first_tile} (64x64 tiles, first_tile ascending) */
int lhrs_transpose_batched(const long* desc, int n, int total_tiles, void* stream);

/* ---- AttnPooler layout (lhrs/models/common_arch.py:134-173: query expand, split, cat(sub_token, sub_image)) ---- */
int lhrs_pooler_build(const void* query, const void* img, void* t, void* kv, int B, int nq0, int nq1, int nq2, int ni0,
                      int ni1, int ni2, int dim, void* stream);
int lhrs_pooler_query_grad(const void* dt0, const void* dkv, float* dquery, int B, int nq0, int nq1, int nq2, int ni0,
                           int ni1, int ni2, int dim, int accumulate, void* stream);
int lhrs_copy_2d(void* dst, long dst_pitch_bytes, const void* src, long src_pitch_bytes, long width_bytes, long height,
                 void* stream);
/* test aid (no reference counterpart): fills the LDS of every CU with `pattern`, so that a kernel which reads LDS it never wrote shows up
 * as a changed result (tools/poison_check.py, tests/test_kernels_gpu.py).  0xFFFFFFFF is a NaN as fp32 and as two bf16. */
int lhrs_debug_poison_lds(unsigned pattern, void* stream);

/* ---- token side ------------------------------------------------------------------------------------- *
 * splice: TextModal.prepare_inputs_for_multimodal (lhrs/models/text_modal.py:296-526); bit-exact.
 * cross_entropy: shifted CE (ignore -100) of HF LlamaForCausalLM.forward, mean over valid targets.       */
int lhrs_splice_fwd(const long* ids, const long* labels, const uint8_t* mask, const void* image, const void* embed,
                    void* out_embeds, long* out_labels, uint8_t* out_mask, int* img_pos, int B, int T, int NI, int dim,
                    int S, int vocab, void* stream);
int lhrs_splice_bwd(const void* d_embeds, const int* img_pos, void* d_image, int B, int NI, int dim, int S, void* stream);
/* general splice - several <image> placeholders per sample (text_modal.py:341-438; the walk over the placeholders and the batch-wide image slot
 * counter run on the host, TextModal.splice_plan_host): per output row the token index it copies (src_tok >= 0) or the row of the flattened image
 * slots [n_slots * NI] (src_img >= 0), neither = zero padding; backward: d_image[r] = d_embeds[inv[r]] (inv[r] < 0: zeros).  Pure copies. */
int lhrs_splice_map_fwd(const long* ids, const int* src_tok, const int* src_img, const void* image, const void* embed, void* out_embeds, int B,
                        int T, int dim, int S, int vocab, void* stream);
int lhrs_splice_map_bwd(const void* d_embeds, const int* inv, void* d_image, int n_rows, int dim, void* stream);
int lhrs_gather_rows(const void* src, long ld_src, const int* idx, void* dst, long ld_dst, int n, int dim, void* stream);
int lhrs_scatter_rows(const void* src, long ld_src, const int* idx, void* dst, long ld_dst, int n, int dim, void* stream);
/* greedy decoding pick (HF GenerationMixin argmax, do_sample=False: main_vqa.py:205-214, cli_qa.py:176-186) */
int lhrs_argmax_rows(const float* x, long ld, long* out, int n, int V, void* stream);
int lhrs_cross_entropy(const void* logits, long ld, const int* target, float* row_loss, float* loss_out, void* dlogits,
                       long ld_d, int n, int V, void* stream);

/* ---- single-token decode (TextModal.generate with use_cache, lhrs/models/text_modal.py:586-627; cli_qa.py:176-186) ---- *
 * gemv: y[B,N] = x[B,K] . W[N,K]^T (+ residual), B <= 8 - HBM-bound weight streaming.  decode_advance / kv_append keep the
 * context length on the device so that one captured hipGraph (lhrs_graph_*) replays for every generated token.        */
int lhrs_gemv_bf16(const void* W, long ldw, const void* x, long ldx, const void* residual, long ldr, void* y, long ldy, int B,
                   int N, int K, int out_f32, void* stream);
/* w_format: 0 = bf16 rows [N, ldw]; 1 = e4m3 rows + per-row wscale; 2 = bf16 re-tiled by lhrs_repack_bf16_mfma into the MFMA operand
 * order [N/16][K/32][64 lanes][8] (batch >= 2, K % 128 == 0; ldw unused) - the batched-evaluation weight stream as consecutive 1-KiB lines */
int lhrs_gemv(const void* W, long ldw, const float* wscale, int w_format, const void* x, long ldx, int prologue, const void* norm_w,
              float eps, const void* residual, long ldr, void* y, long ldy, int B, int N, int K, int out_f32, void* stream);
int lhrs_repack_bf16_mfma(const void* W, long ldw, void* out, int N, int K, void* stream);
/* kernel A/B tests only: rows per wave / 1-KiB chunks per iteration of the batch-1 bf16 GEMV (0, 0 = the built-in shape rule) */
int lhrs_gemv_set_tuning(int rows_per_wave, int chunks_per_iteration);
int lhrs_quant_fp8_rows(const void* W, long ldw, void* W8, long ld8, float* scale, int N, int K, void* stream);
int lhrs_rope_kv_append(void* qkv, long ld, void* kcache, void* vcache, const float* cos_t, const float* sin_t, const int* pos, int B,
                        int H, int D, int max_ctx, void* stream);
int lhrs_decode_advance(int* state, int* desc, int* pos, int B, int max_ctx, int step_inc, void* stream);
/* the same, and cs [B][128] floats = the cos | sin rows (head_dim 128) of the new position, for lhrs_decode_attn_split */
int lhrs_decode_advance_cs(int* state, int* desc, int* pos, const float* cos_t, const float* sin_t, float* cs, int B, int max_ctx,
                           int step_inc, void* stream);
int lhrs_kv_append(const void* qkv, long ld, void* kcache, void* vcache, const int* pos, int B, int d, int max_ctx, void* stream);
int lhrs_decode_emit(const long* next_ids, int* tok32, long* out_ids, int* state, int B, int max_new, void* stream);
int lhrs_graph_begin(void* stream);
int lhrs_graph_end(void* stream, void** exec_out);
int lhrs_graph_launch(void* exec, void* stream);
int lhrs_graph_destroy(void* exec);

/* ---- optimizer -------------------------------------------------------------------------------------- *
 * Adan(no_prox) = timm "adanp" built at lhrs/optimizer/build_optimizer.py:76-86; AdamW + global-norm clip =
 * DeepSpeed engine.step() configured at main_pretrain_stage1.py:28-85 and driven by
 * lhrs/CustomTrainer/hook/deepspeed_hook.py:4-19.                                                         */
int lhrs_sqnorm_nblk(long n); /* host helper */
int lhrs_sqnorm(const float* g, long n, float* partial, float* out, int accumulate, void* stream);
/* gradient accumulation (DeepSpeed "gradient_accumulation_steps", main_pretrain_stage1.py:61,115): y = x or y += x, fp32, n % 4 == 0 */
int lhrs_accum_f32(float* y, const float* x, long n, int copy_only, void* stream);
int lhrs_adan_step(float* param, const float* grad, float* exp_avg, float* exp_avg_diff, float* exp_avg_sq,
                   float* pre_grad, void* shadow_bf16, long n, int step, float lr, float beta1, float beta2, float beta3,
                   float eps, float weight_decay, int no_prox, const float* gnorm_sq, float max_norm, float grad_scale,
                   void* stream);
int lhrs_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16, long n,
                    int step, float lr, float beta1, float beta2, float eps, float weight_decay, const float* gnorm_sq,
                    float max_norm, float grad_scale, void* stream);

/* decode linear with e4m3 weights AND e4m3 activations on the block-scaled MFMA (BASELINE configs[4] "fp8 MFMA weights"):
 * y[B, N] = xscale[b] * wscale[n] * (x8 . W8^T) (+ residual); B <= 16, K % 128 == 0; x8 / xscale from lhrs_quant_fp8_rows,
 * lhrs_rmsnorm_fwd_q or lhrs_swiglu_fwd_q */
int lhrs_gemv_fp8_mfma(const void* W8, long ldw, const float* wscale, const void* x8, long ldx, const float* xscale,
                       const void* residual, long ldr, void* y, long ldy, int B, int N, int K, int out_f32, int w_packed, void* stream);
/* the same with bf16 activations and B <= 2: prologue (0 none, 1 RMSNorm, 2 SwiGLU over x = [B, 2K]) and the per-row e4m3 quantisation of
 * x happen inside the kernel (five launches per layer for a batch-1 token) */
int lhrs_gemv_fp8_mfma_fused(const void* W8, long ldw, const float* wscale, const void* x, long ldx, int prologue, const void* norm_w,
                             float eps, const void* residual, long ldr, void* y, long ldy, int B, int N, int K, int out_f32,
                             int w_packed, void* stream);
/* w_packed != 0 in the two calls above: W8 is the copy this makes - the e4m3 rows re-tiled into the MFMA operand order
 * [N/16][K/128][2][64 lanes][16 B] (ceil(N/16)*16*K bytes), so that the decode weight stream is read as consecutive 1-KiB lines */
int lhrs_repack_fp8_mfma(const void* W8, long ldw, void* out, int N, int K, void* stream);
/* single-token attention for generate() (HF LlamaAttention with a KV cache at q_len = 1, reached from TextModal.generate,
 * lhrs/models/text_modal.py:586-627): RoPE of the new q / k row at device-resident position pos[b], append of (k, v) to the caches
 * [B * max_ctx, H*128] and attention over keys 0..pos[b] (optionally AND an HF attention_mask byte row), in ONE launch. */
int lhrs_decode_attn(const void* qkv, long ld, void* kcache, void* vcache, const float* cos_t, const float* sin_t,
                     const int* pos, const unsigned char* key_mask, long ld_mask, void* out, long ldo, int B, int H, int D,
                     int max_ctx, float scale, void* stream);
/* the same attention with the context split over nsplit (1..16) workgroups per head - slices of 128 keys, partial (max, sum, o[128])
 * exchanged through `part` (fp32 [B][H][nsplit][132]) behind a ticket per head in `tickets` (int32 [B][H]; zero before the first call, left
 * zero by every call) - so that a long context streams through 32 * nsplit CUs instead of 32.  Same result up to the order of the fp32
 * softmax sums.  part / tickets: caller-owned, private to the stream the calls are ordered on.  cs (NULL allowed): the cos | sin rows
 * lhrs_decode_advance_cs wrote for pos[b]; with it the rotation of the new q / k does not wait for pos[b] to index the tables. */
int lhrs_decode_attn_split(const void* qkv, long ld, void* kcache, void* vcache, const float* cos_t, const float* sin_t,
                           const int* pos, const unsigned char* key_mask, long ld_mask, void* out, long ldo, int B, int H, int D,
                           int max_ctx, float scale, int nsplit, float* part, int* tickets, const float* cs, void* stream);
/* ---- data boundary: the image transform of the reference -------------------------------------------
 * CLIPImageProcessor.preprocess as built by build_vlp_transform (lhrs/Dataset/build_transform.py:43-45) for one decoded RGB image:
 * img = uint8 [H][W][3] on the device (row_stride bytes per row) -> out = float32 [3][224][224].  Bit-exact with Pillow's BICUBIC
 * resize (short edge -> 224), center crop, /255, CLIP mean/std.  workspace: lhrs_clip_preprocess_workspace(H, W) bytes, caller-owned. */
long lhrs_clip_preprocess_workspace(int H, int W);
int lhrs_clip_preprocess(const unsigned char* img, int H, int W, long row_stride, float* out, void* workspace,
                         long workspace_bytes, void* stream);
/* General form of the same three kernels: Pillow BICUBIC resize of the short edge to `short_edge` (>= 224), 224x224 centre crop whose
 * offset is floor((n - 224) / 2) (crop_round 0: HF `center_crop`) or Python round((n - 224) / 2.0) (crop_round 1: torchvision
 * `CenterCrop`), byte -> float32 as float32(float64(b) / 255) (rescale_mode 0: HF) or float32(b) / 255.f (rescale_mode 1: torchvision
 * `ToTensor`), then (x - mean[c]) / std[c] in float32.  mean / std: HOST pointers to 3 floats.  Replaces the evaluation transform of the
 * classification caller - lhrs/Dataset/build_transform.py:27-40 `build_cls_transform(is_train=False)` = Resize(256) -> CenterCrop(224) ->
 * ToTensor -> Normalize(ImageNet), reached from lhrs/Dataset/build_loader.py:164-199 `build_zero_shot_loader` (main_cls.py:133) - with
 * short_edge 256, crop_round 1, rescale_mode 1. */
long lhrs_image_preprocess_workspace(int H, int W, int short_edge);
int lhrs_image_preprocess(const unsigned char* img, int H, int W, long row_stride, float* out, void* workspace, long workspace_bytes,
                          int short_edge, int crop_round, int rescale_mode, const float* mean, const float* stdv, void* stream);

/* ---- module-level entry points: one call = one HF LlamaDecoderLayer ----------------------------------------------
 * lhrs_llama_layer_forward : x -> x + o_proj(attn(RoPE(q_proj(n1)), RoPE(k_proj(n1)), v_proj(n1))) =: x_mid -> x_mid + down(silu(gate(n2)) * up(n2)),
 *   n = RMSNorm; replaces the decoder layer inside HF LlamaForCausalLM.forward (lhrs/models/text_modal.py:281-292) for the frozen bf16
 *   base without adapters.  x, x_mid, x_out, o, h: [B*S, d] bf16; qkv [B*S, 3d] (q, k rotated); gu [B*S, 2ff]; act [B*S, ff]; lse f32
 *   [B, heads, LT]; desc int32 [B][8] attention records; LT = S rounded up to 64.  h and act are scratch, the rest is what the backward reads.
 * lhrs_llama_layer_backward: d loss / d x_out -> d loss / d x (activation gradients only: the weights are frozen), what engine.backward
 *   (lhrs/CustomTrainer/hook/deepspeed_hook.py:6-9) does inside one decoder layer.  *_wT = transposed weight copies; gu is overwritten
 *   with d(gate|up); scratch dh, d_o [B*S, d], dqkv [B*S, 3d], delta f32 [B, heads, LT], dact [B*S, ff] (NULL allowed when
 *   lhrs_gemm_swiglu_fusable() == 1); dx_in [B*S, d].
 * Both compose the operator entry points above in the order lhrs_bot_amd/text.py uses (bit-identical results); caller-owned buffers. */
int lhrs_llama_layer_forward(const void* x, const void* ln1_w, const void* qkv_w, const void* o_w, const void* ln2_w, const void* gu_w,
                             const void* down_w, const float* cos_t, const float* sin_t, const int* desc, int B, int S, int LT, int d,
                             int heads, int ff, float eps, void* h, void* qkv, void* o, float* lse, void* x_mid, void* gu,
                             void* act, void* x_out, void* stream);
/* one pre-LN encoder layer of the frozen CLIP ViT IN PLACE on x [B * n, d] (HF CLIPEncoderLayer, reached from VisionModal.encode,
 * lhrs/models/rgb_vision_modal.py:166-179): LN -> qkv (+bias) -> attention (no mask) -> out_proj (+bias, +x) -> LN -> fc1 (+bias, quick_gelu)
 * -> fc2 (+bias, +x).  Scratch: h, o [B*n, d], qkv [B*n, 3d], f [B*n, ff]; desc int32 [B][8]; LT = n rounded up to 64. */
int lhrs_vit_layer_forward(void* x, const void* ln1_w, const void* ln1_b, const void* qkv_w, const void* qkv_b, const void* o_w,
                           const void* o_b, const void* ln2_w, const void* ln2_b, const void* fc1_w, const void* fc1_b, const void* fc2_w,
                           const void* fc2_b, const int* desc, int B, int n, int LT, int d, int heads, int ff, void* h, void* qkv, void* o,
                           void* f, void* stream);
int lhrs_llama_layer_backward(const void* dx_out, const void* x_in, const void* x_mid, const void* qkv, const void* o, const float* lse,
                              void* gu, const void* ln1_w, const void* ln2_w, const void* qkv_wT, const void* o_wT, const void* gu_wT,
                              const void* down_wT, const float* cos_t, const float* sin_t, const int* desc, int B, int S, int LT, int d,
                              int heads, int ff, float eps, void* dh, void* d_o, void* dqkv, float* delta, void* dact, void* dx_in,
                              void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LHRS_HIP_H */
