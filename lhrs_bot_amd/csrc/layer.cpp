// Module-level entry points of the C ABI (include/lhrs_hip.h): ONE call runs a whole LLaMA decoder layer forward or
// its activation-gradient backward - the seven / seven operator launches that lhrs_bot_amd/text.py issues one by one for the reference's
// HF LlamaDecoderLayer (reached from TextModal.decode, /root/reference lhrs/models/text_modal.py:258-294, and from DeepSpeedHook.after_iter's
// engine.backward, lhrs/CustomTrainer/hook/deepspeed_hook.py:6-9).  Host code only: it composes the operator entry points of this same
// library in the order the Python path uses, so the two paths are bit-identical; the caller owns every buffer.
// Covered: the frozen bf16 base with no adapters (stage 1: every layer but the compact last one).  LoRA / 8-bit bases keep the operator path.
#include "../../include/lhrs_hip.h"
#include <math.h>
#include <stdio.h>

extern "C" void lhrs_set_error(const char* msg);
#define LAYER_REQUIRE(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      char buf_[400];                            \
      snprintf(buf_, sizeof(buf_), __VA_ARGS__); \
      lhrs_set_error(buf_);                      \
      return -1;                                 \
    }                                            \
  } while (0)
#define TRY(call)          \
  do {                     \
    if ((call) != 0) return -1; \
  } while (0)

// x [M = B * S, d] bf16 -> x_out [M, d]; saved for the backward: qkv [M, 3d] (q, k rotated), o [M, d], lse [B, heads, LT] f32, x_mid [M, d],
// gu [M, 2 ff].  h [M, d] and act [M, ff] are scratch.  desc: int32 [B][8] attention records (lhrs_attn_fwd).  LT = S rounded up to 64.
extern "C" int lhrs_llama_layer_forward(const void* x, const void* ln1_w, const void* qkv_w, const void* o_w, const void* ln2_w, const void* gu_w,
                                        const void* down_w, const float* cos_t, const float* sin_t, const int* desc, int B, int S, int LT, int d,
                                        int heads, int ff, float eps, void* h, void* qkv, void* o, float* lse, void* x_mid, void* gu,
                                        void* act, void* x_out, void* stream) {
  LAYER_REQUIRE(B > 0 && S > 0 && d > 0 && heads > 0 && d % heads == 0 && ff > 0 && LT >= S, "llama_layer_forward: B=%d S=%d d=%d heads=%d ff=%d LT=%d",
                B, S, d, heads, ff, LT);
  LAYER_REQUIRE(x && ln1_w && qkv_w && o_w && ln2_w && gu_w && down_w && cos_t && sin_t && desc && h && qkv && o && lse && x_mid && gu && act && x_out,
                "llama_layer_forward: null buffer");
  const int M = B * S, hd = d / heads;
  const char* q = (const char*)qkv;
  TRY(lhrs_rmsnorm_fwd(x, d, ln1_w, h, d, nullptr, M, d, eps, stream));
  TRY(lhrs_gemm_rope_fwd(h, d, qkv_w, d, nullptr, 0, nullptr, 0, 0, qkv, 3 * d, M, 3 * d, d, cos_t, sin_t, S, 0, 2 * d, hd, stream));
  TRY(lhrs_attn_fwd(q, 3 * d, q + (long)d * 2, 3 * d, q + (long)2 * d * 2, 3 * d, o, d, lse, desc, B, heads, hd, S, S, LT, 1, 1.0f / sqrtf((float)hd), stream));
  TRY(lhrs_gemm_bf16_nt(o, d, o_w, d, x_mid, d, M, d, d, nullptr, x, d, 0, 0, 0, 1.0f, stream));
  TRY(lhrs_rmsnorm_fwd(x_mid, d, ln2_w, h, d, nullptr, M, d, eps, stream));
  TRY(lhrs_gemm_swiglu_fwd(h, d, gu_w, d, nullptr, 0, nullptr, 0, 0, gu, 2 * ff, act, ff, M, ff, d, stream));
  TRY(lhrs_gemm_bf16_nt(act, ff, down_w, ff, x_out, d, M, d, ff, nullptr, x_mid, d, 0, 0, 0, 1.0f, stream));
  return 0;
}

// dx_out = d loss / d x_out [M, d] -> dx_in = d loss / d x [M, d] through the frozen layer (weights get no gradient: activation gradients only).
// *_wT: the transposed weight copies ([in, out] of the forward weight, i.e. the dX products are NT GEMMs as well).  gu is OVERWRITTEN with
// d(gate|up).  Scratch: dh [M, d], dh1 [M, d], d_o [M, d], dqkv [M, 3d], delta [B, heads, LT] f32, dact [M, ff] (may be NULL when
// lhrs_gemm_swiglu_fusable says the fused kernel runs).
extern "C" int lhrs_llama_layer_backward(const void* dx_out, const void* x_in, const void* x_mid, const void* qkv, const void* o, const float* lse,
                                         void* gu, const void* ln1_w, const void* ln2_w, const void* qkv_wT, const void* o_wT, const void* gu_wT,
                                         const void* down_wT, const float* cos_t, const float* sin_t, const int* desc, int B, int S, int LT, int d,
                                         int heads, int ff, float eps, void* dh, void* d_o, void* dqkv, float* delta, void* dact, void* dx_in,
                                         void* stream) {
  LAYER_REQUIRE(B > 0 && S > 0 && d > 0 && heads > 0 && d % heads == 0 && ff > 0 && LT >= S, "llama_layer_backward: B=%d S=%d d=%d heads=%d ff=%d LT=%d",
                B, S, d, heads, ff, LT);
  LAYER_REQUIRE(dx_out && x_in && x_mid && qkv && o && lse && gu && ln1_w && ln2_w && qkv_wT && o_wT && gu_wT && down_wT && cos_t && sin_t && desc &&
                dh && d_o && dqkv && delta && dx_in, "llama_layer_backward: null buffer");
  const int M = B * S, hd = d / heads;
  const char* q = (const char*)qkv;
  char* dq = (char*)dqkv;
  // MLP: d(gate|up) = swiglu'(gu) * (dx_out . W_down) over gu;  dh = d(gate|up) . W_gu;  dx_mid = RMSNorm'(dh; x_mid) + dx_out  (over dh)
  TRY(lhrs_gemm_swiglu_bwd(dx_out, d, down_wT, d, nullptr, 0, nullptr, 0, 0, gu, gu, 2 * ff, dact, M, ff, d, stream));
  TRY(lhrs_gemm_bf16_nt(gu, 2 * ff, gu_wT, 2 * ff, dh, d, M, d, 2 * ff, nullptr, nullptr, 0, 0, 0, 0, 1.0f, stream));
  TRY(lhrs_rmsnorm_bwd(dh, x_mid, ln2_w, nullptr, dx_out, dh, M, d, eps, stream));
  // attention: d_o = dx_mid . W_o;  (dq, dk, dv) with delta = rowsum(d_o * o) and the inverse RoPE inside;  dh1 = dqkv . W_qkv;
  // dx_in = RMSNorm'(dh1; x_in) + dx_mid
  TRY(lhrs_gemm_bf16_nt(dh, d, o_wT, d, d_o, d, M, d, d, nullptr, nullptr, 0, 0, 0, 0, 1.0f, stream));
  TRY(lhrs_attn_bwd_o(q, 3 * d, q + (long)d * 2, 3 * d, q + (long)2 * d * 2, 3 * d, d_o, d, o, d, lse, delta, dq, 3 * d, dq + (long)d * 2, 3 * d,
                      dq + (long)2 * d * 2, 3 * d, desc, B, heads, hd, S, S, LT, 1, 1.0f / sqrtf((float)hd), cos_t, sin_t, S, 0, M, stream));
  TRY(lhrs_gemm_bf16_nt(dqkv, 3 * d, qkv_wT, 3 * d, dx_in, d, M, d, 3 * d, nullptr, nullptr, 0, 0, 0, 0, 1.0f, stream));
  TRY(lhrs_rmsnorm_bwd(dx_in, x_in, ln1_w, nullptr, dh, dx_in, M, d, eps, stream));
  return 0;
}

// One pre-LN encoder layer of the frozen CLIP ViT (HF CLIPEncoderLayer inside CLIPVisionModel, reached from VisionModal.encode,
// /root/reference lhrs/models/rgb_vision_modal.py:166-179): x <- x + out_proj(attn(q,k,v of LN1(x))), x <- x + fc2(quick_gelu(fc1(LN2(x)))),
// IN PLACE on x [B * n, d] (n = 257 tokens per image).  Scratch: h [B*n, d], qkv [B*n, 3d], o [B*n, d], f [B*n, ff].  No mask, LN eps 1e-5.
extern "C" int lhrs_vit_layer_forward(void* x, const void* ln1_w, const void* ln1_b, const void* qkv_w, const void* qkv_b, const void* o_w,
                                      const void* o_b, const void* ln2_w, const void* ln2_b, const void* fc1_w, const void* fc1_b, const void* fc2_w,
                                      const void* fc2_b, const int* desc, int B, int n, int LT, int d, int heads, int ff, void* h, void* qkv, void* o,
                                      void* f, void* stream) {
  LAYER_REQUIRE(B > 0 && n > 0 && d > 0 && heads > 0 && d % heads == 0 && ff > 0 && LT >= n, "vit_layer_forward: B=%d n=%d d=%d heads=%d ff=%d LT=%d", B, n, d,
                heads, ff, LT);
  LAYER_REQUIRE(x && ln1_w && ln1_b && qkv_w && qkv_b && o_w && o_b && ln2_w && ln2_b && fc1_w && fc1_b && fc2_w && fc2_b && desc && h && qkv && o && f,
                "vit_layer_forward: null buffer");
  const int M = B * n, hd = d / heads;
  const char* q = (const char*)qkv;
  TRY(lhrs_layernorm_fwd(x, d, ln1_w, ln1_b, h, d, nullptr, nullptr, M, d, 1e-5f, stream));
  TRY(lhrs_gemm_bf16_nt(h, d, qkv_w, d, qkv, 3 * d, M, 3 * d, d, qkv_b, nullptr, 0, 0, 0, 0, 1.0f, stream));
  TRY(lhrs_attn_fwd(q, 3 * d, q + (long)d * 2, 3 * d, q + (long)2 * d * 2, 3 * d, o, d, nullptr, desc, B, heads, hd, n, n, LT, 0, 1.0f / sqrtf((float)hd), stream));
  TRY(lhrs_gemm_bf16_nt(o, d, o_w, d, x, d, M, d, d, o_b, x, d, 0, 0, 0, 1.0f, stream));
  TRY(lhrs_layernorm_fwd(x, d, ln2_w, ln2_b, h, d, nullptr, nullptr, M, d, 1e-5f, stream));
  TRY(lhrs_gemm_bf16_nt(h, d, fc1_w, d, f, ff, M, ff, d, fc1_b, nullptr, 0, 1 /* quick_gelu */, 0, 0, 1.0f, stream));
  TRY(lhrs_gemm_bf16_nt(f, ff, fc2_w, ff, x, d, M, d, ff, fc2_b, x, d, 0, 0, 0, 1.0f, stream));
  return 0;
}
