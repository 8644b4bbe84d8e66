// Optimizer step for the trainable projector, fused on device (gfx950, HBM-bound, fp32 master state).
//   Adan   : timm 0.9.12 `Adan(no_prox=True)` == create_optimizer_v2(opt="adanp"), reached from
//            /root/reference lhrs/optimizer/build_optimizer.py:76-86 (stage 1: lr 2e-4, wd 0, betas .98/.92/.99).
//   AdamW  : DeepSpeed FusedAdam(adam_w_mode) configured at main_pretrain_stage1.py:30-39 (stage 2/3).
//   clip   : DeepSpeed global-norm clipping, clip_coef = max_norm / (norm + 1e-6) applied when < 1
//            (`gradient_clipping`, main_pretrain_stage1.py:28-85); the squared norm stays on device.
// Neither timm nor deepspeed is importable in the build container: parity for this file is pinned against an
// independent restatement of the published update rules (oracle/optim_oracle.py) - "parity unpinned" w.r.t. the
// reference's own binaries.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) { const float v = g[i]; s += v * v; }
  s = block_sum<4>(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ out,
                                                           int accumulate) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256) s += partial[i];
  s = block_sum<4>(s, red);
  if (threadIdx.x == 0) *out = accumulate ? *out + s : s;
}

__device__ __forceinline__ float clip_coef(const float* gnorm_sq, float max_norm, float grad_scale) {
  if (!gnorm_sq || max_norm <= 0.f) return grad_scale;
  const float norm = sqrtf(*gnorm_sq) * grad_scale;
  const float c = max_norm / (norm + 1e-6f);
  return (c < 1.f ? c : 1.f) * grad_scale;
}

__global__ __launch_bounds__(256) void adan_kernel(float* __restrict__ p, const float* __restrict__ g_in, float* __restrict__ exp_avg,
                                                   float* __restrict__ exp_avg_diff, float* __restrict__ exp_avg_sq,
                                                   float* __restrict__ pre_grad, bf16_t* __restrict__ shadow, long n, float lr,
                                                   float b1, float b2, float b3, float eps, float wd, float bc1, float bc2,
                                                   float bc3_sqrt, int first_step, int no_prox, const float* gnorm_sq,
                                                   float max_norm, float grad_scale) {
  const float coef = clip_coef(gnorm_sq, max_norm, grad_scale);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
    const float g = g_in[i] * coef;
    const float pg = first_step ? g : pre_grad[i];
    const float diff = g - pg;
    const float m = exp_avg[i] + (g - exp_avg[i]) * (1.f - b1);            // lerp_(grad, 1-b1)
    const float v = exp_avg_diff[i] + (diff - exp_avg_diff[i]) * (1.f - b2);
    const float u = g + b2 * diff;
    const float nn = exp_avg_sq[i] * b3 + u * u * (1.f - b3);
    const float denom = sqrtf(nn) / bc3_sqrt + eps;
    const float upd = (m / bc1 + b2 * v / bc2) / denom;
    float w = p[i];
    if (no_prox) { w = w * (1.f - lr * wd) - lr * upd; }
    else { w = (w - lr * upd) / (1.f + lr * wd); }
    p[i] = w; exp_avg[i] = m; exp_avg_diff[i] = v; exp_avg_sq[i] = nn; pre_grad[i] = g;
    if (shadow) shadow[i] = f2bf(w);
  }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g_in, float* __restrict__ exp_avg,
                                                    float* __restrict__ exp_avg_sq, bf16_t* __restrict__ shadow, long n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2,
                                                    const float* gnorm_sq, float max_norm, float grad_scale) {
  const float coef = clip_coef(gnorm_sq, max_norm, grad_scale);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
    const float g = g_in[i] * coef;
    const float m = b1 * exp_avg[i] + (1.f - b1) * g;
    const float v = b2 * exp_avg_sq[i] + (1.f - b2) * g * g;
    const float denom = sqrtf(v) / sqrtf(bc2) + eps;
    float w = p[i];
    w = w * (1.f - lr * wd) - (lr / bc1) * (m / denom);
    p[i] = w; exp_avg[i] = m; exp_avg_sq[i] = v;
    if (shadow) shadow[i] = f2bf(w);
  }
}

inline int ogrid(long n) { long g = (n + 255) / 256; return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }

// gradient accumulation over micro-batches (DeepSpeed gradient_accumulation_steps): y = x (copy) or y += x, fp32, 16 B per lane
__global__ void accum_f32_kernel(float* __restrict__ y, const float* __restrict__ x, long n4, int copy_only) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(x)[i];
    if (copy_only) { reinterpret_cast<float4*>(y)[i] = a; continue; }
    float4 b = reinterpret_cast<float4*>(y)[i];
    b.x += a.x; b.y += a.y; b.z += a.z; b.w += a.w;
    reinterpret_cast<float4*>(y)[i] = b;
  }
}

}  // namespace

extern "C" int lhrs_sqnorm_nblk(long n) { return ogrid(n); }

// y = x (copy_only) or y += x over n fp32 elements (n % 4 == 0): the accumulation buffer of gradient_accumulation_steps > 1
extern "C" int lhrs_accum_f32(float* y, const float* x, long n, int copy_only, void* stream) {
  LHRS_REQUIRE(n > 0 && n % 4 == 0 && y && x, "accum_f32: n=%ld", n);
  hipLaunchKernelGGL(accum_f32_kernel, dim3(ogrid(n / 4)), dim3(256), 0, (hipStream_t)stream, y, x, n / 4, copy_only);
  LHRS_CHECK_LAUNCH("accum_f32");
  return 0;
}

// out (+)= sum(g[i]^2); partial: lhrs_sqnorm_nblk(n) floats of workspace
extern "C" int lhrs_sqnorm(const float* g, long n, float* partial, float* out, int accumulate, void* stream) {
  LHRS_REQUIRE(n > 0 && partial && out, "sqnorm: n=%ld", n);
  const int nb = ogrid(n);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nb), dim3(256), 0, s, g, n, partial);
  LHRS_CHECK_LAUNCH("sqnorm_partial");
  hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, s, partial, nb, out, accumulate);
  LHRS_CHECK_LAUNCH("sqnorm_final");
  return 0;
}

extern "C" int lhrs_adan_step(float* param, const float* grad, float* exp_avg, float* exp_avg_diff, float* exp_avg_sq,
                              float* pre_grad, void* shadow_bf16, long n, int step, float lr, float beta1, float beta2,
                              float beta3, float eps, float weight_decay, int no_prox, const float* gnorm_sq,
                              float max_norm, float grad_scale, void* stream) {
  LHRS_REQUIRE(n > 0 && step >= 1, "adan_step: n=%ld step=%d", n, step);
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  const float bc3s = sqrtf(1.f - powf(beta3, (float)step));
  hipLaunchKernelGGL(adan_kernel, dim3(ogrid(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_diff,
                     exp_avg_sq, pre_grad, (bf16_t*)shadow_bf16, n, lr, beta1, beta2, beta3, eps, weight_decay, bc1, bc2, bc3s,
                     step == 1 ? 1 : 0, no_prox, gnorm_sq, max_norm, grad_scale);
  LHRS_CHECK_LAUNCH("adan_step");
  return 0;
}

extern "C" int lhrs_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                               long n, int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                               const float* gnorm_sq, float max_norm, float grad_scale, void* stream) {
  LHRS_REQUIRE(n > 0 && step >= 1, "adamw_step: n=%ld step=%d", n, step);
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(ogrid(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                     (bf16_t*)shadow_bf16, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, gnorm_sq, max_norm, grad_scale);
  LHRS_CHECK_LAUNCH("adamw_step");
  return 0;
}
