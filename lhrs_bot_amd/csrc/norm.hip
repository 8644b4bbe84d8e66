// LayerNorm / RMSNorm forward + backward for gfx950.  HBM-bound row kernels: one 64-lane wavefront owns one
// row, 16-byte (8 x bf16) vector loads, statistics in fp32 by wavefront reduction (no LDS on the forward path).
//
// Reference semantics:
//   LayerNorm  - lhrs/models/common_arch.py:253-259 (F.layer_norm, eps 1e-5, cast back to input dtype) and
//                HF CLIP `pre_layrnorm` / `layer_norm1/2` (same op).
//   RMSNorm    - HF LlamaRMSNorm called from lhrs/models/text_modal.py:281-292: variance in fp32,
//                x * rsqrt(var + eps) rounded to the activation dtype, then multiplied by the weight.
#include "common.h"

namespace {

constexpr int LN_WAVES = 4;  // rows per 256-thread block

template <int NCH>  // cols = NCH * 512
__device__ __forceinline__ void load_row(const bf16_t* x, int lane, float (&v)[NCH][8]) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const uint4 u = *reinterpret_cast<const uint4*>(x + (c * 64 + lane) * 8);
    v[c][0] = bflo(u.x); v[c][1] = bfhi(u.x); v[c][2] = bflo(u.y); v[c][3] = bfhi(u.y);
    v[c][4] = bflo(u.z); v[c][5] = bfhi(u.z); v[c][6] = bflo(u.w); v[c][7] = bfhi(u.w);
  }
}
template <int NCH>
__device__ __forceinline__ void store_row(bf16_t* y, int lane, const float (&v)[NCH][8]) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    uint4 u;
    u.x = pack2bf(v[c][0], v[c][1]); u.y = pack2bf(v[c][2], v[c][3]);
    u.z = pack2bf(v[c][4], v[c][5]); u.w = pack2bf(v[c][6], v[c][7]);
    *reinterpret_cast<uint4*>(y + (c * 64 + lane) * 8) = u;
  }
}

// ---------------------------------------------------------------- LayerNorm forward
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ gamma,
                                                            const bf16_t* __restrict__ beta, bf16_t* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int rows, long ldx, long ldy, float eps) {
  constexpr int cols = NCH * 512;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * LN_WAVES + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[NCH][8], g[NCH][8], b[NCH][8];
  load_row<NCH>(x + row * ldx, lane, v);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[c][i];
  const float mean = wave_sum(s) * (1.f / cols);
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float d = v[c][i] - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) * (1.f / cols) + eps);
  load_row<NCH>(gamma, lane, g);
  load_row<NCH>(beta, lane, b);
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) v[c][i] = (v[c][i] - mean) * rstd * g[c][i] + b[c][i];
  store_row<NCH>(y + row * ldy, lane, v);
  if (mean_out && lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// ---------------------------------------------------------------- LayerNorm backward
// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat));  dgamma = sum_rows dy*xhat;  dbeta = sum_rows dy.
// Each block walks rows blockIdx.x*4+w, +gridDim.x*4, ... and keeps per-lane column partials in registers;
// partials go to `partial[gridDim.x][2][cols]` and are folded by layernorm_bwd_finalize (deterministic order).
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                            const bf16_t* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const bf16_t* add, bf16_t* dx,
                                                            float* __restrict__ partial, int rows, long ld_dy, long ldx,
                                                            long ld_dx) {
  constexpr int cols = NCH * 512;
  __shared__ float red[LN_WAVES][cols];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float g[NCH][8];
  load_row<NCH>(gamma, lane, g);
  float dg[NCH][8], db[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) { dg[c][i] = 0.f; db[c][i] = 0.f; }
  for (int row = blockIdx.x * LN_WAVES + w; row < rows; row += gridDim.x * LN_WAVES) {
    float xv[NCH][8], dv[NCH][8];
    load_row<NCH>(x + row * ldx, lane, xv);
    load_row<NCH>(dy + row * ld_dy, lane, dv);
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xh = (xv[c][i] - mu) * rs;
        const float gd = g[c][i] * dv[c][i];
        s1 += gd; s2 += gd * xh;
        dg[c][i] += dv[c][i] * xh; db[c][i] += dv[c][i];
        xv[c][i] = xh;
      }
    s1 = wave_sum(s1) * (1.f / cols);
    s2 = wave_sum(s2) * (1.f / cols);
    if (dx) {
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) dv[c][i] = rs * (g[c][i] * dv[c][i] - s1 - xv[c][i] * s2);
      if (add) {
        float av[NCH][8];
        load_row<NCH>(add + row * ld_dx, lane, av);
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
          for (int i = 0; i < 8; ++i) dv[c][i] += av[c][i];
      }
      store_row<NCH>(dx + row * ld_dx, lane, dv);
    }
  }
  if (!partial) return;
  // fold the 4 waves, then one partial row per block
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int i = 0; i < 8; ++i) red[w][(c * 64 + lane) * 8 + i] = pass ? db[c][i] : dg[c][i];
    __syncthreads();
    for (int j = threadIdx.x; j < cols; j += 256)
      partial[((long)blockIdx.x * 2 + pass) * cols + j] = red[0][j] + red[1][j] + red[2][j] + red[3][j];
  }
}

// 32 columns x 8 row-slices per block: slice sl folds partial blocks sl, sl+8, ... and the slices are added in a fixed order
__global__ __launch_bounds__(256) void layernorm_bwd_finalize(const float* __restrict__ partial, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int nblk, int cols, int accumulate) {
  __shared__ float red[2][8][32];
  const int c = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + c;
  float a = 0.f, b = 0.f;
  if (j < cols)
    for (int k = sl; k < nblk; k += 8) { a += partial[((long)k * 2) * cols + j]; b += partial[((long)k * 2 + 1) * cols + j]; }
  red[0][sl][c] = a; red[1][sl][c] = b;
  __syncthreads();
  if (sl == 0 && j < cols) {
    a = 0.f; b = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a += red[0][i][c]; b += red[1][i][c]; }
    if (accumulate) { dgamma[j] += a; dbeta[j] += b; } else { dgamma[j] = a; dbeta[j] = b; }
  }
}

// per-row e4m3 copy of a row held in registers (values as they were / would be stored in bf16): scale = max|v| / 448 (SURVEY §8 f-4:
// the operand of the next 8-bit-base GEMM, bit-identical to lhrs_quant_fp8_rows of the bf16 row)
template <int NCH>
__device__ __forceinline__ void store_row_e4m3(unsigned char* y8, float* scale_out, int lane, float (&v)[NCH][8]) {
  float m = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[c][i] = bf2f(f2bf(v[c][i]));
      m = fmaxf(m, fabsf(v[c][i]));
    }
  m = wave_max(m);
  const float sc = m > 0.f ? m / 448.f : 1.f;
  if (lane == 0) *scale_out = sc;
  const float inv = 1.f / sc;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][0] * inv, v[c][1] * inv, 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][2] * inv, v[c][3] * inv, lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][4] * inv, v[c][5] * inv, 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][6] * inv, v[c][7] * inv, hi, true);
    *reinterpret_cast<int2*>(y8 + (c * 64 + lane) * 8) = make_int2(lo, hi);
  }
}

// ---------------------------------------------------------------- RMSNorm
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          bf16_t* __restrict__ y, float* __restrict__ rstd_out, int rows,
                                                          long ldx, long ldy, float eps, unsigned char* __restrict__ y8 = nullptr,
                                                          float* __restrict__ y8scale = nullptr) {
  constexpr int cols = NCH * 512;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * LN_WAVES + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[NCH][8], g[NCH][8];
  load_row<NCH>(x + row * ldx, lane, v);
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) q += v[c][i] * v[c][i];
  const float rstd = rsqrtf(wave_sum(q) * (1.f / cols) + eps);
  load_row<NCH>(w, lane, g);
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) v[c][i] = g[c][i] * bf2f(f2bf(v[c][i] * rstd));  // HF rounds xhat before the weight
  if (y) store_row<NCH>(y + row * ldy, lane, v);
  if (rstd_out && lane == 0) rstd_out[row] = rstd;
  if (y8) store_row_e4m3<NCH>(y8 + (long)row * cols, y8scale + row, lane, v);
}

// dx = rstd * (g*dy - xhat * mean(g*dy*xhat)) [+ add]   (activation gradient only: the LLaMA weights are frozen)
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* dy, const bf16_t* __restrict__ x,
                                                          const bf16_t* __restrict__ w, const float* __restrict__ rstd,
                                                          const bf16_t* add, bf16_t* dx, int rows, long ld, float eps,
                                                          unsigned char* __restrict__ dx8 = nullptr, float* __restrict__ dx8scale = nullptr) {
  constexpr int cols = NCH * 512;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * LN_WAVES + (threadIdx.x >> 6);
  if (row >= rows) return;
  float xv[NCH][8], dv[NCH][8], g[NCH][8];
  load_row<NCH>(x + row * ld, lane, xv);
  load_row<NCH>(dy + row * ld, lane, dv);
  load_row<NCH>(w, lane, g);
  float rs;
  if (rstd) {
    rs = rstd[row];
  } else {
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int i = 0; i < 8; ++i) q += xv[c][i] * xv[c][i];
    rs = rsqrtf(wave_sum(q) * (1.f / cols) + eps);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      xv[c][i] *= rs;
      dv[c][i] *= g[c][i];
      s += dv[c][i] * xv[c][i];
    }
  s = wave_sum(s) * (1.f / cols);
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) dv[c][i] = rs * (dv[c][i] - xv[c][i] * s);
  if (add) {
    float av[NCH][8];
    load_row<NCH>(add + row * ld, lane, av);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int i = 0; i < 8; ++i) dv[c][i] += av[c][i];
  }
  store_row<NCH>(dx + row * ld, lane, dv);
  if (dx8) store_row_e4m3<NCH>(dx8 + (long)row * cols, dx8scale + row, lane, dv);
}

}  // namespace

#define DISPATCH_NCH(cols, CALL)                                                  \
  switch ((cols) / 512) {                                                         \
    case 1: { constexpr int NCH = 1; CALL; } break;                               \
    case 2: { constexpr int NCH = 2; CALL; } break;                               \
    case 4: { constexpr int NCH = 4; CALL; } break;                               \
    case 8: { constexpr int NCH = 8; CALL; } break;                               \
    default: LHRS_FAIL("norm: cols=%d not supported (512,1024,2048,4096)", cols); \
  }

extern "C" int lhrs_layernorm_fwd(const void* x, long ldx, const void* gamma, const void* beta, void* y, long ldy,
                                  float* mean, float* rstd, int rows, int cols, float eps, void* stream) {
  LHRS_REQUIRE(rows > 0 && cols % 512 == 0, "layernorm_fwd: rows=%d cols=%d", rows, cols);
  LHRS_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0, "layernorm_fwd: row strides must be multiples of 8");
  LHRS_REQUIRE((mean == nullptr) == (rstd == nullptr), "layernorm_fwd: mean/rstd must both be set or both null");
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_NCH(cols, hipLaunchKernelGGL((layernorm_fwd_kernel<NCH>), dim3(cdiv(rows, LN_WAVES)), dim3(256), 0, s,
                                        (const bf16_t*)x, (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)y, mean,
                                        rstd, rows, ldx, ldy, eps));
  LHRS_CHECK_LAUNCH("layernorm_fwd");
  return 0;
}

extern "C" int lhrs_layernorm_bwd_nblk(int rows) {
  int n = cdiv(rows, LN_WAVES);
  return n < 256 ? n : 256;
}

// partial: fp32 workspace of lhrs_layernorm_bwd_nblk(rows) * 2 * cols floats (may be null when dgamma is null)
// add (optional, same layout as dx, may alias dx): dx = dx_layernorm + add
extern "C" int lhrs_layernorm_bwd(const void* dy, long ld_dy, const void* x, long ldx, const void* gamma,
                                  const float* mean, const float* rstd, const void* add, void* dx, long ld_dx,
                                  float* dgamma, float* dbeta, float* partial, int accumulate, int rows, int cols,
                                  void* stream) {
  LHRS_REQUIRE(rows > 0 && cols % 512 == 0 && cols <= 1024, "layernorm_bwd: rows=%d cols=%d (cols<=1024)", rows, cols);
  LHRS_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "layernorm_bwd: dgamma/dbeta both or neither");
  LHRS_REQUIRE(dgamma == nullptr || partial != nullptr, "layernorm_bwd: workspace missing");
  hipStream_t s = (hipStream_t)stream;
  const int nblk = lhrs_layernorm_bwd_nblk(rows);
  float* part = dgamma ? partial : nullptr;
  switch (cols / 512) {
    case 1:
      hipLaunchKernelGGL((layernorm_bwd_kernel<1>), dim3(nblk), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x,
                         (const bf16_t*)gamma, mean, rstd, (const bf16_t*)add, (bf16_t*)dx, part, rows, ld_dy, ldx, ld_dx);
      break;
    default:
      hipLaunchKernelGGL((layernorm_bwd_kernel<2>), dim3(nblk), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x,
                         (const bf16_t*)gamma, mean, rstd, (const bf16_t*)add, (bf16_t*)dx, part, rows, ld_dy, ldx, ld_dx);
      break;
  }
  LHRS_CHECK_LAUNCH("layernorm_bwd");
  if (dgamma) {
    hipLaunchKernelGGL(layernorm_bwd_finalize, dim3(cdiv(cols, 32)), dim3(256), 0, s, part, dgamma, dbeta, nblk, cols,
                       accumulate);
    LHRS_CHECK_LAUNCH("layernorm_bwd_finalize");
  }
  return 0;
}

extern "C" int lhrs_rmsnorm_fwd(const void* x, long ldx, const void* w, void* y, long ldy, float* rstd, int rows,
                                int cols, float eps, void* stream) {
  LHRS_REQUIRE(rows > 0 && cols % 512 == 0, "rmsnorm_fwd: rows=%d cols=%d", rows, cols);
  LHRS_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0, "rmsnorm_fwd: row strides must be multiples of 8");
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_NCH(cols, hipLaunchKernelGGL((rmsnorm_fwd_kernel<NCH>), dim3(cdiv(rows, LN_WAVES)), dim3(256), 0, s,
                                        (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rstd, rows, ldx, ldy, eps));
  LHRS_CHECK_LAUNCH("rmsnorm_fwd");
  return 0;
}

// dx may alias dy or add.  rstd may be null (recomputed from x with eps).
extern "C" int lhrs_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* add,
                                void* dx, int rows, int cols, float eps, void* stream) {
  LHRS_REQUIRE(rows > 0 && cols % 512 == 0, "rmsnorm_bwd: rows=%d cols=%d", rows, cols);
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_NCH(cols, hipLaunchKernelGGL((rmsnorm_bwd_kernel<NCH>), dim3(cdiv(rows, LN_WAVES)), dim3(256), 0, s,
                                        (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w, rstd, (const bf16_t*)add,
                                        (bf16_t*)dx, rows, (long)cols, eps));
  LHRS_CHECK_LAUNCH("rmsnorm_bwd");
  return 0;
}

// lhrs_rmsnorm_fwd / _bwd that ALSO emit the per-row e4m3 copy of their result (y8 [rows, cols] bytes + scale [rows]) for the next
// 8-bit-base GEMM; y may be NULL in the forward when no adapter needs the bf16 activations
extern "C" int lhrs_rmsnorm_fwd_q(const void* x, long ldx, const void* w, void* y, long ldy, void* y8, float* y8scale, int rows,
                                  int cols, float eps, void* stream) {
  LHRS_REQUIRE(rows > 0 && cols % 512 == 0 && y8 && y8scale, "rmsnorm_fwd_q: rows=%d cols=%d", rows, cols);
  LHRS_REQUIRE(ldx % 8 == 0 && (y == nullptr || ldy % 8 == 0), "rmsnorm_fwd_q: row strides must be multiples of 8");
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_NCH(cols, hipLaunchKernelGGL((rmsnorm_fwd_kernel<NCH>), dim3(cdiv(rows, LN_WAVES)), dim3(256), 0, s,
                                        (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, (float*)nullptr, rows, ldx, ldy, eps,
                                        (unsigned char*)y8, y8scale));
  LHRS_CHECK_LAUNCH("rmsnorm_fwd_q");
  return 0;
}

extern "C" int lhrs_rmsnorm_bwd_q(const void* dy, const void* x, const void* w, const float* rstd, const void* add, void* dx,
                                  void* dx8, float* dx8scale, int rows, int cols, float eps, void* stream) {
  LHRS_REQUIRE(rows > 0 && cols % 512 == 0 && dx && dx8 && dx8scale, "rmsnorm_bwd_q: rows=%d cols=%d", rows, cols);
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_NCH(cols, hipLaunchKernelGGL((rmsnorm_bwd_kernel<NCH>), dim3(cdiv(rows, LN_WAVES)), dim3(256), 0, s,
                                        (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w, rstd, (const bf16_t*)add,
                                        (bf16_t*)dx, rows, (long)cols, eps, (unsigned char*)dx8, dx8scale));
  LHRS_CHECK_LAUNCH("rmsnorm_bwd_q");
  return 0;
}
