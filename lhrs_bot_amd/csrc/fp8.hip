// Producers that emit the e4m3 operand of the NEXT 8-bit-base GEMM directly (SURVEY.md §8 f-4 follow-up): the SwiGLU forward and
// backward of HF LlamaMLP (down(silu(gate(x)) * up(x)), reached from /root/reference lhrs/models/text_modal.py:258-294) with a per-row
// e4m3 quantisation of their result, so that the bf16 tensor is neither written nor read again by a separate quantisation pass.
// One block per token row; the row's results stay in registers between the |max| reduction and the conversion.  HBM-bound:
//   forward : reads gate|up (4*F B/row), writes act8 (F B) [+ act bf16 (2*F B) when an adapter on `down` needs it]
//   backward: reads d_act (2*F), gate|up (4*F), writes d(gate|up)8 (2*F B) [+ bf16 (4*F B) when an adapter on gate|up needs it]
#include "common.h"

namespace {

__device__ __forceinline__ int2 cvt8_e4m3(const float (&v)[8], float inv) {
  int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, 0, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, lo, true);
  int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, 0, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, hi, true);
  return make_int2(lo, hi);
}

constexpr int FCH = 6;  // chunks of 8 columns per thread kept in registers: F <= 6 * 256 * 8 = 12288

// act = silu(g) * u (rounded to bf16 like the unfused kernel), act8 = e4m3(act / scale), scale = max|act| / 448
__global__ __launch_bounds__(256) void swiglu_fwd_q_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ act, uint8_t* __restrict__ act8,
                                                           float* __restrict__ scale, int F) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  const int tid = threadIdx.x, nch = F / 8;
  const bf16_t* g = gu + row * 2 * F;
  float keep[FCH][8];
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < FCH; ++i) {
    const int c = tid + i * 256;
    if (c < nch) {
      float gv[8], uv[8];
      unpack8(*reinterpret_cast<const uint4*>(g + c * 8), gv);
      unpack8(*reinterpret_cast<const uint4*>(g + F + c * 8), uv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        keep[i][e] = bf2f(f2bf(silu(gv[e]) * uv[e]));
        m = fmaxf(m, fabsf(keep[i][e]));
      }
      if (act) *reinterpret_cast<uint4*>(act + row * F + c * 8) = pack8(keep[i]);
    }
  }
  m = block_max<4>(m, red);
  const float sc = m > 0.f ? m / 448.f : 1.f;
  if (tid == 0) scale[row] = sc;
  const float inv = 1.f / sc;
#pragma unroll
  for (int i = 0; i < FCH; ++i) {
    const int c = tid + i * 256;
    if (c < nch) *reinterpret_cast<int2*>(act8 + row * F + c * 8) = cvt8_e4m3(keep[i], inv);
  }
}

// d(gate|up) from d_act and the saved gate|up, quantised per row over all 2*F columns
__global__ __launch_bounds__(256) void swiglu_bwd_q_kernel(const bf16_t* __restrict__ dact, const bf16_t* __restrict__ gu, bf16_t* dgu,
                                                           uint8_t* __restrict__ dgu8, float* __restrict__ scale, int F) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  const int tid = threadIdx.x, nch = F / 8;
  const bf16_t* g = gu + row * 2 * F;
  float kg[FCH][8], ku[FCH][8];
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < FCH; ++i) {
    const int c = tid + i * 256;
    if (c < nch) {
      float gv[8], uv[8], d[8];
      unpack8(*reinterpret_cast<const uint4*>(g + c * 8), gv);
      unpack8(*reinterpret_cast<const uint4*>(g + F + c * 8), uv);
      unpack8(*reinterpret_cast<const uint4*>(dact + row * F + c * 8), d);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float sg = sigmoid_f(gv[e]);
        ku[i][e] = bf2f(f2bf(d[e] * gv[e] * sg));
        kg[i][e] = bf2f(f2bf(d[e] * uv[e] * sg * (1.f + gv[e] * (1.f - sg))));
        m = fmaxf(m, fmaxf(fabsf(kg[i][e]), fabsf(ku[i][e])));
      }
      if (dgu) {  // may alias gu: this thread has read its slots
        *reinterpret_cast<uint4*>(dgu + row * 2 * F + c * 8) = pack8(kg[i]);
        *reinterpret_cast<uint4*>(dgu + row * 2 * F + F + c * 8) = pack8(ku[i]);
      }
    }
  }
  m = block_max<4>(m, red);
  const float sc = m > 0.f ? m / 448.f : 1.f;
  if (tid == 0) scale[row] = sc;
  const float inv = 1.f / sc;
#pragma unroll
  for (int i = 0; i < FCH; ++i) {
    const int c = tid + i * 256;
    if (c < nch) {
      *reinterpret_cast<int2*>(dgu8 + row * 2 * F + c * 8) = cvt8_e4m3(kg[i], inv);
      *reinterpret_cast<int2*>(dgu8 + row * 2 * F + F + c * 8) = cvt8_e4m3(ku[i], inv);
    }
  }
}

}  // namespace

// act (optional, bf16 [rows, F]) and act8 / scale (e4m3 [rows, F], fp32 [rows]) from gate|up [rows, 2F]; F % 8 == 0, F <= 12288
extern "C" int lhrs_swiglu_fwd_q(const void* gate_up, void* act, void* act8, float* scale, long rows, int F, void* stream) {
  LHRS_REQUIRE(rows > 0 && F % 16 == 0 && F <= FCH * 256 * 8 && act8 && scale, "swiglu_fwd_q: rows=%ld F=%d", rows, F);
  hipLaunchKernelGGL(swiglu_fwd_q_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gate_up, (bf16_t*)act,
                     (uint8_t*)act8, scale, F);
  LHRS_CHECK_LAUNCH("swiglu_fwd_q");
  return 0;
}

// dgu (optional bf16 [rows, 2F], may alias gate_up) and dgu8 / scale (e4m3 [rows, 2F], fp32 [rows]) from d_act [rows, F] and gate|up
extern "C" int lhrs_swiglu_bwd_q(const void* dact, const void* gate_up, void* dgu, void* dgu8, float* scale, long rows, int F, void* stream) {
  LHRS_REQUIRE(rows > 0 && F % 16 == 0 && F <= FCH * 256 * 8 && dgu8 && scale, "swiglu_bwd_q: rows=%ld F=%d", rows, F);
  hipLaunchKernelGGL(swiglu_bwd_q_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dact, (const bf16_t*)gate_up,
                     (bf16_t*)dgu, (uint8_t*)dgu8, scale, F);
  LHRS_CHECK_LAUNCH("swiglu_bwd_q");
  return 0;
}
