// Shared device/host helpers for the LHRS-Bot gfx950 hot path.
// Everything here is CDNA4-only: 64-lane wavefronts, bf16 MFMA, 160 KiB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef uint16_t bf16_t;  // raw bf16 bits; matches lhrs_bf16_t in include/lhrs_hip.h

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;    // one 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

// ---- error plumbing (no exceptions cross the C ABI) -------------------------------------------
extern "C" void lhrs_set_error(const char* msg);

#define LHRS_FAIL(...)                                   \
  do {                                                   \
    char _b[512];                                        \
    snprintf(_b, sizeof(_b), __VA_ARGS__);               \
    lhrs_set_error(_b);                                  \
    return -1;                                           \
  } while (0)

#define LHRS_REQUIRE(cond, ...)                          \
  do {                                                   \
    if (!(cond)) LHRS_FAIL(__VA_ARGS__);                 \
  } while (0)

#define LHRS_CHECK_LAUNCH(name)                                                   \
  do {                                                                            \
    hipError_t _e = hipGetLastError();                                            \
    if (_e != hipSuccess) LHRS_FAIL("%s: launch failed: %s", name, hipGetErrorString(_e)); \
  } while (0)

// ---- bf16 <-> f32 -------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {  // RNE, lowers to v_cvt_pk_bf16_f32 on gfx950
  __bf16 x = (__bf16)f;
  return __builtin_bit_cast(bf16_t, x);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
__device__ __forceinline__ float bflo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// ---- wavefront reductions (64 lanes) ------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blockDim.x = 64*NW; `red` is NW floats of LDS scratch. All threads get the result.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  if (NW == 1) return v;
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) t += red[i];
  return t;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  if (NW == 1) return v;
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) t = fmaxf(t, red[i]);
  return t;
}

// HF rotate_half RoPE of one (x[d], x[d + D/2]) pair; separately rounded products so that every kernel that rotates agrees bit for bit
__device__ __forceinline__ void rope_pair(float a, float b, float co, float si, float& o1, float& o2) {
#pragma clang fp contract(off)
  o1 = a * co - b * si;
  o2 = b * co + a * si;
}

// ---- activations --------------------------------------------------------------------------------
// the logistic function with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division (a ten-instruction sequence per element: in the SwiGLU epilogues
// of the four-wave GEMM that was half the write-out's VALU time).  EVERY kernel that needs sigma / silu / quick_gelu goes through here, so fused and unfused paths
// keep agreeing bit for bit (tests/test_kernels_gpu.py: test_gemm_fused_swiglu_bit_identical_to_unfused)
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float quick_gelu(float x) { return x * sigmoid_f(1.702f * x); }
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
__device__ __forceinline__ void unpack8(const uint4& u, float (&v)[8]) {
  v[0] = bflo(u.x); v[1] = bfhi(u.x); v[2] = bflo(u.y); v[3] = bfhi(u.y);
  v[4] = bflo(u.z); v[5] = bfhi(u.z); v[6] = bflo(u.w); v[7] = bfhi(u.w);
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
  return make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
}
// LoRA dropout (peft lora.Linear: lora_B(lora_A(dropout(x))), p = 0.05 in Config/multi_modal_stage2.yaml): counter-based mask - element
// idx of the adapter INPUT is kept iff lowbias32(idx, seed) >= p * 2^32 - so forward, backward and the tests regenerate the same mask
__device__ __forceinline__ bool drop_keep(unsigned seed, long idx, unsigned thresh) {
  unsigned x = ((unsigned)idx * 0x9E3779B1u) ^ ((unsigned)(idx >> 32) * 0x85EBCA77u) ^ seed;
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x >= thresh;
}
__device__ __forceinline__ float silu(float x) { return x * sigmoid_f(x); }

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
