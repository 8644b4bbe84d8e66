// 4-bit storage of the frozen LLaMA linears: `bits: 4` with `quant_type: nf4 | fp4` and `double_quant` of the reference's YAML surface.
//
// The reference loads the decoder through bitsandbytes (lhrs/models/text_modal.py:91-107: BitsAndBytesConfig(load_in_4bit=True,
// bnb_4bit_quant_type=config.quant_type, bnb_4bit_use_double_quant=config.double_quant, bnb_4bit_compute_dtype=compute_dtype)).  What its
// Linear4bit does (bitsandbytes 0.41 series, functional.quantize_4bit / dequantize_4bit and MatMul4Bit):
//   weights, once:  the flattened weight in blocks of 64 consecutive elements: absmax_b = max |w|, code = Q4(w * (1 / absmax_b)) with the 16-entry
//                   NF4 (normal-float) or FP4 (sign + e2m1-like) table, two codes per byte (first element in the HIGH nibble);
//                   double_quant: offset = mean(absmax); (absmax - offset) in blocks of 256 -> 8-bit codes of the "dynamic" data type + fp32 absmax2
//   every product:  y = x . dequant(W)^T with dequant(W) = table[code] * absmax_b cast to the compute dtype - forward AND backward are plain
//                   16-bit products on the dequantised weight; there is no 4-bit arithmetic anywhere.
// So on a 288 GB part the 4-bit base is a weight TRANSFORMATION: this file produces the codes and statistics (kept beside the weight for
// checkpoints) and expands them back into the bf16 weight the ordinary GEMMs read; the bytes of a block never leave a wave.
// bitsandbytes is not installed here: oracle/nf4_oracle.py restates the same rules and is the checker (parity vs the package unpinned).
#include "common.h"

namespace {

// ascending decision thresholds of dQuantizeNF4 (midpoints of neighbouring levels); code = how many of them x exceeds (the package's
// comparison tree, flattened - a NaN from an all-zero block exceeds none and gets code 0 there and here)
__constant__ float NF4_THR[15] = {-0.8480964004993439f, -0.6106329262256622f, -0.4599952697753906f, -0.33967943489551544f, -0.23460740596055984f,
                                  -0.13791173323988914f, -0.045525018125772476f, 0.03979014977812767f, 0.1202552504837513f, 0.2035212516784668f,
                                  0.2920137718319893f, 0.3893125355243683f, 0.5016634166240692f, 0.6427869200706482f, 0.8614784181118011f};
__constant__ float NF4_LEVEL[16] = {-1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f, -0.28444138169288635f,
                                    -0.18477343022823334f, -0.09105003625154495f, 0.0f, 0.07958029955625534f, 0.16093020141124725f,
                                    0.24611230194568634f, 0.33791524171829224f, 0.44070982933044434f, 0.5626170039176941f, 0.7229568362236023f, 1.0f};
// dQuantizeFP4: thresholds on |x| ascending, and the 3-bit code of the magnitude that many thresholds lie below
__constant__ float FP4_THR[7] = {0.00260417f, 0.0859375f, 0.20833333f, 0.29166667f, 0.4166667f, 0.583333f, 0.8333333f};
__constant__ int FP4_CODE[8] = {0b000, 0b001, 0b110, 0b111, 0b100, 0b101, 0b010, 0b011};
__constant__ float FP4_LEVEL[16] = {0.0f, 5.208333333e-03f, 0.66666667f, 1.0f, 0.33333333f, 0.5f, 0.16666667f, 0.25f,
                                    -0.0f, -5.208333333e-03f, -0.66666667f, -1.0f, -0.33333333f, -0.5f, -0.16666667f, -0.25f};

__device__ __forceinline__ int q_nf4(float x) {
  int c = 0;
#pragma unroll
  for (int i = 0; i < 15; ++i) c += x > NF4_THR[i] ? 1 : 0;
  return c;
}
__device__ __forceinline__ int q_fp4(float x) {
  const int sign = x < 0.f ? 0b1000 : 0;
  const float a = fabsf(x);
  int c = 0;
#pragma unroll
  for (int i = 0; i < 7; ++i) c += a > FP4_THR[i] ? 1 : 0;
  return FP4_CODE[c] + sign;
}

// one wave per block of 64 elements, four blocks per workgroup.  n % 64 == 0 (every LLaMA linear: rows of 4096 / 11008 elements).
template <bool FP4>
__global__ __launch_bounds__(256) void quant4_blocks_kernel(const bf16_t* __restrict__ W, uint8_t* __restrict__ packed, float* __restrict__ absmax,
                                                            long nblocks) {
  const long blk = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (blk >= nblocks) return;
  const float x = bf2f(W[blk * 64 + lane]);
  const float m = wave_max(fabsf(x));
  if (lane == 0) absmax[blk] = m;
  const float v = x * (1.0f / m);   // the package multiplies by the reciprocal
  const int code = FP4 ? q_fp4(v) : q_nf4(v);
  const int next = __shfl_down(code, 1);
  if ((lane & 1) == 0) packed[blk * 32 + (lane >> 1)] = (uint8_t)((code << 4) | next);
}

// one thread per byte -> two bf16 values: table[code] * absmax in fp32, one rounding to bf16 (kDequantizeBlockwise, then the cast to T)
template <bool FP4>
__global__ __launch_bounds__(256) void dequant4_blocks_kernel(const uint8_t* __restrict__ packed, const float* __restrict__ absmax,
                                                              bf16_t* __restrict__ W, long nbytes) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nbytes) return;
  const uint8_t b = packed[i];
  const float a = absmax[i >> 5];
  const float hi = (FP4 ? FP4_LEVEL[b >> 4] : NF4_LEVEL[b >> 4]) * a, lo = (FP4 ? FP4_LEVEL[b & 15] : NF4_LEVEL[b & 15]) * a;
  reinterpret_cast<uint32_t*>(W)[i] = pack2bf(hi, lo);
}

// dQuantize<0> of the package: seven halving steps over the sorted 256-entry code from pivot 127, then the nearer of the pivot and the bound
// on x's side of it (strict comparisons against the midpoint)
__device__ __forceinline__ int q_dynamic(const float* code, float x) {
  int pivot = 127, upper_pivot = 255, lower_pivot = 0;
  float lower = -1.0f, upper = 1.0f, val = code[pivot];
#pragma unroll
  for (int i = 64; i > 0; i >>= 1) {
    if (x > val) { lower_pivot = pivot; lower = val; pivot += i; }
    else { upper_pivot = pivot; upper = val; pivot -= i; }
    val = code[pivot];
  }
  if (upper_pivot == 255) upper = code[upper_pivot];
  if (lower_pivot == 0) lower = code[lower_pivot];
  if (x > val) return x > (upper + val) * 0.5f ? upper_pivot : pivot;
  return x < (lower + val) * 0.5f ? lower_pivot : pivot;
}

// blockwise 8-bit quantisation of the (offset-free) absmax statistics: one workgroup per block of 256 values
__global__ __launch_bounds__(256) void quant8_dynamic_kernel(const float* __restrict__ x, long n, const float* __restrict__ code256,
                                                             uint8_t* __restrict__ q, float* __restrict__ absmax2) {
  __shared__ float s_code[256];
  __shared__ float red[4];
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  s_code[threadIdx.x] = code256[threadIdx.x];
  const float v = i < n ? x[i] : 0.f;
  const float m = block_max<4>(fabsf(v), red);   // syncs: s_code is visible afterwards
  if (threadIdx.x == 0) absmax2[blockIdx.x] = m;
  if (i < n) q[i] = (uint8_t)q_dynamic(s_code, v * (1.0f / m));
}
__global__ __launch_bounds__(256) void dequant8_dynamic_kernel(const uint8_t* __restrict__ q, const float* __restrict__ absmax2, long n,
                                                               const float* __restrict__ code256, float offset, float* __restrict__ out) {
#pragma clang fp contract(off)   // two roundings, not one FMA: the package's kernel stores the product, torch adds the offset afterwards
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const float prod = code256[q[i]] * absmax2[blockIdx.x];
    out[i] = prod + offset;
  }
}

}  // namespace

// C ABI ----------------------------------------------------------------------------------------------------------------------------------
// W: n contiguous bf16 values (a weight or a row range of one), n % 64 == 0 -> packed [n / 2] bytes, absmax [n / 64] fp32.  fp4 = 0: NF4
extern "C" int lhrs_quant4_blocks(const void* W, long n, int fp4, void* packed, float* absmax, void* stream) {
  LHRS_REQUIRE(n > 0 && n % 64 == 0, "quant4_blocks: n=%ld must be a positive multiple of the block size 64", n);
  const long nb = n / 64;
  const dim3 grid((unsigned)((nb + 3) / 4));
  if (fp4) hipLaunchKernelGGL(quant4_blocks_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)W, (uint8_t*)packed, absmax, nb);
  else hipLaunchKernelGGL(quant4_blocks_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)W, (uint8_t*)packed, absmax, nb);
  LHRS_CHECK_LAUNCH("quant4_blocks");
  return 0;
}
extern "C" int lhrs_dequant4_blocks(const void* packed, const float* absmax, long n, int fp4, void* W, void* stream) {
  LHRS_REQUIRE(n > 0 && n % 64 == 0, "dequant4_blocks: n=%ld must be a positive multiple of the block size 64", n);
  const long nbytes = n / 2;
  const dim3 grid((unsigned)((nbytes + 255) / 256));
  if (fp4) hipLaunchKernelGGL(dequant4_blocks_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t*)packed, absmax, (bf16_t*)W, nbytes);
  else hipLaunchKernelGGL(dequant4_blocks_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t*)packed, absmax, (bf16_t*)W, nbytes);
  LHRS_CHECK_LAUNCH("dequant4_blocks");
  return 0;
}
// double_quant statistics: x [n] fp32 (absmax - offset) -> q [n] codes of the sorted 256-entry table code256, absmax2 [ceil(n / 256)]
extern "C" int lhrs_quant8_dynamic(const float* x, long n, const float* code256, void* q, float* absmax2, void* stream) {
  LHRS_REQUIRE(n > 0, "quant8_dynamic: n=%ld", n);
  hipLaunchKernelGGL(quant8_dynamic_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n, code256, (uint8_t*)q, absmax2);
  LHRS_CHECK_LAUNCH("quant8_dynamic");
  return 0;
}
extern "C" int lhrs_dequant8_dynamic(const void* q, const float* absmax2, long n, const float* code256, float offset, float* out, void* stream) {
  LHRS_REQUIRE(n > 0, "dequant8_dynamic: n=%ld", n);
  hipLaunchKernelGGL(dequant8_dynamic_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)q, absmax2, n, code256,
                     offset, out);
  LHRS_CHECK_LAUNCH("dequant8_dynamic");
  return 0;
}
