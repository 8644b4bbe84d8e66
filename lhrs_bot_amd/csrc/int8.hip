// LLM.int8() operand preparation for the frozen 8-bit base of stages 2 / 3 (`bits: 8` of Config/multi_modal_stage{2,3}.yaml).
//
// The reference loads the frozen LLaMA through bitsandbytes (lhrs/models/text_modal.py:91-131: BitsAndBytesConfig(load_in_8bit=True,
// llm_int8_threshold=6.0, llm_int8_has_fp16_weight=False)); every decoder linear then runs bitsandbytes' MatMul8bitLt (0.41 series):
//   weights, once:   CB[n, k] = rint(W[n, k] * 127 / absmax_k |W[n, :]|)  (int8),  SCB[n] = absmax                      (vector-wise, per output row)
//   per call:        outlier COLUMNS O = {k : any_t |x[t, k]| >= 6.0};
//                    SCA[t] = max_k { |x[t, k]| : |x[t, k]| < 6.0 };  CA[t, k] = rint(x[t, k] * 127 / SCA[t]), columns in O zeroed
//                    y = (CA . CB^T in int32) * SCA[t] * SCB[n] / 127^2  +  x[:, O] . (CB[:, O] * SCB / 127)^T       (the second term in 16 bit)
//   backward:        dx = dy . (CB * SCB / 127)   - the DEquantised weight in 16 bit (has_fp16_weights = False)
// bitsandbytes is not installed here (parity vs the package is unpinned; oracle/int8_oracle.py restates the same rules and is the checker).
// This file holds the element-wise side: weight quantisation / dequantisation, the outlier-column scan and compaction, the activation
// quantisation and the two column gathers that feed the 16-bit outlier product.  The int8 product itself is gemm_fp8_256_kernel<true>
// (gemm.hip: v_mfma_i32_32x32x32_i8, exact int32 accumulation), with the outlier product appended as bf16 stages of the same launch.
#include "common.h"

namespace {

constexpr int QCH = 12;  // 16-B chunks a thread keeps in registers between the absmax and the conversion (K <= 24576)

__device__ __forceinline__ int q8(float v, float inv) {
  const int q = __float2int_rn(v * inv);  // round half to even, like torch.round / rintf
  return max(-127, min(127, q));
}
__device__ __forceinline__ int pack4(int a, int b, int c, int d) { return (a & 0xff) | ((b & 0xff) << 8) | ((c & 0xff) << 16) | ((d & 0xff) << 24); }

// one block per row: scale[n] = absmax / 127 (dequantisation factor), CB = rint(W * 127 / absmax).  mask8 / thr: the activation variant
// (mask8 != nullptr: one bit per column, bit e of byte c = column 8c + e): entries with |x| >= thr do not count towards the absmax, flagged
// columns are stored as 0.  One mask byte per 16-byte chunk - an int per column tripled the L2 traffic of this kernel (86 -> ~30 us at 8190 x 4096)
__global__ __launch_bounds__(256) void quant_int8_rows_kernel(const bf16_t* __restrict__ W, long ldw, int8_t* __restrict__ Q, long ldq,
                                                              float* __restrict__ scale, int K, const unsigned char* __restrict__ mask8, float thr) {
  __shared__ float red[4];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int nch = K / 8;
  const bf16_t* row = W + (long)n * ldw;
  uint4 keep[QCH];
  float m = 0.f;
  auto upd = [&](const uint4& raw) {
    float v[8];
    unpack8(raw, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float a = fabsf(v[e]);
      if (mask8 == nullptr || a < thr) m = fmaxf(m, a);
    }
  };
#pragma unroll
  for (int i = 0; i < QCH; ++i) {
    const int c = tid + i * 256;
    keep[i] = make_uint4(0, 0, 0, 0);
    if (c < nch) { keep[i] = *reinterpret_cast<const uint4*>(row + c * 8); upd(keep[i]); }
  }
  for (int c = tid + QCH * 256; c < nch; c += 256) upd(*reinterpret_cast<const uint4*>(row + c * 8));
  m = block_max<4>(m, red);
  const float inv = m > 0.f ? 127.f / m : 0.f;
  if (tid == 0) scale[n] = m / 127.f;
  int8_t* orow = Q + (long)n * ldq;
  auto cvt = [&](const uint4& raw, int c) {
    float v[8];
    unpack8(raw, v);
    int q[8];
    const unsigned mk = mask8 != nullptr ? mask8[c] : 0u;
#pragma unroll
    for (int e = 0; e < 8; ++e) q[e] = ((mk >> e) & 1u) ? 0 : q8(v[e], inv);
    *reinterpret_cast<int2*>(orow + c * 8) = make_int2(pack4(q[0], q[1], q[2], q[3]), pack4(q[4], q[5], q[6], q[7]));
  };
#pragma unroll
  for (int i = 0; i < QCH; ++i) {
    const int c = tid + i * 256;
    if (c < nch) cvt(keep[i], c);
  }
  for (int c = tid + QCH * 256; c < nch; c += 256) cvt(*reinterpret_cast<const uint4*>(row + c * 8), c);
}

// W[n, k] = CB[n, k] * scale[n] as bf16 (the 16-bit weight of the backward and of generate())
__global__ __launch_bounds__(256) void dequant_int8_rows_kernel(const int8_t* __restrict__ Q, long ldq, const float* __restrict__ scale,
                                                                bf16_t* __restrict__ W, long ldw, int K) {
  const int n = blockIdx.x;
  const float s = scale[n];
  for (int c = threadIdx.x; c < K / 8; c += 256) {
    const int2 raw = *reinterpret_cast<const int2*>(Q + (long)n * ldq + c * 8);
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = (float)(int8_t)((raw.x >> (8 * e)) & 0xff) * s;
      v[4 + e] = (float)(int8_t)((raw.y >> (8 * e)) & 0xff) * s;
    }
    *reinterpret_cast<uint4*>(W + (long)n * ldw + c * 8) = pack8(v);
  }
}

// bit k of the mask = 1 when any |x[t, k]| >= thr.  Block = 64 column chunks of 8 (one 1-KiB line of a row per wave) x 4 waves that take every
// fourth row of a strip of 32: eight independent 16-B loads per thread; a wave that saw an outlier ORs its byte into the mask word (rare).
__global__ __launch_bounds__(256) void outlier_cols_kernel(const bf16_t* __restrict__ X, long ldx, int M, int K, float thr, unsigned* __restrict__ mask32) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  if (c >= K / 8) return;
  const int r0 = blockIdx.y * 32 + (threadIdx.x >> 6);
  uint4 raw[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) raw[i] = *reinterpret_cast<const uint4*>(X + (long)min(r0 + 4 * i, M - 1) * ldx + c * 8);
  unsigned hit = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float v[8];
    unpack8(raw[i], v);
#pragma unroll
    for (int e = 0; e < 8; ++e) hit |= (fabsf(v[e]) >= thr ? 1u : 0u) << e;
  }
  if (hit) atomicOr(mask32 + (c >> 2), hit << (8 * (c & 3)));
}

// idx[0 .. n) = the flagged columns in ascending order, idx[n .. n_pad) = -1 with n_pad = n rounded up to 64 (the product appends whole
// 64-column bf16 stages); meta[0] = n, meta[1] = n_pad.  idx holds K entries: there is no cap - LLM.int8 takes however many columns carry an
// outlier into the 16-bit product (the inputs of down_proj routinely have hundreds).  One block of 1024 threads; K <= 1024 * 32.
__global__ __launch_bounds__(1024) void compact_cols_kernel(const unsigned char* __restrict__ mask8, int K, int* __restrict__ idx, int* __restrict__ meta) {
  __shared__ int cnt[1024];
  const int tid = threadIdx.x;
  const int per = (K + 1023) / 1024;
  const int k0 = tid * per, k1 = min(K, k0 + per);
  int c = 0;
  for (int k = k0; k < k1; ++k) c += (mask8[k >> 3] >> (k & 7)) & 1;
  cnt[tid] = c;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {   // inclusive scan
    const int v = tid >= off ? cnt[tid - off] : 0;
    __syncthreads();
    cnt[tid] += v;
    __syncthreads();
  }
  int pos = cnt[tid] - c;
  const int total = cnt[1023], npad = (total + 63) / 64 * 64;
  for (int k = k0; k < k1; ++k)
    if ((mask8[k >> 3] >> (k & 7)) & 1) idx[pos++] = k;
  for (int j = total + tid; j < npad; j += 1024) idx[j] = -1;   // npad <= K rounded up to 64 <= the idx buffer
  if (tid == 0) { meta[0] = total; meta[1] = npad; }
}

// A2[t, j] = x[t, idx[j]] (0 where idx[j] < 0), j < n_pad = meta[1]: the 16-bit operand of the outlier product.  One block per row (the column
// count is only known on the device: the threads stride over it)
__global__ __launch_bounds__(256) void gather_cols_x_kernel(const bf16_t* __restrict__ X, long ldx, const int* __restrict__ idx,
                                                            const int* __restrict__ meta, bf16_t* __restrict__ out, long ldo) {
  const int npad = meta[1];
  const long t = blockIdx.x;
  for (int j = threadIdx.x; j < npad; j += 256) {
    const int k = idx[j];
    out[t * ldo + j] = k >= 0 ? X[t * ldx + k] : (bf16_t)0;
  }
}
// B2[n, j] = CB[n, idx[j]] * scale[n] as bf16 (0 where idx[j] < 0): the dequantised weight columns
__global__ __launch_bounds__(256) void gather_cols_w_kernel(const int8_t* __restrict__ Q, long ldq, const float* __restrict__ scale,
                                                            const int* __restrict__ idx, const int* __restrict__ meta, bf16_t* __restrict__ out,
                                                            long ldo) {
  const int npad = meta[1];
  const long n = blockIdx.x;
  const float sc = scale[n];
  for (int j = threadIdx.x; j < npad; j += 256) {
    const int k = idx[j];
    out[n * ldo + j] = k >= 0 ? f2bf((float)Q[n * ldq + k] * sc) : (bf16_t)0;
  }
}

}  // namespace

// bf16 rows -> int8 rows + dequantisation factor absmax / 127 per row (weights; K % 8 == 0)
extern "C" int lhrs_quant_int8_rows(const void* W, long ldw, void* Q, long ldq, float* scale, int N, int K, void* stream) {
  LHRS_REQUIRE(N > 0 && K > 0 && K % 8 == 0 && ldw % 8 == 0 && ldq % 8 == 0, "quant_int8_rows: N=%d K=%d ldw=%ld ldq=%ld", N, K, ldw, ldq);
  hipLaunchKernelGGL(quant_int8_rows_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)W, ldw, (int8_t*)Q, ldq, scale, K,
                     (const unsigned char*)nullptr, 0.f);
  LHRS_CHECK_LAUNCH("quant_int8_rows");
  return 0;
}
extern "C" int lhrs_dequant_int8_rows(const void* Q, long ldq, const float* scale, void* W, long ldw, int N, int K, void* stream) {
  LHRS_REQUIRE(N > 0 && K > 0 && K % 8 == 0 && ldw % 8 == 0 && ldq % 8 == 0, "dequant_int8_rows: N=%d K=%d", N, K);
  hipLaunchKernelGGL(dequant_int8_rows_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, (const int8_t*)Q, ldq, scale, (bf16_t*)W, ldw, K);
  LHRS_CHECK_LAUNCH("dequant_int8_rows");
  return 0;
}

// The activation side of one LLM.int8 product, on `stream`:
//   flags (int [K]: the outlier bit mask lives in its first K / 8 bytes), idx (int [K rounded up to 64]) and meta (int [2]: n, n_pad) are workspaces;
//   XQ int8 [M, ldq] + sx [M] (absmax of the non-outlier entries / 127); A2 bf16 [M, lda2] = x[:, outlier columns] and B2 bf16 [N, ldb2] =
//   dequantised weight columns, n_pad = meta[1] columns of each written (both buffers hold up to K rounded up to 64 columns).
extern "C" int lhrs_int8_prepare(const void* X, long ldx, int M, int K, float thr, const void* WQ, long ldwq, const float* wscale, int N,
                                 void* XQ, long ldq, float* sx, int* flags, int* idx, int* meta, void* A2, long lda2, void* B2, long ldb2,
                                 void* stream) {
  const int kpad = (K + 63) / 64 * 64;
  LHRS_REQUIRE(M > 0 && K > 0 && N > 0 && K % 8 == 0 && K <= 32768 && ldx % 8 == 0 && ldq % 8 == 0 && lda2 >= kpad && ldb2 >= kpad && thr > 0.f,
               "int8_prepare: M=%d K=%d N=%d lda2=%ld ldb2=%ld", M, K, N, lda2, ldb2);
  hipStream_t s = (hipStream_t)stream;
  // `flags` (int [K]) holds the outlier BIT mask in its first K / 8 bytes
  if (hipMemsetAsync(flags, 0, (size_t)(K / 8 + 3) / 4 * 4, s) != hipSuccess) LHRS_FAIL("int8_prepare: memset failed");
  hipLaunchKernelGGL(outlier_cols_kernel, dim3(cdiv(K / 8, 64), cdiv(M, 32)), dim3(256), 0, s, (const bf16_t*)X, ldx, M, K, thr, (unsigned*)flags);
  hipLaunchKernelGGL(compact_cols_kernel, dim3(1), dim3(1024), 0, s, (const unsigned char*)flags, K, idx, meta);
  hipLaunchKernelGGL(quant_int8_rows_kernel, dim3(M), dim3(256), 0, s, (const bf16_t*)X, ldx, (int8_t*)XQ, ldq, sx, K, (const unsigned char*)flags, thr);
  hipLaunchKernelGGL(gather_cols_x_kernel, dim3(M), dim3(256), 0, s, (const bf16_t*)X, ldx, (const int*)idx, (const int*)meta, (bf16_t*)A2, lda2);
  hipLaunchKernelGGL(gather_cols_w_kernel, dim3(N), dim3(256), 0, s, (const int8_t*)WQ, ldwq, wscale, (const int*)idx, (const int*)meta, (bf16_t*)B2,
                     ldb2);
  LHRS_CHECK_LAUNCH("int8_prepare");
  return 0;
}
