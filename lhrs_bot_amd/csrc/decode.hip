// Single-token (batch-1..8) decode kernels for gfx950: HBM-bound weight streaming (one full read of the
// 13.5 GB of bf16 LLaMA weights per generated token bounds cli_qa at ~590 tok/s).
//   gemv_kernel   : y[b, n] = sum_k W[n, k] x[b, k] (+ residual)  - every nn.Linear of HF LlamaDecoderLayer at S_q = 1,
//                   reached from TextModal.generate (/root/reference lhrs/models/text_modal.py:586-627).
//                   One wavefront per output row, 16-B lane loads (1 KiB per wave-instruction, fully coalesced), the
//                   activation vector staged once per block in LDS, fp32 accumulation, wavefront shuffle reduction.
//   decode_advance: bumps the device-resident context length and rewrites the attention descriptor / rope position so
//                   that ONE captured hipGraph replays for every token (no kernel argument changes between tokens).
//   kv_append     : writes the new K / V rows at position ctx of the cache.
#include "common.h"

namespace {

template <int NB>  // batch rows of x handled together (weights are read once for all of them)
__global__ __launch_bounds__(256) void gemv_kernel(const bf16_t* __restrict__ W, long ldw, const bf16_t* __restrict__ x, long ldx,
                                                   const bf16_t* res, long ldr, void* y, long ldy, int N, int K, int out_f32) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);  // [NB][K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nch = K / 8;
  for (int b = 0; b < NB; ++b)
    for (int c = tid; c < nch; c += 256) *reinterpret_cast<uint4*>(xs + b * K + c * 8) = *reinterpret_cast<const uint4*>(x + b * ldx + c * 8);
  __syncthreads();
  const int rows_per_block = 4 * 4;  // 4 waves x 4 rows in flight per wave
  const int row0 = blockIdx.x * rows_per_block + wave * 4;
  float acc[4][NB];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;
  for (int c = lane; c < nch; c += 64) {
    uint4 w[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = min(row0 + r, N - 1);
      const i32x4 t = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(W + (long)row * ldw + c * 8));
      w[r] = make_uint4((unsigned)t[0], (unsigned)t[1], (unsigned)t[2], (unsigned)t[3]);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const uint4 xv = *reinterpret_cast<const uint4*>(xs + b * K + c * 8);
      const float x0 = bflo(xv.x), x1 = bfhi(xv.x), x2 = bflo(xv.y), x3 = bfhi(xv.y), x4 = bflo(xv.z), x5 = bfhi(xv.z), x6 = bflo(xv.w),
                  x7 = bfhi(xv.w);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        acc[r][b] += bflo(w[r].x) * x0 + bfhi(w[r].x) * x1 + bflo(w[r].y) * x2 + bfhi(w[r].y) * x3 + bflo(w[r].z) * x4 +
                     bfhi(w[r].z) * x5 + bflo(w[r].w) * x6 + bfhi(w[r].w) * x7;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float s = wave_sum(acc[r][b]);
      const int row = row0 + r;
      if (lane == 0 && row < N) {
        float v = s;
        if (res) v += bf2f(res[b * ldr + row]);
        if (out_f32) reinterpret_cast<float*>(y)[b * ldy + row] = v;
        else reinterpret_cast<bf16_t*>(y)[b * ldy + row] = f2bf(v);
      }
    }
}

// state: int32 [4] = {ctx, -, -, -}; desc: int32 [B][8]; pos: int32 [B]
__global__ void decode_advance_kernel(int* state, int* desc, int* pos, int B, int max_ctx, int step_inc) {
  const int b = threadIdx.x;
  const int ctx = state[0];
  if (b < B) {
    desc[b * 8 + 0] = b;            // q_off  (one new row per sequence)
    desc[b * 8 + 1] = 1;            // q_len
    desc[b * 8 + 2] = b * max_ctx;  // kv_off
    desc[b * 8 + 3] = ctx + 1;      // kv_len (the new token's key is appended before attention)
    desc[b * 8 + 4] = ctx + 1;
    desc[b * 8 + 5] = ctx;          // causal offset
    pos[b] = ctx;
  }
  __syncthreads();
  if (b == 0) state[0] = ctx + step_inc;
}

__global__ void kv_append_kernel(const bf16_t* __restrict__ qkv, long ld, bf16_t* __restrict__ kc, bf16_t* __restrict__ vc,
                                 const int* __restrict__ pos, int B, int d, int max_ctx) {
  const int b = blockIdx.y;
  const int p = pos[b];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < d / 8) {
    const bf16_t* src = qkv + (long)b * ld;
    const long dst = ((long)b * max_ctx + p) * d + c * 8;
    *reinterpret_cast<uint4*>(kc + dst) = *reinterpret_cast<const uint4*>(src + d + c * 8);
    *reinterpret_cast<uint4*>(vc + dst) = *reinterpret_cast<const uint4*>(src + 2 * d + c * 8);
  }
}

// records the token picked for this step: int32 copy for the next embedding gather + column state[1] of out_ids; state[1]++
__global__ void decode_emit_kernel(const long* __restrict__ next_ids, int* __restrict__ tok32, long* __restrict__ out_ids, int* state,
                                   int B, int max_new) {
  const int b = threadIdx.x;
  const int t = state[1];
  if (b < B) {
    const long v = next_ids[b];
    tok32[b] = (int)v;
    if (t < max_new) out_ids[(long)b * max_new + t] = v;
  }
  __syncthreads();
  if (b == 0) state[1] = t + 1;
}

}  // namespace

extern "C" int lhrs_decode_emit(const long* next_ids, int* tok32, long* out_ids, int* state, int B, int max_new, void* stream) {
  LHRS_REQUIRE(B >= 1 && B <= 64, "decode_emit: B=%d", B);
  hipLaunchKernelGGL(decode_emit_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, next_ids, tok32, out_ids, state, B, max_new);
  LHRS_CHECK_LAUNCH("decode_emit");
  return 0;
}

// y[B, N] = x[B, K] . W[N, K]^T (+ residual[B, N]);  B <= 8, K % 8 == 0
static int gemv_chunk(const bf16_t* W, long ldw, const bf16_t* x, long ldx, const bf16_t* residual, long ldr, void* y, long ldy,
                      int B, int N, int K, int out_f32, hipStream_t s) {
  const dim3 grid(cdiv(N, 16)), blk(256);
  const size_t sm = (size_t)B * K * 2;
#define GEMV_CASE(NB)                                                                                                   \
  case NB:                                                                                                              \
    if (sm > 65536) (void)hipFuncSetAttribute((const void*)gemv_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); \
    hipLaunchKernelGGL((gemv_kernel<NB>), grid, blk, sm, s, W, ldw, x, ldx, residual, ldr, y, ldy, N, K, out_f32);       \
    break;
  switch (B) { GEMV_CASE(1) GEMV_CASE(2) GEMV_CASE(3) GEMV_CASE(4) GEMV_CASE(5) GEMV_CASE(6) GEMV_CASE(7) GEMV_CASE(8) }
#undef GEMV_CASE
  LHRS_CHECK_LAUNCH("gemv_bf16");
  return 0;
}

// y[B, N] = x[B, K] . W[N, K]^T (+ residual[B, N]);  B <= 8, K % 8 == 0.  Batches whose activations exceed the LDS are split.
extern "C" int lhrs_gemv_bf16(const void* W, long ldw, const void* x, long ldx, const void* residual, long ldr, void* y, long ldy,
                              int B, int N, int K, int out_f32, void* stream) {
  LHRS_REQUIRE(B >= 1 && B <= 8 && N > 0 && K % 8 == 0 && ldw % 8 == 0 && ldx % 8 == 0, "gemv: B=%d N=%d K=%d", B, N, K);
  const int bmax = (int)((152L * 1024) / ((long)K * 2));
  LHRS_REQUIRE(bmax >= 1, "gemv: one activation vector does not fit LDS (K=%d)", K);
  const long esz = out_f32 ? 4 : 2;
  for (int b0 = 0; b0 < B; b0 += bmax) {
    const int nb = B - b0 < bmax ? B - b0 : bmax;
    if (gemv_chunk((const bf16_t*)W, ldw, (const bf16_t*)x + b0 * ldx, ldx, residual ? (const bf16_t*)residual + b0 * ldr : nullptr, ldr,
                   (char*)y + b0 * ldy * esz, ldy, nb, N, K, out_f32, (hipStream_t)stream))
      return -1;
  }
  return 0;
}

extern "C" int lhrs_decode_advance(int* state, int* desc, int* pos, int B, int max_ctx, int step_inc, void* stream) {
  LHRS_REQUIRE(B >= 1 && B <= 64, "decode_advance: B=%d", B);
  hipLaunchKernelGGL(decode_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, desc, pos, B, max_ctx, step_inc);
  LHRS_CHECK_LAUNCH("decode_advance");
  return 0;
}

extern "C" int lhrs_kv_append(const void* qkv, long ld, void* kcache, void* vcache, const int* pos, int B, int d, int max_ctx,
                              void* stream) {
  LHRS_REQUIRE(B >= 1 && d % 8 == 0, "kv_append: B=%d d=%d", B, d);
  hipLaunchKernelGGL(kv_append_kernel, dim3(cdiv(d / 8, 256), B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, ld,
                     (bf16_t*)kcache, (bf16_t*)vcache, pos, B, d, max_ctx);
  LHRS_CHECK_LAUNCH("kv_append");
  return 0;
}

// ---- thin hipGraph wrappers: capture the launches enqueued on `stream` between begin/end, replay with launch ----
extern "C" int lhrs_graph_begin(void* stream) {
  hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) LHRS_FAIL("graph_begin: %s", hipGetErrorString(e));
  return 0;
}
extern "C" int lhrs_graph_end(void* stream, void** exec_out) {
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);
  if (e != hipSuccess || !g) LHRS_FAIL("graph_end: %s", hipGetErrorString(e));
  hipGraphExec_t ex = nullptr;
  e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) LHRS_FAIL("graph_end: instantiate: %s", hipGetErrorString(e));
  *exec_out = (void*)ex;
  return 0;
}
extern "C" int lhrs_graph_launch(void* exec, void* stream) {
  hipError_t e = hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)stream);
  if (e != hipSuccess) LHRS_FAIL("graph_launch: %s", hipGetErrorString(e));
  return 0;
}
extern "C" int lhrs_graph_destroy(void* exec) {
  if (exec) (void)hipGraphExecDestroy((hipGraphExec_t)exec);
  return 0;
}
