// bf16 MFMA GEMM for gfx950:  C[M,N] = act(alpha * A[M,K] . B[N,K]^T + bias[N]) + residual[M,N]
//
// This is the kernel that bounds the whole stage-1 step (~96 % of the step's FLOPs are
// LLaMA linears).  Every linear of the reference (`nn.Linear` in HF CLIP / LLaMA and in
// lhrs/models/common_arch.py:276-295) is "NT": activations [M,K] row-major times a weight stored [N,K]
// row-major.  Backward dX through the *frozen* LLaMA uses a pre-transposed copy of each weight (resident in
// HBM; 288 GB makes the second copy free), so dX is NT as well; dW for the projector is NT over transposed
// activations.  One layout, one kernel family.
//
// Structure (CDNA4):
//   * 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 4x4 fragments of v_mfma_f32_16x16x32_bf16.
//   * HBM -> LDS by direct-to-LDS DMA (global_load_lds_dwordx4): no VGPR round trip.  The LDS image of a
//     wave-instruction is lane-linear, so the bank-conflict swizzle (16-B chunk ^= row&7) is applied to the
//     per-lane SOURCE address and to the ds_read_b128 address (both sides, same involution).
//   * two LDS stages; the DMA for tile t+1 is in flight while tile t is multiplied.
//   * operands are fed to the MFMA swapped (weight fragment as A, activation fragment as B) so that each
//     lane ends up with 4 consecutive n of one m: 8-byte bf16 / 16-byte f32 row-contiguous stores, and
//     bias / residual become 8-byte vector loads.
//   * block id -> tile map is XCD-aware (block b runs on XCD b%8; each XCD gets a contiguous run of tiles,
//     rastered in groups of 8 tile-rows) so neighbouring tiles share A/B panels in one XCD's L2.
#include "common.h"

namespace {

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const bf16_t* bias;
  const bf16_t* res;
  int M, N, K;
  int lda, ldb, ldc, ldr;
  float alpha;
  int act;       // 0 none, 1 quick_gelu, 2 gelu(erf), 3 silu
  int out_f32;   // 0 -> bf16 C, 1 -> f32 C
  int accum;     // f32 only: C += result
  int tilesM, tilesN;
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void tile_coords(const GemmArgs& g, int& tm, int& tn) {
  const int bid = blockIdx.x, nblk = gridDim.x;
  const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  constexpr int GM = 8;
  const int per_group = GM * g.tilesN;
  const int group = lin / per_group;
  const int first_m = group * GM;
  const int gsize = min(g.tilesM - first_m, GM);
  const int in_g = lin - group * per_group;
  tm = first_m + in_g % gsize;
  tn = in_g / gsize;
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == 1) return quick_gelu(v);
  if (act == 2) return gelu_erf(v);
  if (act == 3) return silu(v);
  return v;
}

// Epilogue for one lane's 4 consecutive n of row m.
template <int ACT>
__device__ __forceinline__ void store4(const GemmArgs& g, int m, int n, f32x4 acc) {
  float v[4] = {acc[0] * g.alpha, acc[1] * g.alpha, acc[2] * g.alpha, acc[3] * g.alpha};
  if (g.bias) {
    const uint2 b = *reinterpret_cast<const uint2*>(g.bias + n);
    v[0] += bflo(b.x); v[1] += bfhi(b.x); v[2] += bflo(b.y); v[3] += bfhi(b.y);
  }
  if (ACT) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = apply_act(v[i], ACT);
  }
  if (g.res) {
    const uint2 r = *reinterpret_cast<const uint2*>(g.res + (long)m * g.ldr + n);
    v[0] += bflo(r.x); v[1] += bfhi(r.x); v[2] += bflo(r.y); v[3] += bfhi(r.y);
  }
  if (g.out_f32) {
    float* c = reinterpret_cast<float*>(g.C) + (long)m * g.ldc + n;
    if (g.accum) {
      const float4 o = *reinterpret_cast<const float4*>(c);
      v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
    }
    *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    bf16_t* c = reinterpret_cast<bf16_t*>(g.C) + (long)m * g.ldc + n;
    *reinterpret_cast<uint2*>(c) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
  }
}

// ------------------------------------------------------------------------------------------------
// 128x128x64, 4 waves, 2-stage direct-to-LDS pipeline.
// ------------------------------------------------------------------------------------------------
template <int WM_FR, int WN_FR, int ACT>  // fragments per wave in m / n (4,4 -> 128x128 tile; 2,4 -> 64x128; 2,2 -> 64x64)
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs g) {
  constexpr int BM = WM_FR * 32, BN = WN_FR * 32, BK = 64;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  int tm, tn;
  tile_coords(g, tm, tn);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- DMA source addresses: one wave-instruction moves 8 rows x 128 B (1 KiB) ----
  constexpr int A_INSTR = BM / 32;  // per wave
  constexpr int B_INSTR = BN / 32;
  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ lrow;  // logical 16-B chunk this lane fetches (swizzle on the source side)
  const bf16_t* a_src[A_INSTR];
  const bf16_t* b_src[B_INSTR];
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    int row = tm * BM + (wave * A_INSTR + i) * 8 + lrow;
    row = min(row, g.M - 1);
    a_src[i] = g.A + (long)row * g.lda + lchunk * 8;
  }
#pragma unroll
  for (int i = 0; i < B_INSTR; ++i) {
    int row = tn * BN + (wave * B_INSTR + i) * 8 + lrow;
    row = min(row, g.N - 1);
    b_src[i] = g.B + (long)row * g.ldb + lchunk * 8;
  }

  auto issue = [&](int stage, int kt) {
    char* sa = smem + stage * STAGE;
    char* sb = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + (long)kt * BK),
                                       (lptr_t)(sa + (wave * A_INSTR + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(b_src[i] + (long)kt * BK),
                                       (lptr_t)(sb + (wave * B_INSTR + i) * 1024), 16, 0, 0);
  };

  // ---- fragment addressing ----
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const int a_row0 = wm * (WM_FR * 16) + fr;  // + mi*16
  const int b_row0 = wn * (WN_FR * 16) + fr;
  int koff[2];
  koff[0] = ((0 * 4 + fg) ^ (fr & 7)) * 16;
  koff[1] = ((1 * 4 + fg) ^ (fr & 7)) * 16;

  f32x4 acc[WM_FR][WN_FR];
#pragma unroll
  for (int i = 0; i < WM_FR; ++i)
#pragma unroll
    for (int j = 0; j < WN_FR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK;
  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) issue((kt + 1) & 1, kt + 1);
    const char* sa = smem + (kt & 1) * STAGE;
    const char* sb = sa + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[WM_FR], bfr[WN_FR];
#pragma unroll
      for (int mi = 0; mi < WM_FR; ++mi)
        af[mi] = *reinterpret_cast<const bf16x8*>(sa + (a_row0 + mi * 16) * 128 + koff[kk]);
#pragma unroll
      for (int ni = 0; ni < WN_FR; ++ni)
        bfr[ni] = *reinterpret_cast<const bf16x8*>(sb + (b_row0 + ni * 16) * 128 + koff[kk]);
#pragma unroll
      for (int mi = 0; mi < WM_FR; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN_FR; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);
    }
  }

  // ---- epilogue: lane holds C[m = .. + fr][n = .. + fg*4 + 0..3] ----
#pragma unroll
  for (int mi = 0; mi < WM_FR; ++mi) {
    const int m = tm * BM + wm * (WM_FR * 16) + mi * 16 + fr;
    if (m >= g.M) continue;
#pragma unroll
    for (int ni = 0; ni < WN_FR; ++ni) {
      const int n = tn * BN + wn * (WN_FR * 16) + ni * 16 + fg * 4;
      if (n >= g.N) continue;
      store4<ACT>(g, m, n, acc[mi][ni]);
    }
  }
}

}  // namespace

// ---- optional live timing of the GEMM launches (bench.py roofline leg) -----------------------------------
// When enabled, every launch of the 128x128 kernel is bracketed by HIP events ON THE LAUNCH STREAM until the
// event pool is used up; lhrs_gemm_profile_read synchronises those events and returns summed time and flops.
namespace {
struct GemmProf {
  bool on = false;
  int cap = 0, used = 0;
  hipEvent_t* ev = nullptr;   // 2 * cap
  double* flops = nullptr;    // cap
  double total_flops_all = 0; // every GEMM launch while enabled (sampled or not)
  long launches_all = 0;
} g_prof;
}  // namespace

extern "C" int lhrs_gemm_profile_enable(int max_samples) {
  if (g_prof.ev) {
    for (int i = 0; i < 2 * g_prof.cap; ++i) (void)hipEventDestroy(g_prof.ev[i]);
    delete[] g_prof.ev; delete[] g_prof.flops;
    g_prof.ev = nullptr; g_prof.flops = nullptr;
  }
  g_prof.on = max_samples > 0; g_prof.cap = max_samples > 0 ? max_samples : 0; g_prof.used = 0;
  g_prof.total_flops_all = 0; g_prof.launches_all = 0;
  if (g_prof.on) {
    g_prof.ev = new hipEvent_t[2 * g_prof.cap];
    g_prof.flops = new double[g_prof.cap];
    for (int i = 0; i < 2 * g_prof.cap; ++i)
      if (hipEventCreate(&g_prof.ev[i]) != hipSuccess) LHRS_FAIL("gemm_profile_enable: hipEventCreate failed");
  }
  return 0;
}

// out[0] = sampled launches, out[1] = their summed duration (ms), out[2] = their summed flops,
// out[3] = all GEMM launches while enabled, out[4] = flops of all of them
extern "C" int lhrs_gemm_profile_read(double* out) {
  double ms = 0, fl = 0;
  for (int i = 0; i < g_prof.used; ++i) {
    float t = 0;
    if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) LHRS_FAIL("gemm_profile_read: event sync failed");
    if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) LHRS_FAIL("gemm_profile_read: elapsed failed");
    ms += t; fl += g_prof.flops[i];
  }
  out[0] = g_prof.used; out[1] = ms; out[2] = fl; out[3] = (double)g_prof.launches_all; out[4] = g_prof.total_flops_all;
  return 0;
}

// C ABI ------------------------------------------------------------------------------------------
extern "C" int lhrs_gemm_bf16_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N,
                                 int K, const void* bias, const void* residual, int ldr, int act, int out_f32,
                                 int accumulate, float alpha, void* stream) {
  LHRS_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
  LHRS_REQUIRE(K % 64 == 0, "gemm: K=%d must be a multiple of 64 (zero-pad the reduction dim)", K);
  LHRS_REQUIRE(N % 4 == 0, "gemm: N=%d must be a multiple of 4", N);
  LHRS_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, "gemm: lda=%d ldb=%d must be multiples of 8 (16-B rows)", lda, ldb);
  LHRS_REQUIRE(ldc % 4 == 0 && (residual == nullptr || ldr % 4 == 0), "gemm: ldc/ldr must be multiples of 4");
  LHRS_REQUIRE(lda >= K && ldb >= K && ldc >= N, "gemm: leading dims too small");
  LHRS_REQUIRE(!accumulate || out_f32, "gemm: accumulate needs f32 output");
  LHRS_REQUIRE(act >= 0 && act <= 3, "gemm: unknown activation %d", act);
  GemmArgs g;
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = C;
  g.bias = (const bf16_t*)bias; g.res = (const bf16_t*)residual;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr;
  g.alpha = alpha; g.act = act; g.out_f32 = out_f32; g.accum = accumulate;
  hipStream_t s = (hipStream_t)stream;
  // Tile choice: fill the 256 CUs.  Small problems (projector, ViT at small batch) take smaller tiles.
  const long t128 = (long)cdiv(M, 128) * cdiv(N, 128);
  const long t64x128 = (long)cdiv(M, 64) * cdiv(N, 128);
#define LAUNCH_TILE(WM, WN)                                                                            \
  do {                                                                                                 \
    g.tilesM = cdiv(M, WM * 32); g.tilesN = cdiv(N, WN * 32);                                          \
    const dim3 grid(g.tilesM * g.tilesN), blk(256);                                                    \
    switch (act) {                                                                                     \
      case 0: hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, 0>), grid, blk, 0, s, g); break;              \
      case 1: hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, 1>), grid, blk, 0, s, g); break;              \
      case 2: hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, 2>), grid, blk, 0, s, g); break;              \
      default: hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, 3>), grid, blk, 0, s, g); break;             \
    }                                                                                                  \
  } while (0)
  const bool big = t128 >= 384;
  int slot = -1;
  if (g_prof.on) {
    g_prof.launches_all++; g_prof.total_flops_all += 2.0 * M * N * K;
    if (big && g_prof.used < g_prof.cap) {
      slot = g_prof.used++;
      g_prof.flops[slot] = 2.0 * M * N * K;
      (void)hipEventRecord(g_prof.ev[2 * slot], s);
    }
  }
  if (big) LAUNCH_TILE(4, 4);
  else if (t64x128 >= 256) LAUNCH_TILE(2, 4);
  else LAUNCH_TILE(2, 2);
#undef LAUNCH_TILE
  if (slot >= 0) (void)hipEventRecord(g_prof.ev[2 * slot + 1], s);
  LHRS_CHECK_LAUNCH("gemm_bf16_nt");
  return 0;
}
