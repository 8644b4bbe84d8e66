// bf16 MFMA GEMM for gfx950:  C[M,N] = act(alpha * A[M,K] . B[N,K]^T + bias[N]) + residual[M,N]
//
// This is the kernel that bounds the whole stage-1 step (~96 % of the step's FLOPs are
// LLaMA linears).  Every linear of the reference (`nn.Linear` in HF CLIP / LLaMA and in
// lhrs/models/common_arch.py:276-295) is "NT": activations [M,K] row-major times a weight stored [N,K]
// row-major.  Backward dX through the *frozen* LLaMA uses a pre-transposed copy of each weight (resident in
// HBM; 288 GB makes the second copy free), so dX is NT as well; dW for the projector is NT over transposed
// activations.  One layout, one kernel family.
//
// Kernel family (CDNA4), chosen by tile count in gemm_launch:
//   * gemm_nt_256s_kernel<ACT, EPI, K2P> - THE dominant kernel (every LLaMA linear at training batch sizes): 256x256x64 tile, 16 waves (4x4 of
//     64x64 = 4x4 fragments of v_mfma_f32_16x16x32_bf16 - the dense bf16 shape that costs the least power per FLOP), two 64 KiB LDS stages
//     filled by direct-to-LDS DMA (global_load_lds_dwordx4, no VGPR round trip), one barrier per stage, PERSISTENT over tiles (one workgroup
//     per CU; the next tile's first stage is fetched under the epilogue), fused epilogues: bias / activation / residual / dropout mask
//     (EPI 0), SwiGLU forward / backward (EPI 1 / 2), RoPE (EPI 3).
//   * gemm_nt_256p_kernel - K % 64 != 0 fallback of the above (BK = 32, 4-deep ring, 8 waves).
//   * gemm_nt_kernel<WM, WN> - 128x128 / 64x128 / 64x64 tiles, 4 waves, for the small products of the ViT and the projector.
//   * e4m3 siblings (gemm_fp8_256_kernel, gemm_fp8_small_kernel) for the 8-bit frozen base of stages 2/3.
// Common to all:
//   * the LDS image of a DMA wave-instruction is lane-linear, so the bank-conflict swizzle (16-B chunk ^= f(row)) is applied to the
//     per-lane SOURCE address and to the ds_read_b128 address (both sides, same involution);
//   * operands are fed to the MFMA swapped (weight fragment as A, activation fragment as B) so that each lane ends up with consecutive n
//     of one m: row-contiguous stores, and bias / residual become vector loads;
//   * block id -> tile map is XCD-aware (block b runs on XCD b % 8; each XCD gets a contiguous run of tiles, rastered in groups of 8 tile
//     rows) so the 32 tiles an XCD works on at a time share 8 A panels and 4 B panels in its L2.
#include "common.h"
#include <array>
#include <map>
#include <type_traits>

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
  // optional second operand pair, reduced in the same k-loop: C = alpha * (A.B^T + A2.B2^T) ...  (fused LoRA: A2 = s*X*A_lora^T,
  // B2 = B_lora).  K2 = 0 disables it.
  const bf16_t* A2;
  const bf16_t* B2;
  int lda2, ldb2, K2;
  // fused SwiGLU epilogues of the 16-wave 256x256 kernel (LLaMA MLP, HF LlamaMLP: down(silu(gate(x)) * up(x))):
  //   epi 1: B = [gate; up] weight [2*ff, K]; tile tn holds gate AND up of columns [tn*128, +128) (wave wn: 32 gate + 32 up);
  //          C = gate|up [M, 2*ff] in the usual layout, aux_out = silu(gate) * up [M, ff]
  //   epi 2: A.B^T = d_act [M, ff]; aux = gate|up [M, 2*ff]; C = d(gate|up) [M, 2*ff] (may alias aux)
  //   epi 5: epi 2, and row_dot[(tn * 4 + wn) * M + m] = sum over the wave's 64 columns of d(gate|up) * gate|up (fp32) - the partial sums of
  //          the row dot product <d(gate|up), gate|up> that the RMSNorm backward behind the NEXT product needs (epi 4)
  //   epi 4: RMSNorm backward in the epilogue (HF LlamaRMSNorm, no recompute): A.B^T = dh = d loss / d (normalised row) [M, N], N = the whole
  //          normalised width; C = rstd * (w o dh) - x * (rstd^2 * s / N) + add with x = aux [M, ld_aux], add = res [M, ldr] (may be null),
  //          w = bias [N], rstd = sa [M], s = sb [M] = sum_j dh_j w_j x_j * rstd (= <d(gate|up), gate|up> of the linear the norm feeds)
  int epi, ff;
  float* row_dot;
  const bf16_t* aux;
  bf16_t* aux_out;
  long ld_aux;
  // e4m3 operands (gemm_fp8_256_kernel): per-row dequantisation scales of A (length M) and B (length N)
  const float* sa;
  const float* sb;
  // dropout mask on alpha * A.B^T BEFORE bias / residual (LoRA dX path: dx = dy.W + mask * (U.A) / (1 - p)); drop_thresh = 0 -> off
  float drop_scale;
  unsigned drop_seed, drop_thresh;
  // EPI 3 (fused RoPE of the q / k heads of a qkv projection, head_dim 128): fp32 cos / sin tables [pos][64], position of row m =
  // m % rope_mod + rope_pos0, columns [0, rope_cols) are rotated (rope_cols % 256 == 0), the rest stored as computed
  const float* rope_cos;
  const float* rope_sin;
  int rope_mod, rope_pos0, rope_cols;
  // split-K of the small-tile kernel (skinny-N products): block (x, y) reduces K-slice y of length ksplit into f32 slab y of C
  int ksplit;
  // 8-bit kernels: bf16 columns of the second pair whose count is only known on the device (LLM.int8 outlier columns; a multiple of 64),
  // appended behind the K2 host-known ones (nullptr = none)
  const int* k2_dev;
  // persistent 16-wave kernel: the tiles one launch walks are the raster positions [0, ntiles) (ntiles = tilesM * tilesN unless a stream-K
  // launch takes the positions behind them)
  int ntiles;
  // stream-K launch (gemm_nt_256s_kernel<..., SK = true>): the raster positions [sk_tile0, sk_tile0 + sk_tiles) - the tiles of the last,
  // partial round of the CUs - are cut along k into sk_units contiguous ranges of stages, one per workgroup; a workgroup whose range starts
  // inside a tile writes its fp32 accumulators to slab `unit` of sk_ws and raises sk_flags[unit]; the workgroup that holds the tile's stage
  // 0 adds the slabs of the others and runs the epilogue (the caller owns sk_ws / sk_flags; the flags are zero between launches)
  int sk_tile0, sk_tiles, sk_units;
  float* sk_ws;
  int* sk_flags;
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// tile `bid` of `nblk` (the order tiles are dispatched in) -> (tm, tn): consecutive bids alternate over the 8 XCDs (bid & 7), so each XCD
// gets a contiguous run of the m-fastest raster grouped by 8 tile rows - the 32 tiles an XCD works on at a time share 8 A and 4 B panels
// raster position `lin` -> (tm, tn): m-fastest inside groups of 8 tile rows
__device__ __forceinline__ void tile_coords_raster(const GemmArgs& g, int lin, int& tm, int& tn) {
  constexpr int GM = 8;
  const int per_group = GM * g.tilesN;
  const int group = lin / per_group;
  const int first_m = group * GM;
  const int gsize = min(g.tilesM - first_m, GM);
  const int in_g = lin - group * per_group;
  tm = first_m + in_g % gsize;
  tn = in_g / gsize;
}
__device__ __forceinline__ void tile_coords_lin(const GemmArgs& g, int bid, int nblk, int& tm, int& tn) {
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
__device__ __forceinline__ void tile_coords(const GemmArgs& g, int& tm, int& tn) { tile_coords_lin(g, blockIdx.x, gridDim.x, tm, tn); }

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
  if (g.drop_thresh) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = drop_keep(g.drop_seed, (long)m * g.N + n + i, g.drop_thresh) ? v[i] * g.drop_scale : 0.f;
  }
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
  if (g.ksplit > 0) {  // this block's K-slice and its private f32 output slab
    const int sp = blockIdx.y;
    g.A += (long)sp * g.ksplit; g.B += (long)sp * g.ksplit;
    g.C = reinterpret_cast<float*>(g.C) + (long)sp * g.M * g.ldc;
    g.K = min(g.ksplit, g.K - sp * g.ksplit);
  }

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


// ------------------------------------------------------------------------------------------------
// 256x256 tile, 8 waves (2 x 4), BK = 32 stages in a 4-deep LDS ring (128 KiB), DMA two stages ahead; the ds_read of the NEXT k-step is in
// flight while the MFMAs of the current k-step issue, so LDS latency never sits in front of the matrix pipe.
// Fragment reads are inline asm (hipcc would otherwise place an lgkmcnt(0) right before every consumer, i.e.
// behind the reads just issued); every wait is an explicit lgkmcnt(0) placed BEFORE the next batch of reads, so
// it only ever covers reads issued one 8-MFMA block (>= 256 cycles) earlier.
//   per iteration i (stage i+1 is published by the barrier):
//     vmcnt(4) ; barrier ; DMA(stage i+3) ; wait ; read k0(i+1)->set0 ; 8 MFMA set1=k1(i) ; wait ; read k1(i+1)->set1 ;
//     8 MFMA set0
//   WAR on LDS: buffer (i+3)&3 held stage i-1, whose last reads (k1) completed before MFMA k1(i-1) issued, which every
//   wave did before reaching this iteration's barrier.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_read6(bf16x8 (&a)[4], bf16x8 (&b)[2], unsigned a_addr, unsigned b_addr) {
  asm volatile(
      "ds_read_b128 %0, %6\n\t"
      "ds_read_b128 %1, %6 offset:2048\n\t"
      "ds_read_b128 %2, %7\n\t"
      "ds_read_b128 %3, %7 offset:2048\n\t"
      "ds_read_b128 %4, %7 offset:4096\n\t"
      "ds_read_b128 %5, %7 offset:6144"
      : "=&v"(b[0]), "=&v"(b[1]), "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3])
      : "v"(b_addr), "v"(a_addr));
}
__device__ __forceinline__ void lds_wait6(bf16x8 (&a)[4], bf16x8 (&b)[2]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
  __builtin_amdgcn_sched_barrier(0);
}

#define LDS_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))

template <int ACT>
__global__ __launch_bounds__(512, 2) void gemm_nt_256p_kernel(GemmArgs g) {
  constexpr int BM = 256, BN = 256, BK = 32, NS = 4;
  constexpr int A_BYTES = BM * BK * 2, STAGE = A_BYTES + BN * BK * 2;
  __shared__ __attribute__((aligned(16))) char smem[NS * STAGE];

  int tm, tn;
  tile_coords(g, tm, tn);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int lrow = lane >> 2;
  const int lchunk = (lane & 3) ^ ((lane >> 4) & 3);
  const bf16_t* src[4];
  int dst_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = wave * 4 + j;
    if (idx < 16) {
      const int row = min(tm * BM + idx * 16 + lrow, g.M - 1);
      src[j] = g.A + (long)row * g.lda + lchunk * 8;
      dst_off[j] = idx * 1024;
    } else {
      const int row = min(tn * BN + (idx - 16) * 16 + lrow, g.N - 1);
      src[j] = g.B + (long)row * g.ldb + lchunk * 8;
      dst_off[j] = A_BYTES + (idx - 16) * 1024;
    }
  }
  const bf16_t* src2[4];
  const int nk1 = g.K / BK;
  if (g.K2 > 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = wave * 4 + j;
      if (idx < 16) src2[j] = g.A2 + (long)min(tm * BM + idx * 16 + lrow, g.M - 1) * g.lda2 + lchunk * 8;
      else src2[j] = g.B2 + (long)min(tn * BN + (idx - 16) * 16 + lrow, g.N - 1) * g.ldb2 + lchunk * 8;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) src2[j] = src[j];
  }
  auto issue1 = [&](int kt, int j) {
    const bf16_t* p = kt < nk1 ? src[j] + (long)kt * BK : src2[j] + (long)(kt - nk1) * BK;
    __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(smem + ((unsigned)kt % NS) * STAGE + dst_off[j]), 16, 0, 0);
  };

  const int wm = wave >> 2, wn = wave & 3;
  const int fr = lane & 31, fh = lane >> 5;
  const int sw = (lane >> 2) & 3;
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  const unsigned a_base = lds0 + (wm * 128 + fr) * 64;
  const unsigned b_base = lds0 + A_BYTES + (wn * 64 + fr) * 64;
  const unsigned koff0 = ((0 * 2 + fh) ^ sw) * 16, koff1 = ((1 * 2 + fh) ^ sw) * 16;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  bf16x8 a0[4], b0[2], a1[4], b1[2];  // set0 = k-step 0 fragments, set1 = k-step 1 fragments
#define MF(A_, B_, mi, ni) \
  acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B_[ni], A_[mi], acc[mi][ni], 0, 0, 0); __builtin_amdgcn_sched_barrier(0);
#define MFMA8(A_, B_) \
  MF(A_, B_, 0, 0) MF(A_, B_, 0, 1) MF(A_, B_, 1, 0) MF(A_, B_, 1, 1) MF(A_, B_, 2, 0) MF(A_, B_, 2, 1) MF(A_, B_, 3, 0) MF(A_, B_, 3, 1)

  const int nk = (g.K + g.K2) / BK;  // >= 3 (host guarantees)
#pragma unroll
  for (int s = 0; s < NS - 2; ++s)
#pragma unroll
    for (int j = 0; j < 4; ++j) issue1(s, j);
  // NOT a counted vmcnt(4 / 8): the LDS-DMA pieces of one wave do not retire in issue order (round 3: a counted wait on the older of two
  // stages in flight produced wrong tiles in gemm_nt_144s_kernel, 10 of 12 repetitions).  This fallback kernel therefore waits for everything
  // it has issued; its ring still spreads the issue, but a stage has one stage time to land, like in the two-buffer kernels
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int j = 0; j < 4; ++j) issue1(NS - 2, j);
  lds_read6(a0, b0, a_base + koff0, b_base + koff0);
  lds_read6(a1, b1, a_base + koff1, b_base + koff1);
  asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(b0[0]), "+v"(b0[1]), "+v"(a0[0]), "+v"(a0[1]), "+v"(a0[2]), "+v"(a0[3]));
  __builtin_amdgcn_sched_barrier(0);
  MFMA8(a0, b0)

  // One iteration = [publish stage kt+1] + 16 MFMAs (k1 of stage kt, k0 of stage kt+1); behind every MFMA sits one filler:
  // a ds_read of the next k-step's fragments or one DMA piece of stage kt+3.
  auto body = [&](auto dma_c, auto vm_c, int kt) {
    constexpr bool DMA = decltype(dma_c)::value;
    constexpr int VM = decltype(vm_c)::value;  // DMA pieces that may still be in flight: the stages after kt+1
    (void)VM;  // see the prologue: counted waits on LDS-DMA are not safe
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned so = ((unsigned)(kt + 1) % NS) * STAGE;
    const unsigned aa0 = a_base + so + koff0, ba0 = b_base + so + koff0;
    const unsigned aa1 = a_base + so + koff1, ba1 = b_base + so + koff1;
    lds_wait6(a1, b1);
    MF(a1, b1, 0, 0) LDS_RD(b0[0], ba0, 0);    __builtin_amdgcn_sched_barrier(0);
    MF(a1, b1, 0, 1) LDS_RD(b0[1], ba0, 2048); __builtin_amdgcn_sched_barrier(0);
    MF(a1, b1, 1, 0) LDS_RD(a0[0], aa0, 0);    __builtin_amdgcn_sched_barrier(0);
    MF(a1, b1, 1, 1) LDS_RD(a0[1], aa0, 2048); __builtin_amdgcn_sched_barrier(0);
    MF(a1, b1, 2, 0) LDS_RD(a0[2], aa0, 4096); __builtin_amdgcn_sched_barrier(0);
    MF(a1, b1, 2, 1) LDS_RD(a0[3], aa0, 6144); __builtin_amdgcn_sched_barrier(0);
    MF(a1, b1, 3, 0) if constexpr (DMA) issue1(kt + NS - 1, 0); __builtin_amdgcn_sched_barrier(0);
    MF(a1, b1, 3, 1) if constexpr (DMA) issue1(kt + NS - 1, 1); __builtin_amdgcn_sched_barrier(0);
    lds_wait6(a0, b0);
    MF(a0, b0, 0, 0) LDS_RD(b1[0], ba1, 0);    __builtin_amdgcn_sched_barrier(0);
    MF(a0, b0, 0, 1) LDS_RD(b1[1], ba1, 2048); __builtin_amdgcn_sched_barrier(0);
    MF(a0, b0, 1, 0) LDS_RD(a1[0], aa1, 0);    __builtin_amdgcn_sched_barrier(0);
    MF(a0, b0, 1, 1) LDS_RD(a1[1], aa1, 2048); __builtin_amdgcn_sched_barrier(0);
    MF(a0, b0, 2, 0) LDS_RD(a1[2], aa1, 4096); __builtin_amdgcn_sched_barrier(0);
    MF(a0, b0, 2, 1) LDS_RD(a1[3], aa1, 6144); __builtin_amdgcn_sched_barrier(0);
    MF(a0, b0, 3, 0) if constexpr (DMA) issue1(kt + NS - 1, 2); __builtin_amdgcn_sched_barrier(0);
    MF(a0, b0, 3, 1) if constexpr (DMA) issue1(kt + NS - 1, 3); __builtin_amdgcn_sched_barrier(0);
  };
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;
  using V8 = std::integral_constant<int, 8>;
  using V4 = std::integral_constant<int, 4>;
  using V0 = std::integral_constant<int, 0>;
  if constexpr (NS == 4) {
    for (int kt = 0; kt < nk - 3; ++kt) body(T_{}, V4{}, kt);  // steady state: DMA NS-1 stages ahead, vmcnt never 0
    body(F_{}, V4{}, nk - 3);                                  // stage nk-1 is already in flight
  } else {
    for (int kt = 0; kt < nk - 4; ++kt) body(T_{}, V8{}, kt);
    body(F_{}, V8{}, nk - 4);
    body(F_{}, V4{}, nk - 3);
  }
  body(F_{}, V0{}, nk - 2);
  lds_wait6(a1, b1);
  MFMA8(a1, b1)
#undef MFMA8
#undef MF

  if (g.out_f32) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int m = tm * BM + wm * 128 + mi * 32 + fr;
      if (m >= g.M) continue;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = tn * BN + wn * 64 + ni * 32 + q * 8 + fh * 4;
          if (n >= g.N) continue;
          store4<ACT>(g, m, n, f32x4{acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]});
        }
    }
    return;
  }
  // ---- bf16 epilogue through LDS: each wave transposes its own 128 x 64 sub-tile in a private 16 KiB region of the
  // (now idle) ring so that global stores are 16 B per lane, 128 contiguous bytes per row (full cache lines).
  //   stage value = bf16(act(alpha*acc + bias));  final = bf16(stage + residual)  (the reference's rounding order)
  __builtin_amdgcn_s_barrier();  // every wave is done reading the last stages
  char* reg = smem + wave * 16384;
  {
    float bias_v[2][4][4];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = min(tn * BN + wn * 64 + ni * 32 + q * 8 + fh * 4, g.N - 4);
        uint2 bb = make_uint2(0, 0);
        if (g.bias) bb = *reinterpret_cast<const uint2*>(g.bias + n);
        bias_v[ni][q][0] = bflo(bb.x); bias_v[ni][q][1] = bfhi(bb.x); bias_v[ni][q][2] = bflo(bb.y); bias_v[ni][q][3] = bfhi(bb.y);
      }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int row = mi * 32 + fr;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = acc[mi][ni][4 * q + i] * g.alpha + bias_v[ni][q][i];
            if (ACT) v[i] = apply_act(v[i], ACT);
          }
          const int u = ni * 8 + q * 2 + fh;  // 8-byte unit inside the 128-B row
          *reinterpret_cast<uint2*>(reg + row * 128 + ((u ^ ((row & 7) << 1)) << 3)) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
        }
    }
  }
  // a wave reads back only what it wrote itself: its own LDS writes are visible to it once lgkmcnt drains
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  {
    const int rsub = lane >> 3, c = lane & 7;
    const int n = tn * BN + wn * 64 + c * 8;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = i * 8 + rsub;
      const int m = tm * BM + wm * 128 + row;
      uint4 val = *reinterpret_cast<const uint4*>(reg + row * 128 + ((c ^ (row & 7)) << 4));
      if (m < g.M && n < g.N) {
        if (g.res) {
          const uint4 r = *reinterpret_cast<const uint4*>(g.res + (long)m * g.ldr + n);
          val.x = pack2bf(bflo(val.x) + bflo(r.x), bfhi(val.x) + bfhi(r.x));
          val.y = pack2bf(bflo(val.y) + bflo(r.y), bfhi(val.y) + bfhi(r.y));
          val.z = pack2bf(bflo(val.z) + bflo(r.z), bfhi(val.z) + bfhi(r.z));
          val.w = pack2bf(bflo(val.w) + bflo(r.w), bfhi(val.w) + bfhi(r.w));
        }
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(g.C) + (long)m * g.ldc + n) = val;
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// The 16-wave 256x256 kernel (4x4 waves of 64x64, 128 VGPRs -> 4 waves per SIMD: while one wave sits in a DMA issue or at the barrier three
// others feed the SIMD's MFMA pipe).  MFMA shape: v_mfma_f32_16x16x32_bf16, not the 32x32x16 this kernel used through round 2's first half
// (gemm_nt_256r_kernel, in the history).  Per FLOP the 16x16x32 instruction moves half the accumulator data (4 accumulator VGPRs per 16 KFLOP
// instead of 16 per 32 KFLOP) and under the 1400 W package cap a loop of nothing but MFMAs on random bf16 sustains 2.00 PFLOP/s with it
// against 1.78 (tools/mfma_peak.hip) - the GEMM is power-bound, so the cheaper instruction is the faster one: +3.4 .. +4.9 % on the LLaMA
// shapes, +3 % on the step, shader clock 1.87 -> 2.06 GHz under load, results bit-identical (profiles/r02_gemm_mfma_shape_ab.txt).
// A wave's 64x64 is 4x4 fragments of 16x16; one k-block = 32 k = 16 MFMAs, a stage = 2 k-blocks.  Fragment registers are single-buffered and
// ROLL - a fragment is re-read for the next block right behind the last MFMA that uses it (32 VGPRs; double-buffering 8 fragments of 4 VGPRs
// does not fit beside 64 accumulators) - and the MFMA order walks the 2x2 quadrants of the fragment grid (block 0: Q00 Q01 Q11 Q10, block 1:
// Q01 Q00 Q10 Q11) so that every fragment has >= 7 MFMA slots between its re-read and its next use; the waits are counted (LDS returns in
// order).  Order, re-reads, waits, barrier and DMA slots are generated: tools/gen_gemm16_sched.py -> gemm_256s_sched.inc.
// ------------------------------------------------------------------------------------------------
#include "gemm_256s_sched.inc"
__device__ __forceinline__ void lds_wait8(bf16x8 (&a)[4], bf16x8 (&b)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
  __builtin_amdgcn_sched_barrier(0);
}

// SK = true: the stream-K launch for the tiles of the last, partial round (GemmArgs::sk_*): a workgroup computes a contiguous range of
// k-stages that covers the end of one tile and / or the start of the next one instead of whole tiles.  Same main loop: a work item is the
// stages [S0, S0 + nkl) of a tile (SK = false: S0 = 0, nkl = all of them - the compiler folds both).
constexpr int SK_MINSEG = 4;   // no item shorter than this many stages (the pipeline needs 2); cut points closer to a tile edge snap to it
// cut point `uu` of `units` over the sk_tiles * nk stages of the tail tiles (host + device: lhrs_gemm_streamk_plan replays it for the tests)
__host__ __device__ __forceinline__ int sk_cut(int uu, int sk_tiles, int nk, int units) {
  int x = (int)((long)uu * ((long)sk_tiles * nk) / units);
  const int r = x % nk;
  if (r < SK_MINSEG) x -= r;
  else if (r > nk - SK_MINSEG) x += nk - r;
  return x;
}
template <int ACT, int EPI, bool K2P, bool SK = false>
__global__ __launch_bounds__(1024, 1) void gemm_nt_256s_kernel(GemmArgs g) {
  constexpr int BM = 256, BN = 256, BK = 64;
  constexpr int A_BYTES = BM * BK * 2, STAGE = A_BYTES + BN * BK * 2;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int ntiles = g.ntiles;
  int t = blockIdx.x;
  if (!SK && t >= ntiles) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // DMA: 64 pieces of 1 KiB per stage; wave w issues #4w..4w+3 (waves 0-7: A, 8-15: B)
  const bool isA = wave < 8;
  const char* base1 = reinterpret_cast<const char*>(isA ? g.A : g.B);
  const char* base2 = reinterpret_cast<const char*>(isA ? g.A2 : g.B2);
  const long ld1 = isA ? g.lda : g.ldb, ld2 = isA ? g.lda2 : g.ldb2;
  const int rmax = (isA ? g.M : g.N) - 1;
  const int nk1 = g.K / BK;
  const int nk = (g.K + (K2P ? g.K2 : 0)) / BK;  // >= 2 (host guarantees)
  // row of the operand that piece j of this lane reads for tile (tm_, tn_)
  auto dma_row = [&](int tm_, int tn_, int j) {
    const int row0 = isA ? tm_ * BM : tn_ * BN;
    const int ridx = (wave & 7) * 4 + j;
    int row = min(row0 + ridx * 8 + (lane >> 3), rmax);
    if (EPI == 1 && !isA) {  // B tile row r = 64*wn + 32*half + i  <-  weight row half*ff + tn*128 + wn*32 + i
      const int r = ridx * 8 + (lane >> 3);
      row = ((r >> 5) & 1) * g.ff + tn_ * 128 + (r >> 6) * 32 + (r & 31);
    }
    if (EPI == 3 && !isA && tn_ * BN < g.rope_cols) {  // B tile row r = 64*wn + 32*half + i  <-  head 2*tn + (wn >> 1), dim 64*half + 32*(wn & 1) + i
      const int r = ridx * 8 + (lane >> 3);
      row = tn_ * BN + ((r >> 7) << 7) + ((r >> 5) & 1) * 64 + ((r >> 6) & 1) * 32 + (r & 31);
    }
    return row;
  };
  // global byte offsets of this lane's four DMA rows for tile (tm_, tn_) in the FIRST operand pair.  The second pair (K2P: the fused LoRA
  // product, one or two stages at the end of the k-loop) gets its offsets where its pieces are issued: four more registers carried
  // through the main loop spilled there, and a scratch reload in the MFMA stream is followed by a full `vmcnt` wait - behind fresh DMA
  auto dma_rows = [&](int tm_, int tn_, unsigned (&o1)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int lchunk = (lane & 7) ^ ((((j & 1) << 2) + (lane >> 4)) & 7);
      o1[j] = (unsigned)(((long)dma_row(tm_, tn_, j) * ld1 + lchunk * 8) * 2);
    }
  };
  const int dst0 = (isA ? 0 : A_BYTES) + (wave & 7) * 4096;
  // wave-uniform 64-bit base (SGPR pair) + the lane's 32-bit row offset: the saddr form of global_load_lds.  The k offset goes through
  // readfirstlane so that the loop strength reduction cannot fold it into four loop-carried 64-bit VGPR pointers (8 registers).  The
  // instruction itself is inline asm: through the builtin the compiler forms a 64-bit VGPR address with two v_lshl_add_u64 and a v_mov per
  // piece - three VALU instructions in the MFMA stream for every DMA
  auto dma_piece = [&](const char* sp, unsigned vo, int buf, int j) {
    const unsigned lds_dst = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem) + buf * STAGE + dst0 + j * 1024;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(sp), "s"(lds_dst) : "memory", "m0");
  };
  auto issue1 = [&](const unsigned (&o1)[4], int kt, int buf, int j) {  // stage kt < nk1 of the first pair
    const int kb_ = __builtin_amdgcn_readfirstlane(kt * (BK * 2));
    dma_piece(base1 + kb_, o1[j], buf, j);
  };
  auto issue2 = [&](int tm_, int tn_, int kt2, int buf, int j) {        // stage kt2 of the second pair (K2P)
    const int kb_ = __builtin_amdgcn_readfirstlane(kt2 * (BK * 2));
    const int lchunk = (lane & 7) ^ ((((j & 1) << 2) + (lane >> 4)) & 7);
    dma_piece(base2 + kb_, (unsigned)(((long)dma_row(tm_, tn_, j) * ld2 + lchunk * 8) * 2), buf, j);
  };

  // fragment of 16 rows x 32 k: lane -> row (lane & 15), 16-byte chunk kb * 4 + (lane >> 4) of the 128-byte row (swizzled)
  const int wm = wave >> 2, wn = wave & 3;
  const int sw = ((lane & 15) >> 1) & 7;
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  // LDS address of this lane's fragment row in buffer 0, k-block 0 (the fragment index is the instruction's immediate offset).  smem is the
  // kernel's only LDS object (offset 0), so the other k-block is this address ^ 64 (chunk bit 2) and the other buffer ^ STAGE: every fragment
  // address of the main loop is ONE v_xor of these two registers with a loop-variant scalar (nothing for the compiler to hoist and keep)
  const unsigned a0 = lds0 + (wm * 64 + (lane & 15)) * 128 + (((lane >> 4)) ^ sw) * 16;
  const unsigned b0 = lds0 + A_BYTES + (wn * 64 + (lane & 15)) * 128 + (((lane >> 4)) ^ sw) * 16;

  int S0 = 0, nkl = nk;   // the current work item: stages [S0, S0 + nkl) of its tile
  // stream-K: unit u of sk_units owns the stages [bnd(u), bnd(u + 1)) of the sk_tiles * nk stages of the tail tiles (cut points within
  // SK_MINSEG stages of a tile edge snap to it).  Its range is the end of tile sk_i0 (from stage S0: a PARTIAL unless S0 == 0) and / or the
  // first sk_n1 stages of tile sk_i0 + 1; the unit that holds a tile's stage 0 finishes the tile (adds the others' partials, epilogue)
  int sk_u = 0, sk_i0 = 0, sk_n1 = 0, sk_seg = 0;
  auto sk_bnd = [&](int uu) { return sk_cut(uu, g.sk_tiles, nk, g.sk_units); };
  int tm, tn;
  if constexpr (SK) {
    const int U = g.sk_units;
    sk_u = blockIdx.x;
    if ((U & 7) == 0) sk_u = (sk_u & 7) * (U >> 3) + (sk_u >> 3);   // neighbouring units (they exchange partials) on one XCD
    const int b0 = sk_bnd(sk_u), b1 = sk_bnd(sk_u + 1);
    if (b1 <= b0) return;
    sk_i0 = b0 / nk;
    S0 = b0 - sk_i0 * nk;
    const int e0 = min(b1, (sk_i0 + 1) * nk);
    nkl = e0 - b0;
    sk_n1 = b1 - e0;
    tile_coords_raster(g, g.sk_tile0 + sk_i0, tm, tn);
  } else {
    tile_coords_lin(g, t, ntiles, tm, tn);
  }
  unsigned off1[4];
  dma_rows(tm, tn, off1);
  int pb = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) issue1(off1, S0, 0, j);   // (a stream-K item never starts inside the second operand pair: host rule)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  bf16x8 A[4], B[4];
#define RDQ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define SB __builtin_amdgcn_sched_barrier(0);
#define MF(mi, ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B[ni], A[mi], acc[mi][ni], 0, 0, 0); SB

  for (;;) {
    const int tnext = t + (int)gridDim.x;
    const bool has_next = SK ? (sk_seg == 0 && sk_n1 > 0) : (tnext < ntiles);  // workgroup-uniform
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // stage 0 of this tile is in LDS buffer pb; the barrier publishes it and ends the previous tile's epilogue reads of the other buffer
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!K2P || S0 + 1 < nk1) issue1(off1, S0 + 1, pb ^ 1, j);
      else issue2(tm, tn, S0 + 1 - nk1, pb ^ 1, j);
    }
    {
      const unsigned aa = a0 ^ (unsigned)(pb * STAGE), ba = b0 ^ (unsigned)(pb * STAGE);
      RDQ(A[0], aa, 0); RDQ(A[1], aa, 2048); RDQ(A[2], aa, 4096); RDQ(A[3], aa, 6144);
      RDQ(B[0], ba, 0); RDQ(B[1], ba, 2048); RDQ(B[2], ba, 4096); RDQ(B[3], ba, 6144);
      lds_wait8(A, B);
    }
    // one stage = S_BLOCK0 (first 32 k; re-reads every fragment for the second 32 k of the same buffer) + S_BLOCK1 (the stage boundary -
    // full wait, vmcnt, barrier: stage s + 1 is published and the buffer of stage s is free - then the second 32 k, behind whose first
    // MFMAs the DMA of stage s + 2 goes into the freed buffer, and the re-reads from stage s + 1) - gemm_256s_sched.inc
#define STAGE_S(s_)                                                                                             \
    {                                                                                                           \
      const unsigned so = (((s_) + pb) & 1) * STAGE, sn = so ^ STAGE;                                           \
      { const unsigned aa = a0 ^ (so | 64u), ba = b0 ^ (so | 64u); S_BLOCK0(aa, ba) }                           \
      { const unsigned aa = a0 ^ sn, ba = b0 ^ sn; S_BLOCK1(aa, ba) }                                           \
    }
    // local stage s fetches the item's stage s + 2 = stage S0 + s + 2 of the tile: from the first pair while that is < nk1
    const int nA = K2P ? min(nkl - 2, nk1 - 2 - S0) : nkl - 2;
#define ISS(j) issue1(off1, S0 + s + 2, (s + pb) & 1, j);
    for (int s = 0; s < nA; ++s) STAGE_S(s)   // !K2P: every stage but the last two
#undef ISS
    if (K2P) {  // the stages whose DMA slot fetches the second pair
#define ISS(j) issue2(tm, tn, S0 + s + 2 - nk1, (s + pb) & 1, j);
      for (int s = max(nA, 0); s < nkl - 2; ++s) STAGE_S(s)
#undef ISS
    }
    int ntm = 0, ntn = 0;
    if (has_next) {  // this tile's DMA rows are not needed any more (its last stage is in flight): the offsets become the next tile's
      if constexpr (SK) tile_coords_raster(g, g.sk_tile0 + sk_i0 + 1, ntm, ntn);
      else tile_coords_lin(g, tnext, ntiles, ntm, ntn);
      dma_rows(ntm, ntn, off1);
    }
    // stage nkl - 2: the buffer its barrier frees takes the first stage of the NEXT item (always a tile's stage 0)
#define ISS(j) if (has_next) issue1(off1, 0, (nkl + pb) & 1, j);
    STAGE_S(nkl - 2)
#undef ISS
#undef STAGE_S
    {  // last stage: nothing left to fetch behind it
      const unsigned so = ((nkl - 1 + pb) & 1) * STAGE;
      { const unsigned aa = a0 ^ (so | 64u), ba = b0 ^ (so | 64u); S_BLOCK0(aa, ba) }
      S_BLOCK1_FINAL(0, 0)
    }

    bool sk_partial = false;
    if constexpr (SK) {
      // slab layout: [unit][wave][mi][ni][lane] float4 - exactly this lane's accumulator registers, 1 KiB per wave instruction
      if (S0 != 0) {
        // PARTIAL of a tile another unit finishes: accumulators -> slab sk_u, then publish (agent-scope release behind every wave's
        // drained stores, then the flag: cdna_hip_programming.md §6 Guideline 16)
        sk_partial = true;
        float4* dst = reinterpret_cast<float4*>(g.sk_ws + (size_t)sk_u * (BM * BN)) + wave * 1024 + lane;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            dst[(mi * 4 + ni) * 64] = make_float4(acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_store(g.sk_flags + sk_u, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else if (nkl != nk) {
        // this unit holds the tile's stage 0 but not all of it: add the partials of the units that hold the rest (units sk_u + 1 ... whose
        // cut point lies inside this tile), in unit order (a fixed summation order: deterministic results)
        const int tile_end = (sk_i0 + sk_seg + 1) * nk;
        if (tid == 0) {
          for (int p = sk_u + 1; p < g.sk_units && sk_bnd(p) < tile_end; ++p) {
            if (sk_bnd(p + 1) <= sk_bnd(p)) continue;   // an empty unit writes nothing
            while (__hip_atomic_load(g.sk_flags + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(8);
            __hip_atomic_store(g.sk_flags + p, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // flags are zero again when the launch ends
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        for (int p = sk_u + 1; p < g.sk_units && sk_bnd(p) < tile_end; ++p) {
          if (sk_bnd(p + 1) <= sk_bnd(p)) continue;
          const float4* src = reinterpret_cast<const float4*>(g.sk_ws + (size_t)p * (BM * BN)) + wave * 1024 + lane;
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
              const float4 v = src[(mi * 4 + ni) * 64];
              acc[mi][ni][0] += v.x; acc[mi][ni][1] += v.y; acc[mi][ni][2] += v.z; acc[mi][ni][3] += v.w;
            }
        }
      }
    }
    // accumulator element acc[mi][ni][i]: m = wm*64 + mi*16 + fr, n = wn*64 + ni*16 + fg*4 + i.
    // Everything the epilogue derives from the lane id is derived from an opaque copy made HERE, per tile: otherwise those values are
    // loop-invariant across tiles, get hoisted in front of the tile loop and sit in (or spill from) registers all through the main loop
    int le = lane;
    asm volatile("" : "+v"(le));
    const int fr = le & 15, fg = le >> 4;
    if (sk_partial) {
    } else if (g.out_f32) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int m = tm * BM + wm * 64 + mi * 16 + fr;
        if (m >= g.M) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int n = tn * BN + wn * 64 + ni * 16 + fg * 4;
          if (n >= g.N) continue;
          store4<ACT>(g, m, n, acc[mi][ni]);
        }
      }
    } else {
      __builtin_amdgcn_s_barrier();  // every wave has read its last fragments: the last stage's buffer becomes the staging area
      char* reg = smem + ((nkl - 1 + pb) & 1) * STAGE + wave * 4096;  // [32 rows][64 cols] bf16, wave private, one pass per 32 rows
      const bool rope_tile = EPI == 3 && tn * BN < g.rope_cols;
      const int rsub = le >> 3, c = le & 7;
      const int n = EPI == 1   ? (c < 4 ? 0 : g.ff) + tn * 128 + wn * 32 + (c & 3) * 8
                    : rope_tile ? tn * BN + (wn >> 1) * 128 + (c < 4 ? 0 : 64) + (wn & 1) * 32 + (c & 3) * 8
                                : tn * BN + wn * 64 + c * 8;
      constexpr bool SWB = EPI == 2 || EPI == 5;   // SwiGLU-backward epilogue (5: + row dot partials)
      const int nlim = SWB ? g.ff : g.N;
      const bool col_ok = n < nlim;
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        uint4 pre_a[4], pre_b[4];
        if (SWB || EPI == 4 || (EPI == 0 && g.res)) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int m = tm * BM + wm * 64 + ps * 32 + i * 8 + rsub;
            pre_a[i] = make_uint4(0, 0, 0, 0); pre_b[i] = make_uint4(0, 0, 0, 0);
            if (m < g.M && col_ok) {
              if (SWB) {
                pre_a[i] = *reinterpret_cast<const uint4*>(g.aux + (long)m * g.ld_aux + n);
                pre_b[i] = *reinterpret_cast<const uint4*>(g.aux + (long)m * g.ld_aux + g.ff + n);
              } else if (EPI == 4) {
                pre_a[i] = *reinterpret_cast<const uint4*>(g.aux + (long)m * g.ld_aux + n);                        // x
                if (g.res) pre_b[i] = *reinterpret_cast<const uint4*>(g.res + (long)m * g.ldr + n);              // add
              } else {
                pre_a[i] = *reinterpret_cast<const uint4*>(g.res + (long)m * g.ldr + n);
              }
            }
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int mi = ps * 2 + h;
          const int row = h * 16 + fr;  // row inside this 32-row pass
          if (rope_tile) {
            // fragments ni = 0, 1 hold dims d = 32*(wn & 1) + 16*ni + 4*fg + i of the head, ni + 2 their rotate_half partners d + 64
            const int pos = (tm * BM + wm * 64 + mi * 16 + fr) % g.rope_mod + g.rope_pos0;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
              const float4 c4 = *reinterpret_cast<const float4*>(g.rope_cos + (long)pos * 64 + (wn & 1) * 32 + ni * 16 + fg * 4);
              const float4 s4 = *reinterpret_cast<const float4*>(g.rope_sin + (long)pos * 64 + (wn & 1) * 32 + ni * 16 + fg * 4);
              const float cv[4] = {c4.x, c4.y, c4.z, c4.w}, sv[4] = {s4.x, s4.y, s4.z, s4.w};
              float o1[4], o2[4];
#pragma unroll
              for (int i = 0; i < 4; ++i)
                rope_pair(bf2f(f2bf(acc[mi][ni][i] * g.alpha)), bf2f(f2bf(acc[mi][ni + 2][i] * g.alpha)), cv[i], sv[i], o1[i], o2[i]);
              const int u = ni * 4 + fg;
              *reinterpret_cast<uint2*>(reg + row * 128 + ((u ^ ((row & 7) << 1)) << 3)) = make_uint2(pack2bf(o1[0], o1[1]), pack2bf(o1[2], o1[3]));
              *reinterpret_cast<uint2*>(reg + row * 128 + (((8 + u) ^ ((row & 7) << 1)) << 3)) = make_uint2(pack2bf(o2[0], o2[1]), pack2bf(o2[2], o2[3]));
            }
          } else {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
              float v[4];
              uint2 bb = make_uint2(0, 0);
              if (EPI == 0 && g.bias) bb = *reinterpret_cast<const uint2*>(g.bias + min(tn * BN + wn * 64 + ni * 16 + fg * 4, g.N - 4));
              const float bias_v[4] = {bflo(bb.x), bfhi(bb.x), bflo(bb.y), bfhi(bb.y)};
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                v[i] = acc[mi][ni][i] * g.alpha;
                if (EPI == 0 && g.drop_thresh) {
                  const long e = (long)(tm * BM + wm * 64 + mi * 16 + fr) * g.N + tn * BN + wn * 64 + ni * 16 + fg * 4 + i;
                  v[i] = drop_keep(g.drop_seed, e, g.drop_thresh) ? v[i] * g.drop_scale : 0.f;
                }
                v[i] += bias_v[i];
                if (ACT) v[i] = apply_act(v[i], ACT);
              }
              const int u = ni * 4 + fg;
              *reinterpret_cast<uint2*>(reg + row * 128 + ((u ^ ((row & 7) << 1)) << 3)) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
            }
          }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r32 = i * 8 + rsub;
          const int m = tm * BM + wm * 64 + ps * 32 + r32;
          uint4 val = *reinterpret_cast<const uint4*>(reg + r32 * 128 + ((c ^ (r32 & 7)) << 4));
          float rowp = 0.f;   // EPI 5: this lane's 8 columns of <d(gate|up), gate|up> of row m
          if (SWB && m < g.M && col_ok) {
            const uint4 gq = pre_a[i], uq = pre_b[i];
            float d[8], gg[8], uu[8], dg[8], du[8];
            unpack8(val, d); unpack8(gq, gg); unpack8(uq, uu);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float sg = 1.f / (1.f + __expf(-gg[e]));
              du[e] = d[e] * gg[e] * sg;
              dg[e] = d[e] * uu[e] * sg * (1.f + gg[e] * (1.f - sg));
              if (EPI == 5) rowp += bf2f(f2bf(dg[e])) * gg[e] + bf2f(f2bf(du[e])) * uu[e];   // on the values as they are stored
            }
            bf16_t* out = reinterpret_cast<bf16_t*>(g.C) + (long)m * g.ldc + n;
            *reinterpret_cast<uint4*>(out) = pack8(dg);
            *reinterpret_cast<uint4*>(out + g.ff) = pack8(du);
          }
          if (EPI == 5) {   // the 8 lanes c = 0..7 of a row hold its 64 columns of this wave: fold them, lane c == 0 stores the partial
            rowp += __shfl_xor(rowp, 1, 64); rowp += __shfl_xor(rowp, 2, 64); rowp += __shfl_xor(rowp, 4, 64);
            if (c == 0 && m < g.M) g.row_dot[(long)(tn * 4 + wn) * g.M + m] = rowp;
          }
          if (SWB) continue;
          if (m < g.M && col_ok) {
            if (EPI == 4) {
              // dx = rstd * (w o dh) - x * (rstd^2 * s / N) + add, fp32, one rounding (dh as the unfused path sees it: rounded to bf16)
              float dh[8], xv[8], av[8], wv[8], o8[8];
              unpack8(val, dh); unpack8(pre_a[i], xv); unpack8(pre_b[i], av);
              unpack8(*reinterpret_cast<const uint4*>(g.bias + n), wv);
              const float r = g.sa[m], coef = r * r * g.sb[m] / (float)g.N;
#pragma unroll
              for (int e = 0; e < 8; ++e) o8[e] = r * (wv[e] * dh[e]) - xv[e] * coef + av[e];
              *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(g.C) + (long)m * g.ldc + n) = pack8(o8);
              continue;
            }
            if (EPI == 0 && g.res) {
              const uint4 r = pre_a[i];
              val.x = pack2bf(bflo(val.x) + bflo(r.x), bfhi(val.x) + bfhi(r.x));
              val.y = pack2bf(bflo(val.y) + bflo(r.y), bfhi(val.y) + bfhi(r.y));
              val.z = pack2bf(bflo(val.z) + bflo(r.z), bfhi(val.z) + bfhi(r.z));
              val.w = pack2bf(bflo(val.w) + bflo(r.w), bfhi(val.w) + bfhi(r.w));
            }
            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(g.C) + (long)m * g.ldc + n) = val;
          }
        }
      }
      if (EPI == 1) {
        // second output: act = silu(gate) * up on the bf16-rounded gate / up (fragments ni = 0, 1 / ni + 2), staged [64][32]
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const int row = mi * 16 + fr;
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float gv = bf2f(f2bf(acc[mi][ni][i] * g.alpha)), uv = bf2f(f2bf(acc[mi][ni + 2][i] * g.alpha));
              v[i] = silu(gv) * uv;
            }
            const int u = ni * 4 + fg;  // 8-byte unit 0..7 of the 64-byte row
            *reinterpret_cast<uint2*>(reg + row * 64 + ((u ^ ((row & 3) << 1)) << 3)) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int rsub4 = le >> 2, c4 = le & 3;
        const int n2 = tn * 128 + wn * 32 + c4 * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = i * 16 + rsub4;
          const int m = tm * BM + wm * 64 + row;
          const uint4 val = *reinterpret_cast<const uint4*>(reg + row * 64 + ((c4 ^ (row & 3)) << 4));
          if (m < g.M) *reinterpret_cast<uint4*>(g.aux_out + (long)m * g.ld_aux + n2) = val;
        }
      }
    }
    if (!has_next) break;
    if (g.out_f32) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t = tnext; tm = ntm; tn = ntn;
    pb = (pb + nkl) & 1;
    if constexpr (SK) { sk_seg = 1; S0 = 0; nkl = sk_n1; }
  }
#undef MF
#undef SB
#undef RDQ
}



#include "gemm_144s_kernel.inc"

// ------------------------------------------------------------------------------------------------
// e4m3 x e4m3 -> bf16 GEMM for FROZEN base weights in 8-bit (the reference trains stages 2/3 with `bits: 8` base weights,
// lhrs/models/text_modal.py:91-131 -> bitsandbytes LLM.int8): C[m][n] = sa[m] * sb[n] * sum_k A8[m][k] * B8[n][k]
// (+ residual).  Same skeleton as the bf16 256x256 kernels - 256x256 tile, 8 waves, two 64 KiB LDS buffers filled by global_load_lds,
// one barrier per stage - but a stage row is 128 BYTES = 128 k and the product runs on v_mfma_scale_f32_32x32x64_f8f6f4 with unit
// block scales (the only 2x-rate fp8 MFMA of gfx950): a stage is two blocks of 8 MFMAs (64 k each, 16 passes), every lane feeds
// 32 consecutive k-bytes of its row (two ds_read_b128) to both operands.  Twice the FLOPs of the bf16 kernel per byte moved.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) int i32x8;

// I8 = true: the same kernel on LLM.int8 operands (int8.hip; the reference's `bits: 8` base, text_modal.py:91-131 -> bitsandbytes MatMul8bitLt):
// a lane's 32 k-bytes feed TWO v_mfma_i32_32x32x32_i8 (bytes [0,16) and [16,32): both operands are read with the same addressing, so which k
// a byte slot holds does not matter) into int32 accumulators - exact -, converted to f32 once the 8-bit stages are done; the per-row factors
// sa[m] = absmax / 127 and sb[n] = absmax / 127 then turn them into real units, and the optional bf16 stages behind them carry the 16-bit
// outlier-column product (and the LoRA update) on the same accumulators.
template <bool I8>
__global__ __launch_bounds__(512, 2) void gemm_fp8_256_kernel(GemmArgs g) {
  constexpr int BM = 256, BN = 256, BKB = 128;  // bytes (= k) per stage row
  constexpr int A_BYTES = BM * BKB, STAGE = A_BYTES + BN * BKB;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  int tm, tn;
  tile_coords(g, tm, tn);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const bool isA = wave < 4;
  const char* base1 = reinterpret_cast<const char*>(isA ? g.A : g.B);
  const long ld1 = isA ? g.lda : g.ldb;  // bytes
  const int row0 = isA ? tm * BM : tn * BN, rmax = (isA ? g.M : g.N) - 1;
  // optional bf16 pair (fused LoRA update, K2 % 64 == 0): its 64-k stages have the same 128-byte row image and follow the e4m3 stages
  const char* base2 = reinterpret_cast<const char*>(isA ? g.A2 : g.B2);
  const long ld2 = (isA ? g.lda2 : g.ldb2) * 2L;  // bytes
  const int nk1 = g.K / BKB;
  const int K2 = g.K2 + (g.k2_dev ? __builtin_amdgcn_readfirstlane(g.k2_dev[0]) : 0);   // host-known + device-known bf16 columns
  unsigned off1[8], off2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ridx = (wave & 3) * 8 + j;
    const int lchunk = (lane & 7) ^ ((((j & 1) << 2) + (lane >> 4)) & 7);
    const int row = min(row0 + ridx * 8 + (lane >> 3), rmax);
    off1[j] = (unsigned)((long)row * ld1 + lchunk * 16);
    off2[j] = K2 > 0 ? (unsigned)((long)row * ld2 + lchunk * 16) : 0u;
  }
  const int dst0 = (isA ? 0 : A_BYTES) + (wave & 3) * 8192;
  const int nk_all = nk1 + K2 / 64;    // e4m3 stages + bf16 stages of the fused pair
  auto issue1 = [&](int kt, int j) {
    if (kt >= nk_all) return;            // (wave-uniform) nothing left to fetch
    const char* p = kt < nk1 ? base1 + (long)kt * BKB + off1[j] : base2 + (long)(kt - nk1) * BKB + off2[j];
    __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(smem + (kt & 1) * STAGE + dst0 + j * 1024), 16, 0, 0);
  };

  const int wm = wave >> 2, wn = wave & 3;
  const int fr = lane & 31, fh = lane >> 5;
  const int sw = (fr >> 1) & 7;
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  const unsigned a_base = lds0 + (wm * 128 + fr) * 128;
  const unsigned b_base = lds0 + A_BYTES + (wn * 64 + fr) * 128;
  unsigned kofs[2][2];  // [k-step][16-B half]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) kofs[ks][hh] = ((ks * 4 + fh * 2 + hh) ^ sw) * 16;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  i32x8 a0[4], b0[2], a1[4], b1[2];
  // one fragment = two ds_read_b128 (k-bytes [0,16) and [16,32) of the lane's 32) forming one 8-VGPR operand.  Plain LDS loads, not
  // inline asm as in the bf16 kernels: the halves of an 8-register tuple cannot be named as asm outputs, and letting the compiler see
  // the loads keeps its register reuse and s_waitcnt placement correct around the 16-pass MFMAs (sched_barriers pin the interleave)
  typedef const __attribute__((address_space(3))) i32x4* lds4_t;
#define RD8(dst, addr0, addr1, off)                                                                                  \
  do {                                                                                                               \
    const i32x4 lo_ = *reinterpret_cast<lds4_t>((size_t)((addr0) + (off)));                                          \
    const i32x4 hi_ = *reinterpret_cast<lds4_t>((size_t)((addr1) + (off)));                                          \
    dst = i32x8{lo_[0], lo_[1], lo_[2], lo_[3], hi_[0], hi_[1], hi_[2], hi_[3]};                                     \
  } while (0)
  typedef __attribute__((ext_vector_type(16))) int i32x16_t;
#define MF8(A_, B_, mi, ni)                                                                                          \
  if constexpr (I8) {                                                                                                \
    i32x16_t c_ = __builtin_bit_cast(i32x16_t, acc[mi][ni]);                                                         \
    c_ = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_shufflevector(B_[ni], B_[ni], 0, 1, 2, 3),                  \
                                               __builtin_shufflevector(A_[mi], A_[mi], 0, 1, 2, 3), c_, 0, 0, 0);   \
    c_ = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_shufflevector(B_[ni], B_[ni], 4, 5, 6, 7),                  \
                                               __builtin_shufflevector(A_[mi], A_[mi], 4, 5, 6, 7), c_, 0, 0, 0);   \
    acc[mi][ni] = __builtin_bit_cast(f32x16, c_);                                                                    \
  } else {                                                                                                           \
    acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B_[ni], A_[mi], acc[mi][ni], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F); \
  }                                                                                                                  \
  __builtin_amdgcn_sched_barrier(0);
#define SB8 __builtin_amdgcn_sched_barrier(0);
  // one block: 8 MFMAs (64 k); fillers: the 6 fragments (12 reads) of the next block and up to 8 DMA pieces of stage kd
#define BLOCK8(Ac, Bc, An, Bn, aa0, aa1, ba0, ba1, RD, kd, d0, NDMA)                                       \
  MF8(Ac, Bc, 0, 0) if (RD) RD8(Bn[0], ba0, ba1, 0);     if (NDMA > 0) issue1(kd, d0);     SB8             \
  MF8(Ac, Bc, 0, 1) if (RD) RD8(Bn[1], ba0, ba1, 4096);  if (NDMA > 1) issue1(kd, d0 + 1); SB8             \
  MF8(Ac, Bc, 1, 0) if (RD) RD8(An[0], aa0, aa1, 0);     if (NDMA > 2) issue1(kd, d0 + 2); SB8             \
  MF8(Ac, Bc, 1, 1) if (RD) RD8(An[1], aa0, aa1, 4096);  if (NDMA > 3) issue1(kd, d0 + 3); SB8             \
  MF8(Ac, Bc, 2, 0) if (RD) RD8(An[2], aa0, aa1, 8192);  if (NDMA > 4) issue1(kd, d0 + 4); SB8             \
  MF8(Ac, Bc, 2, 1) if (RD) RD8(An[3], aa0, aa1, 12288); if (NDMA > 5) issue1(kd, d0 + 5); SB8             \
  MF8(Ac, Bc, 3, 0)                                      if (NDMA > 6) issue1(kd, d0 + 6); SB8             \
  MF8(Ac, Bc, 3, 1)                                      if (NDMA > 7) issue1(kd, d0 + 7); SB8

  const int nk = nk1;                    // e4m3 stages, >= 2 (host guarantees)
#pragma unroll
  for (int j = 0; j < 8; ++j) issue1(0, j);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int j = 0; j < 8; ++j) issue1(1, j);
  {
    const unsigned aa0 = a_base + kofs[0][0], aa1 = a_base + kofs[0][1], ba0 = b_base + kofs[0][0], ba1 = b_base + kofs[0][1];
    RD8(b0[0], ba0, ba1, 0); RD8(b0[1], ba0, ba1, 4096);
    RD8(a0[0], aa0, aa1, 0); RD8(a0[1], aa0, aa1, 4096); RD8(a0[2], aa0, aa1, 8192); RD8(a0[3], aa0, aa1, 12288);
  }
  {  // block ks0 of stage 0; loads ks1 of stage 0
    const unsigned aa0 = a_base + kofs[1][0], aa1 = a_base + kofs[1][1], ba0 = b_base + kofs[1][0], ba1 = b_base + kofs[1][1];
    BLOCK8(a0, b0, a1, b1, aa0, aa1, ba0, ba1, true, 0, 0, 0)
  }
  auto stage = [&](int kt) {
    constexpr bool DMA = true;  // issue1 is a no-op past the last stage
    const unsigned so = (kt & 1) * STAGE;
    // block ks1(kt-1): retire its fragment reads, then the barrier that publishes stage kt and frees the other buffer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
      const unsigned aa0 = a_base + so + kofs[0][0], aa1 = a_base + so + kofs[0][1], ba0 = b_base + so + kofs[0][0], ba1 = b_base + so + kofs[0][1];
      BLOCK8(a1, b1, a0, b0, aa0, aa1, ba0, ba1, true, kt + 1, 0, (DMA ? 8 : 0))
    }
      {
      const unsigned aa0 = a_base + so + kofs[1][0], aa1 = a_base + so + kofs[1][1], ba0 = b_base + so + kofs[1][0], ba1 = b_base + so + kofs[1][1];
      BLOCK8(a0, b0, a1, b1, aa0, aa1, ba0, ba1, true, 0, 0, 0)
    }
  };
  for (int kt = 1; kt < nk; ++kt) stage(kt);  // the last e4m3 stage already fetches the first bf16 stage of a fused pair
  { BLOCK8(a1, b1, a0, b0, a_base, a_base, b_base, b_base, false, 0, 0, 0) }
#undef BLOCK8
#undef SB8
#undef MF8
#undef RD8

  if constexpr (I8) {  // int32 sums -> f32 (exact below 2^24; |sum| <= 127 * 127 * K stays far inside f32's range, the rounding is 2^-24 relative)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const i32x16_t c_ = __builtin_bit_cast(i32x16_t, acc[mi][ni]);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = (float)c_[r];
      }
  }
  bool scaled = false;
  if (nk_all > nk) {
    // ---- fused LoRA pair: bring the e4m3 sums to real units, then keep accumulating bf16 products (C layout is dtype-independent)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const float sa = g.sa[min(tm * BM + wm * 128 + mi * 32 + fr, g.M - 1)];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 s4 = *reinterpret_cast<const float4*>(g.sb + min(tn * BN + wn * 64 + ni * 32 + q * 8 + fh * 4, g.N - 4));
          acc[mi][ni][4 * q] *= sa * s4.x; acc[mi][ni][4 * q + 1] *= sa * s4.y;
          acc[mi][ni][4 * q + 2] *= sa * s4.z; acc[mi][ni][4 * q + 3] *= sa * s4.w;
        }
    }
    scaled = true;
    typedef const __attribute__((address_space(3))) bf16x8* ldsb_t;
    for (int kt = nk; kt < nk_all; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // stage kt has landed everywhere; every wave is done with the other buffer
      if (kt + 1 < nk_all) {
#pragma unroll
        for (int j = 0; j < 8; ++j) issue1(kt + 1, j);
      }
      const unsigned so = (kt & 1) * STAGE;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const unsigned ko = (((kk * 2 + fh) ^ sw) * 16);
        bf16x8 af[4], bfr[2];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[mi] = *reinterpret_cast<ldsb_t>((size_t)(a_base + so + ko + mi * 4096));
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) bfr[ni] = *reinterpret_cast<ldsb_t>((size_t)(b_base + so + ko + ni * 4096));
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);
      }
    }
  }

  __builtin_amdgcn_s_barrier();
  char* reg = smem + wave * 16384;
  {
    float sbv[2][4][4];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = min(tn * BN + wn * 64 + ni * 32 + q * 8 + fh * 4, g.N - 4);
        float4 s4 = *reinterpret_cast<const float4*>(g.sb + n);
        if (scaled) s4 = make_float4(1.f, 1.f, 1.f, 1.f);
        sbv[ni][q][0] = s4.x; sbv[ni][q][1] = s4.y; sbv[ni][q][2] = s4.z; sbv[ni][q][3] = s4.w;
      }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int row = mi * 32 + fr;
      const float sa = (scaled ? 1.f : g.sa[min(tm * BM + wm * 128 + row, g.M - 1)]) * g.alpha;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = acc[mi][ni][4 * q + i] * sa * sbv[ni][q][i];
          const int u = ni * 8 + q * 2 + fh;
          *reinterpret_cast<uint2*>(reg + row * 128 + ((u ^ ((row & 7) << 1)) << 3)) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
        }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  {
    const int rsub = lane >> 3, c = lane & 7;
    const int n = tn * BN + wn * 64 + c * 8;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = i * 8 + rsub;
      const int m = tm * BM + wm * 128 + row;
      uint4 val = *reinterpret_cast<const uint4*>(reg + row * 128 + ((c ^ (row & 7)) << 4));
      if (m < g.M && n < g.N) {
        if (g.res) {
          const uint4 r = *reinterpret_cast<const uint4*>(g.res + (long)m * g.ldr + n);
          val.x = pack2bf(bflo(val.x) + bflo(r.x), bfhi(val.x) + bfhi(r.x));
          val.y = pack2bf(bflo(val.y) + bflo(r.y), bfhi(val.y) + bfhi(r.y));
          val.z = pack2bf(bflo(val.z) + bflo(r.z), bfhi(val.z) + bfhi(r.z));
          val.w = pack2bf(bflo(val.w) + bflo(r.w), bfhi(val.w) + bfhi(r.w));
        }
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(g.C) + (long)m * g.ldc + n) = val;
      }
    }
  }
}

// sum of `splits` f32 slabs [M, ldc] -> bf16 C[M, N] * alpha
// (+ residual[M, ldr] added in fp32 before the one rounding, as the GEMM epilogues do; res may alias C)
__global__ void splitk_reduce_kernel(const float* __restrict__ part, bf16_t* C, long ldc, int M, int N, int splits, float alpha,
                                     const bf16_t* res = nullptr, long ldr = 0) {
  const long total = (long)M * (N / 4);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / (N / 4);
    const int n = (int)(i % (N / 4)) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < splits; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(part + ((long)k * M + m) * N + n);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    s.x *= alpha; s.y *= alpha; s.z *= alpha; s.w *= alpha;
    if (res) {
      const uint2 r = *reinterpret_cast<const uint2*>(res + m * ldr + n);
      s.x += bflo(r.x); s.y += bfhi(r.x); s.z += bflo(r.y); s.w += bfhi(r.y);
    }
    *reinterpret_cast<uint2*>(C + m * ldc + n) = make_uint2(pack2bf(s.x, s.y), pack2bf(s.z, s.w));
  }
}

// sum of `splits` f32 slabs [M, N] -> f32 C[M, ldc]
__global__ void splitk_reduce_f32_kernel(const float* __restrict__ part, float* __restrict__ C, long ldc, int M, int N, int splits) {
  const long total = (long)M * (N / 4);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / (N / 4);
    const int n = (int)(i % (N / 4)) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < splits; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(part + ((long)k * M + m) * N + n);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(C + m * ldc + n) = s;
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
  int* kind = nullptr;        // cap: epilogue variant of the sampled launch (0 plain <ACT,0>, 1 SwiGLU fwd, 2 SwiGLU bwd, 3 RoPE)
  double total_flops_all = 0; // every GEMM launch while enabled (sampled or not)
  long launches_all = 0;
  int stride = 1;             // every stride-th launch of each epilogue variant is bracketed (lhrs_gemm_profile_stride)
  long seen[7] = {0, 0, 0, 0, 0, 0, 0};  // 0 plain 256-row, 1 SwiGLU fwd, 2 SwiGLU bwd, 3 RoPE, 4 plain 144-row, 5 plain products handed to the vendor library, 6 plain products on gemm_u4_kernel
  bool take(int kind) { return (seen[kind]++ % stride) == 0; }
} g_prof;
}  // namespace

// Sampling stride of the live timing: the two event records around a launch cost the stream ~3 us of idle time each (measured: timing
// EVERY dominant launch slows the step by 1.0 % at micro-batch 30 and 2.2 % at micro-batch 8), so bench.py brackets every 7th launch of
// each variant - 7 is coprime to the period of the launch sequence (2 plain launches per layer forward, 3 backward), so the sample keeps
// the shape mix.  Default 1 (every launch).
extern "C" int lhrs_gemm_profile_stride(int n) { g_prof.stride = n > 0 ? n : 1; return 0; }

extern "C" int lhrs_gemm_profile_enable(int max_samples) {
  if (g_prof.ev) {
    for (int i = 0; i < 2 * g_prof.cap; ++i) (void)hipEventDestroy(g_prof.ev[i]);
    delete[] g_prof.ev; delete[] g_prof.flops; delete[] g_prof.kind;
    g_prof.ev = nullptr; g_prof.flops = nullptr; g_prof.kind = nullptr;
  }
  g_prof.on = max_samples > 0; g_prof.cap = max_samples > 0 ? max_samples : 0; g_prof.used = 0;
  g_prof.total_flops_all = 0; g_prof.launches_all = 0;
  for (int k = 0; k < 7; ++k) g_prof.seen[k] = 0;
  if (g_prof.on) {
    g_prof.ev = new hipEvent_t[2 * g_prof.cap];
    g_prof.flops = new double[g_prof.cap];
    g_prof.kind = new int[g_prof.cap];
    for (int i = 0; i < 2 * g_prof.cap; ++i)
      if (hipEventCreate(&g_prof.ev[i]) != hipSuccess) LHRS_FAIL("gemm_profile_enable: hipEventCreate failed");
  }
  return 0;
}

// out[0] = sampled launches, out[1] = their summed duration (ms), out[2] = their summed flops,
// out[3] = all GEMM launches while enabled, out[4] = flops of all of them
extern "C" int lhrs_gemm_profile_read(double* out) {
  double ms = 0, fl = 0, n = 0;
  for (int i = 0; i < g_prof.used; ++i) {
    if (g_prof.kind[i] != 0 && g_prof.kind[i] != 4) continue;  // the plain-epilogue persistent kernels (256-row + 144-row tiles); per kernel / variant: lhrs_gemm_profile_read_kinds
    float t = 0;
    if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) LHRS_FAIL("gemm_profile_read: event sync failed");
    if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) LHRS_FAIL("gemm_profile_read: elapsed failed");
    ms += t; fl += g_prof.flops[i]; n += 1;
  }
  out[0] = n; out[1] = ms; out[2] = fl; out[3] = (double)g_prof.launches_all; out[4] = g_prof.total_flops_all;
  return 0;
}

// out[7][3]: per kind k - 0 plain <ACT,0> of the 16-wave 256x256 kernel (THE dominant kernel), 1 SwiGLU fwd <0,1>, 2 SwiGLU bwd <0,2>, 3 RoPE <0,3> of
// the same kernel, 4 the plain 144-row persistent kernel (gemm_nt_144s_kernel<ACT, 0>: ViT / projector products, micro-batch 8),
// 5 the plain long-k products handed to the vendor library (vendor.cpp), 6 those on the four-wave gemm_u4_kernel (gemm_u4.hip) -:
// sampled launches, their summed duration (ms), their summed flops
extern "C" int lhrs_gemm_profile_read_kinds(double* out) {
  for (int i = 0; i < 21; ++i) out[i] = 0;
  for (int i = 0; i < g_prof.used; ++i) {
    float t = 0;
    if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) LHRS_FAIL("gemm_profile_read_kinds: event sync failed");
    if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) LHRS_FAIL("gemm_profile_read_kinds: elapsed failed");
    const int k = g_prof.kind[i];
    out[3 * k] += 1; out[3 * k + 1] += t; out[3 * k + 2] += g_prof.flops[i];
  }
  return 0;
}

static int g_gemm_allow_256 = 2;
// the 16-wave 256x256 kernel walks its tiles PERSISTENTLY: at most one workgroup per CU (128 KiB of LDS each: only one fits anyway),
// workgroup b takes tiles b, b + grid, ...  0 = one workgroup per tile (kernel A/B tests: lhrs_gemm_set_persistent)
static int g_gemm_persist = 1;
extern "C" int lhrs_gemm_set_persistent(int on) { g_gemm_persist = on; return 0; }
#define LAUNCH_144(ACT_, EPI_, grid_, s_, g_) hipLaunchKernelGGL((gemm_nt_144s_kernel<ACT_, EPI_>), grid_, dim3(768), 0, s_, g_)
static int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= 0) n = 256;
    else n = prop.multiProcessorCount;
  }
  return n;
}
static dim3 grid_256s(long tiles) { return dim3((unsigned)(g_gemm_persist ? (tiles < num_cus() ? tiles : num_cus()) : tiles)); }

// ---- stream-K for the last, partial round of the persistent 16-wave kernel -------------------------------------------------------------
// T tiles on P CUs run as ceil(T / P) rounds and a nearly empty last round costs a full one (M = 8190, N = 11008: 1376 tiles = 5.375 rounds
// -> 6; M = 8736, N = 4096: 560 = 2.19 -> 3; M = 2184, N = 4096: 144 = 0.56 -> 1).  With a workspace registered the floor(T / P) full rounds
// run as before and the T % P tiles behind them go to a second launch of the same kernel (SK = true) that cuts their k-loops into P
// contiguous ranges of stages, one per CU: the tail then costs (T % P) / P of a round plus the exchange of the fp32 partials.  The
// workspace belongs to the caller (P slabs of 256 x 256 floats behind 4 KiB of flags = 64 MiB + 4 KiB); launches that use it must be
// ordered on ONE stream.  Summation order is fixed by the shape alone: results are deterministic, and differ from the unsplit kernel's
// only by fp32 re-association inside the split tiles.
static struct { float* slabs; int* flags; long units; } g_sk[16];
// MEASURED (round 4, tools/gemm_sk_ab.py, one box, us per launch whole rounds / stream-K): M = 8190: d-down + SwiGLU' (5.375 rounds) 672 / 691-802,
// gate|up + SwiGLU (10.75) 1121-1153 / 1198-1341, lm_head 773-791 / 799-873; M = 8736 (2.19 rounds): o 239 / 285; M = 2184 (0.56 rounds): o 69
// (144-row tiles) / 126, down 171 / 268.  SLOWER everywhere, for two reasons that belong to this chip, not to the code: (1) the 32 tiles an XCD
// walks concurrently share 8 A and 4 B panels through its L2 only while they sit at the SAME k; ranges that start at different stages of
// their tiles stream every panel from the Infinity Cache on their own (2.8 us per stage instead of 1.5); (2) a 256 KiB fp32 slab costs
// 10-30 us to publish (store-issue-bound, then the L2 write-back of the release) and the unit that needs it finishes at the same moment
// as the unit that writes it, so the exchange is exposed.  Therefore OFF by default; kept, with its tests, as the correct starting point for an
// XCD-lockstep variant (units = groups of 16-32 CUs on 16-32 tiles at one k): DESIGN.md §3.1.
static int g_gemm_streamk = 0;   // lhrs_gemm_set_streamk
extern "C" int lhrs_gemm_set_streamk(int on) { g_gemm_streamk = on; return 0; }
extern "C" long lhrs_gemm_streamk_workspace_bytes() { return 4096 + (long)num_cus() * 256 * 256 * 4; }
// ws == nullptr: forget the workspace of the current device (stream-K off)
extern "C" int lhrs_gemm_set_streamk_workspace(void* ws, long bytes) {
  int dev = 0;
  LHRS_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16, "streamk_workspace: device %d", dev);
  if (ws == nullptr) { g_sk[dev].slabs = nullptr; g_sk[dev].flags = nullptr; g_sk[dev].units = 0; return 0; }
  LHRS_REQUIRE(bytes >= lhrs_gemm_streamk_workspace_bytes() && ((size_t)ws & 255) == 0, "streamk_workspace: %ld bytes (need %ld, 256-B aligned)", bytes,
               lhrs_gemm_streamk_workspace_bytes());
  LHRS_REQUIRE(hipMemset(ws, 0, 4096) == hipSuccess, "streamk_workspace: cannot clear the flags");
  g_sk[dev].flags = (int*)ws; g_sk[dev].slabs = (float*)((char*)ws + 4096); g_sk[dev].units = num_cus();
  return 0;
}
// rounds of the P CUs (in units of one full 256x256 tile's k-loop of nk stages) that T tiles cost on the 16-wave kernel, and whether the
// tail goes to the stream-K launch.  Fixed costs of that launch: a kernel boundary, one prologue per range, the slab round trip ~ 8 stages
static double rounds_256(long T, int nk, int nk2, bool drop, bool* use_sk) {
  const long P = num_cus();
  int dev = 0;
  const long tail = T % P;
  bool sk = g_gemm_streamk && g_gemm_persist && tail != 0 && !drop && nk2 < 4 && nk >= 16 && P <= 1020 &&
            hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16 && g_sk[dev].slabs != nullptr && g_sk[dev].units >= P;
  const double cost_sk = (double)(T / P) + (double)tail / P + 8.0 / nk;
  if (sk && cost_sk > (double)(T / P) + 0.92) sk = false;
  if (use_sk) *use_sk = sk;
  return sk ? cost_sk : (double)((T + P - 1) / P);
}
static int sk_units_for(long sk_tiles, int nk) {
  const long W = sk_tiles * nk, P = num_cus();
  const long u = W / 10 < P ? W / 10 : P;          // >= 10 stages per range on average (>= SK_MINSEG after snapping)
  return (int)(u < 1 ? 1 : u);
}
// The stream-K decomposition as the kernels compute it, replayed on the host (tests; no launch).  T tiles of nk stages (nk2 of them from the
// second operand pair), workspace assumed registered when `assume_ws`.  out[0] = tiles of the whole rounds, out[1] = stream-K tiles, out[2] =
// units.  unit >= 0: out[3] = first tile of the unit's range (relative to out[0]), out[4] = its first stage there, out[5] = stages in that
// tile, out[6] = stages in the next tile, out[7] = 1 if the first item is a PARTIAL (another unit finishes the tile), out[8] / out[9] = partials
// the unit adds to its first / second item.  Returns 0, or -1 when stream-K does not apply.
extern "C" int lhrs_gemm_streamk_plan(long T, int nk, int nk2, int assume_ws, int unit, int* out) {
  const long P = num_cus(), tail = T % P;
  bool sk = tail != 0 && nk2 < 4 && nk >= 16 && (double)tail / P + 8.0 / nk <= 0.92;
  if (!assume_ws) (void)rounds_256(T, nk, nk2, false, &sk);
  if (!sk) return -1;
  const int units = sk_units_for(tail, nk), st = (int)tail;
  out[0] = (int)(T - tail); out[1] = st; out[2] = units;
  if (unit < 0) return 0;
  const int b0 = sk_cut(unit, st, nk, units), b1 = sk_cut(unit + 1, st, nk, units);
  for (int i = 3; i < 10; ++i) out[i] = 0;
  if (b1 <= b0) return 0;
  const int i0 = b0 / nk, S0 = b0 - i0 * nk, e0 = b1 < (i0 + 1) * nk ? b1 : (i0 + 1) * nk;
  out[3] = i0; out[4] = S0; out[5] = e0 - b0; out[6] = b1 - e0; out[7] = S0 != 0;
  for (int seg = 0; seg < 2; ++seg) {
    const int n = seg == 0 ? out[5] : out[6], s0 = seg == 0 ? S0 : 0;
    if (n == 0 || s0 != 0 || n == nk) continue;
    const int tile_end = (i0 + seg + 1) * nk;
    for (int p = unit + 1; p < units && sk_cut(p, st, nk, units) < tile_end; ++p)
      if (sk_cut(p + 1, st, nk, units) > sk_cut(p, st, nk, units)) out[8 + seg]++;
  }
  return 0;
}
template <int ACT, int EPI>
static void launch_256s(GemmArgs& g, hipStream_t s) {
  const long T = (long)g.tilesM * g.tilesN, P = num_cus();
  const int nk = (g.K + g.K2) / 64;
  bool sk = false;
  if (EPI < 4) (void)rounds_256(T, nk, g.K2 / 64, g.drop_thresh != 0, &sk);
  g.ntiles = (int)T; g.sk_tile0 = 0; g.sk_tiles = 0; g.sk_units = 0; g.sk_ws = nullptr; g.sk_flags = nullptr;
  if (sk) g.ntiles = (int)(T / P * P);
  if constexpr (EPI >= 4) {   // the RMSNorm-backward epilogues: whole rounds, no second operand pair (their launchers guarantee both)
    hipLaunchKernelGGL((gemm_nt_256s_kernel<ACT, EPI, false, false>), grid_256s(g.ntiles), dim3(1024), 0, s, g);
    return;
  }
  if (g.ntiles > 0) {
    const dim3 grid = grid_256s(g.ntiles);
    if (g.K2 > 0) hipLaunchKernelGGL((gemm_nt_256s_kernel<ACT, EPI, true, false>), grid, dim3(1024), 0, s, g);
    else hipLaunchKernelGGL((gemm_nt_256s_kernel<ACT, EPI, false, false>), grid, dim3(1024), 0, s, g);
  }
  if (sk) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    g.sk_tile0 = g.ntiles; g.sk_tiles = (int)(T - g.ntiles);
    g.sk_units = sk_units_for(g.sk_tiles, nk);
    g.sk_ws = g_sk[dev].slabs; g.sk_flags = g_sk[dev].flags;
    const dim3 grid((unsigned)g.sk_units);
    if (g.K2 > 0) hipLaunchKernelGGL((gemm_nt_256s_kernel<ACT, EPI, true, true>), grid, dim3(1024), 0, s, g);
    else hipLaunchKernelGGL((gemm_nt_256s_kernel<ACT, EPI, false, true>), grid, dim3(1024), 0, s, g);
  }
}
// the 16-wave kernel; a second operand pair (K2 > 0: the fused LoRA product) is a template parameter of it
#define LAUNCH_256(ACT_, EPI_, grid_, s_, g_) launch_256s<ACT_, EPI_>(g_, s_)
// Tile height of the persistent kernels: 256 rows (16 waves) or 144 rows (12 waves, gemm_144s_kernel.inc).  Both walk ceil(tiles / CUs) rounds.  A
// 144-row tile is 0.5625 of the MFMA work of a 256-row tile but costs ~0.8 of its time: more DMA per MFMA (1050 against 1300 TFLOP/s at
// M = 8190), and a 256-row launch that leaves CUs idle runs its busy CUs at a higher clock (measured at M = 2184, one round each: 73 / 174 / 350
// / 69 / 191 us against 98 / 205 / 482 / 86 / 225 us for o / down / d-gate|up / d-o / d-qkv).  Take whichever finishes first: at M = 2184 the five
// N = 4096 products (144 tiles -> 256) go to 144 rows, qkv / gate|up / d-down (432 / 774 / 387 tiles) stay.  0 = never, 1 = cost model, 2 = whenever legal
static int g_gemm_bm144 = 1;
static double g_gemm_bm144_cost = 0.8;
extern "C" int lhrs_gemm_set_bm144(int mode) { g_gemm_bm144 = mode; return 0; }
static bool pick_144(int M, long tiles_n, int K, int K2, bool drop) {
  if (g_gemm_bm144 == 0 || K2 > 0 || drop || K < 192) return false;   // the second operand pair (fused LoRA) and the dropout mask live in the 256-row kernel only
  if (g_gemm_bm144 == 2) return true;
  const long P = num_cus(), t256 = (long)cdiv(M, 256) * tiles_n, t144 = (long)cdiv(M, 144) * tiles_n;
  const int nk = K / 64;
  if (nk <= 32) {
    // short k-loops (the ViT / projector products, K = 1024 .. 2048): a tile's fixed part - cold first stages, epilogue - weighs as much as
    // its stages, and it is smaller for the 144-row tile.  Fitted on tools/gemm_vit_sweep.py (us per round: 256 rows 1.5 nk + 20, 144 rows
    // 1.15 nk + 5): M = 7710: qkv 86.6 -> 68.1 us, o 30.2 -> 25.9; M = 27360: 264.7 -> 214.4, 88.5 -> 69.7; fc1 (496 tiles = 2 rounds) stays
    return (double)((t144 + P - 1) / P) * (1.15 * nk + 5.0) < (double)((t256 + P - 1) / P) * (1.5 * nk + 20.0);
  }
  return (double)((t144 + P - 1) / P) * g_gemm_bm144_cost < rounds_256(t256, nk, 0, drop, nullptr);
}
// fewest 64x128 tiles for which the 64x128 small-tile kernel is taken over the 64x64 one (A/B: lhrs_gemm_set_small_thresh)
static int g_gemm_small_thresh = 256;
extern "C" int lhrs_gemm_set_small_thresh(int n) { g_gemm_small_thresh = n; return 0; }
static int g_gemm_min256 = 128;  // fewest 256x256 tiles (half a round of the 256 CUs) for which the big-tile kernels are chosen: 2184 x 4096 (144
                                 // tiles, the reference's micro-batch 8) runs 20 % faster there than on 576 small tiles; A/B: lhrs_gemm_set_min_tiles
extern "C" int lhrs_gemm_set_min_tiles(int n) { g_gemm_min256 = n; return 0; }
// tile policy switch for A/B measurements: 0 = never a 256x256 kernel, 2 = default (16-wave BK=64 kernel when K % 64 == 0, else the BK=32
// ring kernel), 4 = always the BK=32 ring kernel
extern "C" int lhrs_gemm_set_policy(int allow_256) { g_gemm_allow_256 = allow_256; return 0; }

// C ABI ------------------------------------------------------------------------------------------
static int gemm_launch(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const void* bias,
                       const void* residual, int ldr, int act, int out_f32, int accumulate, float alpha, const void* A2,
                       int lda2, const void* B2, int ldb2, int K2, void* stream);
// epilogue dropout mask of the NEXT gemm_launch on this host thread: set and cleared by lhrs_gemm_bf16_nt_dropmask only (keeps the
// 21-argument launcher signature out of every other call site); thresh = 0 means no mask
static thread_local struct { float scale; unsigned seed, thresh; } t_drop = {1.f, 0u, 0u};
static thread_local bool t_split_ok = true;  // false inside the two launches of a row-split product (see the tail-row rule in gemm_launch)
static int g_gemm_tail_split = 1;            // kernel A/B tests only (lhrs_gemm_set_tail_split)
extern "C" int lhrs_gemm_set_tail_split(int on) { g_gemm_tail_split = on; return 0; }

// C = mask * (alpha * A.B^T) / (1 - p) + residual, mask = the counter-based LoRA dropout mask over the [M, N] result (see common.h)
extern "C" int lhrs_gemm_bf16_nt_dropmask(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                                          const void* residual, int ldr, float alpha, float p, unsigned seed, void* stream) {
  LHRS_REQUIRE(p > 0.f && p < 1.f, "gemm_dropmask: p=%f", p);
  t_drop.scale = 1.f / (1.f - p); t_drop.seed = seed; t_drop.thresh = (unsigned)((double)p * 4294967296.0);
  const int rc = gemm_launch(A, lda, B, ldb, C, ldc, M, N, K, nullptr, residual, ldr, 0, 0, 0, alpha, nullptr, 0, nullptr, 0, 0, stream);
  t_drop.thresh = 0;
  return rc;
}

// ---- plain long-k products: three candidates, chosen per problem by measurement ---------------------------------------------------------
// A product with a plain epilogue (no bias, no activation, bf16 out, alpha 1) on a long k-loop can run (0) gemm_nt_256s_kernel / gemm_nt_144s_kernel of this
// file, (2) the four-wave gemm_u4_kernel (gemm_u4.hip: 128x128 per wave, paced DMA - 5-18 % faster than (0) on these shapes) or (1) the vendor library's
// assembly kernel (vendor.cpp).  Per problem (device, M, N, K, leading dims, residual or not) the FIRST call times all three - every algorithm the library's
// heuristic offers (lhrs_vendor_gemm_tune), 1 untimed + 3 timed launches each on the caller's operands and stream - and later calls repeat the winner.  The
// library's first heuristic answer alone is not safe to follow: tools/gemm_vendor_ab.py has it 1.4-1.8x SLOWER than gemm_nt_256s_kernel at M = 5460
// (K >= 11008) and at M = 2184, K = 22016.  A hand-written kernel keeps the problem unless the library is more than 3 % faster, so that two ranks rarely
// disagree over noise; every choice is a correct bf16 product ((0) and (2) bit-identical unless a residual is added: (1) and (2) round once).  Not timed (kernel (0), nothing cached): a capturing stream, C
// aliasing an input, more than 96 problems in one process.  lhrs_gemm_set_vendor / LHRS_GEMM_VENDOR=0 and lhrs_gemm_set_u4 / LHRS_GEMM_U4=0 remove a candidate.
extern "C" int lhrs_vendor_gemm_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const void* residual, int ldr,
                                   void* workspace, long workspace_bytes, void* stream);
extern "C" int lhrs_vendor_gemm_tune(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const void* residual, int ldr,
                                     void* workspace, long workspace_bytes, int reps, float* best_us, void* stream);
extern "C" int lhrs_gemm_u4_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const void* residual, int ldr,
                               void* stream);
static int g_vendor_on = -1, g_u4_on = -1, g_vendor_min_k = 4096;
static int prof_count(int M, int N, int K, int kind, hipStream_t s);
static void prof_end(int slot, hipStream_t s);
static void plain_env() {
  if (g_vendor_on < 0) {
    const char* e = getenv("LHRS_GEMM_VENDOR");
    g_vendor_on = (e == nullptr || e[0] != '0') ? 1 : 0;
    const char* k = getenv("LHRS_GEMM_VENDOR_MIN_K");
    if (k != nullptr && atoi(k) > 0) g_vendor_min_k = atoi(k);
  }
  if (g_u4_on < 0) {
    const char* e = getenv("LHRS_GEMM_U4");
    g_u4_on = (e == nullptr || e[0] != '0') ? 1 : 0;
  }
}
extern "C" int lhrs_gemm_set_vendor(int on, int min_k) {
  plain_env();
  g_vendor_on = on ? 1 : 0;
  if (min_k > 0) g_vendor_min_k = min_k;
  return 0;
}
extern "C" int lhrs_gemm_set_u4(int on) { plain_env(); g_u4_on = on ? 1 : 0; return 0; }
// 1 when lhrs_gemm_bf16_nt decides this problem by first-call timing (a plain epilogue on a long k-loop, operands every candidate can address)
static int plain_timed(int M, int N, int K, int lda, int ldb, int ldc, int ldr, int has_bias, int act, int out_f32, int accumulate, float alpha) {
  plain_env();
  return (g_vendor_on == 1 || g_u4_on == 1) && !has_bias && act == 0 && !out_f32 && !accumulate && alpha == 1.f && K >= g_vendor_min_k && K % 64 == 0 &&
         M >= 1024 && N >= 1024 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && ldr % 8 == 0;
}
// 1 when the vendor library is among the candidates for this problem
extern "C" int lhrs_gemm_vendor_takes(int M, int N, int K, int lda, int ldb, int ldc, int ldr, int has_bias, int act, int out_f32, int accumulate, float alpha) {
  return plain_timed(M, N, K, lda, ldb, ldc, ldr, has_bias, act, out_f32, accumulate, alpha) && g_vendor_on == 1;
}
static std::map<std::array<long, 11>, int> g_plain_choice;   // -> 0 this file's kernels, 1 library, 2 gemm_u4_kernel
static long g_vendor_stats[3] = {0, 0, 0};                   // problems decided, -> library, -> hand-written (either kernel)
static long g_u4_problems = 0;                               // of the hand-written ones: -> gemm_u4_kernel
extern "C" int lhrs_gemm_vendor_stats(long* out3) { for (int i = 0; i < 3; ++i) out3[i] = g_vendor_stats[i]; return 0; }
extern "C" long lhrs_gemm_u4_problems() { return g_u4_problems; }
static int gemm_launch(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const void* bias,
                       const void* residual, int ldr, int act, int out_f32, int accumulate, float alpha, const void* A2,
                       int lda2, const void* B2, int ldb2, int K2, void* stream);
static bool overlaps(const void* p, long bytes_p, const void* q, long bytes_q) {
  const char* a = (const char*)p; const char* b = (const char*)q;
  return q != nullptr && a < b + bytes_q && b < a + bytes_p;
}

extern "C" int lhrs_gemm_bf16_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N,
                                 int K, const void* bias, const void* residual, int ldr, int act, int out_f32,
                                 int accumulate, float alpha, void* stream) {
  if (plain_timed(M, N, K, lda, ldb, ldc, residual ? ldr : 0, bias != nullptr, act, out_f32, accumulate, alpha) &&
      ((size_t)A % 16 == 0) && ((size_t)B % 16 == 0) && ((size_t)C % 16 == 0) && ((size_t)residual % 16 == 0)) {
    int dev = 0;
    void* ws = nullptr; long ws_bytes = 0;
    hipStream_t s = (hipStream_t)stream;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16 && g_sk[dev].slabs != nullptr) {   // the registered workspace (lhrs_gemm_set_streamk_workspace)
      ws = g_sk[dev].slabs; ws_bytes = g_sk[dev].units * 256L * 256 * 4;
    }
    const std::array<long, 11> key = {dev, M, N, K, lda, ldb, ldc, residual ? ldr : 0, ws_bytes, g_vendor_on, g_u4_on};
    auto it = g_plain_choice.find(key);
    int choice = it == g_plain_choice.end() ? -1 : it->second;
    if (choice < 0) {
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      const long cb = ((long)(M - 1) * ldc + N) * 2;
      const bool tunable = hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone &&
                           !overlaps(C, cb, A, ((long)(M - 1) * lda + K) * 2) && !overlaps(C, cb, B, ((long)(N - 1) * ldb + K) * 2) &&
                           !overlaps(C, cb, residual, ((long)(M - 1) * ldr + N) * 2);
      if (tunable && g_vendor_stats[0] < 96) {   // bounded: a caller that walks through many row counts (ragged prefill batches) stops paying for timing runs
        const bool prof_was = g_prof.on;
        g_prof.on = false;                                     // the timing launches are not part of the step
        float t_lib = 1e30f, t_hand = 1e30f, t_u4 = 1e30f;
        if (g_vendor_on == 1 && lhrs_vendor_gemm_tune(A, lda, B, ldb, C, ldc, M, N, K, residual, ldr, ws, ws_bytes, 3, &t_lib, stream) != 0) t_lib = 1e30f;
        hipEvent_t e0, e1;
        if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
          for (int cand = 0; cand < 2; ++cand) {                // 0: this file's kernels, 1: gemm_u4_kernel
            if (cand == 1 && g_u4_on != 1) break;
            auto run = [&]() {
              return cand == 0 ? gemm_launch(A, lda, B, ldb, C, ldc, M, N, K, nullptr, residual, ldr, 0, 0, 0, 1.f, nullptr, 0, nullptr, 0, 0, stream)
                               : lhrs_gemm_u4_nt(A, lda, B, ldb, C, ldc, M, N, K, residual, ldr, stream);
            };
            int st = run();
            (void)hipEventRecord(e0, s);
            for (int r = 0; r < 3 && st == 0; ++r) st = run();
            (void)hipEventRecord(e1, s);
            float ms = 0.f;
            if (st == 0 && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) (cand == 0 ? t_hand : t_u4) = ms / 3.f * 1e3f;
            if (st < 0) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); g_prof.on = prof_was; return st; }
          }
          (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
        g_prof.on = prof_was;
        const float t_own = t_u4 < t_hand ? t_u4 : t_hand;
        choice = t_lib < 0.97f * t_own ? 1 : (t_u4 < t_hand ? 2 : 0);
        g_plain_choice[key] = choice;
        g_vendor_stats[0]++; g_vendor_stats[choice == 1 ? 1 : 2]++;
        if (choice == 2) g_u4_problems++;
        if (getenv("LHRS_GEMM_VENDOR_LOG") != nullptr)
          fprintf(stderr, "[lhrs gemm] M=%d N=%d K=%d%s: library %.1f us, 16-wave kernel %.1f us, 4-wave kernel %.1f us -> %s\n", M, N, K, residual ? " +residual" : "",
                  t_lib < 1e29f ? t_lib : -1.f, t_hand < 1e29f ? t_hand : -1.f, t_u4 < 1e29f ? t_u4 : -1.f,
                  choice == 1 ? "library" : choice == 2 ? "4-wave kernel" : "16-wave kernel");
      } else {
        choice = 0;
      }
    }
    if (choice == 1) {
      const int slot = prof_count(M, N, K, 5, s);
      const int st = lhrs_vendor_gemm_nt(A, lda, B, ldb, C, ldc, M, N, K, residual, ldr, ws, ws_bytes, stream);
      if (st == 0) { prof_end(slot, s); return 0; }
      if (st < 0) return st;
      if (slot >= 0) { g_prof.used--; g_prof.seen[5]--; }   // not taken after all: the slot goes back (it was the last one handed out)
      if (g_prof.on) { g_prof.launches_all--; g_prof.total_flops_all -= 2.0 * M * N * K; }
    } else if (choice == 2) {
      const int slot = prof_count(M, N, K, 6, s);
      const int st = lhrs_gemm_u4_nt(A, lda, B, ldb, C, ldc, M, N, K, residual, ldr, stream);
      if (st == 0) { prof_end(slot, s); return 0; }
      if (st < 0) return st;
      if (slot >= 0) { g_prof.used--; g_prof.seen[6]--; }
      if (g_prof.on) { g_prof.launches_all--; g_prof.total_flops_all -= 2.0 * M * N * K; }
    }
  }
  return gemm_launch(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, act, out_f32, accumulate, alpha, nullptr, 0, nullptr, 0,
                     0, stream);
}

// C = alpha * (A.B^T + A2.B2^T) + bias + residual: the rank-K2 LoRA update rides in the k-loop of the base GEMM
// (peft lora.Linear forward, y = W x + (alpha/r) B A x, reached from lhrs/models/text_modal.py:133-151).
extern "C" int lhrs_gemm_bf16_nt_lora(const void* A, int lda, const void* B, int ldb, const void* A2, int lda2, const void* B2,
                                      int ldb2, int K2, void* C, int ldc, int M, int N, int K, const void* bias,
                                      const void* residual, int ldr, int out_f32, int accumulate, float alpha, void* stream) {
  LHRS_REQUIRE(A2 && B2 && K2 > 0 && K2 % 64 == 0 && lda2 % 8 == 0 && ldb2 % 8 == 0 && lda2 >= K2 && ldb2 >= K2,
               "gemm_lora: bad second operand pair (K2=%d lda2=%d ldb2=%d)", K2, lda2, ldb2);
  return gemm_launch(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, 0, out_f32, accumulate, alpha, A2, lda2, B2, ldb2, K2,
                     stream);
}

static int gemm_launch(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const void* bias,
                       const void* residual, int ldr, int act, int out_f32, int accumulate, float alpha, const void* A2,
                       int lda2, const void* B2, int ldb2, int K2, void* stream) {
  LHRS_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
  LHRS_REQUIRE(K % 32 == 0, "gemm: K=%d must be a multiple of 32 (zero-pad the reduction dim)", K);
  LHRS_REQUIRE(N % 4 == 0, "gemm: N=%d must be a multiple of 4", N);
  LHRS_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, "gemm: lda=%d ldb=%d must be multiples of 8 (16-B rows)", lda, ldb);
  LHRS_REQUIRE(ldc % 4 == 0 && (residual == nullptr || ldr % 4 == 0), "gemm: ldc/ldr must be multiples of 4");
  LHRS_REQUIRE(lda >= K && ldb >= K && ldc >= N, "gemm: leading dims too small");
  LHRS_REQUIRE(!accumulate || out_f32, "gemm: accumulate needs f32 output");
  LHRS_REQUIRE(act >= 0 && act <= 3, "gemm: unknown activation %d", act);
  GemmArgs g;
  g.epi = 0; g.ff = 0; g.aux = nullptr; g.aux_out = nullptr; g.ld_aux = 0; g.sa = nullptr; g.sb = nullptr; g.ksplit = 0; g.drop_scale = 1.f; g.drop_seed = 0; g.drop_thresh = 0;
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = C;
  g.bias = (const bf16_t*)bias; g.res = (const bf16_t*)residual;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr;
  g.alpha = alpha; g.act = act; g.out_f32 = out_f32; g.accum = accumulate;
  g.A2 = (const bf16_t*)A2; g.B2 = (const bf16_t*)B2; g.lda2 = lda2; g.ldb2 = ldb2; g.K2 = K2;
  g.drop_scale = t_drop.scale; g.drop_seed = t_drop.seed; g.drop_thresh = t_drop.thresh;
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
  const long t256 = (long)cdiv(M, 256) * cdiv(N, 256);
  const bool al16 = out_f32 || (N % 8 == 0 && ldc % 8 == 0 && (residual == nullptr || ldr % 8 == 0));  // 16-B epilogue rows
  bool use256 = g_gemm_allow_256 && t256 >= g_gemm_min256 && K >= 96 && K % 32 == 0 && al16;  // ring prologue needs >= 3 stages
  // just under the big-tile threshold but nearly a full round of 144-row tiles (ViT o / fc2 at micro-batch 30: 124 tiles of 256 rows, 216 of
  // 144): the persistent 144-row kernel beats the small tiles (30.2 -> 25.9, 75.1 -> 68.4 us; tools/gemm_vit_sweep.py)
  if (!use256 && g_gemm_allow_256 == 2 && g_gemm_bm144 == 1 && al16 && !out_f32 && K % 64 == 0 && K >= 192 && K2 == 0 && !g.drop_thresh &&
      t256 >= g_gemm_min256 / 2 && (long)cdiv(M, 144) * cdiv(N, 256) >= 200 && (long)cdiv(M, 144) * cdiv(N, 256) <= num_cus())
    use256 = true;
  if (g.drop_thresh && !(g_gemm_allow_256 == 2 && K % 64 == 0 && K >= 128)) use256 = false;  // the mask lives in the 16-wave kernel and in store4
  if (K2 > 0 && !use256) {  // small problems: base GEMM, then the rank-K2 update accumulated on top of it
    if (gemm_launch(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, act, out_f32, accumulate, alpha, nullptr, 0, nullptr, 0, 0, stream))
      return -1;
    return gemm_launch(A2, lda2, B2, ldb2, C, ldc, M, N, K2, nullptr, out_f32 ? nullptr : C, ldc, 0, out_f32, out_f32 ? 1 : 0, alpha,
                       nullptr, 0, nullptr, 0, 0, stream);
  }
  LHRS_REQUIRE(use256 || K % 64 == 0, "gemm: K=%d must be a multiple of 64 for this problem size (zero-pad the reduction dim)", K);
  // Tail rows: T = tilesM * tilesN 256x256 tiles run as ceil(T / 256) rounds of the 256 CUs, and a nearly empty last round costs as
  // much as a full one (M = 8736, N = 4096: 560 tiles = 2.19 rounds -> 3).  When the tile rows that spill over the last full round
  // are cheaper as a separate small-tile launch (~2.5x the time per FLOP, but no idle CUs), the row range is cut there: whole
  // 256-row tile rows for the 16-wave kernel, the remaining rows for the 64x128 / 128x128 kernel.  Disjoint rows of C, no partials.
  const bool s_kernel = use256 && g_gemm_allow_256 == 2 && K % 64 == 0 && K2 % 64 == 0 && K + K2 >= 128;
  const bool bm144 = s_kernel && !out_f32 && pick_144(M, cdiv(N, 256), K, K2, g.drop_thresh != 0);
  if (s_kernel && !bm144 && t_split_ok && g_gemm_tail_split && !g.drop_thresh) {
    const int tm = cdiv(M, 256), tn = cdiv(N, 256);
    const long T = (long)tm * tn, rounds = (T + 255) / 256, full = T / 256;
    const int tm_main = (int)(full * 256 / tn);
    if (full >= 1 && T % 256 != 0 && tm_main >= 1 && tm_main < tm) {
      const long t_main = (long)tm_main * tn, t_tail = T - t_main;
      const double split_cost = (double)((t_main + 255) / 256) + 2.5 * (double)t_tail / 256.0 + 0.05;
      (void)rounds;
      if (split_cost < rounds_256(T, (K + K2) / 64, K2 / 64, false, nullptr)) {
        const int M_main = tm_main * 256, M_tail = M - M_main;
        const long esz = out_f32 ? 4 : 2;
        t_split_ok = false;
        int rc = gemm_launch(A, lda, B, ldb, C, ldc, M_main, N, K, bias, residual, ldr, act, out_f32, accumulate, alpha, A2, lda2, B2, ldb2, K2, stream);
        // the tail rows of a LONG k-loop (M = 8736: 544 rows x 4096 columns = 160 tiles of 128^2, each walking K = 11008 .. 22016 alone: 100 - 190
        // us at 500 TFLOP/s, tools/gemm_tail_sweep.py): cut K into slabs of ~4096 across blockIdx.y - f32 slabs in the registered workspace,
        // summed in a fixed order with the residual by one small launch - so that every CU has work for the whole tail
        int dev = 0;
        const int splits = K >= 8192 ? (K + 2048) / 4096 : 1;
        if (!rc && splits > 1 && !out_f32 && K2 == 0 && bias == nullptr && act == 0 && N % 128 == 0 && hipGetDevice(&dev) == hipSuccess && dev >= 0 &&
            dev < 16 && g_sk[dev].slabs != nullptr && (long)splits * M_tail * N * 4 <= g_sk[dev].units * 256 * 256 * 4) {
          const int ks = cdiv(K / 64, splits) * 64, used = cdiv(K, ks);
          GemmArgs gt; memset(&gt, 0, sizeof(gt));
          gt.A = (const bf16_t*)A + (long)M_main * lda; gt.B = (const bf16_t*)B; gt.C = g_sk[dev].slabs; gt.M = M_tail; gt.N = N; gt.K = K;
          gt.lda = lda; gt.ldb = ldb; gt.ldc = N; gt.alpha = 1.f; gt.out_f32 = 1; gt.ksplit = ks; gt.drop_scale = 1.f;
          gt.tilesM = cdiv(M_tail, 128); gt.tilesN = cdiv(N, 128);
          if (g_prof.on) { g_prof.launches_all++; g_prof.total_flops_all += 2.0 * M_tail * N * K; }
          hipLaunchKernelGGL((gemm_nt_kernel<4, 4, 0>), dim3(gt.tilesM * gt.tilesN, used), dim3(256), 0, s, gt);
          const long work = (long)M_tail * (N / 4);
          int rg = (int)((work + 255) / 256); if (rg > 8192) rg = 8192;
          hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rg), dim3(256), 0, s, (const float*)g_sk[dev].slabs, (bf16_t*)C + (long)M_main * ldc, (long)ldc, M_tail, N,
                             used, alpha, residual ? (const bf16_t*)residual + (long)M_main * ldr : nullptr, (long)ldr);
          t_split_ok = true;
          LHRS_CHECK_LAUNCH("gemm_tail_splitk");
          return 0;
        }
        if (!rc)
          rc = gemm_launch((const bf16_t*)A + (long)M_main * lda, lda, B, ldb, (char*)C + (long)M_main * ldc * esz, ldc, M_tail, N, K, bias,
                           residual ? (const bf16_t*)residual + (long)M_main * ldr : nullptr, ldr, act, out_f32, accumulate, alpha,
                           A2 ? (const bf16_t*)A2 + (long)M_main * lda2 : nullptr, lda2, B2, ldb2, K2, stream);
        t_split_ok = true;
        return rc;
      }
    }
  }
  const bool big = t128 >= 384;
  int slot = -1;
  if (g_prof.on) {
    g_prof.launches_all++; g_prof.total_flops_all += 2.0 * M * N * (K + K2);
    const bool dominant = use256 && g_gemm_allow_256 == 2 && K % 64 == 0 && K2 % 64 == 0 && K + K2 >= 128;
    const int pkind = bm144 ? 4 : 0;
    if (dominant && g_prof.used < g_prof.cap && g_prof.take(pkind)) {  // time the launches rocprof lists as gemm_nt_256s_kernel<ACT, 0 ...> (kind 0) / gemm_nt_144s_kernel<ACT, 0> (kind 4)
      slot = g_prof.used++;
      g_prof.flops[slot] = 2.0 * M * N * (K + K2);
      g_prof.kind[slot] = pkind;
      (void)hipEventRecord(g_prof.ev[2 * slot], s);
    }
  }
  if (use256) {
    g.tilesM = cdiv(M, 256); g.tilesN = cdiv(N, 256);
    const dim3 grid(g.tilesM * g.tilesN), blk(512);
    if (bm144) {
      g.tilesM = cdiv(M, 144);
      const dim3 grid12 = grid_256s((long)g.tilesM * g.tilesN);
      switch (act) {
        case 0: LAUNCH_144(0, 0, grid12, s, g); break;
        case 1: LAUNCH_144(1, 0, grid12, s, g); break;
        case 2: LAUNCH_144(2, 0, grid12, s, g); break;
        default: LAUNCH_144(3, 0, grid12, s, g); break;
      }
    } else if (s_kernel) {
      const dim3 grid16 = grid_256s((long)g.tilesM * g.tilesN);
      switch (act) {
        case 0: LAUNCH_256(0, 0, grid16, s, g); break;
        case 1: LAUNCH_256(1, 0, grid16, s, g); break;
        case 2: LAUNCH_256(2, 0, grid16, s, g); break;
        default: LAUNCH_256(3, 0, grid16, s, g); break;
      }
    } else {
      switch (act) {
        case 0: hipLaunchKernelGGL((gemm_nt_256p_kernel<0>), grid, blk, 0, s, g); break;
        case 1: hipLaunchKernelGGL((gemm_nt_256p_kernel<1>), grid, blk, 0, s, g); break;
        case 2: hipLaunchKernelGGL((gemm_nt_256p_kernel<2>), grid, blk, 0, s, g); break;
        default: hipLaunchKernelGGL((gemm_nt_256p_kernel<3>), grid, blk, 0, s, g); break;
      }
    }
  } else if (big) LAUNCH_TILE(4, 4);
  else if (t64x128 >= g_gemm_small_thresh) LAUNCH_TILE(2, 4);
  else LAUNCH_TILE(2, 2);
#undef LAUNCH_TILE
  if (slot >= 0) (void)hipEventRecord(g_prof.ev[2 * slot + 1], s);
  LHRS_CHECK_LAUNCH("gemm_bf16_nt");
  return 0;
}


// ---- LLaMA MLP with the SwiGLU fused into the GEMM epilogues (HF LlamaMLP, reached from lhrs/models/text_modal.py:258-294) ----------
// forward : gu[M, 2*ff] = x W_gu^T (+ LoRA pair), act[M, ff] = silu(gu[:, :ff]) * gu[:, ff:]     - one launch instead of GEMM + swiglu_fwd
// backward: dgu[M, 2*ff] = swiglu'(gu) * (dy W_down) (+ LoRA pair); dgu may alias gu             - one launch instead of GEMM + swiglu_bwd
// Results are bit-identical to the unfused sequence (the epilogue rounds gate / up / d_act to bf16 exactly where the unfused path
// stores them).  Shapes the 16-wave 256x256 kernel does not take (K % 64, fewer tiles than the big-tile threshold, ff % 128) fall back to the unfused sequence.
extern "C" int lhrs_swiglu_fwd(const void* gate_up, void* act, long rows, int F, void* stream);
extern "C" int lhrs_swiglu_bwd(const void* dact, const void* gate_up, void* dgate_up, long rows, int F, void* stream);
extern "C" int lhrs_rope(void* x, long ld, int rows, int nheads, int D, const float* cos_t, const float* sin_t, const int* pos_ids, int pos_mod,
                         int pos0, int inverse, void* stream);

static bool swiglu_fusable(long tiles, int ff, int K, int K2, int lda, int ldb) {
  return g_gemm_allow_256 == 2 && ff % 256 == 0 && K % 64 == 0 && K2 % 64 == 0 && K + K2 >= 128 && tiles >= g_gemm_min256 &&
         lda % 8 == 0 && ldb % 8 == 0;
}
// 1 when lhrs_gemm_swiglu_fwd / _bwd will take the fused kernel for this problem (dense operands), else 0 (they fall back)
extern "C" int lhrs_gemm_swiglu_fusable(int M, int ff, int K_fwd, int K_bwd, int K2) {
  const long tf = (long)cdiv(M, 256) * (ff / 128), tb = (long)cdiv(M, 256) * cdiv(ff, 256);
  return swiglu_fusable(tf, ff, K_fwd, K2, 8, 8) && swiglu_fusable(tb, ff, K_bwd, K2, 8, 8);
}

// the fused-epilogue launches count towards the step's GEMM FLOPs but are NOT timed as "the dominant kernel": their epilogues do
// elementwise work (SwiGLU) that has no FLOPs in the GEMM roofline - the live roofline figure is the plain gemm_nt_256s_kernel<ACT, 0, K2P>
static int prof_count(int M, int N, int K, int kind, hipStream_t s) {
  if (!g_prof.on) return -1;
  g_prof.launches_all++; g_prof.total_flops_all += 2.0 * M * N * K;
  if (g_prof.used >= g_prof.cap || !g_prof.take(kind)) return -1;
  const int slot = g_prof.used++;
  g_prof.flops[slot] = 2.0 * M * N * K; g_prof.kind[slot] = kind;
  (void)hipEventRecord(g_prof.ev[2 * slot], s);
  return slot;
}
static void prof_end(int slot, hipStream_t s) {
  if (slot >= 0) (void)hipEventRecord(g_prof.ev[2 * slot + 1], s);
}

extern "C" int lhrs_gemm_swiglu_fwd(const void* X, int ldx, const void* Wgu, int ldw, const void* A2, int lda2, const void* B2, int ldb2,
                                    int K2, void* gu, int ld_gu, void* act, int ld_act, int M, int ff, int K, void* stream) {
  LHRS_REQUIRE(M > 0 && ff > 0 && K > 0 && ff % 8 == 0 && ld_gu >= 2 * ff && ld_act >= ff && ld_gu % 8 == 0 && ld_act % 8 == 0,
               "gemm_swiglu_fwd: M=%d ff=%d K=%d ld_gu=%d ld_act=%d", M, ff, K, ld_gu, ld_act);
  if (!swiglu_fusable((long)cdiv(M, 256) * (ff / 128), ff, K, K2, ldx, ldw)) {
    if (gemm_launch(X, ldx, Wgu, ldw, gu, ld_gu, M, 2 * ff, K, nullptr, nullptr, 0, 0, 0, 0, 1.f, A2, lda2, B2, ldb2, K2, stream)) return -1;
    LHRS_REQUIRE(ld_gu == 2 * ff && ld_act == ff, "gemm_swiglu_fwd: the unfused fallback needs dense gu / act");
    return lhrs_swiglu_fwd(gu, act, M, ff, stream);
  }
  // Tail rows, as in gemm_launch: when the tile rows that spill over the last full round of the 256 CUs are cheaper as a small-tile launch
  // (M = 2184, the reference's micro-batch 8: 9 x 86 = 774 tiles = 3 rounds + SIX tiles), the fused kernel takes the whole tile rows and the
  // remaining rows go through the plain GEMM + the SwiGLU kernel - the same bf16 gate|up rows, the same silu(gate) * up on them
  const bool bm144 = pick_144(M, ff / 128, K, K2, false);
  if (!bm144 && t_split_ok && g_gemm_tail_split && ld_gu == 2 * ff && ld_act == ff) {
    const int tm = cdiv(M, 256), tn = ff / 128;
    const long T = (long)tm * tn, rounds = (T + 255) / 256, full = T / 256;
    const int tm_main = (int)(full * 256 / tn);
    if (full >= 1 && T % 256 != 0 && tm_main >= 1 && tm_main < tm) {
      const long t_main = (long)tm_main * tn, t_tail = T - t_main;
      if ((double)((t_main + 255) / 256) + 2.5 * (double)t_tail / 256.0 + 0.05 < (double)rounds) {
        const int M_main = tm_main * 256, M_tail = M - M_main;
        t_split_ok = false;
        int rc = lhrs_gemm_swiglu_fwd(X, ldx, Wgu, ldw, A2, lda2, B2, ldb2, K2, gu, ld_gu, act, ld_act, M_main, ff, K, stream);
        t_split_ok = true;
        if (rc) return rc;
        bf16_t* gu_t = (bf16_t*)gu + (long)M_main * ld_gu;
        if (gemm_launch((const bf16_t*)X + (long)M_main * ldx, ldx, Wgu, ldw, gu_t, ld_gu, M_tail, 2 * ff, K, nullptr, nullptr, 0, 0, 0, 0, 1.f,
                        A2 ? (const bf16_t*)A2 + (long)M_main * lda2 : nullptr, lda2, B2, ldb2, K2, stream))
          return -1;
        return lhrs_swiglu_fwd(gu_t, (bf16_t*)act + (long)M_main * ld_act, M_tail, ff, stream);
      }
    }
  }
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.A = (const bf16_t*)X; g.B = (const bf16_t*)Wgu; g.C = gu; g.M = M; g.N = 2 * ff; g.K = K; g.lda = ldx; g.ldb = ldw; g.ldc = ld_gu;
  g.alpha = 1.f; g.A2 = (const bf16_t*)A2; g.B2 = (const bf16_t*)B2; g.lda2 = lda2; g.ldb2 = ldb2; g.K2 = K2;
  g.epi = 1; g.ff = ff; g.aux_out = (bf16_t*)act; g.ld_aux = ld_act;
  g.tilesM = cdiv(M, bm144 ? 144 : 256); g.tilesN = ff / 128;
  hipStream_t s = (hipStream_t)stream;
  const int pslot = prof_count(M, 2 * ff, K + K2, 1, s);
  if (bm144) LAUNCH_144(0, 1, grid_256s((long)g.tilesM * g.tilesN), s, g);
  else LAUNCH_256(0, 1, grid_256s((long)g.tilesM * g.tilesN), s, g);
  prof_end(pslot, s);
  LHRS_CHECK_LAUNCH("gemm_swiglu_fwd");
  return 0;
}

// qkv = x . Wqkv^T (+ fused LoRA pair) with the RoPE of the q / k heads (HF apply_rotary_pos_emb, rotate_half convention, head_dim 128)
// applied in the GEMM epilogue: columns [0, rope_cols) are heads of 128 that get rotated with the position m % pos_mod + pos0 of
// their row, the remaining columns (v) are stored as computed.  Bit-identical to lhrs_gemm_bf16_nt(_lora) followed by lhrs_rope; that
// pair is also the fallback when the 256-tile kernel does not apply.
extern "C" int lhrs_gemm_u4_rope(const void* X, int ldx, const void* W, int ldw, void* C, int ldc, int M, int N, int K, const float* cos_t, const float* sin_t,
                                 int pos_mod, int pos0, int rope_cols, void* stream);
static int g_u4_rope = -1;
extern "C" int lhrs_gemm_set_u4_rope(int on) { g_u4_rope = on ? 1 : 0; return 0; }
static bool u4_rope_on() {
  if (g_u4_rope < 0) { const char* e = getenv("LHRS_GEMM_U4_ROPE"); g_u4_rope = (e != nullptr && e[0] == '1') ? 1 : 0; }
  plain_env();
  return g_u4_rope == 1 && g_u4_on == 1;   // LHRS_GEMM_U4=0 / lhrs_gemm_set_u4(0): the 16-wave kernels everywhere
}
extern "C" int lhrs_gemm_rope_fwd(const void* X, int ldx, const void* W, int ldw, const void* A2, int lda2, const void* B2, int ldb2, int K2,
                                  void* C, int ldc, int M, int N, int K, const float* cos_t, const float* sin_t, int pos_mod, int pos0,
                                  int rope_cols, int head_dim, void* stream) {
  LHRS_REQUIRE(M > 0 && N > 0 && K > 0 && head_dim % 16 == 0 && rope_cols >= 0 && rope_cols <= N && rope_cols % head_dim == 0 && pos_mod > 0 &&
                   cos_t && sin_t && ldc % 8 == 0,
               "gemm_rope_fwd: M=%d N=%d K=%d rope_cols=%d head_dim=%d pos_mod=%d", M, N, K, rope_cols, head_dim, pos_mod);
  const long tiles = (long)cdiv(M, 256) * cdiv(N, 256);
  const bool fused = g_gemm_allow_256 == 2 && head_dim == 128 && rope_cols % 256 == 0 && N % 8 == 0 && K % 64 == 0 &&
                     K2 % 64 == 0 && K + K2 >= 128 && tiles >= g_gemm_min256 && ldx % 8 == 0 && ldw % 8 == 0;
  if (!fused) {
    if (gemm_launch(X, ldx, W, ldw, C, ldc, M, N, K, nullptr, nullptr, 0, 0, 0, 0, 1.f, A2, lda2, B2, ldb2, K2, stream)) return -1;
    if (rope_cols == 0) return 0;
    return lhrs_rope(C, ldc, M, rope_cols / head_dim, head_dim, cos_t, sin_t, nullptr, pos_mod, pos0, 0, stream);
  }
  // the four-wave kernel's RoPE variant (gemm_u4.hip), OPT-IN (LHRS_GEMM_U4_ROPE=1): same result bit for bit, 666 vs 722 us at M = 8190 and 218 vs 225 at M = 2184
  // back to back (tools/gemm_u4_rope_ab.py) - but inside the power-capped step 634-636 vs 639 us per launch and no faster a step (156.2-156.9 vs 157.4-157.7
  // samples/s, two same-box pairs): stall removal converts at 15-20 % there, and at K = 4096 this loop has little else to offer
  if (K2 == 0 && u4_rope_on() && M >= 1024) {
    const int pslot = prof_count(M, N, K, 3, (hipStream_t)stream);
    const int st = lhrs_gemm_u4_rope(X, ldx, W, ldw, C, ldc, M, N, K, cos_t, sin_t, pos_mod, pos0, rope_cols, stream);
    if (st == 0) { prof_end(pslot, (hipStream_t)stream); return 0; }
    if (st < 0) return st;
    if (pslot >= 0) { g_prof.used--; g_prof.seen[3]--; }
    if (g_prof.on) { g_prof.launches_all--; g_prof.total_flops_all -= 2.0 * M * N * K; }
  }
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.A = (const bf16_t*)X; g.B = (const bf16_t*)W; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = ldx; g.ldb = ldw; g.ldc = ldc;
  g.alpha = 1.f; g.A2 = (const bf16_t*)A2; g.B2 = (const bf16_t*)B2; g.lda2 = lda2; g.ldb2 = ldb2; g.K2 = K2;
  g.epi = 3; g.rope_cos = cos_t; g.rope_sin = sin_t; g.rope_mod = pos_mod; g.rope_pos0 = pos0; g.rope_cols = rope_cols;
  const bool bm144 = pick_144(M, cdiv(N, 256), K, K2, false);
  g.tilesM = cdiv(M, bm144 ? 144 : 256); g.tilesN = cdiv(N, 256);
  const int pslot = prof_count(M, N, K + K2, 3, (hipStream_t)stream);
  if (bm144) LAUNCH_144(0, 3, grid_256s((long)g.tilesM * g.tilesN), (hipStream_t)stream, g);
  else LAUNCH_256(0, 3, grid_256s((long)g.tilesM * g.tilesN), (hipStream_t)stream, g);
  prof_end(pslot, (hipStream_t)stream);
  LHRS_CHECK_LAUNCH("gemm_rope_fwd");
  return 0;
}

extern "C" int lhrs_gemm_swiglu_bwd(const void* dY, int ldy, const void* WdT, int ldw, const void* A2, int lda2, const void* B2, int ldb2,
                                    int K2, const void* gu, void* dgu, int ld_gu, void* dact_scratch, int M, int ff, int K, void* stream) {
  LHRS_REQUIRE(M > 0 && ff > 0 && K > 0 && ff % 8 == 0 && ld_gu >= 2 * ff && ld_gu % 8 == 0, "gemm_swiglu_bwd: M=%d ff=%d K=%d ld_gu=%d", M, ff, K, ld_gu);
  if (!swiglu_fusable((long)cdiv(M, 256) * cdiv(ff, 256), ff, K, K2, ldy, ldw)) {
    LHRS_REQUIRE(dact_scratch != nullptr && ld_gu == 2 * ff, "gemm_swiglu_bwd: the unfused fallback needs a [M, ff] scratch and dense gu");
    if (gemm_launch(dY, ldy, WdT, ldw, dact_scratch, ff, M, ff, K, nullptr, nullptr, 0, 0, 0, 0, 1.f, A2, lda2, B2, ldb2, K2, stream)) return -1;
    return lhrs_swiglu_bwd(dact_scratch, gu, dgu, M, ff, stream);
  }
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.A = (const bf16_t*)dY; g.B = (const bf16_t*)WdT; g.C = dgu; g.M = M; g.N = ff; g.K = K; g.lda = ldy; g.ldb = ldw; g.ldc = ld_gu;
  g.alpha = 1.f; g.A2 = (const bf16_t*)A2; g.B2 = (const bf16_t*)B2; g.lda2 = lda2; g.ldb2 = ldb2; g.K2 = K2;
  g.epi = 2; g.ff = ff; g.aux = (const bf16_t*)gu; g.ld_aux = ld_gu;
  const bool bm144 = pick_144(M, cdiv(ff, 256), K, K2, false);
  g.tilesM = cdiv(M, bm144 ? 144 : 256); g.tilesN = cdiv(ff, 256);
  hipStream_t s = (hipStream_t)stream;
  const int pslot = prof_count(M, ff, K + K2, 2, s);
  if (bm144) LAUNCH_144(0, 2, grid_256s((long)g.tilesM * g.tilesN), s, g);
  else LAUNCH_256(0, 2, grid_256s((long)g.tilesM * g.tilesN), s, g);
  prof_end(pslot, s);
  LHRS_CHECK_LAUNCH("gemm_swiglu_bwd");
  return 0;
}

// ---- RMSNorm backward without its own pass over HBM (round 4) ------------------------------------------------------------------------
// HF LlamaRMSNorm in front of the MLP: h = w o (x * rstd), gate|up = h W_gu^T.  Its backward needs c = sum_j dh_j w_j x_j over the WHOLE row
// of dh = d(gate|up) W_gu - a full-row reduction that a tiled GEMM epilogue does not have.  But c = (1 / rstd) <d(gate|up), gate|up> (write
// h_j = w_j x_j rstd and pull W_gu through the sum), and both factors sit in the SwiGLU-backward epilogue one launch earlier:
//   lhrs_gemm_swiglu_bwd_rowdot : the fused d-down + SwiGLU' launch, which also writes the 4 * ff/256 per-wave-column partial sums per row
//   lhrs_rowsum_partials        : s[m] = sum of the partials (fixed order: deterministic)
//   lhrs_gemm_rmsnorm_bwd       : dx = rstd * (w o (dgu W_gu)) - x * (rstd^2 s / d) + add in the epilogue of the dX GEMM
// instead of GEMM -> [dh to HBM] -> rmsnorm_bwd (4 row passes, 43 us at M = 8190).  Only where both products are whole rounds of the 256-row
// persistent kernel (lhrs_gemm_rmsnorm_bwd_fusable); everywhere else the callers keep the three-launch sequence.
// MEASURED (round 4, micro-batch 30, rocprofv3): the SwiGLU' launch grows 642 -> 678 us with the row dot, the d-gate|up launch 1118 -> 1133 us with
// the norm epilogue, + the partial sum: +51 us of EXPOSED epilogue per layer for the 43 us rmsnorm_bwd pass it removes - the step is 0.4 ms
// SLOWER (2.3 ms with the first, latency-bound version of the partial-sum kernel).  In this persistent kernel every wave runs the epilogue at
// the same time, nothing overlaps it, and it moves bytes less efficiently than a dedicated bandwidth kernel at 6.2 TB/s.  The callers therefore
// use it only on request (LHRS_FUSE_NORM_BWD=1); correct and tested (test_rmsnorm_backward_inside_the_dx_gemm_*).
namespace {
__global__ __launch_bounds__(512) void rowsum_partials_kernel(const float* __restrict__ part, float* __restrict__ out, int P, int M) {
  // block = 64 rows x 8 slices of the partial index: slice y adds partials y, y + 8, ... (coalesced over the rows), then slice 0 adds the eight slice
  // sums in slice order - a fixed summation order (deterministic), 8 loads in flight per row instead of a chain of P dependent ones
  __shared__ float red[8][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int m = blockIdx.x * 64 + tx;
  float acc = 0.f;
  if (m < M)
    for (int p = ty; p < P; p += 8) acc += part[(long)p * M + m];
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && m < M) {
    float t = red[0][tx];
#pragma unroll
    for (int y = 1; y < 8; ++y) t += red[y][tx];
    out[m] = t;
  }
}
}  // namespace
extern "C" int lhrs_gemm_rmsnorm_bwd_fusable(int M, int d, int ff) {
  const long P = num_cus();
  if (g_gemm_allow_256 != 2 || !g_gemm_persist || d % 256 != 0 || ff % 256 != 0) return 0;
  const long t_down = (long)cdiv(M, 256) * (ff / 256), t_gu = (long)cdiv(M, 256) * (d / 256);
  if (!swiglu_fusable(t_down, ff, d, 0, d, d) || pick_144(M, ff / 256, d, 0, false)) return 0;                 // d-down + SwiGLU': one fused 256-row launch
  if (t_gu < g_gemm_min256 || t_gu % P != 0 || pick_144(M, d / 256, 2 * ff, 0, false)) return 0;               // d-gate|up: whole rounds of the 256-row kernel
  return 1;
}
extern "C" int lhrs_gemm_swiglu_bwd_rowdot(const void* dY, int ldy, const void* WdT, int ldw, const void* gu, void* dgu, int ld_gu, float* row_dot,
                                           int M, int ff, int K, void* stream) {
  LHRS_REQUIRE(M > 0 && ff > 0 && K > 0 && ff % 256 == 0 && ld_gu >= 2 * ff && ld_gu % 8 == 0 && row_dot != nullptr, "gemm_swiglu_bwd_rowdot: M=%d ff=%d K=%d", M, ff, K);
  LHRS_REQUIRE(swiglu_fusable((long)cdiv(M, 256) * (ff / 256), ff, K, 0, ldy, ldw) && !pick_144(M, ff / 256, K, 0, false),
               "gemm_swiglu_bwd_rowdot: not a single fused 256-row launch for M=%d ff=%d K=%d (ask lhrs_gemm_rmsnorm_bwd_fusable first)", M, ff, K);
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.A = (const bf16_t*)dY; g.B = (const bf16_t*)WdT; g.C = dgu; g.M = M; g.N = ff; g.K = K; g.lda = ldy; g.ldb = ldw; g.ldc = ld_gu;
  g.alpha = 1.f; g.epi = 5; g.ff = ff; g.aux = (const bf16_t*)gu; g.ld_aux = ld_gu; g.row_dot = row_dot; g.drop_scale = 1.f;
  g.tilesM = cdiv(M, 256); g.tilesN = ff / 256;
  hipStream_t s = (hipStream_t)stream;
  const int pslot = prof_count(M, ff, K, 2, s);
  LAUNCH_256(0, 5, 0, s, g);
  prof_end(pslot, s);
  LHRS_CHECK_LAUNCH("gemm_swiglu_bwd_rowdot");
  return 0;
}
// s[m] = sum_{p < P} part[p * M + m]
extern "C" int lhrs_rowsum_partials(const float* part, float* out, int P, int M, void* stream) {
  LHRS_REQUIRE(part && out && P > 0 && M > 0, "rowsum_partials: P=%d M=%d", P, M);
  hipLaunchKernelGGL(rowsum_partials_kernel, dim3(cdiv(M, 64)), dim3(512), 0, (hipStream_t)stream, part, out, P, M);
  LHRS_CHECK_LAUNCH("rowsum_partials");
  return 0;
}
// out[M, N] = rstd o (w o (dY . WT^T)) - x o (rstd^2 s / N) + add;  dY [M, K], WT [N, K] (the transposed weight copy), x / add / out [M, N]
extern "C" int lhrs_gemm_rmsnorm_bwd(const void* dY, int ldy, const void* WT, int ldw, const void* x, int ldx, const void* w, const float* rstd,
                                     const float* s_row, const void* add, int ld_add, void* out, int ldo, int M, int N, int K, void* stream) {
  LHRS_REQUIRE(M > 0 && N > 0 && K >= 128 && K % 64 == 0 && N % 256 == 0 && ldy % 8 == 0 && ldw % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0 &&
               (add == nullptr || ld_add % 8 == 0) && x && w && rstd && s_row, "gemm_rmsnorm_bwd: M=%d N=%d K=%d", M, N, K);
  const long T = (long)cdiv(M, 256) * (N / 256);
  LHRS_REQUIRE(g_gemm_allow_256 == 2 && g_gemm_persist && T >= g_gemm_min256 && T % num_cus() == 0 && !pick_144(M, N / 256, K, 0, false),
               "gemm_rmsnorm_bwd: M=%d N=%d is not whole rounds of the 256-row kernel (ask lhrs_gemm_rmsnorm_bwd_fusable first)", M, N);
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.A = (const bf16_t*)dY; g.B = (const bf16_t*)WT; g.C = out; g.M = M; g.N = N; g.K = K; g.lda = ldy; g.ldb = ldw; g.ldc = ldo;
  g.alpha = 1.f; g.epi = 4; g.aux = (const bf16_t*)x; g.ld_aux = ldx; g.res = (const bf16_t*)add; g.ldr = ld_add; g.bias = (const bf16_t*)w;
  g.sa = rstd; g.sb = s_row; g.drop_scale = 1.f;
  g.tilesM = cdiv(M, 256); g.tilesN = N / 256;
  hipStream_t s = (hipStream_t)stream;
  int slot = -1;
  if (g_prof.on) {
    g_prof.launches_all++; g_prof.total_flops_all += 2.0 * M * N * K;
    if (g_prof.used < g_prof.cap && g_prof.take(0)) {   // counted with the plain launches of the 256-row kernel: it is that kernel with a heavier epilogue
      slot = g_prof.used++;
      g_prof.flops[slot] = 2.0 * M * N * K; g_prof.kind[slot] = 0;
      (void)hipEventRecord(g_prof.ev[2 * slot], s);
    }
  }
  LAUNCH_256(0, 4, 0, s, g);
  if (slot >= 0) (void)hipEventRecord(g_prof.ev[2 * slot + 1], s);
  LHRS_CHECK_LAUNCH("gemm_rmsnorm_bwd");
  return 0;
}


// ------------------------------------------------------------------------------------------------
// Small-tile sibling of gemm_fp8_256_kernel (64x128 tile, 4 waves, the 2-stage DMA skeleton of gemm_nt_kernel; a stage row is 128 e4m3
// bytes = one v_mfma_scale_f32_16x16x128_f8f6f4 step per fragment pair).  Takes the tail rows that the 256x256 kernel would run as a
// nearly empty last round (see the tail-row rule in gemm_launch).  Same arithmetic and roundings as the big kernel: scale the e4m3 sum by
// sa[m] * sb[n], add the optional bf16 pair (LoRA) on the same accumulators, * alpha, round to bf16, then add the bf16 residual.
// ------------------------------------------------------------------------------------------------
template <int WM_FR, int WN_FR>
__global__ __launch_bounds__(256) void gemm_fp8_small_kernel(GemmArgs g) {
  constexpr int BM = WM_FR * 32, BN = WN_FR * 32, ROWB = 128;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  int tm, tn;
  tile_coords(g, tm, tn);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int A_INSTR = BM / 32, B_INSTR = BN / 32;  // 1-KiB pieces (8 rows x 128 B) per wave
  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ lrow;
  const char *a_src[A_INSTR], *b_src[B_INSTR], *a2_src[A_INSTR], *b2_src[B_INSTR];
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    const int row = min(tm * BM + (wave * A_INSTR + i) * 8 + lrow, g.M - 1);
    a_src[i] = reinterpret_cast<const char*>(g.A) + (long)row * g.lda + lchunk * 16;
    a2_src[i] = reinterpret_cast<const char*>(g.A2) + ((long)row * g.lda2 + lchunk * 8) * 2;
  }
#pragma unroll
  for (int i = 0; i < B_INSTR; ++i) {
    const int row = min(tn * BN + (wave * B_INSTR + i) * 8 + lrow, g.N - 1);
    b_src[i] = reinterpret_cast<const char*>(g.B) + (long)row * g.ldb + lchunk * 16;
    b2_src[i] = reinterpret_cast<const char*>(g.B2) + ((long)row * g.ldb2 + lchunk * 8) * 2;
  }
  const int nk1 = g.K / 128, nk = nk1 + g.K2 / 64;
  auto issue = [&](int kt) {
    char* sa = smem + (kt & 1) * STAGE;
    char* sb = sa + A_BYTES;
    const bool e4 = kt < nk1;  // wave-uniform
    const long ko = (long)(e4 ? kt : kt - nk1) * ROWB;
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)((e4 ? a_src[i] : a2_src[i]) + ko), (lptr_t)(sa + (wave * A_INSTR + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)((e4 ? b_src[i] : b2_src[i]) + ko), (lptr_t)(sb + (wave * B_INSTR + i) * 1024), 16, 0, 0);
  };
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const int a_row0 = wm * (WM_FR * 16) + fr, b_row0 = wn * (WN_FR * 16) + fr;
  f32x4 acc[WM_FR][WN_FR];
#pragma unroll
  for (int i = 0; i < WM_FR; ++i)
#pragma unroll
    for (int j = 0; j < WN_FR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  issue(0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) issue(kt + 1);
    const char* sa = smem + (kt & 1) * STAGE;
    const char* sb = sa + A_BYTES;
    if (kt < nk1) {  // 128 e4m3 k per row: lane (r, g) feeds bytes [32g, 32g + 32) = chunks 2g, 2g + 1
      const int c0 = ((2 * fg) ^ (fr & 7)) * 16, c1 = ((2 * fg + 1) ^ (fr & 7)) * 16;
      i32x8 af[WM_FR], bfr[WN_FR];
#pragma unroll
      for (int mi = 0; mi < WM_FR; ++mi) {
        const i32x4 lo = *reinterpret_cast<const i32x4*>(sa + (a_row0 + mi * 16) * ROWB + c0), hi = *reinterpret_cast<const i32x4*>(sa + (a_row0 + mi * 16) * ROWB + c1);
        af[mi] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int ni = 0; ni < WN_FR; ++ni) {
        const i32x4 lo = *reinterpret_cast<const i32x4*>(sb + (b_row0 + ni * 16) * ROWB + c0), hi = *reinterpret_cast<const i32x4*>(sb + (b_row0 + ni * 16) * ROWB + c1);
        bfr[ni] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int mi = 0; mi < WM_FR; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN_FR; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      if (kt + 1 == nk1 && g.K2 > 0) {  // the bf16 pair rides on the SCALED sums
#pragma unroll
        for (int mi = 0; mi < WM_FR; ++mi) {
          const float sam = g.sa[min(tm * BM + wm * (WM_FR * 16) + mi * 16 + fr, g.M - 1)];
#pragma unroll
          for (int ni = 0; ni < WN_FR; ++ni) {
            const float4 s4 = *reinterpret_cast<const float4*>(g.sb + min(tn * BN + wn * (WN_FR * 16) + ni * 16 + fg * 4, g.N - 4));
            acc[mi][ni][0] *= sam * s4.x; acc[mi][ni][1] *= sam * s4.y; acc[mi][ni][2] *= sam * s4.z; acc[mi][ni][3] *= sam * s4.w;
          }
        }
      }
    } else {  // 64 bf16 k per row, exactly gemm_nt_kernel's stage
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int ko = ((kk * 4 + fg) ^ (fr & 7)) * 16;
        bf16x8 af[WM_FR], bfr[WN_FR];
#pragma unroll
        for (int mi = 0; mi < WM_FR; ++mi) af[mi] = *reinterpret_cast<const bf16x8*>(sa + (a_row0 + mi * 16) * ROWB + ko);
#pragma unroll
        for (int ni = 0; ni < WN_FR; ++ni) bfr[ni] = *reinterpret_cast<const bf16x8*>(sb + (b_row0 + ni * 16) * ROWB + ko);
#pragma unroll
        for (int mi = 0; mi < WM_FR; ++mi)
#pragma unroll
          for (int ni = 0; ni < WN_FR; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);
      }
    }
  }
  const bool scaled = g.K2 > 0;
#pragma unroll
  for (int mi = 0; mi < WM_FR; ++mi) {
    const int m = tm * BM + wm * (WM_FR * 16) + mi * 16 + fr;
    if (m >= g.M) continue;
    const float sam = (scaled ? 1.f : g.sa[m]) * g.alpha;
#pragma unroll
    for (int ni = 0; ni < WN_FR; ++ni) {
      const int n = tn * BN + wn * (WN_FR * 16) + ni * 16 + fg * 4;
      if (n >= g.N) continue;
      float4 s4 = scaled ? make_float4(1.f, 1.f, 1.f, 1.f) : *reinterpret_cast<const float4*>(g.sb + n);
      uint2 o = make_uint2(pack2bf(acc[mi][ni][0] * sam * s4.x, acc[mi][ni][1] * sam * s4.y), pack2bf(acc[mi][ni][2] * sam * s4.z, acc[mi][ni][3] * sam * s4.w));
      if (g.res) {
        const uint2 r = *reinterpret_cast<const uint2*>(g.res + (long)m * g.ldr + n);
        o.x = pack2bf(bflo(o.x) + bflo(r.x), bfhi(o.x) + bfhi(r.x));
        o.y = pack2bf(bflo(o.y) + bflo(r.y), bfhi(o.y) + bfhi(r.y));
      }
      *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(g.C) + (long)m * g.ldc + n) = o;
    }
  }
}

// C[M, N] (bf16) = alpha * sa[m] * sb[n] * (A8[M, K] . B8[N, K]^T) (+ residual): e4m3 operands with per-row fp32 scales
// (lhrs_quant_fp8_rows).  K % 128 == 0, N % 8 == 0; lda / ldb in BYTES (>= K, multiples of 16).
static int gemm_fp8_launch(const void* A8, long lda, const float* sa, const void* B8, long ldb, const float* sb, const void* A2, int lda2,
                           const void* B2, int ldb2, int K2, void* C, int ldc, int M, int N, int K, const void* residual, int ldr,
                           float alpha, void* stream, bool i8 = false, const int* k2_dev = nullptr) {
  LHRS_REQUIRE(M > 0 && N > 0 && K >= 256 && K % 128 == 0, "gemm_fp8: M=%d N=%d K=%d (K %% 128 == 0, K >= 256)", M, N, K);
  LHRS_REQUIRE(lda % 16 == 0 && ldb % 16 == 0 && lda >= K && ldb >= K && sa && sb, "gemm_fp8: lda=%ld ldb=%ld", lda, ldb);
  LHRS_REQUIRE(N % 8 == 0 && ldc % 8 == 0 && ldc >= N && (residual == nullptr || ldr % 8 == 0), "gemm_fp8: N=%d ldc=%d ldr=%d", N, ldc, ldr);
  LHRS_REQUIRE((K2 == 0 && !k2_dev) || (A2 && B2 && K2 % 64 == 0 && lda2 % 8 == 0 && ldb2 % 8 == 0 && lda2 >= K2 && ldb2 >= K2),
               "gemm_fp8: bad bf16 pair (K2=%d lda2=%d ldb2=%d)", K2, lda2, ldb2);
  // tail-row rule (see gemm_launch): the tile rows that spill over the last full round of the 256 CUs go to the small-tile kernel
  if (!i8 && t_split_ok && g_gemm_tail_split) {   // (the small-tile sibling exists for e4m3 only)
    const int tm = cdiv(M, 256), tn = cdiv(N, 256);
    const long T = (long)tm * tn, rounds = (T + 255) / 256, full = T / 256;
    const int tm_main = (int)(full * 256 / tn);
    if (full >= 1 && T % 256 != 0 && tm_main >= 1 && tm_main < tm) {
      const long t_main = (long)tm_main * tn, t_tail = T - t_main;
      if ((double)((t_main + 255) / 256) + 2.5 * (double)t_tail / 256.0 + 0.05 < (double)rounds) {
        const int M_main = tm_main * 256;
        t_split_ok = false;
        int rc = gemm_fp8_launch(A8, lda, sa, B8, ldb, sb, A2, lda2, B2, ldb2, K2, C, ldc, M_main, N, K, residual, ldr, alpha, stream);
        t_split_ok = true;
        if (rc) return rc;
        GemmArgs h; memset(&h, 0, sizeof(h));
        h.A = (const bf16_t*)((const char*)A8 + (long)M_main * lda); h.B = (const bf16_t*)B8; h.C = (bf16_t*)C + (long)M_main * ldc;
        h.res = residual ? (const bf16_t*)residual + (long)M_main * ldr : nullptr;
        h.M = M - M_main; h.N = N; h.K = K; h.lda = (int)lda; h.ldb = (int)ldb; h.ldc = ldc; h.ldr = ldr; h.alpha = alpha; h.sa = sa + M_main; h.sb = sb;
        h.A2 = A2 ? (const bf16_t*)A2 + (long)M_main * lda2 : nullptr; h.B2 = (const bf16_t*)B2; h.lda2 = lda2; h.ldb2 = ldb2; h.K2 = K2;
        h.tilesM = cdiv(h.M, 64); h.tilesN = cdiv(N, 128);
        if (g_prof.on) { g_prof.launches_all++; g_prof.total_flops_all += 2.0 * h.M * N * (K + K2); }
        hipLaunchKernelGGL((gemm_fp8_small_kernel<2, 4>), dim3(h.tilesM * h.tilesN), dim3(256), 0, (hipStream_t)stream, h);
        LHRS_CHECK_LAUNCH("gemm_fp8_nt (tail rows)");
        return 0;
      }
    }
  }
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.A = (const bf16_t*)A8; g.B = (const bf16_t*)B8; g.C = C; g.res = (const bf16_t*)residual;
  g.M = M; g.N = N; g.K = K; g.lda = (int)lda; g.ldb = (int)ldb; g.ldc = ldc; g.ldr = ldr; g.alpha = alpha; g.sa = sa; g.sb = sb;
  g.A2 = (const bf16_t*)A2; g.B2 = (const bf16_t*)B2; g.lda2 = lda2; g.ldb2 = ldb2; g.K2 = K2; g.k2_dev = k2_dev;
  g.tilesM = cdiv(M, 256); g.tilesN = cdiv(N, 256);
  if (g_prof.on) { g_prof.launches_all++; g_prof.total_flops_all += 2.0 * M * N * (K + K2); }
  if (i8) hipLaunchKernelGGL(gemm_fp8_256_kernel<true>, dim3(g.tilesM * g.tilesN), dim3(512), 0, (hipStream_t)stream, g);
  else hipLaunchKernelGGL(gemm_fp8_256_kernel<false>, dim3(g.tilesM * g.tilesN), dim3(512), 0, (hipStream_t)stream, g);
  LHRS_CHECK_LAUNCH("gemm_fp8_nt");
  return 0;
}

// LLM.int8 product (int8.hip): C[M, N] (bf16) = alpha * (sa[m] * sb[n] * (A8 . B8^T in int32) + A2[M, :K2t] . B2[N, :K2t]^T) (+ residual).
// A8 / B8: int8 rows (lhrs_int8_prepare / lhrs_quant_int8_rows), sa / sb their dequantisation factors absmax / 127.  The bf16 pair holds
// K2 host-known columns (a LoRA update; 0 = none) followed by k2_dev[0] device-known ones (the 16-bit outlier-column product, a multiple of
// 64 written by lhrs_int8_prepare into meta[1]); K2t is their sum, lda2 / ldb2 must cover the worst case.  K % 128 == 0, K2 % 64 == 0.
extern "C" int lhrs_gemm_int8_nt(const void* A8, long lda, const float* sa, const void* B8, long ldb, const float* sb, const void* A2,
                                 int lda2, const void* B2, int ldb2, int K2, const int* k2_dev, void* C, int ldc, int M, int N, int K,
                                 const void* residual, int ldr, float alpha, void* stream) {
  return gemm_fp8_launch(A8, lda, sa, B8, ldb, sb, A2, lda2, B2, ldb2, K2, C, ldc, M, N, K, residual, ldr, alpha, stream, true, k2_dev);
}


extern "C" int lhrs_gemm_fp8_nt(const void* A8, long lda, const float* sa, const void* B8, long ldb, const float* sb, void* C, int ldc,
                                int M, int N, int K, const void* residual, int ldr, float alpha, void* stream) {
  return gemm_fp8_launch(A8, lda, sa, B8, ldb, sb, nullptr, 0, nullptr, 0, 0, C, ldc, M, N, K, residual, ldr, alpha, stream);
}

// ... + alpha * A2[M, K2] . B2[N, K2]^T in bf16 on the same accumulators (the LoRA update of an 8-bit base linear), K2 % 64 == 0
extern "C" int lhrs_gemm_fp8_nt_lora(const void* A8, long lda, const float* sa, const void* B8, long ldb, const float* sb, const void* A2,
                                     int lda2, const void* B2, int ldb2, int K2, void* C, int ldc, int M, int N, int K,
                                     const void* residual, int ldr, float alpha, void* stream) {
  LHRS_REQUIRE(K2 > 0, "gemm_fp8_lora: K2=%d", K2);
  return gemm_fp8_launch(A8, lda, sa, B8, ldb, sb, A2, lda2, B2, ldb2, K2, C, ldc, M, N, K, residual, ldr, alpha, stream);
}


// Skinny-N product C[M, N] (bf16) = alpha * A[M, K] . B[N, K]^T for N <= 384 (the LoRA down-projections x.A^T / dy.B, peft lora.Linear):
// HBM-bound on A, but one 64-row block walking all of K with a two-stage pipeline is latency-bound (74 us at M = 8190, K = 4096);
// here K is split `lhrs_gemm_skinny_splits(K, N)` ways across blockIdx.y into f32 slabs of the workspace, then summed (~15 us).
// workspace: lhrs_gemm_skinny_splits(K, N) * M * N floats.
extern "C" int lhrs_gemm_skinny_splits(int K, int N) {
  int s = K / 512;
  const int cap = N <= 128 ? 16 : 4;  // wide adapters: the f32 slabs (splits * M * N * 4 B) soon cost more than the latency they hide
  return s < 1 ? 1 : (s > cap ? cap : s);
}

extern "C" int lhrs_gemm_bf16_nt_skinny(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, float alpha,
                                        float* workspace, void* stream) {
  LHRS_REQUIRE(M > 0 && N > 0 && N <= 384 && N % 64 == 0 && K % 64 == 0 && workspace != nullptr, "gemm_skinny: M=%d N=%d K=%d", M, N, K);
  LHRS_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K && ldc >= N && ldc % 4 == 0, "gemm_skinny: lda=%d ldb=%d ldc=%d", lda, ldb, ldc);
  const int splits = lhrs_gemm_skinny_splits(K, N);
  int ks = cdiv(K / 64, splits) * 64;
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = workspace; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = N;
  g.alpha = 1.f; g.out_f32 = 1; g.ksplit = ks;
  const bool wide = N % 128 == 0;  // 64x128 tiles: the rows of A are read once per 128 columns, not per 64 (M = 8190: 5-11 % faster)
  g.tilesM = cdiv(M, 64); g.tilesN = cdiv(N, wide ? 128 : 64);
  const int used = cdiv(K, ks);
  hipStream_t s = (hipStream_t)stream;
  if (g_prof.on) { g_prof.launches_all++; g_prof.total_flops_all += 2.0 * M * N * K; }
  if (wide) hipLaunchKernelGGL((gemm_nt_kernel<2, 4, 0>), dim3(g.tilesM * g.tilesN, used), dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm_nt_kernel<2, 2, 0>), dim3(g.tilesM * g.tilesN, used), dim3(256), 0, s, g);
  LHRS_CHECK_LAUNCH("gemm_skinny");
  const long work = (long)M * (N / 4);
  int rg = (int)((work + 255) / 256); if (rg > 8192) rg = 8192;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rg), dim3(256), 0, s, workspace, (bf16_t*)C, (long)ldc, M, N, used, alpha);
  LHRS_CHECK_LAUNCH("gemm_skinny_reduce");
  return 0;
}


// Long-K, few-tile product with f32 output (the projector's weight gradients dW = dY^T X over the token dimension: K = B * 912 = 27392
// against 2048 x 1024 outputs = 128 tiles of 128^2 for 256 CUs): K is split `lhrs_gemm_splitk_splits` ways across blockIdx.y into f32 slabs
// of the workspace (splits * M * N floats), then summed in a fixed order (deterministic).  splits == 1 -> one plain launch.
extern "C" int lhrs_gemm_splitk_splits(int M, int N, int K) {
  const long tiles = (long)cdiv(M, 128) * cdiv(N, 128);
  long s = (512 + tiles / 2) / tiles;
  if (s > K / 1024) s = K / 1024;
  if (s > 8) s = 8;
  return s < 1 ? 1 : (int)s;
}

extern "C" int lhrs_gemm_bf16_nt_splitk_f32(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K,
                                            float* workspace, void* stream) {
  LHRS_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0 && N % 4 == 0 && ldc % 4 == 0 && ldc >= N && lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K,
               "gemm_splitk_f32: M=%d N=%d K=%d lda=%d ldb=%d ldc=%d", M, N, K, lda, ldb, ldc);
  const int splits = lhrs_gemm_splitk_splits(M, N, K);
  if (splits == 1) return gemm_launch(A, lda, B, ldb, C, ldc, M, N, K, nullptr, nullptr, 0, 0, 1, 0, 1.f, nullptr, 0, nullptr, 0, 0, stream);
  LHRS_REQUIRE(workspace != nullptr, "gemm_splitk_f32: %d splits need a workspace of splits * M * N floats", splits);
  const int ks = cdiv(K / 64, splits) * 64;
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = workspace; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = N;
  g.alpha = 1.f; g.out_f32 = 1; g.ksplit = ks;
  g.tilesM = cdiv(M, 128); g.tilesN = cdiv(N, 128);
  const int used = cdiv(K, ks);
  hipStream_t s = (hipStream_t)stream;
  if (g_prof.on) { g_prof.launches_all++; g_prof.total_flops_all += 2.0 * M * N * K; }
  hipLaunchKernelGGL((gemm_nt_kernel<4, 4, 0>), dim3(g.tilesM * g.tilesN, used), dim3(256), 0, s, g);
  LHRS_CHECK_LAUNCH("gemm_splitk_f32");
  const long work = (long)M * (N / 4);
  int rg = (int)((work + 255) / 256); if (rg > 8192) rg = 8192;
  hipLaunchKernelGGL(splitk_reduce_f32_kernel, dim3(rg), dim3(256), 0, s, workspace, C, (long)ldc, M, N, used);
  LHRS_CHECK_LAUNCH("gemm_splitk_f32 reduce");
  return 0;
}
