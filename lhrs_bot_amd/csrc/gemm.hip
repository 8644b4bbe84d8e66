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
//   * gemm_u4_kernel (gemm_u4.hip) - the four-wave 256x256x64 kernel for the plain long-k products.
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
  int epi, ff;
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

template <int ACT, int EPI, bool K2P>
__global__ __launch_bounds__(1024, 1) void gemm_nt_256s_kernel(GemmArgs g) {
  constexpr int BM = 256, BN = 256, BK = 64;
  constexpr int A_BYTES = BM * BK * 2, STAGE = A_BYTES + BN * BK * 2;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int ntiles = g.tilesM * g.tilesN;
  int t = blockIdx.x;
  if (t >= ntiles) return;
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
    // the pointer goes through an SALU move: "VALU writes SGPR -> VMEM reads it" needs 5 wait states the compiler cannot pad inside asm (see gemm_u4.hip: U4_SPTR)
    asm volatile("s_mov_b64 s[100:101], %1\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, s[100:101]" ::"v"(vo), "s"(sp), "s"(lds_dst) : "memory", "m0", "s100", "s101");
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

  const int nkl = nk;   // stages per tile
  int tm, tn;
  tile_coords_lin(g, t, ntiles, tm, tn);
  unsigned off1[4];
  dma_rows(tm, tn, off1);
  int pb = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) issue1(off1, 0, 0, j);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  bf16x8 A[4], B[4];
#define RDQ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define SB __builtin_amdgcn_sched_barrier(0);
#define MF(mi, ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B[ni], A[mi], acc[mi][ni], 0, 0, 0); SB

  for (;;) {
    const int tnext = t + (int)gridDim.x;
    const bool has_next = tnext < ntiles;  // workgroup-uniform
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // stage 0 of this tile is in LDS buffer pb; the barrier publishes it and ends the previous tile's epilogue reads of the other buffer
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!K2P || 1 < nk1) issue1(off1, 1, pb ^ 1, j);
      else issue2(tm, tn, 1 - nk1, pb ^ 1, j);
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
    // stage s fetches stage s + 2 of the tile: from the first pair while that is < nk1
    const int nA = K2P ? min(nkl - 2, nk1 - 2) : nkl - 2;
#define ISS(j) issue1(off1, s + 2, (s + pb) & 1, j);
    for (int s = 0; s < nA; ++s) STAGE_S(s)   // !K2P: every stage but the last two
#undef ISS
    if (K2P) {  // the stages whose DMA slot fetches the second pair
#define ISS(j) issue2(tm, tn, s + 2 - nk1, (s + pb) & 1, j);
      for (int s = max(nA, 0); s < nkl - 2; ++s) STAGE_S(s)
#undef ISS
    }
    int ntm = 0, ntn = 0;
    if (has_next) {  // this tile's DMA rows are not needed any more (its last stage is in flight): the offsets become the next tile's
      tile_coords_lin(g, tnext, ntiles, ntm, ntn);
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

    // accumulator element acc[mi][ni][i]: m = wm*64 + mi*16 + fr, n = wn*64 + ni*16 + fg*4 + i.
    // Everything the epilogue derives from the lane id is derived from an opaque copy made HERE, per tile: otherwise those values are
    // loop-invariant across tiles, get hoisted in front of the tile loop and sit in (or spill from) registers all through the main loop
    int le = lane;
    asm volatile("" : "+v"(le));
    const int fr = le & 15, fg = le >> 4;
    if (g.out_f32) {
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
      constexpr bool SWB = EPI == 2;   // SwiGLU-backward epilogue
      const int nlim = SWB ? g.ff : g.N;
      const bool col_ok = n < nlim;
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        uint4 pre_a[4], pre_b[4];
        if (SWB || (EPI == 0 && g.res)) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int m = tm * BM + wm * 64 + ps * 32 + i * 8 + rsub;
            pre_a[i] = make_uint4(0, 0, 0, 0); pre_b[i] = make_uint4(0, 0, 0, 0);
            if (m < g.M && col_ok) {
              if (SWB) {
                pre_a[i] = *reinterpret_cast<const uint4*>(g.aux + (long)m * g.ld_aux + n);
                pre_b[i] = *reinterpret_cast<const uint4*>(g.aux + (long)m * g.ld_aux + g.ff + n);
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
          if (SWB && m < g.M && col_ok) {
            const uint4 gq = pre_a[i], uq = pre_b[i];
            float d[8], gg[8], uu[8], dg[8], du[8];
            unpack8(val, d); unpack8(gq, gg); unpack8(uq, uu);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float sg = sigmoid_f(gg[e]);
              du[e] = d[e] * gg[e] * sg;
              dg[e] = d[e] * uu[e] * sg * (1.f + gg[e] * (1.f - sg));
            }
            bf16_t* out = reinterpret_cast<bf16_t*>(g.C) + (long)m * g.ldc + n;
            *reinterpret_cast<uint4*>(out) = pack8(dg);
            *reinterpret_cast<uint4*>(out + g.ff) = pack8(du);
          }
          if (SWB) continue;
          if (m < g.M && col_ok) {
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
  long seen[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // kinds: see lhrs_gemm_profile_read_kinds
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
  for (int k = 0; k < 10; ++k) g_prof.seen[k] = 0;
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

// out[10][3] = {sampled launches, their summed duration (ms), their summed flops} per kind - one kind per kernel instantiation a rocprofv3 kernel trace lists:
//   0 gemm_nt_256s_kernel<ACT, 0, ..> plain (16 waves)   1 <0, 1> SwiGLU fwd   2 <0, 2> SwiGLU bwd   3 <0, 3> RoPE   4 gemm_nt_144s_kernel<ACT, 0> plain 144-row tiles
//   5 gemm_u4_kernel<0, true> plain + residual (four waves)   6 gemm_u4_kernel<0, false> plain   7 gemm_u4_kernel<1, false> SwiGLU fwd   8 <2, false> SwiGLU bwd   9 <3, false> RoPE
extern "C" int lhrs_gemm_profile_read_kinds(double* out) {
  for (int i = 0; i < 30; ++i) out[i] = 0;
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

// ---- the GEMM workspace -------------------------------------------------------------------------------------------------------------------
// Caller-owned device memory, registered per device (lhrs_gemm_set_workspace; 64 MiB + 4 KiB for 256 CUs).  One user: the tail rows of a
// row-split product with a long k-loop are computed split-K through f32 slabs in it (gemm_launch below).  Launches that use it must be ordered on
// ONE stream per device.  (Round 4 also ran a stream-K tail of the persistent kernel through it: built, correct, slower at every shape of the path -
// profiles/r04_streamk_ab.txt, DESIGN.md 3.1 - and removed in round 5.)
static struct { float* slabs; long units; } g_sk[16];
extern "C" long lhrs_gemm_workspace_bytes() { return 4096 + (long)num_cus() * 256 * 256 * 4; }
// ws == nullptr: forget the workspace of the current device
extern "C" int lhrs_gemm_set_workspace(void* ws, long bytes) {
  int dev = 0;
  LHRS_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16, "gemm_set_workspace: device %d", dev);
  if (ws == nullptr) { g_sk[dev].slabs = nullptr; g_sk[dev].units = 0; return 0; }
  LHRS_REQUIRE(bytes >= lhrs_gemm_workspace_bytes() && ((size_t)ws & 255) == 0, "gemm_set_workspace: %ld bytes (need %ld, 256-B aligned)", bytes,
               lhrs_gemm_workspace_bytes());
  g_sk[dev].slabs = (float*)((char*)ws + 4096); g_sk[dev].units = num_cus();
  return 0;
}
// rounds of the P CUs that T tiles of the persistent kernels cost
static double rounds_256(long T) {
  const long P = num_cus();
  return (double)((T + P - 1) / P);
}
template <int ACT, int EPI>
static void launch_256s(GemmArgs& g, hipStream_t s) {
  const dim3 grid = grid_256s((long)g.tilesM * g.tilesN);
  if (g.K2 > 0) hipLaunchKernelGGL((gemm_nt_256s_kernel<ACT, EPI, true>), grid, dim3(1024), 0, s, g);
  else hipLaunchKernelGGL((gemm_nt_256s_kernel<ACT, EPI, false>), grid, dim3(1024), 0, s, g);
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
  return (double)((t144 + P - 1) / P) * g_gemm_bm144_cost < rounds_256(t256);
}
// fewest 64x128 tiles for which the 64x128 small-tile kernel is taken over the 64x64 one (A/B: lhrs_gemm_set_small_thresh)
static int g_gemm_small_thresh = 256;
extern "C" int lhrs_gemm_set_small_thresh(int n) { g_gemm_small_thresh = n; return 0; }
static int g_gemm_min256 = 128;  // fewest 256x256 tiles (half a round of the 256 CUs) for which the big-tile kernels are chosen: 2184 x 4096 (144
                                 // tiles, the reference's micro-batch 8) runs 20 % faster there than on 576 small tiles; A/B: lhrs_gemm_set_min_tiles
extern "C" int lhrs_gemm_set_min_tiles(int n) { g_gemm_min256 = n; return 0; }
// tile policy switch for A/B measurements: 0 = never a 256x256 / 144x256 persistent kernel (small tiles only), 2 = default
extern "C" int lhrs_gemm_set_policy(int allow_256) { g_gemm_allow_256 = allow_256 ? 2 : 0; return 0; }

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

// ---- plain long-k products: the four-wave kernel by a shape rule ------------------------------------------------------------------------
// A product with a plain epilogue (no bias, no activation, bf16 out, alpha 1; optional bf16 residual) on a long k-loop runs the four-wave
// gemm_u4_kernel (gemm_u4.hip: 128x128 per wave, paced DMA - 5-18 % faster than the 16-wave kernel on these shapes) whenever its 256x256 tiles
// fill at least ~80 % of one round of the CUs; everything else takes this file's kernels (gemm_launch).  The rule is a function of the SHAPE only -
// no timing, no cache, no state: a run is bit-reproducible, every data-parallel rank runs the same kernels, ragged row counts (padded batches,
// the supervised rows of lm_head) cost nothing, and a capturing stream sees the same kernels as an eager one.  Where the threshold comes from
// (profiles/r04_plain_first_call_timing.txt, us, 16-wave / four-wave): M = 8190, N = 4096 (512 tiles): K = 4096 204.8 / 196.0, 12288 584.4 / 525.0,
// 22016 1135.8 / 976.6; M = 3840, N = 4096 (240 tiles): K = 4096 100.9 / 96.8, 22016 560.8 / 476.7; M = 3840, N = 32000: 776.0 / 766.7;
// M = 4320, N = 1024 (68 tiles): 58.4 / 76.0 - the four-wave kernel has one tile height, so a launch that cannot fill the chip stays on the 144-row /
// small-tile kernels (M = 2184, N = 4096, the reference's micro-batch 8: 144 tiles).  lhrs_gemm_set_u4(0) / LHRS_GEMM_U4=0: the 16-wave kernels everywhere (kernel A/B tests).
extern "C" int lhrs_gemm_u4_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const void* residual, int ldr,
                               void* stream);
static int g_u4_on = -1;
static int prof_count(int M, int N, int K, int kind, hipStream_t s);
static void prof_end(int slot, hipStream_t s);
static void prof_undo(int slot, int kind, double flops);
static void plain_env() {
  if (g_u4_on < 0) {
    const char* e = getenv("LHRS_GEMM_U4");
    g_u4_on = (e == nullptr || e[0] != '0') ? 1 : 0;
  }
}
extern "C" int lhrs_gemm_set_u4(int on) { plain_env(); g_u4_on = on ? 1 : 0; return 0; }
// 1 when lhrs_gemm_bf16_nt runs this problem on gemm_u4_kernel (pure function of the arguments and of lhrs_gemm_set_u4)
extern "C" int lhrs_gemm_u4_takes(int M, int N, int K, int lda, int ldb, int ldc, int ldr, int has_bias, int act, int out_f32, int accumulate, float alpha) {
  plain_env();
  const long tiles = (long)cdiv(M, 256) * cdiv(N, 256);
  return g_u4_on == 1 && g_gemm_allow_256 != 0 && !has_bias && act == 0 && !out_f32 && !accumulate && alpha == 1.f && K >= 4096 && K % 64 == 0 && M >= 1024 && N >= 1024 &&
         5 * tiles >= 4L * num_cus() && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && ldr % 8 == 0;
}
static int gemm_launch(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const void* bias,
                       const void* residual, int ldr, int act, int out_f32, int accumulate, float alpha, const void* A2,
                       int lda2, const void* B2, int ldb2, int K2, void* stream);
// The fused-epilogue products (qkv + RoPE, gate|up + SwiGLU, d-down + SwiGLU') on the four-wave kernel: a LoRA pair in the k-loop only for RoPE (stage 3's
// q|k|v adapters; the SwiGLU products with a pair live in the 16-wave kernel), K >= 4096, and a tile walk whose rounds leave at most 15 % (SwiGLU backward: 5 %) of the CU-rounds idle (idle share over ALL rounds, R / T <= 1.15) - the four-wave kernel has one tile height and no tail-row
// split for the fused epilogues.  RoPE and SwiGLU forward: 15 % (micro-batch 30: 6.0 / 10.75 rounds, micro-batch 60: 12 / 21.5: taken; the reference's micro-batch 8,
// M = 2184: 1.69 / 3.02 rounds: the 16-wave kernels with their 144-row tiles and tail-row rules).  SwiGLU backward: 5 % - its write-out is VALU-bound on four waves
// (the sigmoid of 256 elements per lane beside one stage of MFMAs), so it only wins where the tile walk fits: micro-batch 60 (10.75 rounds) 1261 us against the 16-wave
// kernel's ~1307; micro-batch 30 (5.375 rounds) 678 against 634 on the same box (profiles/r05_bench_line_b30_ab_16wave.json) - stays on the 16-wave kernel.
// kind: 0 RoPE (tiles_n = N / 256), 1 SwiGLU forward (ff / 128), 2 SwiGLU backward (ff / 256).  A pure function of the shape.
extern "C" int lhrs_gemm_u4_fused_takes(int kind, int M, int tiles_n, int K, int K2) {
  plain_env();
  const long P = num_cus(), T = (long)cdiv(M, 256) * tiles_n, R = (T + P - 1) / P * P;
  const long idle_ok = kind == 2 ? 100 * R <= 105 * T : 20 * R <= 23 * T;
  return g_u4_on == 1 && g_gemm_allow_256 != 0 && (kind == 0 ? K2 % 64 == 0 : K2 == 0) && K >= 4096 && K % 64 == 0 && M >= 1024 && 5 * T >= 4 * P && idle_ok;
}
extern "C" int lhrs_gemm_u4_rope(const void* X, int ldx, const void* W, int ldw, void* C, int ldc, int M, int N, int K, const float* cos_t, const float* sin_t,
                                 int pos_mod, int pos0, int rope_cols, void* stream);
extern "C" int lhrs_gemm_u4_rope_lora(const void* X, int ldx, const void* W, int ldw, const void* A2, int lda2, const void* B2, int ldb2, int K2, void* C, int ldc,
                                      int M, int N, int K, const float* cos_t, const float* sin_t, int pos_mod, int pos0, int rope_cols, void* stream);
extern "C" int lhrs_gemm_u4_nt_lora(const void* A, int lda, const void* B, int ldb, const void* A2, int lda2, const void* B2, int ldb2, int K2, void* C, int ldc,
                                    int M, int N, int K, const void* residual, int ldr, void* stream);
extern "C" int lhrs_gemm_u4_swiglu_fwd(const void* X, int ldx, const void* Wgu, int ldw, void* gu, int ld_gu, void* act, int ld_act, int M, int ff, int K, void* stream);
extern "C" int lhrs_gemm_u4_swiglu_bwd(const void* dY, int ldy, const void* WdT, int ldw, const void* gu, void* dgu, int ld_gu, int M, int ff, int K, void* stream);
#define U4_FUSED_TRY(kind_, flopsN_, call_)                                                                          \
  {                                                                                                                  \
    const int pslot_ = prof_count(M, flopsN_, K, kind_, (hipStream_t)stream);                                        \
    const int st_ = call_;                                                                                           \
    if (st_ == 0) { prof_end(pslot_, (hipStream_t)stream); return 0; }                                               \
    if (st_ < 0) return st_;                                                                                         \
    prof_undo(pslot_, kind_, 2.0 * M * (double)(flopsN_) * K);   /* declined (addressing limits): as if never counted */ \
  }

static int tail_rows_splitk(const bf16_t* A, int lda, const bf16_t* B, int ldb, bf16_t* C, int ldc, int M_tail, int N, int K, const bf16_t* residual, int ldr,
                            float alpha, hipStream_t s);
// The four-wave kernel walks whole 256x256 tiles in ceil(T / P) rounds of the P CUs.  When the last round would be mostly empty (M = 8736, BASELINE configs[3]'s
// micro-batch 32: 35 x 16 = 560 tiles = 2.19 rounds -> 3) the rows are cut behind the last tile row the whole rounds cover: those rows go to gemm_u4_kernel, the
// remaining rows (544 of 8736) to gemm_launch's kernels (small tiles, split-K over the workspace for the long k-loops) - disjoint rows of C.  Returns the number of
// rows for gemm_u4_kernel (M itself: no cut).  A pure function of the shape.
static int u4_main_rows(int M, long tiles_n) {
  const long P = num_cus(), tm = cdiv(M, 256), T = tm * tiles_n, full = T / P;
  if (T % P == 0 || full < 1 || 20 * ((T + P - 1) / P * P) <= 23 * T) return M;        // whole rounds, or the idle share of the last round is <= 15 %
  const long tm_main = full * P / tiles_n;
  return tm_main >= 1 && tm_main < tm ? (int)(tm_main * 256) : M;
}
extern "C" int lhrs_gemm_u4_main_rows(int M, int N) { return u4_main_rows(M, cdiv(N, 256)); }

// the plain-epilogue product on gemm_u4_kernel when the shape rule takes it (optionally with the LoRA pair in the k-loop): 0 done, 1 not taken, -1 error
static int plain_u4_try(const void* A, int lda, const void* B, int ldb, const void* A2, int lda2, const void* B2, int ldb2, int K2, void* C, int ldc, int M, int N,
                        int K, const void* bias, const void* residual, int ldr, int act, int out_f32, int accumulate, float alpha, void* stream) {
  if (!lhrs_gemm_u4_takes(M, N, K, lda, ldb, ldc, residual ? ldr : 0, bias != nullptr, act, out_f32, accumulate, alpha) || t_drop.thresh != 0 ||
      ((size_t)A | (size_t)B | (size_t)C | (size_t)residual | (size_t)A2 | (size_t)B2) % 16 != 0)
    return 1;
  hipStream_t s = (hipStream_t)stream;
  const int Mu = u4_main_rows(M, cdiv(N, 256));
  const int ukind = residual ? 5 : 6;
  const int slot = prof_count(Mu, N, K + K2, ukind, s);
  const int st = K2 > 0 ? lhrs_gemm_u4_nt_lora(A, lda, B, ldb, A2, lda2, B2, ldb2, K2, C, ldc, Mu, N, K, residual, ldr, stream)
                        : lhrs_gemm_u4_nt(A, lda, B, ldb, C, ldc, Mu, N, K, residual, ldr, stream);
  if (st == 0) {
    prof_end(slot, s);
    if (Mu == M) return 0;
    const bf16_t* At = (const bf16_t*)A + (long)Mu * lda;
    const bf16_t* Rt = residual ? (const bf16_t*)residual + (long)Mu * ldr : nullptr;
    bf16_t* Ct = (bf16_t*)C + (long)Mu * ldc;
    if (K2 == 0) {   // the split-K tail has no second operand pair: a product with one takes the small-tile kernels over its whole k-loop
      const int ts = tail_rows_splitk(At, lda, (const bf16_t*)B, ldb, Ct, ldc, M - Mu, N, K, Rt, ldr, 1.f, s);
      if (ts <= 0) return ts;
    }
    return gemm_launch(At, lda, B, ldb, Ct, ldc, M - Mu, N, K, nullptr, Rt, ldr, 0, 0, 0, 1.f, K2 > 0 ? (const bf16_t*)A2 + (long)Mu * lda2 : nullptr, lda2, B2, ldb2,
                       K2, stream) ? -1 : 0;
  }
  if (st < 0) return st;
  prof_undo(slot, ukind, 2.0 * Mu * N * (K + K2));   // not its problem after all (addressing limits): as if never counted
  return 1;
}

extern "C" int lhrs_gemm_bf16_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N,
                                 int K, const void* bias, const void* residual, int ldr, int act, int out_f32,
                                 int accumulate, float alpha, void* stream) {
  const int st = plain_u4_try(A, lda, B, ldb, nullptr, 0, nullptr, 0, 0, C, ldc, M, N, K, bias, residual, ldr, act, out_f32, accumulate, alpha, stream);
  if (st <= 0) return st;
  return gemm_launch(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, act, out_f32, accumulate, alpha, nullptr, 0, nullptr, 0,
                     0, stream);
}

// C = alpha * (A.B^T + A2.B2^T) + bias + residual: the rank-K2 LoRA update rides in the k-loop of the base GEMM
// (peft lora.Linear forward, y = W x + (alpha/r) B A x, reached from lhrs/models/text_modal.py:133-151).  The same shape rule as lhrs_gemm_bf16_nt decides
// between gemm_u4_kernel and the 16-wave kernels (the pair adds K2 / 64 stages to either k-loop; the two are bit-identical).
extern "C" int lhrs_gemm_bf16_nt_lora(const void* A, int lda, const void* B, int ldb, const void* A2, int lda2, const void* B2,
                                      int ldb2, int K2, void* C, int ldc, int M, int N, int K, const void* bias,
                                      const void* residual, int ldr, int out_f32, int accumulate, float alpha, void* stream) {
  LHRS_REQUIRE(A2 && B2 && K2 > 0 && K2 % 64 == 0 && lda2 % 8 == 0 && ldb2 % 8 == 0 && lda2 >= K2 && ldb2 >= K2,
               "gemm_lora: bad second operand pair (K2=%d lda2=%d ldb2=%d)", K2, lda2, ldb2);
  const int st = plain_u4_try(A, lda, B, ldb, A2, lda2, B2, ldb2, K2, C, ldc, M, N, K, bias, residual, ldr, 0, out_f32, accumulate, alpha, stream);
  if (st <= 0) return st;
  return gemm_launch(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, 0, out_f32, accumulate, alpha, A2, lda2, B2, ldb2, K2,
                     stream);
}

// The tail rows of a row-split product with a LONG k-loop (M = 8736: 544 rows x 4096 columns = 160 tiles of 128^2, each walking K = 11008 .. 22016 alone: 100 - 190
// us at 500 TFLOP/s, tools/gemm_tail_sweep.py): K is cut into slabs of ~4096 across blockIdx.y - f32 slabs in the registered workspace, summed in a fixed order with
// the residual by one small launch - so that every CU has work for the whole tail.  0 launched, 1 not applicable (short k-loop, no workspace), -1 error.
static int tail_rows_splitk(const bf16_t* A, int lda, const bf16_t* B, int ldb, bf16_t* C, int ldc, int M_tail, int N, int K, const bf16_t* residual, int ldr,
                            float alpha, hipStream_t s) {
  int dev = 0;
  const int splits = K >= 8192 ? (K + 2048) / 4096 : 1;
  if (splits <= 1 || N % 128 != 0 || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || g_sk[dev].slabs == nullptr ||
      (long)splits * M_tail * N * 4 > g_sk[dev].units * 256 * 256 * 4)
    return 1;
  const int ks = cdiv(K / 64, splits) * 64, used = cdiv(K, ks);
  GemmArgs gt; memset(&gt, 0, sizeof(gt));
  gt.A = A; gt.B = B; gt.C = g_sk[dev].slabs; gt.M = M_tail; gt.N = N; gt.K = K;
  gt.lda = lda; gt.ldb = ldb; gt.ldc = N; gt.alpha = 1.f; gt.out_f32 = 1; gt.ksplit = ks; gt.drop_scale = 1.f;
  gt.tilesM = cdiv(M_tail, 128); gt.tilesN = cdiv(N, 128);
  if (g_prof.on) { g_prof.launches_all++; g_prof.total_flops_all += 2.0 * M_tail * N * K; }
  hipLaunchKernelGGL((gemm_nt_kernel<4, 4, 0>), dim3(gt.tilesM * gt.tilesN, used), dim3(256), 0, s, gt);
  const long work = (long)M_tail * (N / 4);
  int rg = (int)((work + 255) / 256); if (rg > 8192) rg = 8192;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rg), dim3(256), 0, s, (const float*)g_sk[dev].slabs, C, (long)ldc, M_tail, N, used, alpha, residual, (long)ldr);
  LHRS_CHECK_LAUNCH("gemm_tail_splitk");
  return 0;
}

static int gemm_launch(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const void* bias,
                       const void* residual, int ldr, int act, int out_f32, int accumulate, float alpha, const void* A2,
                       int lda2, const void* B2, int ldb2, int K2, void* stream) {
  LHRS_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
  LHRS_REQUIRE(K % 64 == 0, "gemm: K=%d must be a multiple of 64 (zero-pad the reduction dim)", K);
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
  bool use256 = g_gemm_allow_256 && t256 >= g_gemm_min256 && K >= 128 && al16;  // the persistent kernels' pipeline needs >= 2 stages
  // just under the big-tile threshold but nearly a full round of 144-row tiles (ViT o / fc2 at micro-batch 30: 124 tiles of 256 rows, 216 of
  // 144): the persistent 144-row kernel beats the small tiles (30.2 -> 25.9, 75.1 -> 68.4 us; tools/gemm_vit_sweep.py)
  if (!use256 && g_gemm_allow_256 == 2 && g_gemm_bm144 == 1 && al16 && !out_f32 && K % 64 == 0 && K >= 192 && K2 == 0 && !g.drop_thresh &&
      t256 >= g_gemm_min256 / 2 && (long)cdiv(M, 144) * cdiv(N, 256) >= 200 && (long)cdiv(M, 144) * cdiv(N, 256) <= num_cus())
    use256 = true;
  if (K2 > 0 && !use256) {  // small problems: base GEMM, then the rank-K2 update accumulated on top of it
    if (gemm_launch(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, act, out_f32, accumulate, alpha, nullptr, 0, nullptr, 0, 0, stream))
      return -1;
    return gemm_launch(A2, lda2, B2, ldb2, C, ldc, M, N, K2, nullptr, out_f32 ? nullptr : C, ldc, 0, out_f32, out_f32 ? 1 : 0, alpha,
                       nullptr, 0, nullptr, 0, 0, stream);
  }
  // Tail rows: T = tilesM * tilesN 256x256 tiles run as ceil(T / 256) rounds of the 256 CUs, and a nearly empty last round costs as
  // much as a full one (M = 8736, N = 4096: 560 tiles = 2.19 rounds -> 3).  When the tile rows that spill over the last full round
  // are cheaper as a separate small-tile launch (~2.5x the time per FLOP, but no idle CUs), the row range is cut there: whole
  // 256-row tile rows for the 16-wave kernel, the remaining rows for the 64x128 / 128x128 kernel.  Disjoint rows of C, no partials.
  const bool s_kernel = use256;
  const bool bm144 = s_kernel && !out_f32 && pick_144(M, cdiv(N, 256), K, K2, g.drop_thresh != 0);
  if (s_kernel && !bm144 && t_split_ok && g_gemm_tail_split && !g.drop_thresh) {
    const int tm = cdiv(M, 256), tn = cdiv(N, 256);
    const long T = (long)tm * tn, rounds = (T + 255) / 256, full = T / 256;
    const int tm_main = (int)(full * 256 / tn);
    if (full >= 1 && T % 256 != 0 && tm_main >= 1 && tm_main < tm) {
      const long t_main = (long)tm_main * tn, t_tail = T - t_main;
      const double split_cost = (double)((t_main + 255) / 256) + 2.5 * (double)t_tail / 256.0 + 0.05;
      (void)rounds;
      if (split_cost < rounds_256(T)) {
        const int M_main = tm_main * 256, M_tail = M - M_main;
        const long esz = out_f32 ? 4 : 2;
        t_split_ok = false;
        int rc = gemm_launch(A, lda, B, ldb, C, ldc, M_main, N, K, bias, residual, ldr, act, out_f32, accumulate, alpha, A2, lda2, B2, ldb2, K2, stream);
        if (!rc && !out_f32 && K2 == 0 && bias == nullptr && act == 0) {
          const int ts = tail_rows_splitk((const bf16_t*)A + (long)M_main * lda, lda, (const bf16_t*)B, ldb, (bf16_t*)C + (long)M_main * ldc, ldc, M_tail, N, K,
                                          residual ? (const bf16_t*)residual + (long)M_main * ldr : nullptr, ldr, alpha, s);
          if (ts <= 0) { t_split_ok = true; return ts; }
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
    const bool dominant = s_kernel;
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
    if (bm144) {
      g.tilesM = cdiv(M, 144);
      const dim3 grid12 = grid_256s((long)g.tilesM * g.tilesN);
      switch (act) {
        case 0: LAUNCH_144(0, 0, grid12, s, g); break;
        case 1: LAUNCH_144(1, 0, grid12, s, g); break;
        case 2: LAUNCH_144(2, 0, grid12, s, g); break;
        default: LAUNCH_144(3, 0, grid12, s, g); break;
      }
    } else {
      const dim3 grid16 = grid_256s((long)g.tilesM * g.tilesN);
      switch (act) {
        case 0: LAUNCH_256(0, 0, grid16, s, g); break;
        case 1: LAUNCH_256(1, 0, grid16, s, g); break;
        case 2: LAUNCH_256(2, 0, grid16, s, g); break;
        default: LAUNCH_256(3, 0, grid16, s, g); break;
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
// a launch prof_count counted was declined by its kernel wrapper: everything prof_count did is taken back - the slot (the last one handed out), the totals, and the
// sampling phase seen[kind], which take() advanced exactly when a slot was free (whether or not this launch was the stride's sample)
static void prof_undo(int slot, int kind, double flops) {
  if (!g_prof.on) return;
  g_prof.launches_all--; g_prof.total_flops_all -= flops;
  if (slot >= 0) g_prof.used--;
  if (slot >= 0 || g_prof.used < g_prof.cap) g_prof.seen[kind]--;
}

extern "C" int lhrs_gemm_swiglu_fwd(const void* X, int ldx, const void* Wgu, int ldw, const void* A2, int lda2, const void* B2, int ldb2,
                                    int K2, void* gu, int ld_gu, void* act, int ld_act, int M, int ff, int K, void* stream) {
  LHRS_REQUIRE(M > 0 && ff > 0 && K > 0 && ff % 8 == 0 && ld_gu >= 2 * ff && ld_act >= ff && ld_gu % 8 == 0 && ld_act % 8 == 0,
               "gemm_swiglu_fwd: M=%d ff=%d K=%d ld_gu=%d ld_act=%d", M, ff, K, ld_gu, ld_act);
  if (ff % 128 == 0 && lhrs_gemm_u4_fused_takes(1, M, ff / 128, K, K2))
    U4_FUSED_TRY(7, 2 * ff, lhrs_gemm_u4_swiglu_fwd(X, ldx, Wgu, ldw, gu, ld_gu, act, ld_act, M, ff, K, stream))
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
  if (head_dim == 128 && rope_cols % 256 == 0 && lhrs_gemm_u4_fused_takes(0, M, cdiv(N, 256), K, K2))
    U4_FUSED_TRY(9, N, K2 > 0 ? lhrs_gemm_u4_rope_lora(X, ldx, W, ldw, A2, lda2, B2, ldb2, K2, C, ldc, M, N, K, cos_t, sin_t, pos_mod, pos0, rope_cols, stream)
                              : lhrs_gemm_u4_rope(X, ldx, W, ldw, C, ldc, M, N, K, cos_t, sin_t, pos_mod, pos0, rope_cols, stream))
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
  if (lhrs_gemm_u4_fused_takes(2, M, cdiv(ff, 256), K, K2))
    U4_FUSED_TRY(8, ff, lhrs_gemm_u4_swiglu_bwd(dY, ldy, WdT, ldw, gu, dgu, ld_gu, M, ff, K, stream))
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
