// Weight-gradient GEMM for gfx950:  C[out, in] (f32) = sum over tokens t of  dY[t, out] * X[t, in]      ("TN": both operands TOKEN-major)
//
// What autograd computes for every trainable nn.Linear / nn.MultiheadAttention projection of the AttnPooler
// (/root/reference lhrs/models/common_arch.py:93-173, 302-333): dW = dY^T X, reduced over all B * L tokens.  The NT GEMM family wants the
// reduction index contiguous in memory, which for these operands meant writing transposed copies of dY and X first (62 transposes per
// step, ~1.6 ms at micro-batch 30).  This kernel takes them as they lie:
//   * a stage = 64 tokens of a 128-column slice of each operand, DMA'd straight into LDS (global_load_lds, 1 KiB = 4 token rows per wave
//     instruction) in the swizzled row-major image of the attention kernels ([64 tokens][128 cols], 16-B chunk ^= f(token));
//   * BOTH MFMA operands are formed by transposing LDS reads (ds_read_b64_tr_b16): lane (L, g) of a fragment gets column 16*blk + L at
//     tokens {4g..4g+3, 16+4g..16+4g+3} (+32 for the second k-step) - the same token permutation on both sides, so the dot product is
//     over the same 32 tokens;
//   * 128 x 128 output tile, 4 waves (2 x 2) of 4 x 4 v_mfma_f32_16x16x32_bf16 fragments; the X fragment is fed as the A operand and the
//     dY fragment as B, so a lane ends up with 4 consecutive `in` of one `out` row: 16-B f32 stores;
//   * two LDS stages (64 KiB): 2 workgroups per CU; the token range is split across blockIdx.y into f32 slabs (summed in a fixed order by
//     the caller's reduction: bit-reproducible), the last partial 64-token stage is staged through registers with zero fill.
#include "common.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int TK = 64;                 // tokens per stage
constexpr int TILE_B = TK * 128 * 2;   // bytes of one operand's stage image

__device__ __forceinline__ int swz128(int row) { return ((row & 7) << 1) | ((row >> 3) & 1); }

// per d-block byte offsets of this lane's transposing reads inside a [64][128] image (k-step 0, lower token half)
struct TrAddr4 {
  unsigned base[4];
  __device__ __forceinline__ void init(unsigned tile, int lane, int blk0) {
    const int L = lane & 15, g = lane >> 4;
    const int r0 = 4 * g + (L >> 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) base[i] = tile + r0 * 256 + (((2 * (blk0 + i) + ((L & 3) >> 1)) ^ swz128(r0)) << 4) + (L & 1) * 8;
  }
};
#define TN_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))

__device__ __forceinline__ bf16x8 join4(const bf16x4& lo, const bf16x4& hi) { return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; }

struct TnArgs {
  const bf16_t* P;  // dY [T, ldp], columns [0, Mo)
  const bf16_t* Q;  // X  [T, ldq], columns [0, No)
  float* C;         // slab s at C + s * Mo * No (ldc = No), or the final matrix when splits == 1 (ldc given)
  long ldp, ldq, ldc;
  int T, Mo, No, stages_per_split;
};

// DMA one 64-token stage of a 128-column slice: wave w issues 4 pieces (rows 16w..16w+15), lane -> (row l / 16, chunk l % 16), swizzled SOURCE
__device__ __forceinline__ void dma_stage(char* lds, const bf16_t* base, long ld, int t0, int col0, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r0 = wave * 16 + j * 4;
    const int row = r0 + (lane >> 4);
    const int c = (lane & 15) ^ swz128(row);
    __builtin_amdgcn_global_load_lds((gptr_t)(base + (long)(t0 + row) * ld + col0 + c * 8), (lptr_t)(lds + r0 * 256), 16, 0, 0);
  }
}
// the last, partial stage: through registers, tokens >= T read as zero
__device__ __forceinline__ void reg_stage(char* lds, const bf16_t* base, long ld, int t0, int T, int col0, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + i * 256;
    const int row = idx >> 4, c = idx & 15;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (t0 + row < T) v = *reinterpret_cast<const uint4*>(base + (long)(t0 + row) * ld + col0 + c * 8);
    *reinterpret_cast<uint4*>(lds + row * 256 + ((c ^ swz128(row)) << 4)) = v;
  }
}

__global__ __launch_bounds__(256, 2) void gemm_tn_f32_kernel(TnArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * TILE_B];  // [stage][P | Q]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = a.No / 128;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;   // out tile, in tile
  const int wo = wave >> 1, wi = wave & 1;                           // wave's 64 x 64 quadrant: out half, in half
  const int total_stages = (a.T + TK - 1) / TK;
  const int s0 = blockIdx.y * a.stages_per_split, s1 = min(s0 + a.stages_per_split, total_stages);
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);

  f32x4 acc[4][4];  // [out block][in block]: lane (L, g) holds C[out = 16 ob + L][in = 16 ib + 4 g + 0..3]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto load = [&](int s, int buf) {
    char* p = smem + buf * 2 * TILE_B;
    const int t0 = s * TK;
    if (t0 + TK <= a.T) {
      dma_stage(p, a.P, a.ldp, t0, tm * 128, wave, lane);
      dma_stage(p + TILE_B, a.Q, a.ldq, t0, tn * 128, wave, lane);
    } else {
      reg_stage(p, a.P, a.ldp, t0, a.T, tm * 128, tid);
      reg_stage(p + TILE_B, a.Q, a.ldq, t0, a.T, tn * 128, tid);
    }
  };
  if (s0 < s1) load(s0, 0);
  for (int s = s0; s < s1; ++s) {
    const int buf = (s - s0) & 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();                       // stage s is complete in `buf`; everybody is done reading the other buffer
    if (s + 1 < s1) load(s + 1, buf ^ 1);
    TrAddr4 tp, tq;
    tp.init(lds0 + buf * 2 * TILE_B, lane, wo * 4);
    tq.init(lds0 + buf * 2 * TILE_B + TILE_B, lane, wi * 4);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {       // two k-steps of 32 tokens
      bf16x4 plo[4], phi[4], qlo[4], qhi[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (ks == 0) { TN_TR(plo[i], tp.base[i], 0); TN_TR(phi[i], tp.base[i], 16 * 256); TN_TR(qlo[i], tq.base[i], 0); TN_TR(qhi[i], tq.base[i], 16 * 256); }
        else { TN_TR(plo[i], tp.base[i], 32 * 256); TN_TR(phi[i], tp.base[i], 48 * 256); TN_TR(qlo[i], tq.base[i], 32 * 256); TN_TR(qhi[i], tq.base[i], 48 * 256); }
      }
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(plo[0]), "+v"(plo[1]), "+v"(plo[2]), "+v"(plo[3]), "+v"(phi[0]), "+v"(phi[1]), "+v"(phi[2]), "+v"(phi[3]),
                     "+v"(qlo[0]), "+v"(qlo[1]), "+v"(qlo[2]), "+v"(qlo[3]), "+v"(qhi[0]), "+v"(qhi[1]), "+v"(qhi[2]), "+v"(qhi[3]));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)   // A = X fragment (m = in column), B = dY fragment (n = out column): C[m = in 4g + r][n = out L]
          acc[ob][ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join4(qlo[ib], qhi[ib]), join4(plo[ob], phi[ob]), acc[ob][ib], 0, 0, 0);
    }
  }
  // store: lane (L, g): out row 16 ob + L, in columns 16 ib + 4 g .. + 3
  const int L = lane & 15, g = lane >> 4;
  float* c = a.C + (long)blockIdx.y * a.Mo * a.No * (gridDim.y > 1 ? 1 : 0);
  const long ldc = gridDim.y > 1 ? a.No : a.ldc;
#pragma unroll
  for (int ob = 0; ob < 4; ++ob) {
    const int orow = tm * 128 + wo * 64 + ob * 16 + L;
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) {
      const int icol = tn * 128 + wi * 64 + ib * 16 + g * 4;
      *reinterpret_cast<float4*>(c + (long)orow * ldc + icol) = make_float4(acc[ob][ib][0], acc[ob][ib][1], acc[ob][ib][2], acc[ob][ib][3]);
    }
  }
}

// sum of `splits` f32 slabs [Mo, No] -> C[Mo, ldc] in slab order (deterministic)
__global__ void tn_slab_reduce_kernel(const float* __restrict__ part, float* __restrict__ C, long ldc, int Mo, int No, int splits) {
  const long total = (long)Mo * (No / 4);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / (No / 4);
    const int n = (int)(i % (No / 4)) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < splits; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(part + ((long)k * Mo + m) * No + n);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(C + m * ldc + n) = s;
  }
}

}  // namespace

// token splits that fill the 256 CUs twice over (two workgroups per CU) without making a split shorter than 8 stages
extern "C" int lhrs_gemm_tn_splits(int T, int Mo, int No) {
  const long tiles = (long)(Mo / 128) * (No / 128);
  if (tiles < 1 || T < 1) return 1;  // (shapes lhrs_gemm_tn_f32 rejects)
  const int stages = (T + TK - 1) / TK;
  long s = (512 + tiles - 1) / tiles;
  if (s > stages / 8) s = stages / 8;
  if (s > 32) s = 32;
  return s < 1 ? 1 : (int)s;
}

// C[Mo, No] f32 = P[T, Mo]^T . Q[T, No]   (P = dY, Q = X, both token-major bf16; Mo, No multiples of 128; ldp / ldq multiples of 8).
// workspace: lhrs_gemm_tn_splits(T, Mo, No) * Mo * No floats (may be NULL when that is 1).
extern "C" int lhrs_gemm_tn_f32(const void* P, long ldp, const void* Q, long ldq, float* C, long ldc, int T, int Mo, int No,
                                float* workspace, void* stream) {
  LHRS_REQUIRE(T > 0 && Mo > 0 && No > 0 && Mo % 128 == 0 && No % 128 == 0, "gemm_tn_f32: T=%d Mo=%d No=%d (Mo, No multiples of 128)", T, Mo, No);
  LHRS_REQUIRE(ldp % 8 == 0 && ldq % 8 == 0 && ldp >= Mo && ldq >= No && ldc % 4 == 0 && ldc >= No, "gemm_tn_f32: ldp=%ld ldq=%ld ldc=%ld", ldp, ldq, ldc);
  LHRS_REQUIRE((((size_t)P | (size_t)Q) & 15) == 0 && ((size_t)C & 15) == 0, "gemm_tn_f32: operands must be 16-byte aligned");
  const int splits = lhrs_gemm_tn_splits(T, Mo, No);
  LHRS_REQUIRE(splits == 1 || workspace != nullptr, "gemm_tn_f32: %d token splits need a workspace of splits * Mo * No floats", splits);
  const int stages = (T + TK - 1) / TK;
  TnArgs a;
  a.P = (const bf16_t*)P; a.Q = (const bf16_t*)Q; a.ldp = ldp; a.ldq = ldq; a.T = T; a.Mo = Mo; a.No = No;
  a.stages_per_split = (stages + splits - 1) / splits;
  const int used = (stages + a.stages_per_split - 1) / a.stages_per_split;
  a.C = used > 1 ? workspace : C; a.ldc = ldc;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gemm_tn_f32_kernel, dim3((Mo / 128) * (No / 128), used), dim3(256), 0, s, a);
  LHRS_CHECK_LAUNCH("gemm_tn_f32");
  if (used > 1) {
    const long work = (long)Mo * (No / 4);
    int rg = (int)((work + 255) / 256); if (rg > 8192) rg = 8192;
    hipLaunchKernelGGL(tn_slab_reduce_kernel, dim3(rg), dim3(256), 0, s, workspace, C, ldc, Mo, No, used);
    LHRS_CHECK_LAUNCH("gemm_tn_f32 reduce");
  }
  return 0;
}
