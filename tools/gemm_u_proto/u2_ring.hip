#include <hip/hip_runtime.h>
#include "agpr.inc"
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
struct Args { const unsigned short* A; const unsigned short* B; unsigned short* C; int M, N, K, lda, ldb, ldc, tilesM, tilesN; };

// prototype 2: 256x256 tile, FOUR waves (2x2 of 128x128, 256 AGPR accumulators), a RING of five 32-k half-stages (5 x 32 KiB), one boundary
// (fragment wait, counted vmcnt, barrier) per 64-MFMA block, the DMA of half-stage b + 5 issued at the boundary of block b
__global__ __launch_bounds__(256, 1) void gemm_u2(Args g) {
  constexpr int BM = 256, BN = 256, BKH = 32, NB = 5, A_BYTES = BM * BKH * 2, HS = A_BYTES + BN * BKH * 2;
  __shared__ __attribute__((aligned(16))) char smem[NB * HS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tm, tn;
  {
    const int nblk = g.tilesM * g.tilesN, bid = blockIdx.x;
    const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    constexpr int GM = 8;
    const int per_group = GM * g.tilesN, grp = lin / per_group, rem = lin - grp * per_group;
    const int rows = min(GM, g.tilesM - grp * GM);
    tm = grp * GM + rem % rows; tn = rem / rows;
  }
  const bool isA = wave < 2;
  const char* base = reinterpret_cast<const char*>(isA ? g.A : g.B);
  const long ld = isA ? g.lda : g.ldb;
  const int rmax = (isA ? g.M : g.N) - 1;
  const int row0 = isA ? tm * BM : tn * BN;
  const int nh = g.K / BKH;
  unsigned off[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int r = ((wave & 1) * 8 + j) * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((r >> 1) & 3);
    const int row = min(row0 + r, rmax);
    off[j] = (unsigned)(((long)row * ld + chunk * 8) * 2);
  }
  const int dst0 = (isA ? 0 : A_BYTES) + (wave & 1) * 8192;
  auto issue = [&](int h, int buf, int j) {
    if (h >= nh) return;
    const int kb_ = __builtin_amdgcn_readfirstlane(h * (BKH * 2));
    unsigned vo = off[j];
    asm volatile("" : "+v"(vo));
    __builtin_amdgcn_global_load_lds((gptr_t)(base + kb_ + vo), (lptr_t)(smem + buf * HS + dst0 + j * 1024), 16, 0, 0);
  };
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  const unsigned a0 = lds0 + (wm * 128 + (lane & 15)) * 64 + (((lane >> 4) ^ ((lane >> 1) & 3)) * 16);
  const unsigned b0 = lds0 + A_BYTES + (wn * 128 + (lane & 15)) * 64 + (((lane >> 4) ^ ((lane >> 1) & 3)) * 16);

#pragma unroll
  for (int h = 0; h < NB; ++h)
#pragma unroll
    for (int j = 0; j < 8; ++j) issue(h, h, j);
#define ZR(mi, ni) asm volatile("v_accvgpr_write_b32 " AS_##mi##_##ni##_0 ", 0\n v_accvgpr_write_b32 " AS_##mi##_##ni##_1 ", 0\n v_accvgpr_write_b32 " AS_##mi##_##ni##_2 ", 0\n v_accvgpr_write_b32 " AS_##mi##_##ni##_3 ", 0" ::: CL_##mi##_##ni);
#define ZROW(mi) ZR(mi, 0) ZR(mi, 1) ZR(mi, 2) ZR(mi, 3) ZR(mi, 4) ZR(mi, 5) ZR(mi, 6) ZR(mi, 7)
  ZROW(0) ZROW(1) ZROW(2) ZROW(3) ZROW(4) ZROW(5) ZROW(6) ZROW(7)
  asm volatile("s_waitcnt vmcnt(32)" ::: "memory");  // half-stage 0 has landed (the other four may be in flight)
  __builtin_amdgcn_s_barrier();

  bf16x8 A0[8], B0[8], A1[8], B1[8];
#define RDQ(dst, addr, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off_))
#define SB __builtin_amdgcn_sched_barrier(0);
#define RD8(X, ad) RDQ(X[0], ad, 0); RDQ(X[1], ad, 1024); RDQ(X[2], ad, 2048); RDQ(X[3], ad, 3072); RDQ(X[4], ad, 4096); RDQ(X[5], ad, 5120); RDQ(X[6], ad, 6144); RDQ(X[7], ad, 7168);
  { RD8(A0, a0) RD8(B0, b0) }
  auto wait16 = [&](bf16x8 (&a)[8], bf16x8 (&b)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                 "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
    __builtin_amdgcn_sched_barrier(0);
  };
#define MFM(Ac, Bc, mi, ni) asm volatile("v_mfma_f32_16x16x32_bf16 " AR_##mi##_##ni ", %0, %1, " AR_##mi##_##ni :: "v"(Bc[ni]), "v"(Ac[mi]) : CL_##mi##_##ni); SB
  // row mi of a block: 8 MFMAs; in rows 0..3 four fragment reads of the next block and two DMA pieces
#define ROW(Ac, Bc, An, Bn, aa, ba, mi, RD)                                                                        \
  MFM(Ac, Bc, mi, 0) if (RD && mi < 4) RDQ(An[2 * mi], aa, (2 * mi) * 1024); SB                                     \
  MFM(Ac, Bc, mi, 1) if (mi < 4) { issue(hh, bb, mi * 2); } SB                                                     \
  MFM(Ac, Bc, mi, 2) if (RD && mi < 4) RDQ(An[2 * mi + 1], aa, (2 * mi + 1) * 1024); SB                             \
  MFM(Ac, Bc, mi, 3)                                                                                               \
  MFM(Ac, Bc, mi, 4) if (RD && mi < 4) RDQ(Bn[2 * mi], ba, (2 * mi) * 1024); SB                                     \
  MFM(Ac, Bc, mi, 5) if (mi < 4) { issue(hh, bb, mi * 2 + 1); } SB                                                 \
  MFM(Ac, Bc, mi, 6) if (RD && mi < 4) RDQ(Bn[2 * mi + 1], ba, (2 * mi + 1) * 1024); SB                             \
  MFM(Ac, Bc, mi, 7)
#define BLOCKU(Ac, Bc, An, Bn, aa, ba, RD)                                                                         \
  ROW(Ac, Bc, An, Bn, aa, ba, 0, RD) ROW(Ac, Bc, An, Bn, aa, ba, 1, RD) ROW(Ac, Bc, An, Bn, aa, ba, 2, RD)          \
  ROW(Ac, Bc, An, Bn, aa, ba, 3, RD) ROW(Ac, Bc, An, Bn, aa, ba, 4, RD) ROW(Ac, Bc, An, Bn, aa, ba, 5, RD)          \
  ROW(Ac, Bc, An, Bn, aa, ba, 6, RD) ROW(Ac, Bc, An, Bn, aa, ba, 7, RD)
  // boundary of block b: its fragments are in registers, half-stage b + 1 has landed (counted: up to three younger half-stages in flight),
  // the barrier publishes it and frees buffer b % 5 for half-stage b + 5
#define BOUNDARY(Ac, Bc, b)                                                                                         \
  wait16(Ac, Bc);                                                                                                  \
  if (b + 4 < nh) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");                                                 \
  else if (b + 3 < nh) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                                            \
  else if (b + 2 < nh) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                             \
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                             \
  __builtin_amdgcn_s_barrier();

  int buf = 0;  // buffer of half-stage b
  for (int b = 0; b < nh; b += 2) {
    {
      const int hh = b + NB, bb = buf;
      const int nb1 = buf + 1 == NB ? 0 : buf + 1;
      const unsigned aa = a0 + nb1 * HS, ba = b0 + nb1 * HS;
      BOUNDARY(A0, B0, b)
      BLOCKU(A0, B0, A1, B1, aa, ba, true)
      buf = nb1;
    }
    {
      const int bo = b + 1;  // odd block (nh is even)
      const int hh = bo + NB, bb = buf;
      const int nb1 = buf + 1 == NB ? 0 : buf + 1;
      const unsigned aa = a0 + nb1 * HS, ba = b0 + nb1 * HS;
      BOUNDARY(A1, B1, bo)
      if (bo + 1 < nh) { BLOCKU(A1, B1, A0, B0, aa, ba, true) } else { BLOCKU(A1, B1, A0, B0, aa, ba, false) }
      buf = nb1;
    }
  }
  asm volatile("s_nop 15\n s_nop 15" ::: "memory");
#define RDACC(mi, ni, v) asm volatile("v_accvgpr_read_b32 %0, " AS_##mi##_##ni##_0 "\n v_accvgpr_read_b32 %1, " AS_##mi##_##ni##_1 "\n v_accvgpr_read_b32 %2, " AS_##mi##_##ni##_2 "\n v_accvgpr_read_b32 %3, " AS_##mi##_##ni##_3 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
#define ST(mi, ni)                                                                                                  \
  {                                                                                                                 \
    float v[4]; RDACC(mi, ni, v)                                                                                    \
    const int m = tm * BM + wm * 128 + mi * 16 + (lane & 15), n = tn * BN + wn * 128 + ni * 16 + (lane >> 4) * 4;   \
    if (m < g.M && n < g.N) {                                                                                       \
      unsigned short h[4];                                                                                          \
      for (int r = 0; r < 4; ++r) { unsigned u = __float_as_uint(v[r]); u += 0x7fffu + ((u >> 16) & 1u); h[r] = (unsigned short)(u >> 16); } \
      *reinterpret_cast<uint2*>(g.C + (long)m * g.ldc + n) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));      \
    }                                                                                                               \
  }
#define STROW(mi) ST(mi, 0) ST(mi, 1) ST(mi, 2) ST(mi, 3) ST(mi, 4) ST(mi, 5) ST(mi, 6) ST(mi, 7)
  STROW(0) STROW(1) STROW(2) STROW(3) STROW(4) STROW(5) STROW(6) STROW(7)
}

extern "C" int gemm_u_launch(const void* A, const void* B, void* C, int M, int N, int K, void* stream) {
  Args g{(const unsigned short*)A, (const unsigned short*)B, (unsigned short*)C, M, N, K, K, K, N, (M + 255) / 256, (N + 255) / 256};
  hipLaunchKernelGGL(gemm_u2, dim3(g.tilesM * g.tilesN), dim3(256), 0, (hipStream_t)stream, g);
  return hipGetLastError() != hipSuccess;
}
