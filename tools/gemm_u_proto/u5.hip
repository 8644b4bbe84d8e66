#include <hip/hip_runtime.h>
#include "agpr.inc"
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
struct Args { const unsigned short* A; const unsigned short* B; unsigned short* C; int M, N, K, lda, ldb, ldc, tilesM, tilesN; };

// prototype: 256x256x64 tile, FOUR waves (2x2 of 128x128 = 8x8 fragments of v_mfma_f32_16x16x32_bf16, 256 accumulator registers),
// two 64 KiB DMA stages, fragments double-buffered per 32-k block; u5: stage kt+2's DMA starts inside block 0 of stage kt (barrier X), the
// wait for stage kt+1 sits inside block 1 (barrier Y): every piece has 0.9-1.3 stages to land instead of 0.5-1.0 (gen_u5.py)
__global__ __launch_bounds__(256, 1) void gemm_u(Args g) {
  constexpr int BM = 256, BN = 256, BK = 64, A_BYTES = BM * BK * 2, STAGE = A_BYTES + BN * BK * 2;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
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
  const int nk = g.K / BK;
  unsigned off[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int ridx = (wave & 1) * 16 + j;
    const int lchunk = (lane & 7) ^ ((((j & 1) << 2) + (lane >> 4)) & 7);
    const int row = min(row0 + ridx * 8 + (lane >> 3), rmax);
    off[j] = (unsigned)(((long)row * ld + lchunk * 8) * 2);
  }
  const int dst0 = (isA ? 0 : A_BYTES) + (wave & 1) * 16384;
#ifdef DMA_MUBUF
  // the DMA through the MUBUF path: buffer_load_dwordx4 ... offen lds with a raw buffer descriptor over the whole operand (the k offset of the stage in the
  // SGPR soffset) instead of global_load_lds_dwordx4 in its saddr form
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  u32x4 rsrc;
  {
    const unsigned long long b = (unsigned long long)(size_t)base;
    rsrc.x = __builtin_amdgcn_readfirstlane((unsigned)b);
    rsrc.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);   // stride 0
    rsrc.z = 0xffffffffu;                                                       // num_records: rows are clamped by hand
    rsrc.w = 0x00020000u;
  }
  auto issue = [&](int kt, int buf, int j) {
    const unsigned soff = (unsigned)kt * (BK * 2);
    const unsigned lds_dst = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem) + buf * STAGE + dst0 + j * 1024;
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(off[j]), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory", "m0");
  };
#else
  auto issue = [&](int kt, int buf, int j) {
#ifdef ABL_NODMA
    return;
#endif
    const char* sp = base + (long)kt * (BK * 2);
    const unsigned lds_dst = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem) + buf * STAGE + dst0 + j * 1024;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off[j]), "s"(sp), "s"(lds_dst) : "memory", "m0");
  };
#endif
  const int wm = wave >> 1, wn = wave & 1;
  const int sw = ((lane & 15) >> 1) & 7;
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  const unsigned a0 = lds0 + (wm * 128 + (lane & 15)) * 128 + (((lane >> 4)) ^ sw) * 16;
  const unsigned b0 = lds0 + A_BYTES + (wn * 128 + (lane & 15)) * 128 + (((lane >> 4)) ^ sw) * 16;

#pragma unroll
  for (int j = 0; j < 16; ++j) issue(0, 0, j);
#define ZR(mi, ni) asm volatile("v_accvgpr_write_b32 " AS_##mi##_##ni##_0 ", 0\n v_accvgpr_write_b32 " AS_##mi##_##ni##_1 ", 0\n v_accvgpr_write_b32 " AS_##mi##_##ni##_2 ", 0\n v_accvgpr_write_b32 " AS_##mi##_##ni##_3 ", 0" ::: CL_##mi##_##ni);
#define ZROW(mi) ZR(mi, 0) ZR(mi, 1) ZR(mi, 2) ZR(mi, 3) ZR(mi, 4) ZR(mi, 5) ZR(mi, 6) ZR(mi, 7)
  ZROW(0) ZROW(1) ZROW(2) ZROW(3) ZROW(4) ZROW(5) ZROW(6) ZROW(7)
  if (nk > 1) {
#pragma unroll
    for (int j = 0; j < 16; ++j) issue(1, 1, j);
  }
  if (nk > 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  bf16x8 A0[8], B0[8], A1[8], B1[8];
#define RDQ(dst, addr, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off_))
#define SB __builtin_amdgcn_sched_barrier(0);
#define RD8(X, ad) RDQ(X[0], ad, 0); RDQ(X[1], ad, 2048); RDQ(X[2], ad, 4096); RDQ(X[3], ad, 6144); RDQ(X[4], ad, 8192); RDQ(X[5], ad, 10240); RDQ(X[6], ad, 12288); RDQ(X[7], ad, 14336);
  { RD8(A0, a0) RD8(B0, b0) }
  auto wait16 = [&](bf16x8 (&a)[8], bf16x8 (&b)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                 "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
    __builtin_amdgcn_sched_barrier(0);
  };
#define MFM(Ac, Bc, mi, ni) asm volatile("v_mfma_f32_16x16x32_bf16 " AR_##mi##_##ni ", %0, %1, " AR_##mi##_##ni :: "v"(Bc[ni]), "v"(Ac[mi]) : CL_##mi##_##ni); SB
#define BARX __builtin_amdgcn_s_barrier(); SB
  int kt = 0;
  // steady state: stages kt+1 and kt+2 exist - no conditions inside the 128-MFMA body
  // one DMA piece = s_add on m0 behind one MFMA, the load behind the next: never more than two non-MFMA instructions between two MFMAs (one wave per SIMD issues
  // one instruction per 4 cycles; a 16-cycle MFMA leaves three slots)
#define M0P(p) if (p == 0) { asm volatile("s_mov_b32 m0, %0" ::"s"(lds0 + (kt & 1) * STAGE + dst0) : "m0"); } else { asm volatile("s_add_u32 m0, m0, 0x400" ::: "m0", "scc"); }
#define GLDS(p) asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(off[p]), "s"(sp2) : "memory")
#define RDN(dst, ad, off_) RDQ(dst, ad, off_)
  // stage kt+1 must have landed (this wave's pieces: vmcnt; everybody's: the barrier); the 16 younger pieces (stage kt+2) stay in flight
#define WAITY(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory"); __builtin_amdgcn_s_barrier(); SB
  for (; kt + 2 < nk; ++kt) {
    const unsigned so = (kt & 1) * STAGE, sn = so ^ STAGE;
    const unsigned aa1 = a0 ^ (so | 64u), ba1 = b0 ^ (so | 64u), aa0 = a0 ^ sn, ba0 = b0 ^ sn;
    const char* sp2 = base + (long)(kt + 2) * (BK * 2);
    wait16(A0, B0);
#include "u5_body.inc"
  }
#undef M0P
#undef GLDS
#undef RDN
#undef WAITY
  // the last two stages: nothing left to request; the last one has nothing to read ahead
#define M0P(p)
#define GLDS(p)
#define RDN(dst, ad, off_) if (more) { RDQ(dst, ad, off_); }
#define WAITY(n) if (more) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } SB
  for (; kt < nk; ++kt) {
    const unsigned so = (kt & 1) * STAGE, sn = so ^ STAGE;
    const bool more = kt + 1 < nk;
    const unsigned aa1 = a0 ^ (so | 64u), ba1 = b0 ^ (so | 64u), aa0 = a0 ^ sn, ba0 = b0 ^ sn;
    wait16(A0, B0);
#include "u5_body.inc"
  }
  asm volatile("s_nop 15\n s_nop 15" ::: "memory");
  // plain epilogue straight from the accumulators (prototype)
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
  hipLaunchKernelGGL(gemm_u, dim3(g.tilesM * g.tilesN), dim3(256), 0, (hipStream_t)stream, g);
  return hipGetLastError() != hipSuccess;
}
