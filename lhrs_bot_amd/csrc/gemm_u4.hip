// gemm_u4_kernel: the plain bf16 NT product out[M,N] = a[M,K] . b[N,K]^T (+ residual) on a 256x256x64 tile walked by FOUR waves (one per SIMD),
// 128x128 of the tile per wave - HF LlamaDecoderLayer's nn.Linear calls with a plain epilogue (lhrs/models/text_modal.py:133-151), as gemm.hip.
//
// Why a second kernel next to the 16-wave gemm_nt_256s_kernel: a 128x128 wave tile reads 256 B of LDS per MFMA instead of 512 B, four waves meet at a
// barrier instead of sixteen, and with 512 registers per wave all fragments of a 32-k half sit in registers early enough to free the LDS buffer a fifth of
// the way into a stage - the DMA of stage kt+2 then has 0.9-1.7 stages to land instead of one (tools/gemm_u_proto/README.md, the time line the vendor
// library's assembly kernel uses).  What made it fast in HIP source:
//   * accumulators are not C++ variables: every MFMA is an `asm volatile` naming its AGPR tuple (gemm_u4_agpr.inc) - the register allocator cannot hold 64
//     accumulator tuples in place (docs/design_notes_r01_r02.md §4); fragments, addresses and control flow stay compiler-managed (164 VGPRs, no scratch)
//   * the 128-MFMA stage body is generated (tools/gemm_u_proto/gen_u5.py 1 21 6 108 1 -> gemm_u4_body.inc): fragment reads of the second 32-k half behind MFMAs
//     0..15, barrier X behind MFMA 21, ONE DMA piece every 6 MFMAs from there (16 pieces per wave), wait + barrier Y behind MFMA 108, the next stage's first
//     fragments behind MFMAs 109..124.  The pacing is the point: 64 pieces of 1 KiB per stage are 1024 cycles of the CU's address path - half the stage; issued one
//     per 2 MFMAs and wave they back up there and stall the issuing wave ~46 cycles each (1270-1345 TFLOP/s); one per 6 MFMAs: 1430-1570
//   * persistent over tiles (XCD-aware raster as in gemm.hip); the first two stages of the NEXT tile are requested before this tile's epilogue stores
// Same k order and fp32 accumulation as gemm_nt_256s_kernel: bit-identical results without a residual; a residual is added to the fp32 sum before the one
// rounding (as the vendor library and the split-K tail do; gemm_nt_256s_kernel's staged epilogue rounds the sum first) - tests/test_kernels_gpu.py.
// M = 8190, random operands, TFLOP/s (this kernel's main loop in tools/gemm_u_proto, 16-wave kernel, vendor library): 4096x22016 1530-1567 / 1320-1331 / 1575-1587,
// 4096x11008 1427-1456 / 1390-1414 / 1599-1614, 22016x4096 1431-1457 / 1406-1444 / 1400-1426, 11008x4096 1355-1383 / 1319-1347 / 1324-1334.
#include "common.h"
#include "gemm_u4_agpr.inc"

namespace {
typedef __attribute__((ext_vector_type(8))) short bf16x8;
struct U4Args {
  const bf16_t* A; const bf16_t* B; bf16_t* C; const bf16_t* res;
  int M, N, K, lda, ldb, ldc, ldr, tilesM, tilesN;
  const float* rope_cos; const float* rope_sin; int rope_mod, rope_pos0, rope_cols;   // ROPE variant: columns [0, rope_cols) are heads of 128, tables [pos][64] f32
};

__device__ __forceinline__ void u4_tile(const U4Args& g, int t, int& tm, int& tn) {
  const int nblk = g.tilesM * g.tilesN;
  const int xcd = t & 7, q = nblk >> 3, r = nblk & 7;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
  constexpr int GM = 8;
  const int per_group = GM * g.tilesN, grp = lin / per_group, rem = lin - grp * per_group;
  const int rows = min(GM, g.tilesM - grp * GM);
  tm = grp * GM + rem % rows; tn = rem / rows;
}

template <bool ROPE>
__global__ __launch_bounds__(256, 1) void gemm_u4_kernel(U4Args g) {
  constexpr int BM = 256, BN = 256, BK = 64, A_BYTES = BM * BK * 2, STAGE = A_BYTES + BN * BK * 2;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = g.tilesM * g.tilesN;
  // waves 0,1 bring the activation rows of a stage, waves 2,3 the weight rows: 16 pieces of 8 rows x 128 B each, chunk-swizzled as gemm.hip's image
  const bool isA = wave < 2;
  const char* base = reinterpret_cast<const char*>(isA ? g.A : g.B);
  const long ld = isA ? g.lda : g.ldb;
  const int rmax = (isA ? g.M : g.N) - 1;
  const int nk = g.K / BK;
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  const int dst0 = (isA ? 0 : A_BYTES) + (wave & 1) * 16384;
  unsigned off[16];
  auto offsets = [&](int tm, int tn) {
    const int row0 = isA ? tm * BM : tn * BN;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int ridx = (wave & 1) * 16 + j;
      const int lchunk = (lane & 7) ^ ((((j & 1) << 2) + (lane >> 4)) & 7);
      const int row = min(row0 + ridx * 8 + (lane >> 3), rmax);
      off[j] = (unsigned)(((long)row * ld + lchunk * 8) * 2);
    }
  };
  auto issue = [&](int kt, int buf, int j) {
    const char* sp = base + (long)kt * (BK * 2);
    const unsigned lds_dst = lds0 + buf * STAGE + dst0 + j * 1024;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off[j]), "s"(sp), "s"(lds_dst) : "memory", "m0");
  };
  auto first_stages = [&]() {
#pragma unroll
    for (int j = 0; j < 16; ++j) issue(0, 0, j);
    if (nk > 1) {
#pragma unroll
      for (int j = 0; j < 16; ++j) issue(1, 1, j);
    }
  };
  const int wm = wave >> 1, wn = wave & 1;
  const int sw = ((lane & 15) >> 1) & 7;
  const unsigned a0 = lds0 + (wm * 128 + (lane & 15)) * 128 + (((lane >> 4)) ^ sw) * 16;
  const unsigned b0 = lds0 + A_BYTES + (wn * 128 + (lane & 15)) * 128 + (((lane >> 4)) ^ sw) * 16;

  int t = blockIdx.x, tm, tn;
  u4_tile(g, t, tm, tn);
  offsets(tm, tn);
  first_stages();

  bf16x8 A0[8], B0[8], A1[8], B1[8];
#define RDQ(dst, addr, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off_))
#define SB __builtin_amdgcn_sched_barrier(0);
#define RD8(X, ad) RDQ(X[0], ad, 0); RDQ(X[1], ad, 2048); RDQ(X[2], ad, 4096); RDQ(X[3], ad, 6144); RDQ(X[4], ad, 8192); RDQ(X[5], ad, 10240); RDQ(X[6], ad, 12288); RDQ(X[7], ad, 14336);
  auto wait16 = [&](bf16x8 (&a)[8], bf16x8 (&b)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                 "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
    __builtin_amdgcn_sched_barrier(0);
  };
#define ZR(mi, ni) asm volatile("v_accvgpr_write_b32 " AS_##mi##_##ni##_0 ", 0\n v_accvgpr_write_b32 " AS_##mi##_##ni##_1 ", 0\n v_accvgpr_write_b32 " AS_##mi##_##ni##_2 ", 0\n v_accvgpr_write_b32 " AS_##mi##_##ni##_3 ", 0" ::: CL_##mi##_##ni);
#define ZROW(mi) ZR(mi, 0) ZR(mi, 1) ZR(mi, 2) ZR(mi, 3) ZR(mi, 4) ZR(mi, 5) ZR(mi, 6) ZR(mi, 7)
#define MFM(Ac, Bc, mi, ni) asm volatile("v_mfma_f32_16x16x32_bf16 " AR_##mi##_##ni ", %0, %1, " AR_##mi##_##ni :: "v"(Bc[ni]), "v"(Ac[mi]) : CL_##mi##_##ni); SB
#define BARX __builtin_amdgcn_s_barrier(); SB
#define RDACC(mi, ni, v) asm volatile("v_accvgpr_read_b32 %0, " AS_##mi##_##ni##_0 "\n v_accvgpr_read_b32 %1, " AS_##mi##_##ni##_1 "\n v_accvgpr_read_b32 %2, " AS_##mi##_##ni##_2 "\n v_accvgpr_read_b32 %3, " AS_##mi##_##ni##_3 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));

  while (true) {
    ZROW(0) ZROW(1) ZROW(2) ZROW(3) ZROW(4) ZROW(5) ZROW(6) ZROW(7)
    // everything this wave has in flight - the two first stages and the previous tile's stores - has landed; the barrier says so for all four waves
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    { RD8(A0, a0) RD8(B0, b0) }
    int kt = 0;
    // steady state: stages kt+1 and kt+2 exist - no conditions inside the 128-MFMA body.  One DMA piece = an s_add on m0 behind one MFMA, the load behind the
    // next: never more than two other instructions between two MFMAs (one wave per SIMD issues one instruction per 4 cycles; a 16-cycle MFMA leaves three slots)
#define M0P(p) if (p == 0) { asm volatile("s_mov_b32 m0, %0" ::"s"(lds0 + (kt & 1) * STAGE + dst0) : "m0"); } else { asm volatile("s_add_u32 m0, m0, 0x400" ::: "m0", "scc"); }
#define GLDS(p) asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(off[p]), "s"(sp2) : "memory")
#define RDN(dst, ad, off_) RDQ(dst, ad, off_)
    // stage kt+1 has landed (this wave's pieces: vmcnt - the n younger pieces of stage kt+2 stay in flight; everybody's: the barrier)
#define WAITY(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory"); __builtin_amdgcn_s_barrier(); SB
    for (; kt + 2 < nk; ++kt) {
      const unsigned so = (kt & 1) * STAGE, sn = so ^ STAGE;
      const unsigned aa1 = a0 ^ (so | 64u), ba1 = b0 ^ (so | 64u), aa0 = a0 ^ sn, ba0 = b0 ^ sn;
      const char* sp2 = base + (long)(kt + 2) * (BK * 2);
      wait16(A0, B0);
#include "gemm_u4_body.inc"
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
#include "gemm_u4_body.inc"
    }
#undef M0P
#undef GLDS
#undef RDN
#undef WAITY
    // every wave is done with both LDS buffers: the next tile's first two stages go out before this tile's stores
    const int cm = tm, cn = tn;
    t += gridDim.x;
    const bool next = t < ntiles;
    __builtin_amdgcn_s_barrier();
    if (next) {
      u4_tile(g, t, tm, tn);
      offsets(tm, tn);
      first_stages();
    }
    // plain epilogue straight from the accumulators: a lane holds 4 consecutive n of one m per fragment (weight fragment as the MFMA's A operand).  The residual
    // of fragment row mi + 1 is requested before row mi is converted and stored: eight 8-byte loads per lane in flight instead of a load -> add -> store chain per
    // fragment (o-projection at M = 8190: 226 -> ~200 us)
    const int n_lane = cn * BN + wn * 128 + (lane >> 4) * 4, m_lane = cm * BM + wm * 128 + (lane & 15);
    uint2 rcur[8], rnxt[8];
    auto load_res = [&](int mi, uint2 (&r)[8]) {
      const int m = m_lane + mi * 16;
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) {
        const int n = n_lane + ni * 16;
        r[ni] = (g.res != nullptr && m < g.M && n < g.N) ? *reinterpret_cast<const uint2*>(g.res + (long)m * g.ldr + n) : make_uint2(0u, 0u);
      }
    };
#define ST(mi, ni)                                                                                                  \
    {                                                                                                               \
      float v[4]; RDACC(mi, ni, v)                                                                                  \
      const int m = m_lane + mi * 16, n = n_lane + ni * 16;                                                         \
      if (m < g.M && n < g.N) {                                                                                     \
        if (g.res != nullptr) {                                                                                     \
          const uint2 rr = rcur[ni];                                                                                \
          v[0] += bf2f((bf16_t)(rr.x & 0xffffu)); v[1] += bf2f((bf16_t)(rr.x >> 16));                               \
          v[2] += bf2f((bf16_t)(rr.y & 0xffffu)); v[3] += bf2f((bf16_t)(rr.y >> 16));                               \
        }                                                                                                           \
        const unsigned lo = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16), hi = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16); \
        *reinterpret_cast<uint2*>(g.C + (long)m * g.ldc + n) = make_uint2(lo, hi);                                  \
      }                                                                                                             \
    }
#define STROW(mi)                                                                                                   \
    if (mi + 1 < 8) load_res(mi + 1, rnxt);                                                                         \
    ST(mi, 0) ST(mi, 1) ST(mi, 2) ST(mi, 3) ST(mi, 4) ST(mi, 5) ST(mi, 6) ST(mi, 7)                                 \
    _Pragma("unroll") for (int q = 0; q < 8; ++q) rcur[q] = rnxt[q];
    // RoPE (HF apply_rotary_pos_emb on q and k, text_modal.py:258-294 via LlamaAttention): a wave's 128 columns are one head; dim d sits in fragment ni = d / 16
    // (d < 64) and its rotate_half partner d + 64 in fragment ni + 4 of the SAME lane - no exchange.  Both are rounded to bf16 first, exactly as the
    // unfused pair lhrs_gemm_bf16_nt + lhrs_rope sees them (bit-identical: tests/test_kernels_gpu.py)
#define ROPE_ST(mi, ni, nj)                                                                                         \
    {                                                                                                               \
      float v1[4], v2[4]; RDACC(mi, ni, v1) RDACC(mi, nj, v2)                                                       \
      const int m = m_lane + mi * 16, n = n_lane + ni * 16;                                                         \
      if (m < g.M) {                                                                                                \
        const int pos = m % g.rope_mod + g.rope_pos0, d = ni * 16 + (lane >> 4) * 4;                                \
        const float4 c4 = *reinterpret_cast<const float4*>(g.rope_cos + (long)pos * 64 + d);                        \
        const float4 s4 = *reinterpret_cast<const float4*>(g.rope_sin + (long)pos * 64 + d);                        \
        const float cv[4] = {c4.x, c4.y, c4.z, c4.w}, sv[4] = {s4.x, s4.y, s4.z, s4.w};                             \
        float o1[4], o2[4];                                                                                         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) rope_pair(bf2f(f2bf(v1[i])), bf2f(f2bf(v2[i])), cv[i], sv[i], o1[i], o2[i]); \
        *reinterpret_cast<uint2*>(g.C + (long)m * g.ldc + n) =                                                      \
            make_uint2((unsigned)f2bf(o1[0]) | ((unsigned)f2bf(o1[1]) << 16), (unsigned)f2bf(o1[2]) | ((unsigned)f2bf(o1[3]) << 16)); \
        *reinterpret_cast<uint2*>(g.C + (long)m * g.ldc + n + 64) =                                                 \
            make_uint2((unsigned)f2bf(o2[0]) | ((unsigned)f2bf(o2[1]) << 16), (unsigned)f2bf(o2[2]) | ((unsigned)f2bf(o2[3]) << 16)); \
      }                                                                                                             \
    }
#define ROPE_ROW(mi) ROPE_ST(mi, 0, 4) ROPE_ST(mi, 1, 5) ROPE_ST(mi, 2, 6) ROPE_ST(mi, 3, 7)
    if (ROPE && cn * BN < g.rope_cols) {   // tile-uniform (rope_cols % 256 == 0)
      ROPE_ROW(0) ROPE_ROW(1) ROPE_ROW(2) ROPE_ROW(3) ROPE_ROW(4) ROPE_ROW(5) ROPE_ROW(6) ROPE_ROW(7)
    } else {
      load_res(0, rcur);
      STROW(0) STROW(1) STROW(2) STROW(3) STROW(4) STROW(5) STROW(6) STROW(7)
    }
    if (!next) break;
  }
}
}  // namespace

static bool u4_addressable(const void* A, int lda, const void* B, int ldb, const void* C, int ldc, int M, int N, int K) {
  return M > 0 && N > 0 && K >= 128 && K % 64 == 0 && N % 4 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && lda >= K && ldb >= K && ldc >= N &&
         ((size_t)A | (size_t)B) % 16 == 0 && (size_t)C % 8 == 0 && (long)M * lda * 2 < (1L << 32) && (long)N * ldb * 2 < (1L << 32);   // 32-bit lane offsets
}
static dim3 u4_grid(const U4Args& g) {
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int tiles = g.tilesM * g.tilesN;
  return dim3(tiles < cus ? tiles : cus);
}

// 0 launched; 1 not this kernel's problem (the caller takes gemm.hip's kernels); -1 error.  Plain epilogue only: bf16 out, optional bf16 residual.
extern "C" int lhrs_gemm_u4_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const void* residual, int ldr,
                               void* stream) {
  if (!u4_addressable(A, lda, B, ldb, C, ldc, M, N, K) || (residual != nullptr && (ldr % 4 != 0 || (size_t)residual % 8 != 0))) return 1;
  U4Args g{(const bf16_t*)A, (const bf16_t*)B, (bf16_t*)C, (const bf16_t*)residual, M, N, K, lda, ldb, ldc, ldr, (M + 255) / 256, (N + 255) / 256,
           nullptr, nullptr, 1, 0, 0};
  hipLaunchKernelGGL(gemm_u4_kernel<false>, u4_grid(g), dim3(256), 0, (hipStream_t)stream, g);
  LHRS_CHECK_LAUNCH("gemm_u4_nt");
  return 0;
}

// q|k|v projection with RoPE in the epilogue (the semantics of lhrs_gemm_rope_fwd without a LoRA pair): columns [0, rope_cols) are heads of 128 rotated with
// the position m % pos_mod + pos0 of their row, the rest stored as computed.  0 launched; 1 not this kernel's problem.
extern "C" int lhrs_gemm_u4_rope(const void* X, int ldx, const void* W, int ldw, void* C, int ldc, int M, int N, int K, const float* cos_t, const float* sin_t,
                                 int pos_mod, int pos0, int rope_cols, void* stream) {
  if (!u4_addressable(X, ldx, W, ldw, C, ldc, M, N, K) || rope_cols % 256 != 0 || rope_cols > N || pos_mod <= 0 || cos_t == nullptr || sin_t == nullptr ||
      ((size_t)cos_t | (size_t)sin_t) % 16 != 0)
    return 1;
  U4Args g{(const bf16_t*)X, (const bf16_t*)W, (bf16_t*)C, nullptr, M, N, K, ldx, ldw, ldc, 0, (M + 255) / 256, (N + 255) / 256,
           cos_t, sin_t, pos_mod, pos0, rope_cols};
  hipLaunchKernelGGL(gemm_u4_kernel<true>, u4_grid(g), dim3(256), 0, (hipStream_t)stream, g);
  LHRS_CHECK_LAUNCH("gemm_u4_rope");
  return 0;
}
