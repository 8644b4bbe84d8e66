// gemm_u4_kernel<EPI, RES>: the bf16 NT product out[M,N] = a[M,K] . b[N,K]^T on a 256x256x64 tile walked by FOUR waves (one per SIMD), 128x128 of the tile per
// wave - HF LlamaDecoderLayer's nn.Linear calls (lhrs/models/text_modal.py:133-151, 258-294), as gemm.hip - with the epilogues of the decoder layer:
//   EPI 0  plain, optional bf16 residual (RES)            o / down projections, the dX products, lm_head
//   EPI 1  SwiGLU forward: gate|up + act = silu(gate) * up  (HF LlamaMLP)        EPI 2  SwiGLU backward: d(gate|up) = swiglu'(gate|up) * d_act
//   EPI 3  RoPE on the q / k heads of the qkv projection (HF apply_rotary_pos_emb, rotate_half convention)
//
// Main loop (round 4; docs/design_notes_r03_r04.md): a 128x128 wave tile reads 256 B of LDS per MFMA instead of the 16-wave kernel's 512 B, four waves meet at a
// barrier instead of sixteen, and with 512 registers per wave all fragments of a 32-k half sit in registers early enough to free the LDS buffer a fifth of the way
// into a stage - the DMA of stage kt+2 then has 0.9-1.7 stages to land.  Accumulators are not C++ variables: every MFMA is an `asm volatile` naming its AGPR tuple
// (gemm_u4_agpr.inc); the 128-MFMA stage body is generated (tools/gen_u4.py): fragment reads of the second 32-k half behind MFMAs 0..15, barrier X behind MFMA 21,
// ONE DMA piece every 6 MFMAs from there (64 pieces of 1 KiB per stage are half a stage of the CU's address path: issued faster they stall the issuing wave),
// counted vmcnt + barrier Y behind MFMA 108, the next stage's first fragments behind MFMAs 109..124.
//
// Round 5: the tiles of a workgroup are ONE stream of stages.
//   * The last two stages of a tile request the first two stages of the NEXT tile (the stage body does not know where a tile ends: only the DMA source changes), and
//     the last stage reads the next tile's first fragments: no cold start between tiles (round 4: two stages requested, vmcnt(0), barrier - ~2 us per tile).
//   * The finished tile is written out INSIDE the next tile's first stage (gemm_u4_flush*.inc): right in front of each first-half MFMA - which takes the constant 0
//     as its C operand, so nothing is zeroed either - the accumulators it overwrites are read into a few VGPRs, and their conversion and stores sit between the
//     following MFMAs.  The stores of a tile overlap the MFMAs of the next one instead of 256 CUs bursting at once with the matrix pipes idle (round 4: ~10 us of a
//     ~100 us tile at K = 4096; 19 us for SwiGLU').  Operands an epilogue reads (residual rows, gate|up, cos / sin) are requested a few units ahead - the first ones in
//     the previous tile's last stage - and waited for with counted vmcnt.
//   * For 16-byte stores straight from registers the weight rows of a tile sit in LDS in a permuted order (the DMA computes a source address per lane anyway): lane
//     group fg of fragment ni holds columns (ni / 4) * 64 + ((ni / 2) & 1) * 32 + fg * 8 + (ni & 1) * 4 + r of the wave's 128, so two fragments give a lane 8
//     consecutive columns (a unit: one global_store_dwordx4, 64 B contiguous per row and instruction) and fragment ni + 4 holds column c + 64: the rotate_half
//     partner of RoPE, and - with the weight image [64 gate | 64 up] per wave - the up column of a gate column.  All epilogues are lane-local.
//   * Counted waits: every vmcnt in the generated bodies counts the vector-memory operations issued behind the one waited for.  That is sound because a wave's
//     loads, stores and LDS-DMA pieces share one counter and retire in issue order (MI355X_MICROARCH.md: `vmcnt(N)` waits for the outstanding - N oldest) and because
//     every memory operation of a flush body is unconditional: only INTERIOR tiles are written out that way; edge tiles (ragged M / N) and a workgroup's last tile
//     take the exposed epilogue below, whose loads and stores are ordinary C++ (the compiler's own waits are conservative under the same in-order rule).
//     tests/test_kernels_gpu.py: 200-launch soak per shape under memory load, every result identical to the first and to the 16-wave kernel (vmcnt(0) only).
// Same k order and fp32 accumulation as gemm_nt_256s_kernel: bit-identical results without a residual; a residual is added to the fp32 sum before the one rounding
// (gemm_nt_256s_kernel's staged epilogue rounds the sum first).  SwiGLU / RoPE epilogues round gate / up / d_act / q / k to bf16 exactly where the unfused kernel
// pairs store them: bit-identical to those (tests/test_kernels_gpu.py).
#include "common.h"
#include "gemm_u4_agpr.inc"

namespace {
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
struct U4Args {
  const bf16_t* A; const bf16_t* B; bf16_t* C; const bf16_t* res;
  int M, N, K, lda, ldb, ldc, ldr, tilesM, tilesN;
  const float* rope_cos; const float* rope_sin; int rope_mod, rope_pos0, rope_cols;   // EPI 3: columns [0, rope_cols) are heads of 128, tables [pos][64] f32
  int ff; const bf16_t* aux; bf16_t* aux_out; int ld_aux;                             // EPI 1: aux_out = act [M, ff] (ld_aux); EPI 2: aux = gate|up [M, 2 ff] (ld_aux)
  // optional second operand pair, reduced in the same k-loop behind the first: out = A.B^T + A2.B2^T (the fused LoRA update: A2 = s * x * A_lora^T [M, K2], B2 = B_lora
  // [N, K2]; peft lora.Linear, lhrs/models/text_modal.py:133-151).  K2 = 0: none.  Same rows, same tile image: only the DMA source of the last K2 / 64 stages changes
  const bf16_t* A2; const bf16_t* B2; int lda2, ldb2, K2;
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

// ---- the arithmetic of a unit (8 consecutive columns of one row in a lane), shared by the in-stream flush and the exposed epilogue ---------------------------
typedef __attribute__((ext_vector_type(2))) float u4_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 u4_bf16x2;
__device__ __forceinline__ unsigned u4_pk(float lo, float hi) {   // ONE v_cvt_pk_bf16_f32 (RNE, the instruction f2bf lowers to): same bits as pack2bf, a quarter of its instructions
  return __builtin_bit_cast(unsigned, __builtin_convertvector((u4_f32x2{lo, hi}), u4_bf16x2));
}
__device__ __forceinline__ u32x4 u4_pack(const float* v) {
  return u32x4{u4_pk(v[0], v[1]), u4_pk(v[2], v[3]), u4_pk(v[4], v[5]), u4_pk(v[6], v[7])};
}
__device__ __forceinline__ void u4_unpack(const u32x4& u, float* v) {
  v[0] = bflo(u.x); v[1] = bfhi(u.x); v[2] = bflo(u.y); v[3] = bfhi(u.y); v[4] = bflo(u.z); v[5] = bfhi(u.z); v[6] = bflo(u.w); v[7] = bfhi(u.w);
}
__device__ __forceinline__ u32x4 u4_plain_res(const float* t, u32x4 r) {   // fp32 sum + residual -> one rounding
  float v[8], rv[8];
  u4_unpack(r, rv);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = t[i] + rv[i];
  return u4_pack(v);
}
// SwiGLU forward: gate / up rounded to bf16 as stored, act = silu(gate) * up on the rounded values (lhrs_swiglu_fwd, elementwise.hip)
__device__ __forceinline__ void u4_swiglu_fwd(const float* tg, const float* tu, u32x4& gq, u32x4& uq, u32x4& aq) {
  gq = u4_pack(tg); uq = u4_pack(tu);
  float gv[8], uv[8], a[8];
  u4_unpack(gq, gv); u4_unpack(uq, uv);
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = silu(gv[i]) * uv[i];
  aq = u4_pack(a);
}
// SwiGLU backward: d_act rounded to bf16 as the unfused product stores it, then lhrs_swiglu_bwd's expressions (elementwise.hip)
__device__ __forceinline__ void u4_swiglu_bwd(const float* t, const u32x4& gq, const u32x4& uq, u32x4& dgq, u32x4& duq) {
  const u32x4 dq = u4_pack(t);
  float d[8], gg[8], uu[8], dg[8], du[8];
  u4_unpack(dq, d); u4_unpack(gq, gg); u4_unpack(uq, uu);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float sg = sigmoid_f(gg[e]);
    du[e] = d[e] * gg[e] * sg;
    dg[e] = d[e] * uu[e] * sg * (1.f + gg[e] * (1.f - sg));
  }
  dgq = u4_pack(dg); duq = u4_pack(du);
}
// RoPE: x1 = dims d .. d+7, x2 = dims d+64 .. d+71 of one head, rounded to bf16 first (as lhrs_gemm_bf16_nt + lhrs_rope see them); cs = cos[d..d+7], sn = sin[d..d+7]
__device__ __forceinline__ void u4_rope(const float* t1, const float* t2, const u32x4& c0, const u32x4& c1, const u32x4& s0, const u32x4& s1, u32x4& o1q, u32x4& o2q) {
  const u32x4 x1q = u4_pack(t1), x2q = u4_pack(t2);
  float x1[8], x2[8], o1[8], o2[8];
  u4_unpack(x1q, x1); u4_unpack(x2q, x2);
  const float cs[8] = {__uint_as_float(c0.x), __uint_as_float(c0.y), __uint_as_float(c0.z), __uint_as_float(c0.w),
                       __uint_as_float(c1.x), __uint_as_float(c1.y), __uint_as_float(c1.z), __uint_as_float(c1.w)};
  const float sn[8] = {__uint_as_float(s0.x), __uint_as_float(s0.y), __uint_as_float(s0.z), __uint_as_float(s0.w),
                       __uint_as_float(s1.x), __uint_as_float(s1.y), __uint_as_float(s1.z), __uint_as_float(s1.w)};
#pragma unroll
  for (int i = 0; i < 8; ++i) rope_pair(x1[i], x2[i], cs[i], sn[i], o1[i], o2[i]);
  o1q = u4_pack(o1); o2q = u4_pack(o2);
}

constexpr int u4_units_paired(int epi) { return epi == 1 || epi == 3; }
// prefetch depth (units) of the operands a flush reads from memory; must match the generator arguments in the Makefile
constexpr int U4_DEPTH_RES = 8, U4_DEPTH_SWB = 6, U4_DEPTH_ROPE = 3;

template <int EPI, bool RES>
__global__ __launch_bounds__(256, 1) void gemm_u4_kernel(U4Args g) {
  static_assert(!RES || EPI == 0, "a residual only with the plain epilogue");
  constexpr int BM = 256, BN = 256, BK = 64, A_BYTES = BM * BK * 2, STAGE = A_BYTES + BN * BK * 2;
  constexpr bool PAIRED = EPI == 1 || EPI == 3;
  constexpr int NLOAD = RES ? 1 : EPI == 2 ? 2 : EPI == 3 ? 4 : 0;                     // 16-byte loads per unit
  constexpr int DEPTH = RES ? U4_DEPTH_RES : EPI == 2 ? U4_DEPTH_SWB : EPI == 3 ? U4_DEPTH_ROPE : 1;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = g.tilesM * g.tilesN;
  // waves 0,1 bring the activation rows of a stage, waves 2,3 the weight rows: 16 pieces of 8 rows x 128 B each, chunk-swizzled as gemm.hip's image
  const bool isA = wave < 2;
  const char* base = reinterpret_cast<const char*>(isA ? g.A : g.B);
  const char* base_p2 = reinterpret_cast<const char*>(isA ? g.A2 : g.B2);
  const long ld_p1 = isA ? g.lda : g.ldb, ld_p2 = isA ? g.lda2 : g.ldb2;
  const int rmax = (isA ? g.M : (EPI == 1 ? 2 * g.ff : g.N)) - 1;
  const int nk1 = g.K / BK;                                                            // >= 4 (host)
  const int nk = nk1 + g.K2 / BK;                                                      // stages per tile: the first pair's, then the second pair's
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  const int dst0 = (isA ? 0 : A_BYTES) + (wave & 1) * 16384;
  unsigned off[16];
  auto offsets = [&](int tm, int tn, bool pair2 = false) {
    const long ld = pair2 ? ld_p2 : ld_p1;
    // everything derived from the lane id comes from an opaque copy made HERE: otherwise those values are invariant across the three call sites (launch start, second
    // pair, next tile), get hoisted in front of the tile loop and sit in registers all through the main loop (the write-out variants have none to spare)
    int le = lane;
    asm volatile("" : "+v"(le));
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int lchunk = (le & 7) ^ ((((j & 1) << 2) + (le >> 4)) & 7);
      int row;
      if (isA) {
        row = min(tm * BM + ((wave & 1) * 16 + j) * 8 + (le >> 3), rmax);
      } else {
        // LDS row wn * 128 + ni * 16 + a of the weight tile holds column cw of the wave's 128 (see the header): ni = j / 2, a = (j & 1) * 8 + lane / 8
        const int ni = j >> 1, a = ((j & 1) << 3) + (le >> 3);
        const int cw = (ni >> 2) * 64 + ((ni >> 1) & 1) * 32 + (a >> 2) * 8 + (ni & 1) * 4 + (a & 3);
        if (EPI == 1) row = (ni >> 2) * g.ff + tn * 128 + (wave & 1) * 64 + (cw & 63);   // [gate; up] weight: 64 gate + 64 up columns per wave
        else row = min(tn * BN + (wave & 1) * 128 + cw, rmax);
      }
      off[j] = (unsigned)(((long)row * ld + lchunk * 8) * 2);
    }
  };
  auto issue = [&](int kt, int buf, int j) {
    const char* sp = base + (long)kt * (BK * 2);
    const unsigned lds_dst = lds0 + buf * STAGE + dst0 + j * 1024;
    asm volatile("s_mov_b64 s[100:101], %1\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, s[100:101]" ::"v"(off[j]), "s"(sp), "s"(lds_dst) : "memory", "m0", "s100", "s101");   // the pointer through an SALU move: see U4_SPTR below
  };
  const int wm = wave >> 1, wn = wave & 1;
  const int sw = ((lane & 15) >> 1) & 7;
  const unsigned a0 = lds0 + (wm * 128 + (lane & 15)) * 128 + (((lane >> 4)) ^ sw) * 16;
  const unsigned b0 = lds0 + A_BYTES + (wn * 128 + (lane & 15)) * 128 + (((lane >> 4)) ^ sw) * 16;

  int t = blockIdx.x, tm, tn;
  u4_tile(g, t, tm, tn);
  offsets(tm, tn);
#pragma unroll
  for (int j = 0; j < 16; ++j) issue(0, 0, j);
#pragma unroll
  for (int j = 0; j < 16; ++j) issue(1, 1, j);

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
#define ZALL ZROW(0) ZROW(1) ZROW(2) ZROW(3) ZROW(4) ZROW(5) ZROW(6) ZROW(7)
#define MFM(Ac, Bc, mi, ni) asm volatile("v_mfma_f32_16x16x32_bf16 " AR_##mi##_##ni ", %0, %1, " AR_##mi##_##ni :: "v"(Bc[ni]), "v"(Ac[mi]) : CL_##mi##_##ni); SB
#define MFM0(Ac, Bc, mi, ni) asm volatile("v_mfma_f32_16x16x32_bf16 " AR_##mi##_##ni ", %0, %1, 0" :: "v"(Bc[ni]), "v"(Ac[mi]) : CL_##mi##_##ni); SB
#define BARX __builtin_amdgcn_s_barrier(); SB
#define RDACC4(mi, ni, v, o) asm volatile("v_accvgpr_read_b32 %0, " AS_##mi##_##ni##_0 "\n v_accvgpr_read_b32 %1, " AS_##mi##_##ni##_1 "\n v_accvgpr_read_b32 %2, " AS_##mi##_##ni##_2 "\n v_accvgpr_read_b32 %3, " AS_##mi##_##ni##_3 : "=v"(v[o]), "=v"(v[o + 1]), "=v"(v[o + 2]), "=v"(v[o + 3]));
// the s_nop: a VALU write to the data registers of a > 64-bit store needs two wait states behind it on gfx950 (the compiler's hazard recognizer cannot see into
// asm statements, and the next thing the flush bodies do is read accumulators into those very registers)
// Every scalar pointer of a VMEM instruction in these asm statements goes through an SALU move into s[100:101] first: "VALU writes SGPR -> VMEM reads that SGPR"
// needs 5 wait states on gfx9, the hazard recognizer pads it for the compiler's own instructions only, and the register allocator is free to reload a spilled pointer
// (v_readlane_b32) or make one uniform (v_readfirstlane_b32) right in front of an asm statement - the load / store then goes out with the OLD register contents
// (round 5: memory access faults of the SwiGLU' variant once its pointers spilled).  SALU -> VMEM has no such hazard.  tools/check_asm_sgpr_hazard.py scans the ISA.
#define U4_SPTR "s100", "s101"
#define GST(voff_, data_, sptr_, imm_) asm volatile("s_mov_b64 s[100:101], %2\n\tglobal_store_dwordx4 %0, %1, s[100:101] offset:%3\n\ts_nop 1" ::"v"(voff_), "v"(data_), "s"(sptr_), "n"(imm_) : "memory", U4_SPTR)
#define GLD(dst_, voff_, sptr_, imm_) asm volatile("s_mov_b64 s[100:101], %2\n\tglobal_load_dwordx4 %0, %1, s[100:101] offset:%3" : "=v"(dst_) : "v"(voff_), "s"(sptr_), "n"(imm_) : "memory", U4_SPTR)

  // ---- per-lane constants of the write-out: byte offset of (row wm * 128 + fr, column fg * 8 of the wave's columns) in C / aux; rows advance by 16 per mi (scalar) ----
  const int fr = lane & 15, fg = lane >> 4;
  const int wcol = (EPI == 1 ? wn * 64 : wn * 128) + fg * 8;                           // first column of this lane inside the tile (EPI 1: inside the tile's 128 ff columns)
  const unsigned voff_c = (unsigned)(((long)(wm * 128 + fr) * g.ldc + wcol) * 2);
  const unsigned voff_x = RES ? (unsigned)(((long)(wm * 128 + fr) * g.ldr + wcol) * 2)
                              : (EPI == 1 || EPI == 2) ? (unsigned)(((long)(wm * 128 + fr) * g.ld_aux + wcol) * 2) : 0u;
  const long row16_c = (long)16 * g.ldc * 2, row16_x = RES ? (long)16 * g.ldr * 2 : (long)16 * g.ld_aux * 2;
  const int tile_cols = EPI == 1 ? 128 : 256;                                          // output columns per tile in the first output
  // the finished tile that is being written out (wave-uniform): origin pointers of its outputs / memory operands
  int fm = 0, fn = 0;
  const char* f_c = nullptr; const char* f_c2 = nullptr; const char* f_x = nullptr; const char* f_x2 = nullptr;
  bool f_rope = false;
  unsigned voff_cs[8];                                                                 // EPI 3: byte offset of (pos(row mi), dim fg * 8) in the cos / sin tables
  auto flush_origin = [&](int tm_, int tn_) {
    fm = tm_; fn = tn_;
    f_c = reinterpret_cast<const char*>(g.C) + ((long)tm_ * BM * g.ldc + (long)tn_ * tile_cols) * 2;
    if (EPI == 1) { f_c2 = f_c + (long)g.ff * 2; f_x = reinterpret_cast<const char*>(g.aux_out) + ((long)tm_ * BM * g.ld_aux + (long)tn_ * 128) * 2; }
    if (EPI == 2) { f_c2 = f_c + (long)g.ff * 2; f_x = reinterpret_cast<const char*>(g.aux) + ((long)tm_ * BM * g.ld_aux + (long)tn_ * 256) * 2; f_x2 = f_x + (long)g.ff * 2; }
    if (RES) f_x = reinterpret_cast<const char*>(g.res) + ((long)tm_ * BM * g.ldr + (long)tn_ * 256) * 2;
    if (EPI == 3) {
      f_rope = tn_ * BN < g.rope_cols;
      int pos = (tm_ * BM + wm * 128 + fr) % g.rope_mod;                               // rope_mod >= 16 (host): one conditional subtraction per 16 rows
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        voff_cs[mi] = (unsigned)(((pos + g.rope_pos0) * 64 + fg * 8) * 4);
        pos += 16; if (pos >= g.rope_mod) pos -= g.rope_mod;
      }
    }
  };
  float T[2][PAIRED ? 16 : 8];
  u32x4 LB[DEPTH][NLOAD > 0 ? NLOAD : 1];

  // ---- the write-out of one unit, as macros over literal (mi, sub): used by the generated flush bodies (unconditional, asm memory operations) ----
#define ROWP(p_, mi) ((p_) + (long)(mi) * row16_c)
#define ROWX(p_, mi) ((p_) + (long)(mi) * row16_x)
#define IMM1(sub) (((sub) >> 1) * 128 + ((sub) & 1) * 64)     /* plain units: sub = p = ni0 / 2: columns (p / 2) * 64 + (p & 1) * 32 */
#define IMM2(sub) ((sub) * 64)                                /* paired units: sub = q: columns q * 32 (and + 64) */
#define FL_ACC1(u, s, mi, n0, n1) RDACC4(mi, n0, T[s], 0) RDACC4(mi, n1, T[s], 4)
#define FL_ACC2(u, s, mi, n0, n1, n2, n3) RDACC4(mi, n0, T[s], 0) RDACC4(mi, n1, T[s], 4) RDACC4(mi, n2, T[s], 8) RDACC4(mi, n3, T[s], 12)
#define FL_WAITN(lb, n)                                                                                                      \
  if constexpr (NLOAD == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(LB[lb][0]) : "n"(n));                                 \
  else if constexpr (NLOAD == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(LB[lb][0]), "+v"(LB[lb][1]) : "n"(n));         \
  else if constexpr (NLOAD == 4) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(LB[lb][0]), "+v"(LB[lb][1]), "+v"(LB[lb][2]), "+v"(LB[lb][3]) : "n"(n));
#define WAITY(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory"); __builtin_amdgcn_s_barrier(); SB

  ZALL
  // the two first stages have landed; the barrier says so for all four waves
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  { RD8(A0, a0) RD8(B0, b0) }
  bool pend = false;     // the previous tile of this workgroup still sits in the accumulators (it is written out inside this tile's first stage)
  int gs = 0;            // stages this workgroup has gone through: stage buffer = gs & 1

  while (true) {
    const int tnext = t + (int)gridDim.x;
    const bool has_next = tnext < ntiles;
    int ntm = 0, ntn = 0;
    if (has_next) u4_tile(g, tnext, ntm, ntn);
    const bool interior = (tm + 1) * BM <= g.M && (EPI == 1 || (tn + 1) * BN <= g.N);
    int kt = 0;
    // one DMA piece = an s_add on m0 behind one MFMA, the load behind the next: never more than two other instructions between two MFMAs
#define M0P(p) if (p == 0) { asm volatile("s_mov_b32 m0, %0" ::"s"(lds0 + so + dst0) : "m0"); } else { asm volatile("s_add_u32 m0, m0, 0x400" ::: "m0", "scc"); }
#define GLDS(p) asm volatile("s_mov_b64 s[100:101], %1\n\tglobal_load_lds_dwordx4 %0, s[100:101]" ::"v"(off[p]), "s"(sp2) : "memory", U4_SPTR)
#define RDN(dst, ad, off_) RDQ(dst, ad, off_)
#define STAGE_ADDRS                                                                                            \
      const unsigned so = (gs & 1) * STAGE, sn = so ^ STAGE;                                                   \
      const unsigned aa1 = a0 ^ (so | 64u), ba1 = b0 ^ (so | 64u), aa0 = a0 ^ sn, ba0 = b0 ^ sn;
    if (pend) {
      // stage 0 of this tile + the write-out of the previous one
      STAGE_ADDRS
      const char* sp2 = base + (long)2 * (BK * 2);
      wait16(A0, B0);
      if constexpr (EPI == 0 && !RES) {
#define FL_ACC(u, s, mi, n0, n1) FL_ACC1(u, s, mi, n0, n1)
#define FL_OP(u, s, lb, mi, sub) { const u32x4 d_ = u4_pack(T[s]); GST(voff_c, d_, ROWP(f_c, mi), IMM1(sub)); }
#include "gemm_u4_flush_p.inc"
#undef FL_ACC
#undef FL_OP
      } else if constexpr (EPI == 0 && RES) {
#define FL_ACC(u, s, mi, n0, n1) FL_ACC1(u, s, mi, n0, n1)
#define FL_WAIT(u, lb, n) FL_WAITN(lb, n)
#define FL_LOAD(u, lb, mi, sub) GLD(LB[lb][0], voff_x, ROWX(f_x, mi), IMM1(sub))
#define FL_OP(u, s, lb, mi, sub) { const u32x4 d_ = u4_plain_res(T[s], LB[lb][0]); GST(voff_c, d_, ROWP(f_c, mi), IMM1(sub)); }
#include "gemm_u4_flush_r.inc"
#undef FL_ACC
#undef FL_WAIT
#undef FL_LOAD
#undef FL_OP
      } else if constexpr (EPI == 1) {
#define FL_ACC(u, s, mi, n0, n1, n2, n3) FL_ACC2(u, s, mi, n0, n1, n2, n3)
#define FL_OP(u, s, lb, mi, sub) { u32x4 gq_, uq_, aq_; u4_swiglu_fwd(T[s], T[s] + 8, gq_, uq_, aq_);                                  \
    GST(voff_c, gq_, ROWP(f_c, mi), IMM2(sub)); GST(voff_c, uq_, ROWP(f_c2, mi), IMM2(sub)); GST(voff_x, aq_, ROWX(f_x, mi), IMM2(sub)); }
#include "gemm_u4_flush_f.inc"
#undef FL_ACC
#undef FL_OP
      } else if constexpr (EPI == 2) {
#define FL_ACC(u, s, mi, n0, n1) FL_ACC1(u, s, mi, n0, n1)
#define FL_WAIT(u, lb, n) FL_WAITN(lb, n)
#define FL_LOAD(u, lb, mi, sub) GLD(LB[lb][0], voff_x, ROWX(f_x, mi), IMM1(sub)); GLD(LB[lb][1], voff_x, ROWX(f_x2, mi), IMM1(sub))
#define FL_OP(u, s, lb, mi, sub) { u32x4 dg_, du_; u4_swiglu_bwd(T[s], LB[lb][0], LB[lb][1], dg_, du_);                               \
    GST(voff_c, dg_, ROWP(f_c, mi), IMM1(sub)); GST(voff_c, du_, ROWP(f_c2, mi), IMM1(sub)); }
#include "gemm_u4_flush_b.inc"
#undef FL_ACC
#undef FL_WAIT
#undef FL_LOAD
#undef FL_OP
      } else {
#define FL_ACC(u, s, mi, n0, n1, n2, n3) FL_ACC2(u, s, mi, n0, n1, n2, n3)
#define FL_WAIT(u, lb, n) FL_WAITN(lb, n)
#define FL_LOAD(u, lb, mi, sub) GLD(LB[lb][0], voff_cs[mi], g.rope_cos, (sub) * 128); GLD(LB[lb][1], voff_cs[mi], g.rope_cos, (sub) * 128 + 16);   \
    GLD(LB[lb][2], voff_cs[mi], g.rope_sin, (sub) * 128); GLD(LB[lb][3], voff_cs[mi], g.rope_sin, (sub) * 128 + 16)
#define FL_OP(u, s, lb, mi, sub) { u32x4 o1_, o2_;                                                                                    \
    if (f_rope) u4_rope(T[s], T[s] + 8, LB[lb][0], LB[lb][1], LB[lb][2], LB[lb][3], o1_, o2_);                                        \
    else { o1_ = u4_pack(T[s]); o2_ = u4_pack(T[s] + 8); }                                                                            \
    GST(voff_c, o1_, ROWP(f_c, mi), IMM2(sub)); GST(voff_c, o2_, ROWP(f_c, mi), IMM2(sub) + 128); }
#include "gemm_u4_flush_o.inc"
#undef FL_ACC
#undef FL_WAIT
#undef FL_LOAD
#undef FL_OP
      }
      pend = false;
      ++gs; kt = 1;
    }
    // steady state: stage kt requests stage kt + 2 - of this tile, or (its last two stages) stages 0 and 1 of the next tile
    const int kend = has_next ? nk - ((NLOAD > 0 && interior) ? 1 : 0) : nk - 2;
    for (; kt < kend; ++kt, ++gs) {
      // the stage requested here is stage kt + 2 of the stream: of this tile's first pair, of its second pair (from stage nk1 on), or stage 0 / 1 of the next tile -
      // the lane offsets follow the (tile, pair) they address; the rows they replace are not needed any more (their last stage is in flight)
      const int t2 = kt + 2;
      if (t2 == nk1 && nk1 < nk) offsets(tm, tn, true);
      if (t2 == nk) offsets(ntm, ntn);
      STAGE_ADDRS
      const char* sp2 = t2 < nk1 ? base + (long)t2 * (BK * 2) : t2 < nk ? base_p2 + (long)(t2 - nk1) * (BK * 2) : base + (long)(t2 - nk) * (BK * 2);
      wait16(A0, B0);
#include "gemm_u4_body.inc"
    }
    if (has_next && NLOAD > 0 && interior) {
      // last stage of a tile whose write-out reads memory operands: the first DEPTH units' loads go out here
      flush_origin(tm, tn);
      STAGE_ADDRS
      const char* sp2 = base + (long)1 * (BK * 2);
      wait16(A0, B0);
      if constexpr (RES) {
#define FL_LOAD(u, lb, mi, sub) GLD(LB[lb][0], voff_x, ROWX(f_x, mi), IMM1(sub))
#include "gemm_u4_last_r.inc"
#undef FL_LOAD
      } else if constexpr (EPI == 2) {
#define FL_LOAD(u, lb, mi, sub) GLD(LB[lb][0], voff_x, ROWX(f_x, mi), IMM1(sub)); GLD(LB[lb][1], voff_x, ROWX(f_x2, mi), IMM1(sub))
#include "gemm_u4_last_b.inc"
#undef FL_LOAD
      } else if constexpr (EPI == 3) {
#define FL_LOAD(u, lb, mi, sub) GLD(LB[lb][0], voff_cs[mi], g.rope_cos, (sub) * 128); GLD(LB[lb][1], voff_cs[mi], g.rope_cos, (sub) * 128 + 16);   \
    GLD(LB[lb][2], voff_cs[mi], g.rope_sin, (sub) * 128); GLD(LB[lb][3], voff_cs[mi], g.rope_sin, (sub) * 128 + 16)
#include "gemm_u4_last_o.inc"
#undef FL_LOAD
      }
      ++gs; ++kt;
    }
#undef M0P
#undef GLDS
#undef RDN
#undef WAITY
    if (!has_next) {
      // the last two stages of the launch: nothing left to request; the last one has nothing to read ahead
#define M0P(p)
#define GLDS(p)
#define RDN(dst, ad, off_) if (more) { RDQ(dst, ad, off_); }
#define WAITY(n) if (more) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } SB
      for (; kt < nk; ++kt, ++gs) {
        STAGE_ADDRS
        const bool more = kt + 1 < nk;
        wait16(A0, B0);
#include "gemm_u4_body.inc"
      }
#undef M0P
#undef GLDS
#undef RDN
#undef WAITY
#define WAITY(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory"); __builtin_amdgcn_s_barrier(); SB
    }
    if (has_next && interior) {
      if (NLOAD == 0) flush_origin(tm, tn);
      pend = true;                                              // written out inside the next tile's first stage
    } else {
      // exposed epilogue: edge tiles (ragged M / N) and the workgroup's last tile.  Ordinary C++ loads and stores with bounds checks.
      const int m_lane = tm * BM + wm * 128 + fr;
      float E[PAIRED ? 16 : 8];
      if constexpr (!PAIRED) {
        const int n_lane = tn * BN + wn * 128 + fg * 8;
#define EX1(mi, p, n0, n1)                                                                                                                     \
        {                                                                                                                              \
          RDACC4(mi, n0, E, 0) RDACC4(mi, n1, E, 4)                                                                                    \
          const int m = m_lane + mi * 16, n = n_lane + (p >> 1) * 64 + (p & 1) * 32;                                                   \
          if (m < g.M && n < g.N) {                                                                                                    \
            if constexpr (EPI == 2) {                                                                                                  \
              const u32x4 gq_ = *reinterpret_cast<const u32x4*>(g.aux + (long)m * g.ld_aux + n);                                       \
              const u32x4 uq_ = *reinterpret_cast<const u32x4*>(g.aux + (long)m * g.ld_aux + g.ff + n);                                \
              u32x4 dg_, du_; u4_swiglu_bwd(E, gq_, uq_, dg_, du_);                                                                    \
              *reinterpret_cast<u32x4*>(g.C + (long)m * g.ldc + n) = dg_;                                                              \
              *reinterpret_cast<u32x4*>(g.C + (long)m * g.ldc + g.ff + n) = du_;                                                       \
            } else if (RES) {                                                                                                          \
              const u32x4 r_ = *reinterpret_cast<const u32x4*>(g.res + (long)m * g.ldr + n);                                           \
              *reinterpret_cast<u32x4*>(g.C + (long)m * g.ldc + n) = u4_plain_res(E, r_);                                                \
            } else {                                                                                                                   \
              *reinterpret_cast<u32x4*>(g.C + (long)m * g.ldc + n) = u4_pack(E);                                                      \
            }                                                                                                                          \
          }                                                                                                                            \
        }
#define EXROW1(mi) EX1(mi, 0, 0, 1) EX1(mi, 1, 2, 3) EX1(mi, 2, 4, 5) EX1(mi, 3, 6, 7)
        EXROW1(0) EXROW1(1) EXROW1(2) EXROW1(3) EXROW1(4) EXROW1(5) EXROW1(6) EXROW1(7)
#undef EXROW1
#undef EX1
      } else {
        const int n_lane = (EPI == 1 ? tn * 128 + wn * 64 : tn * BN + wn * 128) + fg * 8;
        const bool rope_tile = EPI == 3 && tn * BN < g.rope_cols;
#define EX2(mi, q, n0, n1, n2, n3)                                                                                                                     \
        {                                                                                                                              \
          RDACC4(mi, n0, E, 0) RDACC4(mi, n1, E, 4) RDACC4(mi, n2, E, 8) RDACC4(mi, n3, E, 12)                                         \
          const int m = m_lane + mi * 16, n = n_lane + q * 32;                                                                         \
          if (m < g.M) {                                                                                                               \
            if constexpr (EPI == 1) {                                                                                                  \
              u32x4 gq_, uq_, aq_; u4_swiglu_fwd(E, E + 8, gq_, uq_, aq_);                                                             \
              *reinterpret_cast<u32x4*>(g.C + (long)m * g.ldc + n) = gq_;                                                              \
              *reinterpret_cast<u32x4*>(g.C + (long)m * g.ldc + g.ff + n) = uq_;                                                       \
              *reinterpret_cast<u32x4*>(g.aux_out + (long)m * g.ld_aux + n) = aq_;                                                     \
            } else {                                                                                                                   \
              u32x4 o1_, o2_;                                                                                                          \
              if (rope_tile) {                                                                                                         \
                const long cs_ = (long)(m % g.rope_mod + g.rope_pos0) * 64 + q * 32 + fg * 8;                                          \
                const u32x4 c0_ = *reinterpret_cast<const u32x4*>(g.rope_cos + cs_), c1_ = *reinterpret_cast<const u32x4*>(g.rope_cos + cs_ + 4);   \
                const u32x4 s0_ = *reinterpret_cast<const u32x4*>(g.rope_sin + cs_), s1_ = *reinterpret_cast<const u32x4*>(g.rope_sin + cs_ + 4);   \
                u4_rope(E, E + 8, c0_, c1_, s0_, s1_, o1_, o2_);                                                                       \
              } else { o1_ = u4_pack(E); o2_ = u4_pack(E + 8); }                                                                       \
              if (n < g.N) *reinterpret_cast<u32x4*>(g.C + (long)m * g.ldc + n) = o1_;                                                 \
              if (n + 64 < g.N) *reinterpret_cast<u32x4*>(g.C + (long)m * g.ldc + n + 64) = o2_;                                       \
            }                                                                                                                          \
          }                                                                                                                            \
        }
#define EXROW2(mi) EX2(mi, 0, 0, 1, 4, 5) EX2(mi, 1, 2, 3, 6, 7)
        EXROW2(0) EXROW2(1) EXROW2(2) EXROW2(3) EXROW2(4) EXROW2(5) EXROW2(6) EXROW2(7)
#undef EXROW2
#undef EX2
      }
      if (!has_next) break;
      ZALL
    }
    t = tnext; tm = ntm; tn = ntn;
  }
}
}  // namespace

static bool u4_addressable(const void* A, int lda, const void* B, int ldb, const void* C, int ldc, int M, int N, int K) {
  return M > 0 && N > 0 && K >= 256 && K % 64 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && lda >= K && ldb >= K && ldc >= N &&
         ((size_t)A | (size_t)B | (size_t)C) % 16 == 0 && (long)M * lda * 2 < (1L << 32) && (long)N * ldb * 2 < (1L << 32) &&      // 32-bit lane offsets of the operands
         (long)144 * ldc * 2 < (1L << 32);                                                                                           // ... and of a tile's output rows
}
// the optional second operand pair (fused LoRA update): rows of A2 / B2 as the rows of A / B; K2 a multiple of 64.  false: not addressable by this kernel
struct U4Pair { const void* A2; int lda2; const void* B2; int ldb2; int K2; };
static bool u4_pair(U4Args& g, const U4Pair* p, int rowsB) {
  if (p == nullptr || p->K2 == 0) return true;
  if (p->K2 < 0 || p->K2 % 64 != 0 || p->A2 == nullptr || p->B2 == nullptr || p->lda2 % 8 != 0 || p->ldb2 % 8 != 0 || p->lda2 < p->K2 || p->ldb2 < p->K2 ||
      ((size_t)p->A2 | (size_t)p->B2) % 16 != 0 || (long)g.M * p->lda2 * 2 >= (1L << 32) || (long)rowsB * p->ldb2 * 2 >= (1L << 32))
    return false;
  g.A2 = (const bf16_t*)p->A2; g.B2 = (const bf16_t*)p->B2; g.lda2 = p->lda2; g.ldb2 = p->ldb2; g.K2 = p->K2;
  return true;
}
static dim3 u4_grid(const U4Args& g) {
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int tiles = g.tilesM * g.tilesN;
  return dim3(tiles < cus ? tiles : cus);
}

// 0 launched; 1 not this kernel's problem (the caller takes gemm.hip's kernels); -1 error.  Plain epilogue: bf16 out, optional bf16 residual.
static int u4_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const void* residual, int ldr, const U4Pair* pair,
                 void* stream) {
  if (!u4_addressable(A, lda, B, ldb, C, ldc, M, N, K) || (residual != nullptr && (ldr % 8 != 0 || ldr < N || (size_t)residual % 16 != 0 || (long)144 * ldr * 2 >= (1L << 32))))
    return 1;
  U4Args g{};
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = (bf16_t*)C; g.res = (const bf16_t*)residual;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr; g.tilesM = (M + 255) / 256; g.tilesN = (N + 255) / 256; g.rope_mod = 1;
  if (!u4_pair(g, pair, N)) return 1;
  if (residual != nullptr) hipLaunchKernelGGL((gemm_u4_kernel<0, true>), u4_grid(g), dim3(256), 0, (hipStream_t)stream, g);
  else hipLaunchKernelGGL((gemm_u4_kernel<0, false>), u4_grid(g), dim3(256), 0, (hipStream_t)stream, g);
  LHRS_CHECK_LAUNCH("gemm_u4_nt");
  return 0;
}
extern "C" int lhrs_gemm_u4_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const void* residual, int ldr,
                               void* stream) {
  return u4_nt(A, lda, B, ldb, C, ldc, M, N, K, residual, ldr, nullptr, stream);
}
// ... + A2 . B2^T in the same k-loop (lhrs_gemm_bf16_nt_lora's semantics with a plain epilogue; bit-identical to the 16-wave kernel's fused pair)
extern "C" int lhrs_gemm_u4_nt_lora(const void* A, int lda, const void* B, int ldb, const void* A2, int lda2, const void* B2, int ldb2, int K2, void* C, int ldc,
                                    int M, int N, int K, const void* residual, int ldr, void* stream) {
  const U4Pair p{A2, lda2, B2, ldb2, K2};
  return u4_nt(A, lda, B, ldb, C, ldc, M, N, K, residual, ldr, &p, stream);
}

// q|k|v projection with RoPE in the epilogue (the semantics of lhrs_gemm_rope_fwd without a LoRA pair): columns [0, rope_cols) are heads of 128 rotated with
// the position m % pos_mod + pos0 of their row, the rest stored as computed.  0 launched; 1 not this kernel's problem.
static int u4_rope(const void* X, int ldx, const void* W, int ldw, void* C, int ldc, int M, int N, int K, const float* cos_t, const float* sin_t,
                   int pos_mod, int pos0, int rope_cols, const U4Pair* pair, void* stream) {
  if (!u4_addressable(X, ldx, W, ldw, C, ldc, M, N, K) || rope_cols % 256 != 0 || rope_cols > N || pos_mod < 16 || pos0 < 0 || cos_t == nullptr || sin_t == nullptr ||
      ((size_t)cos_t | (size_t)sin_t) % 16 != 0 || (long)(pos_mod + pos0) * 64 * 4 >= (1L << 31))
    return 1;
  U4Args g{};
  g.A = (const bf16_t*)X; g.B = (const bf16_t*)W; g.C = (bf16_t*)C; g.M = M; g.N = N; g.K = K; g.lda = ldx; g.ldb = ldw; g.ldc = ldc;
  g.tilesM = (M + 255) / 256; g.tilesN = (N + 255) / 256;
  g.rope_cos = cos_t; g.rope_sin = sin_t; g.rope_mod = pos_mod; g.rope_pos0 = pos0; g.rope_cols = rope_cols;
  if (!u4_pair(g, pair, N)) return 1;
  hipLaunchKernelGGL((gemm_u4_kernel<3, false>), u4_grid(g), dim3(256), 0, (hipStream_t)stream, g);
  LHRS_CHECK_LAUNCH("gemm_u4_rope");
  return 0;
}
extern "C" int lhrs_gemm_u4_rope(const void* X, int ldx, const void* W, int ldw, void* C, int ldc, int M, int N, int K, const float* cos_t, const float* sin_t,
                                 int pos_mod, int pos0, int rope_cols, void* stream) {
  return u4_rope(X, ldx, W, ldw, C, ldc, M, N, K, cos_t, sin_t, pos_mod, pos0, rope_cols, nullptr, stream);
}
extern "C" int lhrs_gemm_u4_rope_lora(const void* X, int ldx, const void* W, int ldw, const void* A2, int lda2, const void* B2, int ldb2, int K2, void* C, int ldc,
                                      int M, int N, int K, const float* cos_t, const float* sin_t, int pos_mod, int pos0, int rope_cols, void* stream) {
  const U4Pair p{A2, lda2, B2, ldb2, K2};
  return u4_rope(X, ldx, W, ldw, C, ldc, M, N, K, cos_t, sin_t, pos_mod, pos0, rope_cols, &p, stream);
}

// LLaMA MLP, forward half: gu [M, 2 ff] = X . Wgu^T (Wgu = [gate; up] weight [2 ff, K]) and act [M, ff] = silu(gate) * up in one launch (the semantics of
// lhrs_gemm_swiglu_fwd without a LoRA pair; bit-identical).  0 launched; 1 not this kernel's problem.
extern "C" int lhrs_gemm_u4_swiglu_fwd(const void* X, int ldx, const void* Wgu, int ldw, void* gu, int ld_gu, void* act, int ld_act, int M, int ff, int K,
                                       void* stream) {
  if (ff <= 0 || ff % 128 != 0 || !u4_addressable(X, ldx, Wgu, ldw, gu, ld_gu, M, 2 * ff, K) || ld_act % 8 != 0 || ld_act < ff || (size_t)act % 16 != 0 ||
      (long)144 * ld_act * 2 >= (1L << 32) || (long)2 * ff * ldw * 2 >= (1L << 32))
    return 1;
  U4Args g{};
  g.A = (const bf16_t*)X; g.B = (const bf16_t*)Wgu; g.C = (bf16_t*)gu; g.M = M; g.N = 2 * ff; g.K = K; g.lda = ldx; g.ldb = ldw; g.ldc = ld_gu;
  g.tilesM = (M + 255) / 256; g.tilesN = ff / 128; g.rope_mod = 1;
  g.ff = ff; g.aux_out = (bf16_t*)act; g.ld_aux = ld_act;
  hipLaunchKernelGGL((gemm_u4_kernel<1, false>), u4_grid(g), dim3(256), 0, (hipStream_t)stream, g);
  LHRS_CHECK_LAUNCH("gemm_u4_swiglu_fwd");
  return 0;
}

// LLaMA MLP, backward half: dgu [M, 2 ff] = swiglu'(gu) * (dY . WdT^T) in one launch (lhrs_gemm_swiglu_bwd without a LoRA pair; bit-identical); dgu may alias gu
// (a lane reads the gate / up values of exactly the elements it overwrites, before it writes them).  0 launched; 1 not this kernel's problem.
extern "C" int lhrs_gemm_u4_swiglu_bwd(const void* dY, int ldy, const void* WdT, int ldw, const void* gu, void* dgu, int ld_gu, int M, int ff, int K, void* stream) {
  if (ff <= 0 || ff % 8 != 0 || !u4_addressable(dY, ldy, WdT, ldw, dgu, ld_gu, M, ff, K) || ld_gu < 2 * ff || ((size_t)gu) % 16 != 0) return 1;
  U4Args g{};
  g.A = (const bf16_t*)dY; g.B = (const bf16_t*)WdT; g.C = (bf16_t*)dgu; g.M = M; g.N = ff; g.K = K; g.lda = ldy; g.ldb = ldw; g.ldc = ld_gu;
  g.tilesM = (M + 255) / 256; g.tilesN = (ff + 255) / 256; g.rope_mod = 1;
  g.ff = ff; g.aux = (const bf16_t*)gu; g.ld_aux = ld_gu;
  hipLaunchKernelGGL((gemm_u4_kernel<2, false>), u4_grid(g), dim3(256), 0, (hipStream_t)stream, g);
  LHRS_CHECK_LAUNCH("gemm_u4_swiglu_bwd");
  return 0;
}
