// Flash-style attention forward / backward on bf16 MFMA for gfx950 (64-lane wavefronts, LDS-staged K/V tiles,
// online softmax in fp32 by wavefront reduction).  One kernel family serves the three attentions on the
// LHRS-Bot hot path:
//   * CLIP ViT self-attention      (16 heads x 64, N = 257, no mask)           - HF CLIPAttention
//   * AttnPooler cross-attention   (16 heads x 64, (Lq,Lkv) = (64,320),(48,304),(32,288); forward + backward)
//                                  - nn.MultiheadAttention in lhrs/models/common_arch.py:276,302-313
//   * LLaMA causal self-attention  (32 heads x 128, causal + key-padding length; forward + backward)
//                                  - HF LlamaAttention called from lhrs/models/text_modal.py:281-292
//
// Sequences are described by 8-int records {q_off, q_len, kv_off, kv_len, kv_rows, causal_off, -, -}: token
// offsets into the row-major activations, so ragged batches (the pooler's three query groups) run in one launch.
//
// MFMA data flow.  Every product is v_mfma_f32_16x16x32_bf16, whose A and B operands both hold, per lane,
// 8 k-contiguous values of row/col (lane&15), and whose result holds 4 consecutive rows of column (lane&15).
// Products are therefore arranged so that (a) the softmax row index is always (lane&15) - row statistics are
// per-lane scalars plus two xor-shuffles - and (b) a result fragment is reused directly as the next MFMA's B
// operand: the MFMA k-slot <-> key (or query) assignment is a fixed permutation, applied identically to the
// register operand and to the LDS operand, which is legal because k is summed over.
//   forward :  S^T = K . Q^T  (A = K tile, B = Q regs)   ->  P  ->  O^T  = V^T . P^T (A = V^T tile, B = P regs)
//   dQ      :  S^T, dP^T = V . dO^T -> dS ->  dQ^T = K^T . dS^T (A = K^T tile, B = dS regs)
//   dK, dV  :  S = Q . K^T (A = Q tile, B = K regs), dP = dO . V^T -> P, dS ->
//              dV^T = dO^T . P (A = dO^T tile, B = P regs),  dK^T = Q^T . dS (A = Q^T tile, B = dS regs)
// The operands that must be contiguous along the token axis (V^T, K^T, Q^T, dO^T) are read straight out of the SAME
// row-major LDS tiles with gfx950's transposing LDS read (ds_read_b64_tr_b16): inside a 16-lane group, lane i receives
// element (i&3) of the 8-byte chunk addressed by lane 4j + (i>>2), for j = 0..3 - so when lane L addresses
// tile[token t0 + (L>>2)][d0 + 4*(L&3)], lane i ends up with tile[t0 + j][d0 + i]: a [4 tokens x 16 d] transpose per
// group and instruction, which is exactly one half of an MFMA operand in the k-slot permutation above.
// No transposed copies exist in HBM.  K/V (or Q/dO) tiles are prefetched global->registers while the previous tile is
// multiplied, then written to LDS behind one barrier (latency hidden behind the MFMAs).
#include "common.h"
#include <type_traits>

namespace {

struct AttnArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* dout;  // row-major [tokens, ld]
  long ldq, ldk, ldv, ldo, ld_do, ld_dq, ld_dk, ld_dv;
  bf16_t* o; bf16_t* dq; bf16_t* dk; bf16_t* dv;
  int LTq;            // padded row length of lse / delta (multiple of 64)
  float* lse;         // [seq][H][LTq]
  const float* delta; // [seq][H][LTq]
  const int* desc;    // [nseq][8]
  int H;
  float scale;
  const unsigned char* kmask;  // optional key-visibility bytes [seq][ld_kmask] (HF attention_mask semantics), tiled forward only
  long ld_kmask;
  // backward only (lhrs_attn_bwd_rope): the INVERSE rotary embedding of the dq / dk rows applied in the store (resident kernels): fp32
  // cos / sin tables [pos][D / 2], position of token row m = m % rope_mod + rope_pos0; nullptr = plain stores
  const float* rope_cos;
  const float* rope_sin;
  int rope_mod, rope_pos0;
  // backward, resident dQ kernel only (lhrs_attn_bwd_o): the forward output rows - delta = rowsum(dO * O) is then computed by the kernel for the
  // rows it loads anyway and written to delta_w (which the dK/dV kernel, launched behind it, reads) instead of by a launch of its own
  const bf16_t* o_in;
  long ld_oin;
  float* delta_w;
  int wide;  // resident kernels: result rows go out as 16-byte stores (store_row_wide); the host clears it when a row pointer would not be 16-byte aligned
};

constexpr float NEG_INF = -__builtin_huge_valf();

// ---- LDS tile image ----------------------------------------------------------------------------
// row-major tile [64 tokens][D]: 16-B chunk index XOR f(row).  D=128 (16 chunks, 256-B rows): f = ((row&7)<<1)|((row>>3)&1)
// - a bijection on 0..15 (ds_read_b128 over 16 rows conflict-free) whose low rows land on distinct 32-B pairs (the
// 8 tokens x 32 B of a transposing read conflict-free).  D=64 (8 chunks, 128-B rows): f = row&7.
template <int D>
__device__ __forceinline__ int swz(int row) {
  return D == 128 ? (((row & 7) << 1) | ((row >> 3) & 1)) : (row & 7);
}
template <int D>
__device__ __forceinline__ int rm_off(int row, int c) { return row * (D * 2) + ((c ^ swz<D>(row)) << 4); }

template <int D> struct TileRegs { uint4 v[64 * (D / 8) / 256]; };

template <int D>
__device__ __forceinline__ void tile_load(TileRegs<D>& r, const bf16_t* base, long ld, int row0, int len, int tid) {
  constexpr int CH = D / 8, N = 64 * CH / 256;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int idx = tid + i * 256;
    const int gr = min(row0 + idx / CH, len - 1);
    r.v[i] = *reinterpret_cast<const uint4*>(base + (long)gr * ld + (idx % CH) * 8);
  }
}
template <int D>
__device__ __forceinline__ void tile_store(char* lds, const TileRegs<D>& r, int tid) {
  constexpr int CH = D / 8, N = 64 * CH / 256;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int idx = tid + i * 256;
    *reinterpret_cast<uint4*>(lds + rm_off<D>(idx / CH, idx % CH)) = r.v[i];
  }
}
template <int D>
__device__ __forceinline__ bf16x8 frag_rm(const char* lds, int row, int ks, int fg) {
  return *reinterpret_cast<const bf16x8*>(lds + rm_off<D>(row, ks * 4 + fg));
}

// Transposing fragment reads.  For k-step t of a 64-token tile, the fragment of d-block db holds, for lane (L = lane&15,
// g = lane>>4): k-slot j<4 <-> token 32t + 4g + j, j>=4 <-> token 32t + 16 + 4g + (j-4), all at d = 16 db + L.
// Lane address: token r0 = 4g + (L>>2) (+32t, +16), 16-B chunk 2 db + ((L&3)>>1), 8-B half (L&1).
template <int D>
struct TrAddr {
  unsigned base[D / 16];  // per d-block byte address inside the tile for (t = 0, lower half)
  __device__ __forceinline__ void init(const char* lds, int lane) {
    const int L = lane & 15, g = lane >> 4;
    const int r0 = 4 * g + (L >> 2);
    const unsigned tile = (unsigned)(size_t)((__attribute__((address_space(3))) const char*)lds);
#pragma unroll
    for (int db = 0; db < D / 16; ++db)
      base[db] = tile + r0 * (D * 2) + (((2 * db + ((L & 3) >> 1)) ^ swz<D>(r0)) << 4) + (L & 1) * 8;
  }
};
#define TR_RD(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))

// issue the 2 * D/16 transposing reads of k-step T (results are NOT valid until tr_wait)
template <int D, int T>
__device__ __forceinline__ void tr_issue(const TrAddr<D>& a, bf16x4 (&lo)[D / 16], bf16x4 (&hi)[D / 16]) {
#pragma unroll
  for (int db = 0; db < D / 16; ++db) {
    TR_RD(lo[db], a.base[db], (32 * T) * (D * 2));
    TR_RD(hi[db], a.base[db], (32 * T + 16) * (D * 2));
  }
}
template <int N>
__device__ __forceinline__ void tr_wait(bf16x4 (&lo)[N], bf16x4 (&hi)[N]) {
  if constexpr (N == 8) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(lo[4]), "+v"(lo[5]), "+v"(lo[6]), "+v"(lo[7]),
                   "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]), "+v"(hi[4]), "+v"(hi[5]), "+v"(hi[6]), "+v"(hi[7]));
  } else if constexpr (N == 4) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
  } else {
    static_assert(N == 2, "tr_wait: 2, 4 or 8 d-blocks");
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(lo[1]), "+v"(hi[0]), "+v"(hi[1]));
  }
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ bf16x8 join(const bf16x4& lo, const bf16x4& hi) {
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ bf16x8 pack_frag(const f32x4& a, const f32x4& b) {
  bf16x8 r;
  r[0] = (short)f2bf(a[0]); r[1] = (short)f2bf(a[1]); r[2] = (short)f2bf(a[2]); r[3] = (short)f2bf(a[3]);
  r[4] = (short)f2bf(b[0]); r[5] = (short)f2bf(b[1]); r[6] = (short)f2bf(b[2]); r[7] = (short)f2bf(b[3]);
  return r;
}
// Reductions across the 4 lanes that share (lane & 15), i.e. across the four 16-lane rows of the wavefront, WITHOUT the LDS crossbar
// (ds_bpermute + s_waitcnt in the middle of a dependent softmax chain): gfx950's v_permlane16_swap (odd rows of the first operand <-> even
// rows of the second) and v_permlane32_swap (upper half of the first <-> lower half of the second) applied to two copies of the value
// leave in every lane the partner's value next to its own.
__device__ __forceinline__ float group_max(float v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float group_sum(float v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ void store4bf(bf16_t* p, const f32x4& v, float s) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack2bf(v[0] * s, v[1] * s), pack2bf(v[2] * s, v[3] * s));
}
// One token row of a [16 x D] result as the lanes hold it (lane (fr, fg): DB blocks of the four dims db * 16 + fg * 4 + 0..3 of row fr) stored with 16-byte stores
// (round 6).  Eight-byte stores - DB per lane, 32 B per row and instruction - made the epilogues store-ISSUE bound (MI355X_MICROARCH: "attention epilogue store tail").
// v_permlane16_swap exchanges, between the blocks db and db + 1, the odd 16-lane rows of the first with the even rows of the second: afterwards lane fg = 0 / 2 holds
// dims 0..7 / 8..15 of block db and lane fg = 1 / 3 those of block db + 1 - eight consecutive bf16 per lane, 64 contiguous bytes per row and instruction, half the
// store instructions.  Same values, same rounding: bit-identical to store4bf per block.  `row` points at dim 0 of the row (16-byte aligned, D a multiple of 32).
template <int DB>
__device__ __forceinline__ void store_row_wide(bf16_t* row, const f32x4 (&v)[DB], float s, int fg) {
#pragma unroll
  for (int db = 0; db < DB; db += 2) {
    const uint32_t a0 = pack2bf(v[db][0] * s, v[db][1] * s), a1 = pack2bf(v[db][2] * s, v[db][3] * s);
    const uint32_t b0 = pack2bf(v[db + 1][0] * s, v[db + 1][1] * s), b1 = pack2bf(v[db + 1][2] * s, v[db + 1][3] * s);
    const auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
    *reinterpret_cast<uint4*>(row + (db + (fg & 1)) * 16 + (fg >> 1) * 8) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
  }
}

// d(q) / d(k) of one token row, as the lane holds it (DB f32x4 blocks: dims db * 16 + fg * 4 + 0..3), through the transpose of the rotary
// embedding: the gradient of (x1 cos - x2 sin, x2 cos + x1 sin) is rope_pair with -sin.  The values are rounded to bf16 first, exactly
// what the stand-alone kernel reads back from HBM: attention-backward + lhrs_rope(inverse) and this fused store agree bit for bit.
template <int DB> struct RopeRow { float4 c[DB / 2], s[DB / 2]; };
template <int DB>
__device__ __forceinline__ void rope_row_load(RopeRow<DB>& rr, const AttnArgs& a, long token_row, int fg) {
  const int pos = (int)(token_row % a.rope_mod) + a.rope_pos0;
  const float* cs = a.rope_cos + (long)pos * (DB * 8) + fg * 4;
  const float* sn = a.rope_sin + (long)pos * (DB * 8) + fg * 4;
#pragma unroll
  for (int db = 0; db < DB / 2; ++db) { rr.c[db] = *reinterpret_cast<const float4*>(cs + db * 16); rr.s[db] = *reinterpret_cast<const float4*>(sn + db * 16); }
}
template <int DB>
__device__ __forceinline__ void rope_row_apply(f32x4 (&g)[DB], const RopeRow<DB>& rr) {
#pragma unroll
  for (int db = 0; db < DB / 2; ++db) {
    const float cv[4] = {rr.c[db].x, rr.c[db].y, rr.c[db].z, rr.c[db].w}, sv[4] = {rr.s[db].x, rr.s[db].y, rr.s[db].z, rr.s[db].w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float o1, o2;
      rope_pair(bf2f(f2bf(g[db][i])), bf2f(f2bf(g[db + DB / 2][i])), cv[i], -sv[i], o1, o2);
      g[db][i] = o1; g[db + DB / 2][i] = o2;
    }
  }
}
template <int DB>
__device__ __forceinline__ void rope_bwd_inplace(f32x4 (&g)[DB], const AttnArgs& a, long token_row, int fg) {
  RopeRow<DB> rr;
  rope_row_load<DB>(rr, a, token_row, fg);
  rope_row_apply<DB>(g, rr);
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

// ---------------------------------------------------------------- forward
template <int D, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
  constexpr int KS = D / 32, DB = D / 16;
  __shared__ __attribute__((aligned(16))) char lds_k[64 * D * 2];
  __shared__ __attribute__((aligned(16))) char lds_v[64 * D * 2];
  const int seq = blockIdx.z, h = blockIdx.y;
  const int* ds = a.desc + seq * 8;
  const int q_off = ds[0], q_len = ds[1], kv_off = ds[2], kv_len = ds[3], coff = ds[5];
  const int q0 = blockIdx.x * 64;
  if (q0 >= q_len) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
  const int qrow = q0 + wave * 16 + fr;
  const bf16_t* qp = a.q + (long)(q_off + min(qrow, q_len - 1)) * a.ldq + h * D;
  bf16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 32 + fg * 8);

  float m = NEG_INF, l = 0.f;
  f32x4 o[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned char* km = a.kmask ? a.kmask + (long)seq * a.ld_kmask : nullptr;
  int kv_end = kv_len;
  if (CAUSAL) kv_end = max(0, min(kv_len, q0 + 64 + coff));
  const int ntiles = (kv_end + 63) >> 6;
  const bf16_t* kbase = a.k + (long)kv_off * a.ldk + h * D;
  const bf16_t* vbase = a.v + (long)kv_off * a.ldv + h * D;
  TrAddr<D> tv;
  tv.init(lds_v, lane);
  TileRegs<D> kr, vr;
  if (ntiles > 0) { tile_load<D>(kr, kbase, a.ldk, 0, kv_len, tid); tile_load<D>(vr, vbase, a.ldv, 0, kv_len, tid); }

  for (int j = 0; j < ntiles; ++j) {
    __syncthreads();  // every wave is done reading tile j-1
    tile_store<D>(lds_k, kr, tid);
    tile_store<D>(lds_v, vr, tid);
    __syncthreads();
    if (j + 1 < ntiles) {  // prefetch tile j+1 into registers; the loads fly while tile j is multiplied
      tile_load<D>(kr, kbase, a.ldk, (j + 1) * 64, kv_len, tid);
      tile_load<D>(vr, vbase, a.ldv, (j + 1) * 64, kv_len, tid);
    }
    f32x4 s[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      s[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) s[nb] = MFMA(frag_rm<D>(lds_k, nb * 16 + fr, ks, fg), qf[ks], s[nb]);
    }
    bf16x4 v0lo[DB], v0hi[DB], v1lo[DB], v1hi[DB];
    tr_issue<D, 0>(tv, v0lo, v0hi);  // V^T fragments of k-step 0 land while the softmax runs
    float mx = NEG_INF;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = j * 64 + nb * 16 + fg * 4 + r;
        bool ok = key < kv_len && (!CAUSAL || key <= qrow + coff);
        if (km != nullptr && ok) ok = km[key] != 0;
        s[nb][r] = ok ? s[nb][r] * a.scale : NEG_INF;
        mx = fmaxf(mx, s[nb][r]);
      }
    mx = group_max(mx);
    const float m_new = fmaxf(m, mx);
    const float m_use = (m_new == NEG_INF) ? 0.f : m_new;
    const float alpha = __expf(m - m_use);
    float rs = 0.f;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[nb][r] = __expf(s[nb][r] - m_use);
        rs += s[nb][r];
      }
    rs = group_sum(rs);
    l = l * alpha + rs;
    m = m_new;
#pragma unroll
    for (int i = 0; i < DB; ++i) o[i] *= alpha;
    const bf16x8 p0 = pack_frag(s[0], s[1]), p1 = pack_frag(s[2], s[3]);
    tr_wait<DB>(v0lo, v0hi);
    tr_issue<D, 1>(tv, v1lo, v1hi);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int db = 0; db < DB; ++db) o[db] = MFMA(join(v0lo[db], v0hi[db]), p0, o[db]);
    tr_wait<DB>(v1lo, v1hi);
#pragma unroll
    for (int db = 0; db < DB; ++db) o[db] = MFMA(join(v1lo[db], v1hi[db]), p1, o[db]);
  }
  if (qrow < q_len) {
    const float inv = l > 0.f ? 1.f / l : 0.f;
    bf16_t* op = a.o + (long)(q_off + qrow) * a.ldo + h * D + fg * 4;
#pragma unroll
    for (int db = 0; db < DB; ++db) store4bf(op + db * 16, o[db], inv);
    if (a.lse && fg == 0) a.lse[(long)(seq * a.H + h) * a.LTq + qrow] = (l > 0.f) ? m + __logf(l) : NEG_INF;
  }
}

// ---------------------------------------------------------------- backward: dQ
template <int D, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnArgs a) {
  constexpr int KS = D / 32, DB = D / 16;
  __shared__ __attribute__((aligned(16))) char lds_k[64 * D * 2];
  __shared__ __attribute__((aligned(16))) char lds_v[64 * D * 2];
  const int seq = blockIdx.z, h = blockIdx.y;
  const int* ds = a.desc + seq * 8;
  const int q_off = ds[0], q_len = ds[1], kv_off = ds[2], kv_len = ds[3], coff = ds[5];
  const int q0 = blockIdx.x * 64;
  if (q0 >= q_len) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
  const int qrow = q0 + wave * 16 + fr;
  const int qrow_c = min(qrow, q_len - 1);
  const bf16_t* qp = a.q + (long)(q_off + qrow_c) * a.ldq + h * D;
  const bf16_t* dop = a.dout + (long)(q_off + qrow_c) * a.ld_do + h * D;
  bf16x8 qf[KS], dof[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 32 + fg * 8);
    dof[ks] = *reinterpret_cast<const bf16x8*>(dop + ks * 32 + fg * 8);
  }
  const long stat = (long)(seq * a.H + h) * a.LTq + qrow_c;
  const float lse_q = a.lse[stat], delta_q = a.delta[stat];
  f32x4 dq[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i) dq[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  int kv_end = kv_len;
  if (CAUSAL) kv_end = max(0, min(kv_len, q0 + 64 + coff));
  const int ntiles = (kv_end + 63) >> 6;
  const bf16_t* kbase = a.k + (long)kv_off * a.ldk + h * D;
  const bf16_t* vbase = a.v + (long)kv_off * a.ldv + h * D;
  TrAddr<D> tk;
  tk.init(lds_k, lane);
  TileRegs<D> kr, vr;
  if (ntiles > 0) { tile_load<D>(kr, kbase, a.ldk, 0, kv_len, tid); tile_load<D>(vr, vbase, a.ldv, 0, kv_len, tid); }

  for (int j = 0; j < ntiles; ++j) {
    __syncthreads();
    tile_store<D>(lds_k, kr, tid);
    tile_store<D>(lds_v, vr, tid);
    __syncthreads();
    if (j + 1 < ntiles) {
      tile_load<D>(kr, kbase, a.ldk, (j + 1) * 64, kv_len, tid);
      tile_load<D>(vr, vbase, a.ldv, (j + 1) * 64, kv_len, tid);
    }
    f32x4 s[4], dp[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      s[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
      dp[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        s[nb] = MFMA(frag_rm<D>(lds_k, nb * 16 + fr, ks, fg), qf[ks], s[nb]);
        dp[nb] = MFMA(frag_rm<D>(lds_v, nb * 16 + fr, ks, fg), dof[ks], dp[nb]);
      }
    }
    bf16x4 k0lo[DB], k0hi[DB], k1lo[DB], k1hi[DB];
    tr_issue<D, 0>(tk, k0lo, k0hi);  // K^T fragments of k-step 0 land while dS is formed
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = j * 64 + nb * 16 + fg * 4 + r;
        const bool ok = key < kv_len && (!CAUSAL || key <= qrow + coff) && qrow < q_len;
        const float p = ok ? __expf(s[nb][r] * a.scale - lse_q) : 0.f;
        s[nb][r] = ok ? p * (dp[nb][r] - delta_q) * a.scale : 0.f;  // dS
      }
    const bf16x8 d0 = pack_frag(s[0], s[1]), d1 = pack_frag(s[2], s[3]);
    tr_wait<DB>(k0lo, k0hi);
    tr_issue<D, 1>(tk, k1lo, k1hi);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int db = 0; db < DB; ++db) dq[db] = MFMA(join(k0lo[db], k0hi[db]), d0, dq[db]);
    tr_wait<DB>(k1lo, k1hi);
#pragma unroll
    for (int db = 0; db < DB; ++db) dq[db] = MFMA(join(k1lo[db], k1hi[db]), d1, dq[db]);
  }
  if (qrow < q_len) {
    bf16_t* p = a.dq + (long)(q_off + qrow) * a.ld_dq + h * D + fg * 4;
#pragma unroll
    for (int db = 0; db < DB; ++db) store4bf(p + db * 16, dq[db], 1.f);
  }
}

// ---------------------------------------------------------------- backward: dK, dV
template <int D, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnArgs a) {
  constexpr int KS = D / 32, DB = D / 16;
  __shared__ __attribute__((aligned(16))) char lds_q[64 * D * 2];
  __shared__ __attribute__((aligned(16))) char lds_do[64 * D * 2];
  __shared__ __attribute__((aligned(16))) float lds_lse[64];
  __shared__ __attribute__((aligned(16))) float lds_delta[64];
  const int seq = blockIdx.z, h = blockIdx.y;
  const int* ds = a.desc + seq * 8;
  const int q_off = ds[0], q_len = ds[1], kv_off = ds[2], kv_len = ds[3], kv_rows = ds[4], coff = ds[5];
  const int k0 = blockIdx.x * 64;
  if (k0 >= kv_rows) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
  const int key = k0 + wave * 16 + fr;
  const int key_c = min(key, kv_rows - 1);
  const bf16_t* kp = a.k + (long)(kv_off + key_c) * a.ldk + h * D;
  const bf16_t* vp = a.v + (long)(kv_off + key_c) * a.ldv + h * D;
  bf16x8 kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    kf[ks] = *reinterpret_cast<const bf16x8*>(kp + ks * 32 + fg * 8);
    vf[ks] = *reinterpret_cast<const bf16x8*>(vp + ks * 32 + fg * 8);
  }
  f32x4 dk[DB], dv[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i) { dk[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int nq_tiles = (q_len + 63) >> 6;
  int i0 = 0;
  if (CAUSAL) i0 = max(0, (k0 - coff) >> 6);  // first q tile that can see key k0 (q >= key - coff)
  const bf16_t* qbase = a.q + (long)q_off * a.ldq + h * D;
  const bf16_t* dobase = a.dout + (long)q_off * a.ld_do + h * D;
  const float* lsebase = a.lse + (long)(seq * a.H + h) * a.LTq;
  const float* deltabase = a.delta + (long)(seq * a.H + h) * a.LTq;
  TrAddr<D> tq, tdo;
  tq.init(lds_q, lane);
  tdo.init(lds_do, lane);
  TileRegs<D> qr, dor;
  float stat_r = 0.f;
  auto prefetch = [&](int i) {
    tile_load<D>(qr, qbase, a.ldq, i * 64, q_len, tid);
    tile_load<D>(dor, dobase, a.ld_do, i * 64, q_len, tid);
    if (tid < 64) stat_r = lsebase[i * 64 + tid];
    else if (tid < 128) stat_r = deltabase[i * 64 + tid - 64];
  };
  if (i0 < nq_tiles) prefetch(i0);

  for (int i = i0; i < nq_tiles; ++i) {
    __syncthreads();
    tile_store<D>(lds_q, qr, tid);
    tile_store<D>(lds_do, dor, tid);
    if (tid < 64) lds_lse[tid] = stat_r;
    else if (tid < 128) lds_delta[tid - 64] = stat_r;
    __syncthreads();
    if (i + 1 < nq_tiles) prefetch(i + 1);
    f32x4 s[4], dp[4];
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
      s[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
      dp[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        s[qb] = MFMA(frag_rm<D>(lds_q, qb * 16 + fr, ks, fg), kf[ks], s[qb]);
        dp[qb] = MFMA(frag_rm<D>(lds_do, qb * 16 + fr, ks, fg), vf[ks], dp[qb]);
      }
    }
    bf16x4 alo[DB], ahi[DB], blo[DB], bhi[DB];
    tr_issue<D, 0>(tdo, alo, ahi);  // dO^T, k-step 0
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
      const f32x4 lq = *reinterpret_cast<const f32x4*>(&lds_lse[qb * 16 + fg * 4]);
      const f32x4 dl = *reinterpret_cast<const f32x4*>(&lds_delta[qb * 16 + fg * 4]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = i * 64 + qb * 16 + fg * 4 + r;
        const bool ok = q < q_len && key < kv_len && (!CAUSAL || key <= q + coff);
        const float p = ok ? __expf(s[qb][r] * a.scale - lq[r]) : 0.f;
        dp[qb][r] = ok ? p * (dp[qb][r] - dl[r]) * a.scale : 0.f;  // dS
        s[qb][r] = p;                                               // P
      }
    }
    const bf16x8 p0 = pack_frag(s[0], s[1]), p1 = pack_frag(s[2], s[3]);
    const bf16x8 d0 = pack_frag(dp[0], dp[1]), d1 = pack_frag(dp[2], dp[3]);
    tr_wait<DB>(alo, ahi);
    tr_issue<D, 0>(tq, blo, bhi);   // Q^T, k-step 0
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int db = 0; db < DB; ++db) dv[db] = MFMA(join(alo[db], ahi[db]), p0, dv[db]);
    tr_wait<DB>(blo, bhi);
    tr_issue<D, 1>(tdo, alo, ahi);  // dO^T, k-step 1
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int db = 0; db < DB; ++db) dk[db] = MFMA(join(blo[db], bhi[db]), d0, dk[db]);
    tr_wait<DB>(alo, ahi);
    tr_issue<D, 1>(tq, blo, bhi);   // Q^T, k-step 1
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int db = 0; db < DB; ++db) dv[db] = MFMA(join(alo[db], ahi[db]), p1, dv[db]);
    tr_wait<DB>(blo, bhi);
#pragma unroll
    for (int db = 0; db < DB; ++db) dk[db] = MFMA(join(blo[db], bhi[db]), d1, dk[db]);
  }
  if (key < kv_rows) {
    bf16_t* pk = a.dk + (long)(kv_off + key) * a.ld_dk + h * D + fg * 4;
    bf16_t* pv = a.dv + (long)(kv_off + key) * a.ld_dv + h * D + fg * 4;
#pragma unroll
    for (int db = 0; db < DB; ++db) { store4bf(pk + db * 16, dk[db], 1.f); store4bf(pv + db * 16, dv[db], 1.f); }
  }
}

// ================================================================================================
// "Resident" variants: 160 KiB of LDS per CU holds BOTH operand matrices of one (sequence, head) completely
// (K and V for forward / dQ, Q and dO for dK/dV) when they have at most 320 rows (D = 128) or 640 rows (D = 64) -
// every attention on the LHRS-Bot training path (S = 273, ViT 257, pooler <= 320).  One 8-wave workgroup per
// (sequence, head): the matrices are DMA'd once (global_load_lds, swizzle on the source address), one barrier, and then
// each wave walks 16-row groups on its own - no further barriers, no per-tile global loads.  Groups are dealt to waves
// in a zig-zag over descending causal cost so that every wave gets (almost) the same number of 64-key tiles.
// ================================================================================================
template <int D> constexpr int res_rows() { return 163840 / (2 * D * 2); }

// DMA `rows` (multiple of 64, <= res_rows) rows of a [*, D] bf16 matrix into the swizzled resident image
template <int D>
__device__ __forceinline__ void res_load(char* lds, const bf16_t* base, long ld, int rows, int len, int lane, int wave) {
  constexpr int CH = D / 8, R = 1024 / (D * 2);  // rows per wave-instruction
  typedef const __attribute__((address_space(1))) void* gp;
  typedef __attribute__((address_space(3))) void* lp;
  for (int r0 = wave * R; r0 < rows; r0 += 8 * R) {
    const int row = r0 + lane / CH;
    const int c = (lane % CH) ^ swz<D>(row);
    const bf16_t* src = base + (long)min(row, len - 1) * ld + c * 8;
    __builtin_amdgcn_global_load_lds((gp)src, (lp)(lds + r0 * (D * 2)), 16, 0, 0);
  }
}
__device__ __forceinline__ int zigzag_group(int p, int wave, int G, bool descending) {
  const int k = p * 8 + ((p & 1) ? 7 - wave : wave);
  if (k >= G) return -1;
  return descending ? G - 1 - k : k;
}

// phase stamps for tools/attn_diag.py (a -DATTN_DIAG build only; compiled out of the product)
#ifdef ATTN_DIAG
__device__ unsigned long long* g_attn_dbg = nullptr;
__device__ int g_attn_dbg_sel = 0;  // 0 forward, 1 dQ, 2 dK/dV
#define DIAG_T(slot) do { if (g_attn_dbg && g_attn_dbg_sel == DIAG_KID && lane == 0) g_attn_dbg[((long)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 8 + (slot)] = wall_clock64(); } while (0)
#define DIAG_DECL(id) constexpr int DIAG_KID = id; unsigned long long dg_t = 0, dg_acc[3] = {0, 0, 0}; int dg_units = 0
#define DIAG_MARK() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); dg_t = wall_clock64(); } while (0)
#define DIAG_ACC(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = wall_clock64(); dg_acc[k] += t_ - dg_t; dg_t = t_; } while (0)
#define DIAG_UNITS(n) dg_units += (n)
#define DIAG_FLUSH() do { if (g_attn_dbg && g_attn_dbg_sel == DIAG_KID && lane == 0) { unsigned long long* d_ = g_attn_dbg + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 8; d_[2] = dg_acc[0]; d_[3] = dg_acc[1]; d_[5] = dg_acc[2]; d_[6] = dg_units; } } while (0)
#else
#define DIAG_T(slot) do { } while (0)
#define DIAG_DECL(id)
#define DIAG_MARK() do { } while (0)
#define DIAG_ACC(k) do { } while (0)
#define DIAG_UNITS(n) do { } while (0)
#define DIAG_FLUSH() do { } while (0)
#endif

template <int D, bool CAUSAL>
__global__ __launch_bounds__(512, 2) void attn_fwd_res_kernel(AttnArgs a) {
  constexpr int KS = D / 32, DB = D / 16, TILE = 64 * D * 2;
  __shared__ __attribute__((aligned(16))) char smem[163840];
  char* lds_k = smem;
  char* lds_v = smem + res_rows<D>() * D * 2;
  const int seq = blockIdx.y, h = blockIdx.x;
  const int* ds = a.desc + seq * 8;
  const int q_off = ds[0], q_len = ds[1], kv_off = ds[2], kv_len = ds[3], coff = ds[5];
  const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  DIAG_DECL(0);
  DIAG_T(0);
  int kv_need = kv_len;
  if (CAUSAL) kv_need = max(0, min(kv_len, q_len + coff));
  const int rows = ((kv_need + 63) >> 6) << 6;
  if (rows > 0) {
    res_load<D>(lds_k, a.k + (long)kv_off * a.ldk + h * D, a.ldk, rows, kv_len, lane, wave);
    res_load<D>(lds_v, a.v + (long)kv_off * a.ldv + h * D, a.ldv, rows, kv_len, lane, wave);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  DIAG_T(1);
  TrAddr<D> tv;
  tv.init(lds_v, lane);
  const float c2 = a.scale * 1.4426950408889634f;  // scale * log2(e)
  const int G = (q_len + 15) >> 4;
  for (int p = 0;; ++p) {
    const int g = zigzag_group(p, wave, G, CAUSAL);
    if (g < 0) break;
    DIAG_MARK();
    const int qrow = g * 16 + fr;
    const bf16_t* qp = a.q + (long)(q_off + min(qrow, q_len - 1)) * a.ldq + h * D;
    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 32 + fg * 8);
    DIAG_ACC(0);
    float m = NEG_INF, l = 0.f;
    f32x4 o[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int kv_end = kv_len;
    if (CAUSAL) kv_end = max(0, min(kv_len, g * 16 + 16 + coff));
    const int ntiles = (kv_end + 63) >> 6;
    for (int j = 0; j < ntiles; ++j) {
      const char* kt = lds_k + j * TILE;
      f32x4 s[4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        s[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s[nb] = MFMA(frag_rm<D>(kt, nb * 16 + fr, ks, fg), qf[ks], s[nb]);
      }
      TrAddr<D> tj;
#pragma unroll
      for (int db = 0; db < DB; ++db) tj.base[db] = tv.base[db] + j * TILE;
      bf16x4 v0lo[DB], v0hi[DB], v1lo[DB], v1hi[DB];
      tr_issue<D, 0>(tj, v0lo, v0hi);
      // The kernel is VALU-bound (the softmax of a 64-key tile was ~200 VALU instructions against 32 MFMAs), so: (1) only a tile that
      // crosses the causal diagonal or the end of the keys computes the mask - a wave-uniform test; (2) the running maximum m is kept in
      // RAW score units and exp(scale * s - scale * m) is ONE fma into v_exp_f32 (exp2) with c2 = scale * log2(e)
      const bool full = (j + 1) * 64 <= kv_len && (!CAUSAL || j * 64 + 63 <= g * 16 + coff);
      if (!full) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = j * 64 + nb * 16 + fg * 4 + r;
            const bool ok = key < kv_len && (!CAUSAL || key <= qrow + coff);
            s[nb][r] = ok ? s[nb][r] : NEG_INF;
          }
      }
      float mx = NEG_INF;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[nb][r]);
      mx = group_max(mx);
      const float m_new = fmaxf(m, mx);
      const float m_use = (m_new == NEG_INF) ? 0.f : m_new;
      const float mc = m_use * c2;
      const float alpha = __builtin_amdgcn_exp2f(m * c2 - mc);
      float rs = 0.f;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[nb][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[nb][r], c2, -mc));
          rs += s[nb][r];
        }
      l = l * alpha + rs;  // this lane's 16 keys of the tile: the four lanes of a row are added up once, behind the last tile
      m = m_new;
#pragma unroll
      for (int i = 0; i < DB; ++i) o[i] *= alpha;
      const bf16x8 p0 = pack_frag(s[0], s[1]), p1 = pack_frag(s[2], s[3]);
      tr_wait<DB>(v0lo, v0hi);
      tr_issue<D, 1>(tj, v1lo, v1hi);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int db = 0; db < DB; ++db) o[db] = MFMA(join(v0lo[db], v0hi[db]), p0, o[db]);
      tr_wait<DB>(v1lo, v1hi);
#pragma unroll
      for (int db = 0; db < DB; ++db) o[db] = MFMA(join(v1lo[db], v1hi[db]), p1, o[db]);
    }
    DIAG_ACC(1);
    DIAG_UNITS(ntiles);
    l = group_sum(l);
    if (qrow < q_len) {
      const float inv = l > 0.f ? 1.f / l : 0.f;
      bf16_t* op = a.o + (long)(q_off + qrow) * a.ldo + h * D;
      if (a.wide) store_row_wide<DB>(op, o, inv, fg);
      else {
#pragma unroll
        for (int db = 0; db < DB; ++db) store4bf(op + fg * 4 + db * 16, o[db], inv);
      }
      if (a.lse && fg == 0) a.lse[(long)(seq * a.H + h) * a.LTq + qrow] = (l > 0.f) ? m * a.scale + __logf(l) : NEG_INF;
    }
    DIAG_ACC(2);
  }
  DIAG_T(4);
  DIAG_FLUSH();
}

template <int D, bool CAUSAL>
__global__ __launch_bounds__(512, 2) void attn_bwd_dq_res_kernel(AttnArgs a) {
  constexpr int KS = D / 32, DB = D / 16, TILE = 64 * D * 2;
  __shared__ __attribute__((aligned(16))) char smem[163840];
  char* lds_k = smem;
  char* lds_v = smem + res_rows<D>() * D * 2;
  const int seq = blockIdx.y, h = blockIdx.x;
  const int* ds = a.desc + seq * 8;
  const int q_off = ds[0], q_len = ds[1], kv_off = ds[2], kv_len = ds[3], coff = ds[5];
  const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  DIAG_DECL(1);
  DIAG_T(0);
  int kv_need = kv_len;
  if (CAUSAL) kv_need = max(0, min(kv_len, q_len + coff));
  const int rows = ((kv_need + 63) >> 6) << 6;
  if (rows > 0) {
    res_load<D>(lds_k, a.k + (long)kv_off * a.ldk + h * D, a.ldk, rows, kv_len, lane, wave);
    res_load<D>(lds_v, a.v + (long)kv_off * a.ldv + h * D, a.ldv, rows, kv_len, lane, wave);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  DIAG_T(1);
  TrAddr<D> tk;
  tk.init(lds_k, lane);
  const int G = (q_len + 15) >> 4;
  for (int p = 0;; ++p) {
    const int g = zigzag_group(p, wave, G, CAUSAL);
    if (g < 0) break;
    DIAG_MARK();
    const int qrow = g * 16 + fr;
    const int qrow_c = min(qrow, q_len - 1);
    const bf16_t* qp = a.q + (long)(q_off + qrow_c) * a.ldq + h * D;
    const bf16_t* dop = a.dout + (long)(q_off + qrow_c) * a.ld_do + h * D;
    bf16x8 qf[KS], dof[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 32 + fg * 8);
      dof[ks] = *reinterpret_cast<const bf16x8*>(dop + ks * 32 + fg * 8);
    }
    const long stat = (long)(seq * a.H + h) * a.LTq + qrow_c;
    float delta_q;
    if (a.o_in != nullptr) {  // delta of this row from the dO fragments already in registers and the matching O fragments
      const bf16_t* op = a.o_in + (long)(q_off + qrow_c) * a.ld_oin + h * D;
      float part = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 of = *reinterpret_cast<const bf16x8*>(op + ks * 32 + fg * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) part += bf2f((bf16_t)dof[ks][e]) * bf2f((bf16_t)of[e]);
      }
      delta_q = group_sum(part);
      if (fg == 0 && qrow < q_len) a.delta_w[stat] = delta_q;
    } else {
      delta_q = a.delta[stat];
    }
    const float lse2 = a.lse[stat] * 1.4426950408889634f, c2 = a.scale * 1.4426950408889634f;
    DIAG_ACC(0);
    f32x4 dq[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i) dq[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int kv_end = kv_len;
    if (CAUSAL) kv_end = max(0, min(kv_len, g * 16 + 16 + coff));
    const int ntiles = (kv_end + 63) >> 6;
    for (int j = 0; j < ntiles; ++j) {
      const char* kt = lds_k + j * TILE;
      const char* vt = lds_v + j * TILE;
      f32x4 s[4], dp[4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        s[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          s[nb] = MFMA(frag_rm<D>(kt, nb * 16 + fr, ks, fg), qf[ks], s[nb]);
          dp[nb] = MFMA(frag_rm<D>(vt, nb * 16 + fr, ks, fg), dof[ks], dp[nb]);
        }
      }
      TrAddr<D> tj;
#pragma unroll
      for (int db = 0; db < DB; ++db) tj.base[db] = tk.base[db] + j * TILE;
      bf16x4 k0lo[DB], k0hi[DB], k1lo[DB], k1hi[DB];
      tr_issue<D, 0>(tj, k0lo, k0hi);
      // VALU diet (see the forward kernel): P = exp2(s * c2 - lse * log2 e) is one fma + v_exp_f32, dS / scale = P * (dP - delta) (the
      // scale goes on the finished dQ rows), and only a tile on the causal diagonal / past the last key computes the mask.  Rows beyond q_len compute
      // garbage in their own lanes and are not stored.
      const bool full = (j + 1) * 64 <= kv_len && (!CAUSAL || j * 64 + 63 <= g * 16 + coff);
      if (full) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[nb][r], c2, -lse2));
            s[nb][r] = pv * (dp[nb][r] - delta_q);
          }
      } else {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = j * 64 + nb * 16 + fg * 4 + r;
            const bool ok = key < kv_len && (!CAUSAL || key <= qrow + coff) && qrow < q_len;
            const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[nb][r], c2, -lse2));
            s[nb][r] = ok ? pv * (dp[nb][r] - delta_q) : 0.f;
          }
      }
      const bf16x8 d0 = pack_frag(s[0], s[1]), d1 = pack_frag(s[2], s[3]);
      tr_wait<DB>(k0lo, k0hi);
      tr_issue<D, 1>(tj, k1lo, k1hi);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int db = 0; db < DB; ++db) dq[db] = MFMA(join(k0lo[db], k0hi[db]), d0, dq[db]);
      tr_wait<DB>(k1lo, k1hi);
#pragma unroll
      for (int db = 0; db < DB; ++db) dq[db] = MFMA(join(k1lo[db], k1hi[db]), d1, dq[db]);
    }
    DIAG_ACC(1);
    DIAG_UNITS(ntiles);
    if (qrow < q_len) {
#pragma unroll
      for (int db = 0; db < DB; ++db) dq[db] *= a.scale;
      if (a.rope_cos != nullptr) rope_bwd_inplace<DB>(dq, a, (long)q_off + qrow, fg);
      bf16_t* pq = a.dq + (long)(q_off + qrow) * a.ld_dq + h * D;
      if (a.wide) store_row_wide<DB>(pq, dq, 1.f, fg);
      else {
#pragma unroll
        for (int db = 0; db < DB; ++db) store4bf(pq + fg * 4 + db * 16, dq[db], 1.f);
      }
    }
    DIAG_ACC(2);
  }
  DIAG_T(4);
  DIAG_FLUSH();
}

template <int D, bool CAUSAL>
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv_res_kernel(AttnArgs a) {
  constexpr int KS = D / 32, DB = D / 16, TILE = 64 * D * 2;
  __shared__ __attribute__((aligned(16))) char smem[163840];
  char* lds_q = smem;
  char* lds_do = smem + res_rows<D>() * D * 2;
  const int seq = blockIdx.y, h = blockIdx.x;
  const int* ds = a.desc + seq * 8;
  const int q_off = ds[0], q_len = ds[1], kv_off = ds[2], kv_len = ds[3], kv_rows = ds[4], coff = ds[5];
  const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  DIAG_DECL(2);
  DIAG_T(0);
  const int nq_tiles = (q_len + 63) >> 6;
  // Round 6.  (1) A last query tile with at most 32 rows (S = 273: 17) is a HALF tile: its rows 32..63 are neither loaded nor multiplied (their P and dS are zero:
  // skipping the products changes no bit).  (2) The space that frees behind the Q image holds the row statistics lse | delta of the whole sequence (2 x LTq floats):
  // round 5 read them from L2 / HBM in front of every tile - a dependent global round trip of 1-2 us per tile on a loaded chip, 2.8 us per tile against the forward
  // kernel's 0.9 (tools/attn_diag.py) - now they are LDS reads behind the same barrier as the operands, read where they are used.  The host sends a sequence whose
  // image leaves no room for them (more than 288 queries at head_dim 128) to the tiled kernel (dkv_res_fits).
  const bool half_last = nq_tiles > 0 && q_len - (nq_tiles - 1) * 64 <= 32;
  const int rows_alloc = nq_tiles * 64 - (half_last ? 32 : 0);
  float* lds_lse = reinterpret_cast<float*>(smem + rows_alloc * D * 2);
  float* lds_delta = lds_lse + a.LTq;
  const float* lsebase = a.lse + (long)(seq * a.H + h) * a.LTq;
  const float* deltabase = a.delta + (long)(seq * a.H + h) * a.LTq;
  res_load<D>(lds_q, a.q + (long)q_off * a.ldq + h * D, a.ldq, rows_alloc, q_len, lane, wave);
  res_load<D>(lds_do, a.dout + (long)q_off * a.ld_do + h * D, a.ld_do, rows_alloc, q_len, lane, wave);
  for (int t = tid; t < a.LTq; t += 512) { lds_lse[t] = lsebase[t]; lds_delta[t] = deltabase[t]; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  DIAG_T(1);
  TrAddr<D> tq, tdo;
  tq.init(lds_q, lane);
  tdo.init(lds_do, lane);
  const int G = (kv_rows + 15) >> 4;
  for (int p = 0;; ++p) {
    const int g = zigzag_group(p, wave, G, !CAUSAL);  // causal: low key groups see the most query tiles
    if (g < 0) break;
    DIAG_MARK();
    const int key = g * 16 + fr;
    const int key_c = min(key, kv_rows - 1);
    const bf16_t* kp = a.k + (long)(kv_off + key_c) * a.ldk + h * D;
    const bf16_t* vp = a.v + (long)(kv_off + key_c) * a.ldv + h * D;
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = *reinterpret_cast<const bf16x8*>(kp + ks * 32 + fg * 8);
      vf[ks] = *reinterpret_cast<const bf16x8*>(vp + ks * 32 + fg * 8);
    }
    f32x4 dk[DB], dv[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i) { dk[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    int i0 = 0;
    if (CAUSAL) i0 = max(0, (g * 16 - coff) >> 6);
    DIAG_ACC(0);
    DIAG_UNITS(nq_tiles - i0);
    for (int i = i0; i < nq_tiles; ++i) {
      const bool hf = half_last && i == nq_tiles - 1;  // the half tile: query blocks 0, 1 and k-step 0 only
      const char* qt = lds_q + i * TILE;
      const char* dot = lds_do + i * TILE;
      f32x4 s[4], dp[4];
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        s[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          s[qb] = MFMA(frag_rm<D>(qt, qb * 16 + fr, ks, fg), kf[ks], s[qb]);
          dp[qb] = MFMA(frag_rm<D>(dot, qb * 16 + fr, ks, fg), vf[ks], dp[qb]);
        }
      }
      if (!hf) {
#pragma unroll
        for (int qb = 2; qb < 4; ++qb) {
          s[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
          dp[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            s[qb] = MFMA(frag_rm<D>(qt, qb * 16 + fr, ks, fg), kf[ks], s[qb]);
            dp[qb] = MFMA(frag_rm<D>(dot, qb * 16 + fr, ks, fg), vf[ks], dp[qb]);
          }
        }
      }
      TrAddr<D> tqi, tdi;
#pragma unroll
      for (int db = 0; db < DB; ++db) { tqi.base[db] = tq.base[db] + i * TILE; tdi.base[db] = tdo.base[db] + i * TILE; }
      bf16x4 alo[DB], ahi[DB];
      // only a tile on the causal diagonal, past the last query or past the last key computes the mask (a wave-uniform test: this wave's keys
      // are g*16 .. g*16+15): the 16 index computations + compares + selects were a third of the loop's VALU work
      const bool full = (i + 1) * 64 <= q_len && g * 16 + 15 < kv_len && (!CAUSAL || g * 16 + 15 <= i * 64 + coff);
      auto probs = [&](const int qb) {  // s[qb] <- P, dp[qb] <- dS of query block qb; its row statistics come out of the LDS here
        const f32x4 lq = *reinterpret_cast<const f32x4*>(lds_lse + i * 64 + qb * 16 + fg * 4);
        const f32x4 dl = *reinterpret_cast<const f32x4*>(lds_delta + i * 64 + qb * 16 + fg * 4);
        if (full) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = __expf(s[qb][r] * a.scale - lq[r]);
            dp[qb][r] = pv * (dp[qb][r] - dl[r]) * a.scale;
            s[qb][r] = pv;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int q = i * 64 + qb * 16 + fg * 4 + r;
            const bool ok = q < q_len && key < kv_len && (!CAUSAL || key <= q + coff);
            const float pv = ok ? __expf(s[qb][r] * a.scale - lq[r]) : 0.f;
            dp[qb][r] = ok ? pv * (dp[qb][r] - dl[r]) * a.scale : 0.f;
            s[qb][r] = pv;
          }
        }
      };
      probs(0); probs(1);
      // one register set: the next transposed read goes out right behind the MFMAs that consume the previous one (an MFMA has read its
      // operands long before an LDS read returns) and its latency runs under those eight MFMAs
      const bf16x8 p0 = pack_frag(s[0], s[1]), d0 = pack_frag(dp[0], dp[1]);
      tr_issue<D, 0>(tdi, alo, ahi);
      tr_wait<DB>(alo, ahi);
#pragma unroll
      for (int db = 0; db < DB; ++db) dv[db] = MFMA(join(alo[db], ahi[db]), p0, dv[db]);
      __builtin_amdgcn_sched_barrier(0);
      tr_issue<D, 0>(tqi, alo, ahi);
      tr_wait<DB>(alo, ahi);
#pragma unroll
      for (int db = 0; db < DB; ++db) dk[db] = MFMA(join(alo[db], ahi[db]), d0, dk[db]);
      __builtin_amdgcn_sched_barrier(0);
      if (!hf) {
        probs(2); probs(3);
        const bf16x8 p1 = pack_frag(s[2], s[3]), d1 = pack_frag(dp[2], dp[3]);
        tr_issue<D, 1>(tdi, alo, ahi);
        tr_wait<DB>(alo, ahi);
#pragma unroll
        for (int db = 0; db < DB; ++db) dv[db] = MFMA(join(alo[db], ahi[db]), p1, dv[db]);
        __builtin_amdgcn_sched_barrier(0);
        tr_issue<D, 1>(tqi, alo, ahi);
        tr_wait<DB>(alo, ahi);
#pragma unroll
        for (int db = 0; db < DB; ++db) dk[db] = MFMA(join(alo[db], ahi[db]), d1, dk[db]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    RopeRow<DB> rr;
    const bool rope = a.rope_cos != nullptr;
    if (rope) rope_row_load<DB>(rr, a, (long)kv_off + key_c, fg);
    DIAG_ACC(1);
    if (key < kv_rows) {
      if (rope) rope_row_apply<DB>(dk, rr);
      bf16_t* pk = a.dk + (long)(kv_off + key) * a.ld_dk + h * D;
      bf16_t* pv = a.dv + (long)(kv_off + key) * a.ld_dv + h * D;
      if (a.wide) { store_row_wide<DB>(pk, dk, 1.f, fg); store_row_wide<DB>(pv, dv, 1.f, fg); }
      else {
#pragma unroll
        for (int db = 0; db < DB; ++db) { store4bf(pk + fg * 4 + db * 16, dk[db], 1.f); store4bf(pv + fg * 4 + db * 16, dv[db], 1.f); }
      }
    }
    DIAG_ACC(2);
  }
  DIAG_T(4);
  DIAG_FLUSH();
}

// ---------------------------------------------------------------- delta = rowsum(dO * O)
template <int D>
__global__ void attn_delta_kernel(const bf16_t* __restrict__ o, long ldo, const bf16_t* __restrict__ dout, long ld_do,
                                  float* __restrict__ delta, const int* __restrict__ desc, int H, int LTq) {
  constexpr int TPR = D / 8;  // threads per (token, head)
  const int seq = blockIdx.z, h = blockIdx.y;
  const int q_off = desc[seq * 8 + 0], q_len = desc[seq * 8 + 1];
  const int t = blockIdx.x * (256 / TPR) + threadIdx.x / TPR;
  const int c = threadIdx.x % TPR;
  float s = 0.f;
  if (t < q_len) {
    const uint4 a = *reinterpret_cast<const uint4*>(o + (long)(q_off + t) * ldo + h * D + c * 8);
    const uint4 b = *reinterpret_cast<const uint4*>(dout + (long)(q_off + t) * ld_do + h * D + c * 8);
    s = bflo(a.x) * bflo(b.x) + bfhi(a.x) * bfhi(b.x) + bflo(a.y) * bflo(b.y) + bfhi(a.y) * bfhi(b.y) +
        bflo(a.z) * bflo(b.z) + bfhi(a.z) * bfhi(b.z) + bflo(a.w) * bflo(b.w) + bfhi(a.w) * bfhi(b.w);
  }
#pragma unroll
  for (int off = TPR / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (t < q_len && c == 0) delta[(long)(seq * H + h) * LTq + t] = s;
}

// ---------------------------------------------------------------- per-sequence transpose with zero padding
// out[seq][c][t] = in[off[seq] + t][c]  (t < len[seq]) else 0,  t in [0, LT)
__global__ __launch_bounds__(256) void seq_transpose_kernel(const bf16_t* __restrict__ in, long ld_in, bf16_t* __restrict__ out,
                                                            int cols, int LT, const int* __restrict__ desc, int off_idx,
                                                            int len_idx) {
  __shared__ bf16_t tile[64][66];
  const int seq = blockIdx.z;
  const int off = desc[seq * 8 + off_idx], len = desc[seq * 8 + len_idx];
  const int t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int t = t0 + r, c = c0 + tx;
    tile[r][tx] = (t < len && c < cols) ? in[(long)(off + t) * ld_in + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r, t = t0 + tx;
    if (c < cols && t < LT) out[((long)seq * cols + c) * LT + t] = tile[tx][r];
  }
}

int check_common(const AttnArgs& a, int D, int nseq, const char* who) {
  LHRS_REQUIRE(D == 64 || D == 128, "%s: head_dim %d unsupported (64 or 128)", who, D);
  LHRS_REQUIRE(nseq > 0 && a.H > 0, "%s: nseq=%d H=%d", who, nseq, a.H);
  LHRS_REQUIRE(a.LTq % 64 == 0, "%s: LTq must be a multiple of 64", who);
  return 0;
}

}  // namespace

// C ABI ------------------------------------------------------------------------------------------
#ifdef ATTN_DIAG
extern "C" int lhrs_attn_set_dbg(void* p, int sel) {
  unsigned long long* v = (unsigned long long*)p;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_attn_dbg_sel), &sel, sizeof(sel)) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_dbg), &v, sizeof(v)) == hipSuccess ? 0 : -1;
}
#endif
extern "C" int lhrs_seq_transpose(const void* in, long ld_in, void* out, int cols, int LT, const int* desc, int nseq,
                                  int use_kv, void* stream) {
  LHRS_REQUIRE(LT % 64 == 0 && cols > 0 && nseq > 0, "seq_transpose: LT=%d cols=%d nseq=%d", LT, cols, nseq);
  hipLaunchKernelGGL(seq_transpose_kernel, dim3(LT / 64, cdiv(cols, 64), nseq), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)in, ld_in, (bf16_t*)out, cols, LT, desc, use_kv ? 2 : 0, use_kv ? (use_kv == 2 ? 4 : 3) : 1);
  LHRS_CHECK_LAUNCH("seq_transpose");
  return 0;
}

static int attn_fwd_impl(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo,
                         float* lse, const int* desc, int nseq, int H, int D, int max_q, int max_kv, int LTq,
                         int causal, float scale, const unsigned char* key_mask, long ld_mask, void* stream) {
  AttnArgs a; memset(&a, 0, sizeof(a));
  a.kmask = key_mask; a.ld_kmask = ld_mask;
  if (key_mask != nullptr) max_kv = 1 << 30;  // the mask lives in the tiled kernel only
  a.q = (const bf16_t*)q; a.ldq = ldq; a.k = (const bf16_t*)k; a.ldk = ldk; a.v = (const bf16_t*)v; a.ldv = ldv;
  a.o = (bf16_t*)o; a.ldo = ldo; a.lse = lse; a.desc = desc; a.H = H; a.LTq = LTq; a.scale = scale;
  a.wide = (ldo % 8 == 0 && (size_t)o % 16 == 0) ? 1 : 0;
  if (check_common(a, D, nseq, "attn_fwd")) return -1;
  const dim3 grid(cdiv(max_q, 64), H, nseq), blk(256);
  hipStream_t s = (hipStream_t)stream;
  if (max_kv > 0 && max_kv <= (D == 128 ? res_rows<128>() : res_rows<64>())) {  // both operands fit one CU's LDS
    const dim3 rg(H, nseq), rb(512);
    if (D == 128) {
      if (causal) hipLaunchKernelGGL((attn_fwd_res_kernel<128, true>), rg, rb, 0, s, a);
      else hipLaunchKernelGGL((attn_fwd_res_kernel<128, false>), rg, rb, 0, s, a);
    } else {
      if (causal) hipLaunchKernelGGL((attn_fwd_res_kernel<64, true>), rg, rb, 0, s, a);
      else hipLaunchKernelGGL((attn_fwd_res_kernel<64, false>), rg, rb, 0, s, a);
    }
    LHRS_CHECK_LAUNCH("attn_fwd_res");
    return 0;
  }
  if (D == 128) {
    if (causal) hipLaunchKernelGGL((attn_fwd_kernel<128, true>), grid, blk, 0, s, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<128, false>), grid, blk, 0, s, a);
  } else {
    if (causal) hipLaunchKernelGGL((attn_fwd_kernel<64, true>), grid, blk, 0, s, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<64, false>), grid, blk, 0, s, a);
  }
  LHRS_CHECK_LAUNCH("attn_fwd");
  return 0;
}

extern "C" int lhrs_attn_fwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo,
                             float* lse, const int* desc, int nseq, int H, int D, int max_q, int max_kv, int LTq,
                             int causal, float scale, void* stream) {
  return attn_fwd_impl(q, ldq, k, ldk, v, ldv, o, ldo, lse, desc, nseq, H, D, max_q, max_kv, LTq, causal, scale, nullptr, 0, stream);
}

// forward with an HF-style attention_mask: key j of sequence s is visible iff key_mask[s * ld_mask + j] != 0 (AND causal, AND
// j < kv_len).  A query row with no visible key returns 0 (HF would return the mean of V: such rows are pad positions).
extern "C" int lhrs_attn_fwd_kmask(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo,
                                   float* lse, const int* desc, int nseq, int H, int D, int max_q, int max_kv, int LTq,
                                   int causal, float scale, const unsigned char* key_mask, long ld_mask, void* stream) {
  LHRS_REQUIRE(key_mask != nullptr && ld_mask > 0, "attn_fwd_kmask: key_mask=%p ld_mask=%ld", (const void*)key_mask, ld_mask);
  return attn_fwd_impl(q, ldq, k, ldk, v, ldv, o, ldo, lse, desc, nseq, H, D, max_q, max_kv, LTq, causal, scale, key_mask, ld_mask, stream);
}

extern "C" int lhrs_attn_delta(const void* o, long ldo, const void* dout, long ld_do, float* delta, const int* desc,
                               int nseq, int H, int D, int max_q, int LTq, void* stream) {
  LHRS_REQUIRE(D == 64 || D == 128, "attn_delta: head_dim %d", D);
  const int rows_per_blk = 256 / (D / 8);
  const dim3 grid(cdiv(max_q, rows_per_blk), H, nseq);
  if (D == 128)
    hipLaunchKernelGGL((attn_delta_kernel<128>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)o, ldo,
                       (const bf16_t*)dout, ld_do, delta, desc, H, LTq);
  else
    hipLaunchKernelGGL((attn_delta_kernel<64>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)o, ldo,
                       (const bf16_t*)dout, ld_do, delta, desc, H, LTq);
  LHRS_CHECK_LAUNCH("attn_delta");
  return 0;
}

extern "C" int lhrs_rope(void* x, long ld, int rows, int nheads, int D, const float* cos_t, const float* sin_t, const int* pos_ids, int pos_mod,
                         int pos0, int inverse, void* stream);

static int attn_bwd_impl(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                         const void* dout, long ld_do, const float* lse, const float* delta, void* dq, long ld_dq,
                         void* dk, long ld_dk, void* dv, long ld_dv, const int* desc, int nseq, int H, int D,
                         int max_q, int max_kv, int LTq, int causal, float scale, const float* rope_cos, const float* rope_sin,
                         int rope_mod, int rope_pos0, long rope_rows, void* stream, const void* o = nullptr, long ldo = 0) {
  AttnArgs a; memset(&a, 0, sizeof(a));
  if (o != nullptr) {
    // delta is an OUTPUT here: by the resident dQ kernel when it runs (it is launched in front of the dK/dV kernel), by a launch of the
    // delta kernel otherwise
    const int rmax0 = D == 128 ? res_rows<128>() : res_rows<64>();
    if (max_kv <= rmax0) { a.o_in = (const bf16_t*)o; a.ld_oin = ldo; a.delta_w = const_cast<float*>(delta); }
    else if (lhrs_attn_delta(o, ldo, dout, ld_do, const_cast<float*>(delta), desc, nseq, H, D, max_q, LTq, stream)) return -1;
  }
  const bool rope = rope_cos != nullptr;
  const int rmax_ = D == 128 ? res_rows<128>() : res_rows<64>();
  // the resident dK/dV kernel keeps lse | delta of the sequence behind its Q image (rows rounded up to 32): head_dim 128 up to 288 queries
  const bool dkv_fits = max_q <= rmax_ && (long)((max_q + 31) / 32 * 32) * D * 2 + 2L * LTq * 4 <= (long)rmax_ * D * 2;
  const bool fuse_rope = rope && max_kv <= rmax_ && dkv_fits;  // both resident kernels run: the rotation rides in their stores
  if (fuse_rope) { a.rope_cos = rope_cos; a.rope_sin = rope_sin; a.rope_mod = rope_mod; a.rope_pos0 = rope_pos0; }
  a.q = (const bf16_t*)q; a.ldq = ldq; a.k = (const bf16_t*)k; a.ldk = ldk; a.v = (const bf16_t*)v; a.ldv = ldv;
  a.dout = (const bf16_t*)dout; a.ld_do = ld_do; a.lse = (float*)lse; a.delta = delta;
  a.dq = (bf16_t*)dq; a.ld_dq = ld_dq; a.dk = (bf16_t*)dk; a.ld_dk = ld_dk; a.dv = (bf16_t*)dv; a.ld_dv = ld_dv;
  a.desc = desc; a.H = H; a.LTq = LTq; a.scale = scale;
  a.wide = (ld_dq % 8 == 0 && ld_dk % 8 == 0 && ld_dv % 8 == 0 && ((size_t)dq | (size_t)dk | (size_t)dv) % 16 == 0) ? 1 : 0;
  if (check_common(a, D, nseq, "attn_bwd")) return -1;
  hipStream_t s = (hipStream_t)stream;
  const dim3 gq(cdiv(max_q, 64), H, nseq), gk(cdiv(max_kv, 64), H, nseq), blk(256);
  const int rmax = D == 128 ? res_rows<128>() : res_rows<64>();
  const dim3 rg(H, nseq), rb(512);
  bool dq_done = false, dkv_done = false;
  if (max_kv <= rmax) {
    if (D == 128) { if (causal) hipLaunchKernelGGL((attn_bwd_dq_res_kernel<128, true>), rg, rb, 0, s, a); else hipLaunchKernelGGL((attn_bwd_dq_res_kernel<128, false>), rg, rb, 0, s, a); }
    else { if (causal) hipLaunchKernelGGL((attn_bwd_dq_res_kernel<64, true>), rg, rb, 0, s, a); else hipLaunchKernelGGL((attn_bwd_dq_res_kernel<64, false>), rg, rb, 0, s, a); }
    dq_done = true;
  }
  if (dkv_fits) {
    if (D == 128) { if (causal) hipLaunchKernelGGL((attn_bwd_dkv_res_kernel<128, true>), rg, rb, 0, s, a); else hipLaunchKernelGGL((attn_bwd_dkv_res_kernel<128, false>), rg, rb, 0, s, a); }
    else { if (causal) hipLaunchKernelGGL((attn_bwd_dkv_res_kernel<64, true>), rg, rb, 0, s, a); else hipLaunchKernelGGL((attn_bwd_dkv_res_kernel<64, false>), rg, rb, 0, s, a); }
    dkv_done = true;
  }
  if (dq_done && dkv_done) { LHRS_CHECK_LAUNCH("attn_bwd_res"); return 0; }
  // (never reached with fuse_rope: it requires both resident kernels)
#define LAUNCH_TILED(KERN, GRID)                                                                 \
  do {                                                                                           \
    if (D == 128) { if (causal) hipLaunchKernelGGL((KERN<128, true>), GRID, blk, 0, s, a); else hipLaunchKernelGGL((KERN<128, false>), GRID, blk, 0, s, a); } \
    else { if (causal) hipLaunchKernelGGL((KERN<64, true>), GRID, blk, 0, s, a); else hipLaunchKernelGGL((KERN<64, false>), GRID, blk, 0, s, a); }          \
  } while (0)
  if (!dq_done) LAUNCH_TILED(attn_bwd_dq_kernel, gq);
  if (!dkv_done) LAUNCH_TILED(attn_bwd_dkv_kernel, gk);
#undef LAUNCH_TILED
  LHRS_CHECK_LAUNCH("attn_bwd");
  if (rope) {  // long sequences (tiled kernels): the rotation as its own pass over the dq and dk rows, same numbers
    if (lhrs_rope(dq, ld_dq, (int)rope_rows, H, D, rope_cos, rope_sin, nullptr, rope_mod, rope_pos0, 1, stream)) return -1;
    if (lhrs_rope(dk, ld_dk, (int)rope_rows, H, D, rope_cos, rope_sin, nullptr, rope_mod, rope_pos0, 1, stream)) return -1;
  }
  return 0;
}

extern "C" int lhrs_attn_bwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                             const void* dout, long ld_do, const float* lse, const float* delta, void* dq, long ld_dq,
                             void* dk, long ld_dk, void* dv, long ld_dv, const int* desc, int nseq, int H, int D,
                             int max_q, int max_kv, int LTq, int causal, float scale, void* stream) {
  return attn_bwd_impl(q, ldq, k, ldk, v, ldv, dout, ld_do, lse, delta, dq, ld_dq, dk, ld_dk, dv, ld_dv, desc, nseq, H, D, max_q, max_kv, LTq,
                       causal, scale, nullptr, nullptr, 1, 0, 0, stream);
}

// attention backward of ROTATED q / k (HF LlamaAttention: apply_rotary_pos_emb before the scores): dq and dk come out as gradients of the
// UN-rotated projections - the inverse rotation (position of token row m = m % pos_mod + pos0, fp32 cos / sin tables [pos][D / 2]) is
// applied where the rows are stored.  Bit-identical to lhrs_attn_bwd followed by lhrs_rope(inverse) on the dq and dk rows [0, rows).
extern "C" int lhrs_attn_bwd_rope(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                                  const void* dout, long ld_do, const float* lse, const float* delta, void* dq, long ld_dq,
                                  void* dk, long ld_dk, void* dv, long ld_dv, const int* desc, int nseq, int H, int D,
                                  int max_q, int max_kv, int LTq, int causal, float scale, const float* cos_t, const float* sin_t,
                                  int pos_mod, int pos0, long rows, void* stream) {
  LHRS_REQUIRE(cos_t && sin_t && pos_mod > 0 && rows > 0, "attn_bwd_rope: cos/sin tables, pos_mod=%d, rows=%ld", pos_mod, rows);
  return attn_bwd_impl(q, ldq, k, ldk, v, ldv, dout, ld_do, lse, delta, dq, ld_dq, dk, ld_dk, dv, ld_dv, desc, nseq, H, D, max_q, max_kv, LTq,
                       causal, scale, cos_t, sin_t, pos_mod, pos0, rows, stream);
}

// lhrs_attn_delta + lhrs_attn_bwd / lhrs_attn_bwd_rope as ONE call: o [tokens, ldo] are the forward output rows, delta [nseq][H][LTq] is
// WRITTEN (workspace).  When both operand matrices fit the LDS (every attention of the training path) the resident dQ kernel computes
// delta = rowsum(dO * O) for the rows it loads anyway - one launch and one pass over O and dO per layer less; cos_t == nullptr: no rotation.
extern "C" int lhrs_attn_bwd_o(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* dout, long ld_do,
                               const void* o, long ldo, const float* lse, float* delta, void* dq, long ld_dq, void* dk, long ld_dk,
                               void* dv, long ld_dv, const int* desc, int nseq, int H, int D, int max_q, int max_kv, int LTq, int causal,
                               float scale, const float* cos_t, const float* sin_t, int pos_mod, int pos0, long rows, void* stream) {
  LHRS_REQUIRE(o != nullptr && delta != nullptr, "attn_bwd_o: o=%p delta=%p", o, (void*)delta);
  LHRS_REQUIRE(cos_t == nullptr || (sin_t && pos_mod > 0 && rows > 0), "attn_bwd_o: cos/sin tables, pos_mod=%d, rows=%ld", pos_mod, rows);
  return attn_bwd_impl(q, ldq, k, ldk, v, ldv, dout, ld_do, lse, delta, dq, ld_dq, dk, ld_dk, dv, ld_dv, desc, nseq, H, D, max_q, max_kv, LTq,
                       causal, scale, cos_t, sin_t, cos_t ? pos_mod : 1, cos_t ? pos0 : 0, cos_t ? rows : 0, stream, o, ldo);
}
