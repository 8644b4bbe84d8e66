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

typedef __attribute__((ext_vector_type(2))) float f32x2_t;

// PRO: what the block does to the activation rows while staging them in LDS (every block redoes it: 8-22 KB from L2)
//   0 plain copy            1 RMSNorm (HF LlamaRMSNorm: w * bf16(x * rsqrt(mean x^2 + eps)))       2 SwiGLU: silu(x[k]) * x[K + k]
// FP8: W is OCP e4m3 with one fp32 scale per output row (the 6.7 GB / token weight stream of SURVEY.md §8d)
// RPW rows per wave; UNR 1-KiB row chunks per lane-iteration (bf16 weights): RPW * UNR 16-B loads per lane are prefetched one iteration
// ahead, so a wave keeps RPW * UNR .. 2 * RPW * UNR KiB in flight.  The per-row summation order does not depend on RPW / UNR.
template <int NB, int PRO, bool FP8, int RPW = 4, int UNR = 1>
__global__ __launch_bounds__(256) void gemv_kernel(const void* __restrict__ Wv, long ldw, const float* __restrict__ wscale,
                                                   const bf16_t* __restrict__ x, long ldx, const bf16_t* __restrict__ norm_w, float eps,
                                                   const bf16_t* res, long ldr, void* y, long ldy, int N, int K, int out_f32) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float red[4];
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);  // [NB][K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nch = K / 8;
  const int row0 = (blockIdx.x * 4 + wave) * RPW;  // 4 waves x RPW rows per block
  // the first weight loads are issued BEFORE the activation prologue (which needs two block-wide syncs): HBM latency overlaps it
  uint4 wcur[UNR][RPW];
  if (!FP8) {
    const bf16_t* W = reinterpret_cast<const bf16_t*>(Wv);
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const i32x4 t = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(W + (long)min(row0 + r, N - 1) * ldw + min(lane + 64 * u, nch - 1) * 8));
        wcur[u][r] = make_uint4((unsigned)t[0], (unsigned)t[1], (unsigned)t[2], (unsigned)t[3]);
      }
  }
  for (int b = 0; b < NB; ++b) {
    if (PRO == 2) {
      for (int c = tid; c < nch; c += 256) {
        const uint4 g = *reinterpret_cast<const uint4*>(x + b * ldx + c * 8);
        const uint4 u = *reinterpret_cast<const uint4*>(x + b * ldx + K + c * 8);
        uint4 o;
        o.x = pack2bf(silu(bflo(g.x)) * bflo(u.x), silu(bfhi(g.x)) * bfhi(u.x));
        o.y = pack2bf(silu(bflo(g.y)) * bflo(u.y), silu(bfhi(g.y)) * bfhi(u.y));
        o.z = pack2bf(silu(bflo(g.z)) * bflo(u.z), silu(bfhi(g.z)) * bfhi(u.z));
        o.w = pack2bf(silu(bflo(g.w)) * bflo(u.w), silu(bfhi(g.w)) * bfhi(u.w));
        *reinterpret_cast<uint4*>(xs + b * K + c * 8) = o;
      }
    } else {
      float q = 0.f;
      for (int c = tid; c < nch; c += 256) {
        const uint4 v = *reinterpret_cast<const uint4*>(x + b * ldx + c * 8);
        *reinterpret_cast<uint4*>(xs + b * K + c * 8) = v;
        if (PRO == 1)
          q += bflo(v.x) * bflo(v.x) + bfhi(v.x) * bfhi(v.x) + bflo(v.y) * bflo(v.y) + bfhi(v.y) * bfhi(v.y) + bflo(v.z) * bflo(v.z) +
               bfhi(v.z) * bfhi(v.z) + bflo(v.w) * bflo(v.w) + bfhi(v.w) * bfhi(v.w);
      }
      if (PRO == 1) {
        const float rstd = rsqrtf(block_sum<4>(q, red) / (float)K + eps);
        __syncthreads();
        for (int c = tid; c < K; c += 256) xs[b * K + c] = f2bf(bf2f(norm_w[c]) * bf2f(f2bf(bf2f(xs[b * K + c]) * rstd)));
      }
    }
  }
  __syncthreads();
  float acc[RPW][NB];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;
  if (!FP8) {
    const bf16_t* W = reinterpret_cast<const bf16_t*>(Wv);
    for (int c = lane; c < nch; c += 64 * UNR) {  // software pipeline: the loads of the next UNR chunks fly while these are multiplied
      uint4 wnext[UNR][RPW];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int cn = min(c + 64 * (UNR + u), nch - 1);
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
          const i32x4 t = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(W + (long)min(row0 + r, N - 1) * ldw + cn * 8));
          wnext[u][r] = make_uint4((unsigned)t[0], (unsigned)t[1], (unsigned)t[2], (unsigned)t[3]);
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int cc = c + 64 * u;
        if (cc < nch) {
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const uint4 xv = *reinterpret_cast<const uint4*>(xs + b * K + cc * 8);
            const float x0 = bflo(xv.x), x1 = bfhi(xv.x), x2 = bflo(xv.y), x3 = bfhi(xv.y), x4 = bflo(xv.z), x5 = bfhi(xv.z), x6 = bflo(xv.w),
                        x7 = bfhi(xv.w);
#pragma unroll
            for (int r = 0; r < RPW; ++r)
              acc[r][b] += bflo(wcur[u][r].x) * x0 + bfhi(wcur[u][r].x) * x1 + bflo(wcur[u][r].y) * x2 + bfhi(wcur[u][r].y) * x3 +
                           bflo(wcur[u][r].z) * x4 + bfhi(wcur[u][r].z) * x5 + bflo(wcur[u][r].w) * x6 + bfhi(wcur[u][r].w) * x7;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u)
#pragma unroll
        for (int r = 0; r < RPW; ++r) wcur[u][r] = wnext[u][r];
    }
  } else {
    const uint8_t* W = reinterpret_cast<const uint8_t*>(Wv);
    const int nch16 = K / 16;  // 16 fp8 per lane load
    for (int c = lane; c < nch16; c += 64) {
      i32x4 w[RPW];
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const int row = min(row0 + r, N - 1);
        w[r] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(W + (long)row * ldw + c * 16));
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        float xf[16];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const uint4 xv = *reinterpret_cast<const uint4*>(xs + b * K + c * 16 + h * 8);
          xf[h * 8 + 0] = bflo(xv.x); xf[h * 8 + 1] = bfhi(xv.x); xf[h * 8 + 2] = bflo(xv.y); xf[h * 8 + 3] = bfhi(xv.y);
          xf[h * 8 + 4] = bflo(xv.z); xf[h * 8 + 5] = bfhi(xv.z); xf[h * 8 + 6] = bflo(xv.w); xf[h * 8 + 7] = bfhi(xv.w);
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8(w[r][j], false);
            const f32x2_t hi = __builtin_amdgcn_cvt_pk_f32_fp8(w[r][j], true);
            acc[r][b] += lo[0] * xf[j * 4] + lo[1] * xf[j * 4 + 1] + hi[0] * xf[j * 4 + 2] + hi[1] * xf[j * 4 + 3];
          }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float s = wave_sum(acc[r][b]);
      const int row = row0 + r;
      if (lane == 0 && row < N) {
        float v = FP8 ? s * wscale[row] : s;
        if (res) v += bf2f(res[b * ldr + row]);
        if (out_f32) reinterpret_cast<float*>(y)[b * ldy + row] = v;
        else reinterpret_cast<bf16_t*>(y)[b * ldy + row] = f2bf(v);
      }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// gemv_mfma_kernel: the same weight stream for 2..16 sequences at once (batched evaluation, main_vqa.py:205-214).  The per-weight
// VALU work of gemv_kernel grows with the batch (8 FMAs per weight element at B = 8: VALU-bound at ~3x the B = 1 rate); here the
// batch is the N dimension of v_mfma_f32_16x16x32_bf16, so every weight element still crosses the CU once and costs one MFMA lane.
//   block = 16 output rows; the 4 waves split K into 4 contiguous quarters (rows are read in 64-B pieces straight into the MFMA A
//   layout: lane (r, g) holds W[row0 + r][k0 + 8g .. +8]); x rows sit in LDS after the RMSNorm / SwiGLU prologue and feed the B
//   operand (batch columns >= B read zeros); the 4 partial 16x16 tiles are summed through LDS.
// ------------------------------------------------------------------------------------------------------------------
// PK: W is the copy lhrs_repack_bf16_mfma made - [N/16][K/32][64 lanes][8 bf16], the A-operand order - so every wave instruction reads
// 1 KiB of consecutive bytes and a wave's K quarter is one contiguous run (row-major: 16 rows x 64-B segments at the row stride).
// NW waves per block split K (8 when K % 256 == 0: the 4096-row projections are 256 blocks - one per CU - and need the loads of 8 waves in flight)
template <int PRO, bool PK, int NW>
__global__ __launch_bounds__(NW * 64) void gemv_mfma_kernel(const bf16_t* __restrict__ W, long ldw, const bf16_t* __restrict__ x, long ldx,
                                                        const bf16_t* __restrict__ norm_w, float eps, const bf16_t* res, long ldr, void* y,
                                                        long ldy, int NB, int N, int K, int out_f32) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float red[NW];
  __shared__ float part[NW][16][17];
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);  // [NB][K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
  const int nch = K / 8;
  const int row0 = blockIdx.x * 16;
  const int kq = K / NW;                     // this wave's slice of K (K % (32 NW) == 0)
  const bf16_t* wp = PK ? W + ((long)blockIdx.x * (K / 32) + wave * (kq / 32)) * 512 + lane * 8
                        : W + (long)min(row0 + fr, N - 1) * ldw + wave * kq + fg * 8;
  constexpr int WSTEP = PK ? 512 : 32;       // elements between consecutive 32-k steps of this lane
  constexpr int U = 4;                       // k-steps (of 32) in flight per wave
  i32x4 wreg[U];
#pragma unroll
  for (int u = 0; u < U; ++u) wreg[u] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp + u * WSTEP));  // kq >= 128
  for (int b = 0; b < (PRO == 0 ? 0 : NB); ++b) {  // PRO 0: x is read straight from L2 into the B fragments, nothing to stage
    if (PRO == 2) {
      for (int c = tid; c < nch; c += NW * 64) {
        const uint4 g = *reinterpret_cast<const uint4*>(x + b * ldx + c * 8);
        const uint4 u = *reinterpret_cast<const uint4*>(x + b * ldx + K + c * 8);
        uint4 o;
        o.x = pack2bf(silu(bflo(g.x)) * bflo(u.x), silu(bfhi(g.x)) * bfhi(u.x));
        o.y = pack2bf(silu(bflo(g.y)) * bflo(u.y), silu(bfhi(g.y)) * bfhi(u.y));
        o.z = pack2bf(silu(bflo(g.z)) * bflo(u.z), silu(bfhi(g.z)) * bfhi(u.z));
        o.w = pack2bf(silu(bflo(g.w)) * bflo(u.w), silu(bfhi(g.w)) * bfhi(u.w));
        *reinterpret_cast<uint4*>(xs + b * K + c * 8) = o;
      }
    } else {
      float q = 0.f;
      for (int c = tid; c < nch; c += NW * 64) {
        const uint4 v = *reinterpret_cast<const uint4*>(x + b * ldx + c * 8);
        *reinterpret_cast<uint4*>(xs + b * K + c * 8) = v;
        if (PRO == 1)
          q += bflo(v.x) * bflo(v.x) + bfhi(v.x) * bfhi(v.x) + bflo(v.y) * bflo(v.y) + bfhi(v.y) * bfhi(v.y) + bflo(v.z) * bflo(v.z) +
               bfhi(v.z) * bfhi(v.z) + bflo(v.w) * bflo(v.w) + bfhi(v.w) * bfhi(v.w);
      }
      if (PRO == 1) {
        const float rstd = rsqrtf(block_sum<NW>(q, red) / (float)K + eps);
        __syncthreads();
        for (int c = tid; c < K; c += NW * 64) xs[b * K + c] = f2bf(bf2f(norm_w[c]) * bf2f(f2bf(bf2f(xs[b * K + c]) * rstd)));
      }
    }
  }
  __syncthreads();
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const bf16_t* xq = (PRO == 0 ? x + (long)min(fr, NB - 1) * ldx : xs + (long)min(fr, NB - 1) * K) + wave * kq + fg * 8;
  const bool live = fr < NB;
  const int nsteps = kq / 32;
  for (int s0 = 0; s0 < nsteps; s0 += U) {
    i32x4 wnext[U];
    bf16x8 xfrag[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int sn = min(s0 + U + u, nsteps - 1);
      wnext[u] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp + (long)sn * WSTEP));
      xfrag[u] = *reinterpret_cast<const bf16x8*>(xq + min(s0 + u, nsteps - 1) * 32);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (s0 + u < nsteps) {
        bf16x8 bfrag = xfrag[u];
        if (!live) bfrag = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wreg[u]), bfrag, acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) wreg[u] = wnext[u];
  }
  // acc[r] = partial y[batch = fr][row0 + 4*fg + r] of this wave's K quarter
#pragma unroll
  for (int r = 0; r < 4; ++r) part[wave][fg * 4 + r][fr] = acc[r];
  __syncthreads();
  const int i = tid >> 4, b = tid & 15;  // the first 256 threads = 16 rows x 16 batch columns
  const int row = row0 + i;
  if (tid < 256 && b < NB && row < N) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += part[w][i][b];
    if (res) v += bf2f(res[b * ldr + row]);
    if (out_f32) reinterpret_cast<float*>(y)[b * ldy + row] = v;
    else reinterpret_cast<bf16_t*>(y)[b * ldy + row] = f2bf(v);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// gemv_fp8_mfma_kernel: e4m3 weights x e4m3 activations on the block-scaled MFMA (BASELINE configs[4]: "fp8 MFMA weights").
// y[b][n] = sx[b] * sw[n] * sum_k x8[b][k] * W8[n][k] (+ residual) for 1..16 sequences.  16 output rows per block; the eight waves
// split the 128-k steps of K, four steps (8 KiB per wave) in flight, so that even the 4096-row projections (256 blocks) keep
// ~16 MB outstanding chip-wide.  Lane (r, g) holds bytes [16g, 16g+16) and [64+16g, 64+16g+16) of row r of a step - a k permutation
// applied to BOTH operands of v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales), which makes every load instruction read whole
// 64-B segments of the 16 rows.  The VALU e4m3 GEMV spends ~2.5 instructions per weight byte; this one is one MFMA per 2 KiB.
// PK: the weights were re-tiled by lhrs_repack_fp8_mfma into [N/16][K/128][2][64 lanes][16 B] - the operand order itself - so a wave
// instruction reads 1 KiB of consecutive bytes and a wave's k-slice is one contiguous run (row-strided 64-B segments reach ~3.5 TB/s,
// and a 4096-B row stride additionally camps on few channels; the tiled stream matches the bf16 GEMV's contiguous rows).
// ------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
constexpr int GF_WAVES = 8, GF_THREADS = GF_WAVES * 64, GF_MAXC = 3;  // fused prologue: K <= GF_THREADS * GF_MAXC * 8 = 12288

// PRO < 0: x8 / xscale come from global memory (already quantised; any batch <= 16).
// PRO 0 / 1 / 2 (batch <= 2): x is bf16; the block applies the prologue (copy / RMSNorm / SwiGLU) AND the per-row e4m3 quantisation
// itself while the first weight lines are in flight - the row stays in registers between the reductions, only the e4m3 bytes go to
// LDS.  Every block recomputes the same few KB, which keeps the decode step at five launches per layer.
template <int PRO, bool PK>
__global__ __launch_bounds__(GF_THREADS) void gemv_fp8_mfma_kernel(const uint8_t* __restrict__ W, long ldw, const float* __restrict__ wscale,
                                                                   const void* __restrict__ xin, long ldx, const float* __restrict__ xscale,
                                                                   const bf16_t* __restrict__ norm_w, float eps, const bf16_t* res, long ldr,
                                                                   void* y, long ldy, int NB, int N, int K, int out_f32) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // PRO >= 0: [NB][K] e4m3 bytes
  __shared__ float part[GF_WAVES][16][17];
  __shared__ float red[GF_WAVES];
  __shared__ float s_scale[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
  const int row0 = blockIdx.x * 16;
  const int nsteps = K / 128, per = (nsteps + GF_WAVES - 1) / GF_WAVES;
  const int s_begin = min(wave * per, nsteps), s_end = min(nsteps, s_begin + per);
  const uint8_t* wp = PK ? W + (long)blockIdx.x * nsteps * 2048 + lane * 16 : W + (long)min(row0 + fr, N - 1) * ldw + fg * 16;
  constexpr long WSTEP = PK ? 2048 : 128, WHALF = PK ? 1024 : 64;
  constexpr int U = 4;  // 128-k steps in flight per wave (2 x 16 B of weights per lane each)
  i32x4 wlo[U], whi[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {  // first weight lines before the prologue
    const long o = (long)min(s_begin + u, max(s_end - 1, 0)) * WSTEP;
    wlo[u] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp + o));
    whi[u] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp + o + WHALF));
  }
  const uint8_t* xp;
  if (PRO >= 0) {
    uint8_t* x8s = reinterpret_cast<uint8_t*>(smem);
    const bf16_t* x = reinterpret_cast<const bf16_t*>(xin);
    const int nch = K / 8;
    for (int b = 0; b < NB; ++b) {
      float v[GF_MAXC][8], g[GF_MAXC][8];
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < GF_MAXC; ++i) {
        const int c = tid + i * GF_THREADS;
        if (c < nch) {
          uint4 t = *reinterpret_cast<const uint4*>(x + b * ldx + c * 8);
          if (PRO == 2) {
            const uint4 u = *reinterpret_cast<const uint4*>(x + b * ldx + K + c * 8);
            t.x = pack2bf(silu(bflo(t.x)) * bflo(u.x), silu(bfhi(t.x)) * bfhi(u.x));
            t.y = pack2bf(silu(bflo(t.y)) * bflo(u.y), silu(bfhi(t.y)) * bfhi(u.y));
            t.z = pack2bf(silu(bflo(t.z)) * bflo(u.z), silu(bfhi(t.z)) * bfhi(u.z));
            t.w = pack2bf(silu(bflo(t.w)) * bflo(u.w), silu(bfhi(t.w)) * bfhi(u.w));
          }
          unpack8(t, v[i]);
          if (PRO == 1) {
            unpack8(*reinterpret_cast<const uint4*>(norm_w + c * 8), g[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) q += v[i][e] * v[i][e];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
        }
      }
      if (PRO == 1) {
        const float rstd = rsqrtf(block_sum<GF_WAVES>(q, red) / (float)K + eps);
#pragma unroll
        for (int i = 0; i < GF_MAXC; ++i)
          if (tid + i * GF_THREADS < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = bf2f(f2bf(g[i][e] * bf2f(f2bf(v[i][e] * rstd))));  // HF LlamaRMSNorm roundings
          }
      }
      float m = 0.f;
#pragma unroll
      for (int i = 0; i < GF_MAXC; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[i][e]));
      m = block_max<GF_WAVES>(m, red);
      const float sc = m > 0.f ? m / 448.f : 1.f;
      if (tid == 0) s_scale[b] = sc;
      const float inv = 1.f / sc;
#pragma unroll
      for (int i = 0; i < GF_MAXC; ++i) {
        const int c = tid + i * GF_THREADS;
        if (c < nch) {
          int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][0] * inv, v[i][1] * inv, 0, false);
          lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][2] * inv, v[i][3] * inv, lo, true);
          int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][4] * inv, v[i][5] * inv, 0, false);
          hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][6] * inv, v[i][7] * inv, hi, true);
          *reinterpret_cast<int2*>(x8s + (size_t)b * K + c * 8) = make_int2(lo, hi);
        }
      }
    }
    __syncthreads();
    xp = x8s + (size_t)min(fr, NB - 1) * K + fg * 16;
  } else {
    xp = reinterpret_cast<const uint8_t*>(xin) + (long)min(fr, NB - 1) * ldx + fg * 16;
  }
  const bool live = fr < NB;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s0 = s_begin; s0 < s_end; s0 += U) {
    const bool more = s0 + U < s_end;  // wave-uniform
    i32x4 wnlo[U], wnhi[U], xlo[U], xhi[U];
    if (more) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long on = (long)min(s0 + U + u, s_end - 1) * WSTEP;
        wnlo[u] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp + on));
        wnhi[u] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp + on + WHALF));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long o = (long)min(s0 + u, s_end - 1) * 128;
      xlo[u] = *reinterpret_cast<const i32x4*>(xp + o);
      xhi[u] = *reinterpret_cast<const i32x4*>(xp + o + 64);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (s0 + u < s_end) {
        const i32x8_t a = {wlo[u][0], wlo[u][1], wlo[u][2], wlo[u][3], whi[u][0], whi[u][1], whi[u][2], whi[u][3]};
        i32x8_t b = {xlo[u][0], xlo[u][1], xlo[u][2], xlo[u][3], xhi[u][0], xhi[u][1], xhi[u][2], xhi[u][3]};
        if (!live) b = i32x8_t{0, 0, 0, 0, 0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      }
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < U; ++u) { wlo[u] = wnlo[u]; whi[u] = wnhi[u]; }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) part[wave][fg * 4 + r][fr] = acc[r];
  __syncthreads();
  const int i = tid >> 4, b = tid & 15;
  const int row = row0 + i;
  if (tid < 256 && b < NB && row < N) {
    const float xs_b = PRO >= 0 ? s_scale[b] : xscale[b];
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < GF_WAVES; ++w) v += part[w][i][b];
    v *= wscale[row] * xs_b;
    if (res) v += bf2f(res[b * ldr + row]);
    if (out_f32) reinterpret_cast<float*>(y)[b * ldy + row] = v;
    else reinterpret_cast<bf16_t*>(y)[b * ldy + row] = f2bf(v);
  }
}

// e4m3 [N, ldw] rows -> the tiled operand order of gemv_fp8_mfma_kernel<*, true>: piece p = ((rg * K/128 + step) * 2 + half) * 64 + lane
// holds bytes [64 half + 16 (lane >> 4), +16) of step `step` of row 16 rg + (lane & 15); rows past N are zero.
__global__ __launch_bounds__(256) void repack_fp8_mfma_kernel(const uint8_t* __restrict__ W, long ldw, uint8_t* __restrict__ out, int N, int K) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  const int nsteps = K / 128;
  const long total = (long)((N + 15) / 16) * nsteps * 128;
  if (p >= total) return;
  const int lane = (int)(p & 63), half = (int)((p >> 6) & 1);
  const long t = p >> 7;
  const int step = (int)(t % nsteps);
  const long row = (t / nsteps) * 16 + (lane & 15);
  i32x4 v = {0, 0, 0, 0};
  if (row < N) v = *reinterpret_cast<const i32x4*>(W + row * ldw + (long)step * 128 + half * 64 + (lane >> 4) * 16);
  *reinterpret_cast<i32x4*>(out + p * 16) = v;
}

// bf16 [N, ldw] rows -> the tiled operand order of gemv_mfma_kernel<*, true>: piece p = (rg * K/32 + step) * 64 + lane holds
// W[16 rg + (lane & 15)][32 step + 8 (lane >> 4) .. +8]; rows past N are zero.
__global__ __launch_bounds__(256) void repack_bf16_mfma_kernel(const bf16_t* __restrict__ W, long ldw, bf16_t* __restrict__ out, int N, int K) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  const int nsteps = K / 32;
  const long total = (long)((N + 15) / 16) * nsteps * 64;
  if (p >= total) return;
  const int lane = (int)(p & 63);
  const long t = p >> 6;
  const int step = (int)(t % nsteps);
  const long row = (t / nsteps) * 16 + (lane & 15);
  i32x4 v = {0, 0, 0, 0};
  if (row < N) v = *reinterpret_cast<const i32x4*>(W + row * ldw + (long)step * 32 + (lane >> 4) * 8);
  *reinterpret_cast<i32x4*>(out + p * 8) = v;
}

// per-row e4m3 quantisation: scale[n] = max|W[n,:]| / 448, W8 = round(W / scale).  One block per row, 16-B loads; the row stays in
// registers between the max and the conversion (up to 12 chunks of 8 per thread = K <= 24576; longer rows are re-read from L2).
constexpr int QCH = 12;
__global__ __launch_bounds__(256) void quant_fp8_rows_kernel(const bf16_t* __restrict__ W, long ldw, uint8_t* __restrict__ W8, long ld8,
                                                             float* __restrict__ scale, int K) {
  __shared__ float red[4];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int nch = K / 8;  // K % 16 == 0
  const bf16_t* row = W + (long)n * ldw;
  uint4 keep[QCH];
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < QCH; ++i) {
    const int c = tid + i * 256;
    keep[i] = make_uint4(0, 0, 0, 0);
    if (c < nch) {
      keep[i] = *reinterpret_cast<const uint4*>(row + c * 8);
      float v[8];
      unpack8(keep[i], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[e]));
    }
  }
  for (int c = tid + QCH * 256; c < nch; c += 256) {
    float v[8];
    unpack8(*reinterpret_cast<const uint4*>(row + c * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[e]));
  }
  m = block_max<4>(m, red);
  const float sc = m > 0.f ? m / 448.f : 1.f;
  if (tid == 0) scale[n] = sc;
  const float inv = 1.f / sc;
  uint8_t* orow = W8 + (long)n * ld8;
  auto cvt8 = [&](const uint4& raw, int c) {
    float v[8];
    unpack8(raw, v);
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, hi, true);
    *reinterpret_cast<int2*>(orow + c * 8) = make_int2(lo, hi);
  };
#pragma unroll
  for (int i = 0; i < QCH; ++i) {
    const int c = tid + i * 256;
    if (c < nch) cvt8(keep[i], c);
  }
  for (int c = tid + QCH * 256; c < nch; c += 256) cvt8(*reinterpret_cast<const uint4*>(row + c * 8), c);
}

// RoPE on the new q / k rows + append of (rotated k, v) to the cache at the device-resident position
__global__ void rope_kv_append_kernel(bf16_t* qkv, long ld, bf16_t* __restrict__ kc, bf16_t* __restrict__ vc,
                                      const float* __restrict__ cos_t, const float* __restrict__ sin_t, const int* __restrict__ pos,
                                      int H, int D, int max_ctx) {
  const int b = blockIdx.y;
  const int p = pos[b];
  const int half = D / 2, d = H * D;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over 2*H*half/8 rope chunks, then d/8 v chunks
  const int n_rope = 2 * H * (half / 8);
  bf16_t* row = qkv + (long)b * ld;
  if (idx < n_rope) {
    const int c = idx % (half / 8), h = idx / (half / 8);  // h in [0, 2H): q heads then k heads
    bf16_t* p1 = row + (long)h * D + c * 8;
    bf16_t* p2 = p1 + half;
    const uint4 a = *reinterpret_cast<const uint4*>(p1), bb = *reinterpret_cast<const uint4*>(p2);
    const float av[8] = {bflo(a.x), bfhi(a.x), bflo(a.y), bfhi(a.y), bflo(a.z), bfhi(a.z), bflo(a.w), bfhi(a.w)};
    const float bv[8] = {bflo(bb.x), bfhi(bb.x), bflo(bb.y), bfhi(bb.y), bflo(bb.z), bfhi(bb.z), bflo(bb.w), bfhi(bb.w)};
    float o1[8], o2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float co = cos_t[(long)p * half + c * 8 + i], si = sin_t[(long)p * half + c * 8 + i];
      o1[i] = av[i] * co - bv[i] * si;
      o2[i] = bv[i] * co + av[i] * si;
    }
    const uint4 r1 = make_uint4(pack2bf(o1[0], o1[1]), pack2bf(o1[2], o1[3]), pack2bf(o1[4], o1[5]), pack2bf(o1[6], o1[7]));
    const uint4 r2 = make_uint4(pack2bf(o2[0], o2[1]), pack2bf(o2[2], o2[3]), pack2bf(o2[4], o2[5]), pack2bf(o2[6], o2[7]));
    *reinterpret_cast<uint4*>(p1) = r1;
    *reinterpret_cast<uint4*>(p2) = r2;
    if (h >= H) {  // key head: also into the cache
      bf16_t* kd = kc + ((long)b * max_ctx + p) * d + (long)(h - H) * D + c * 8;
      *reinterpret_cast<uint4*>(kd) = r1;
      *reinterpret_cast<uint4*>(kd + half) = r2;
    }
  } else if (idx < n_rope + d / 8) {
    const int c = idx - n_rope;
    *reinterpret_cast<uint4*>(vc + ((long)b * max_ctx + p) * d + c * 8) = *reinterpret_cast<const uint4*>(row + 2 * d + c * 8);
  }
}

// state: int32 [4] = {ctx, -, -, -}; desc: int32 [B][8]; pos: int32 [B]
// cs (optional): float [B][128] = cos / sin row of the new position, so that the attention kernel's q / k rotation does not have to wait for
// the position before it can ask for its table row (one dependent round trip less on the latency-bound decode attention)
__global__ void decode_advance_kernel(int* state, int* desc, int* pos, int B, int max_ctx, int step_inc, const float* __restrict__ cos_t = nullptr,
                                      const float* __restrict__ sin_t = nullptr, float* __restrict__ cs = nullptr) {
  const int b = threadIdx.x;
  const int ctx = state[0];
  if (cs != nullptr) {
    const float c = cos_t[(long)ctx * 64 + threadIdx.x], sn = sin_t[(long)ctx * 64 + threadIdx.x];
    for (int i = 0; i < B; ++i) { cs[i * 128 + threadIdx.x] = c; cs[i * 128 + 64 + threadIdx.x] = sn; }
  }
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


// ------------------------------------------------------------------------------------------------------------------
// decode_attn_kernel: the whole attention of ONE new token per sequence in a single launch (head_dim 128):
//   RoPE of the new q / k row (HF rotate_half), append of (rotated k, v) to the cache, attention over the cache.
//   Latency-bound (ctx * 512 B per head), so the design minimises dependent memory round trips: one 16-wave workgroup per
//   (sequence, head); a pass covers 512 keys (wave w: keys 32w..32w+31, 16 lanes x 16 B per key row = whole 256-B rows per
//   wave-instruction) and issues ALL its K and V row loads before anything else; waves keep online-softmax partials
//   (m, l, o[128]) across passes and merge them through LDS - no global atomics or fences (a cross-workgroup split was tried:
//   device-scope release/acquire between XCD-private L2s cost more than the serial pass loop).
// ------------------------------------------------------------------------------------------------------------------
constexpr int DA_WAVES = 8, DA_KPG = 16, DA_PASS = DA_WAVES * 4 * DA_KPG;   // 8 waves x 4 lane groups x 16 keys = 512 keys per pass; 8 waves (not 16 x 8 keys): 256 VGPRs per lane hold the 32 row loads without scratch (16 waves: 46 spilled registers)

__device__ __forceinline__ float group16_sum(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  return v;
}

__global__ __launch_bounds__(DA_WAVES * 64) void decode_attn_kernel(const bf16_t* __restrict__ qkv, long ld, bf16_t* kc, bf16_t* vc,
                                                           const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                           const int* __restrict__ pos, const unsigned char* __restrict__ kmask,
                                                           long ld_kmask, bf16_t* __restrict__ out, long ldo, int H, int max_ctx,
                                                           float scale) {
  constexpr int D = 128, HALF = 64;
  __shared__ float part[DA_WAVES][132];
  const int h = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, grp = lane >> 4;
  const int d_model = H * D;
  const long cache_row0 = (long)b * max_ctx;
  const bf16_t* kbase = kc + cache_row0 * d_model + h * D + l16 * 8;
  const bf16_t* vbase = vc + cache_row0 * d_model + h * D + l16 * 8;
  int key0 = wave * (4 * DA_KPG) + grp;  // this 16-lane group: keys key0, key0+4, ... (DA_KPG of them) of the current pass
  // first pass: every global load is issued up front - none depends on another (any cache row < max_ctx is readable; rows past the
  // context are discarded below)
  uint4 kr[DA_KPG], vr[DA_KPG];
#pragma unroll
  for (int i = 0; i < DA_KPG; ++i) {
    const long key = min(key0 + i * 4, max_ctx - 1);
    kr[i] = *reinterpret_cast<const uint4*>(kbase + key * d_model);
    vr[i] = *reinterpret_cast<const uint4*>(vbase + key * d_model);
  }
  const bf16_t* row = qkv + (long)b * ld;
  const uint4 q_raw = *reinterpret_cast<const uint4*>(row + h * D + l16 * 8);
  const uint4 k_raw = *reinterpret_cast<const uint4*>(row + d_model + h * D + l16 * 8);
  const uint4 v_raw = *reinterpret_cast<const uint4*>(row + 2 * d_model + h * D + l16 * 8);
  const int p = pos[b];  // position of the new token; keys 0..p are visible
  // ---- rotated q (pre-scaled) and rotated new k for dims [l16*8, +8); the partner half lives in lane l16 ^ 8
  float q[8], kn[8], vn[8];
  {
    float qa[8], ka[8], qb[8], kb[8];
    unpack8(q_raw, qa); unpack8(k_raw, ka); unpack8(v_raw, vn);
    const int f0 = (l16 & 7) * 8;
    const bool hi = l16 >= 8;
    const float4 c0 = *reinterpret_cast<const float4*>(cos_t + (long)p * HALF + f0), c1 = *reinterpret_cast<const float4*>(cos_t + (long)p * HALF + f0 + 4);
    const float4 s0 = *reinterpret_cast<const float4*>(sin_t + (long)p * HALF + f0), s1 = *reinterpret_cast<const float4*>(sin_t + (long)p * HALF + f0 + 4);
    const float cv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      qb[i] = __shfl_xor(qa[i], 8, 64);
      kb[i] = __shfl_xor(ka[i], 8, 64);
      // rotate_half: x[d] * cos - x[d+64] * sin (d < 64);  x[d] * cos + x[d-64] * sin (d >= 64); rounded to bf16 like the stored rows
      q[i] = bf2f(f2bf(hi ? qa[i] * cv[i] + qb[i] * sv[i] : qa[i] * cv[i] - qb[i] * sv[i])) * scale;
      kn[i] = bf2f(f2bf(hi ? ka[i] * cv[i] + kb[i] * sv[i] : ka[i] * cv[i] - kb[i] * sv[i]));
    }
  }
  float m = -INFINITY, l = 0.f, o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
  for (int base = 0; base <= p; base += DA_PASS) {
    if (base > 0) {  // later passes (context > 512): same load pattern, issued together at the top of the pass
      key0 = base + wave * (4 * DA_KPG) + grp;
#pragma unroll
      for (int i = 0; i < DA_KPG; ++i) {
        const long key = min(key0 + i * 4, max_ctx - 1);
        kr[i] = *reinterpret_cast<const uint4*>(kbase + key * d_model);
        vr[i] = *reinterpret_cast<const uint4*>(vbase + key * d_model);
      }
    }
    float sc[DA_KPG];
    float mp = -INFINITY;
#pragma unroll
    for (int i = 0; i < DA_KPG; ++i) {
      const int key = key0 + i * 4;
      float kv[8];
      unpack8(kr[i], kv);
      if (key == p) {
#pragma unroll
        for (int e = 0; e < 8; ++e) kv[e] = kn[e];
        *reinterpret_cast<uint4*>(kc + (cache_row0 + p) * d_model + h * D + l16 * 8) = pack8(kn);  // append: this group owns the new key
        *reinterpret_cast<uint4*>(vc + (cache_row0 + p) * d_model + h * D + l16 * 8) = v_raw;
      }
      float dot = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) dot += q[e] * kv[e];
      dot = group16_sum(dot);
      bool ok = key <= p;
      if (ok && kmask != nullptr) ok = kmask[(long)b * ld_kmask + key] != 0;
      sc[i] = ok ? dot : -INFINITY;
      mp = fmaxf(mp, sc[i]);
    }
    mp = fmaxf(mp, __shfl_xor(mp, 16, 64));
    mp = fmaxf(mp, __shfl_xor(mp, 32, 64));
    const float m_new = fmaxf(m, mp);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = __expf(m - m_use);  // 0 on the first contribution (m = -inf)
    l *= alpha;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] *= alpha;
    m = m_new;
#pragma unroll
    for (int i = 0; i < DA_KPG; ++i) {
      if (sc[i] == -INFINITY) continue;  // masked / absent key: its (possibly uninitialised) cache row must not touch the sum
      const float pr = __expf(sc[i] - m_use);
      float vv[8];
      unpack8(vr[i], vv);
      if (key0 + i * 4 == p) {
#pragma unroll
        for (int e = 0; e < 8; ++e) vv[e] = vn[e];
      }
      l += pr;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += pr * vv[e];
    }
  }
  // ---- fold the 4 key groups of the wave (each lane group accumulated with the wave-wide max, so plain sums), then the waves
  l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64);
#pragma unroll
  for (int e = 0; e < 8; ++e) { o[e] += __shfl_xor(o[e], 16, 64); o[e] += __shfl_xor(o[e], 32, 64); }
  if (lane < 16) {
#pragma unroll
    for (int e = 0; e < 8; ++e) part[wave][2 + l16 * 8 + e] = o[e];
    if (lane == 0) { part[wave][0] = m; part[wave][1] = l; }
  }
  __syncthreads();
  if (tid < D) {
    float M = -INFINITY;
#pragma unroll
    for (int i = 0; i < DA_WAVES; ++i) M = fmaxf(M, part[i][0]);
    float L = 0.f, acc = 0.f;
#pragma unroll
    for (int i = 0; i < DA_WAVES; ++i) {
      const float mi = part[i][0];
      const float w = mi == -INFINITY ? 0.f : __expf(mi - M);
      L += w * part[i][1];
      acc += w * part[i][2 + tid];
    }
    out[(long)b * ldo + h * D + tid] = f2bf(L > 0.f ? acc / L : 0.f);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// decode_attn_split_kernel (round 4): the same attention with the CONTEXT split over workgroups.  One workgroup streams at most ~27 GB/s
// from HBM (10-12 B / clk / CU: the per-CU miss path), so the single-workgroup-per-head form above keeps 32 of 256 CUs busy and needs
// ctx * 512 B / 27 GB/s per head: 13.8 us at ctx ~ 700 - 13 % of a decoded token for 7 MB of K / V.  Here workgroup (head, sp) owns the
// keys [128 j + 128 sp NS', ...) - 128-key slices with stride NS - and issues ALL loads of its first slice before it knows the context
// length (any cache row < max_ctx is readable), so the only dependent round trips are pos -> cos / sin and the exchange below.
// Exchange: every workgroup that owns at least one visible key publishes (m, l, o[128]) with write-through (sc1) stores, drains them,
// and takes a ticket; the LAST arriver reads all partials with sc1 loads (both sides sc1: no fence needed, cdna_hip_programming.md §6 G16
// R1), merges and writes the bf16 row; it resets the ticket (tickets are zero between launches: hipGraph replays need no memset node).
// ------------------------------------------------------------------------------------------------------------------
constexpr int DS_WAVES = 4, DS_KPG = 8, DS_SLICE = DS_WAVES * 4 * DS_KPG;   // 4 waves x 4 lane groups x 8 keys = 128 keys per slice
constexpr int DS_PART = 132;                                                 // floats per partial: m, l, -, -, o[128]

__global__ __launch_bounds__(DS_WAVES * 64) void decode_attn_split_kernel(const bf16_t* __restrict__ qkv, long ld, bf16_t* kc, bf16_t* vc,
                                                                 const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                                 const int* __restrict__ pos, const unsigned char* __restrict__ kmask,
                                                                 long ld_kmask, bf16_t* __restrict__ out, long ldo, int H, int max_ctx,
                                                                 float scale, int NS, float* part_g, int* tickets, const float* __restrict__ cs) {
  constexpr int D = 128, HALF = 64;
  __shared__ float part[DS_WAVES][DS_PART];
  __shared__ int s_ticket;
  const int h = blockIdx.x / NS, sp = blockIdx.x - h * NS, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, grp = lane >> 4;
  const int d_model = H * D;
  const long cache_row0 = (long)b * max_ctx;
  const bf16_t* kbase = kc + cache_row0 * d_model + h * D + l16 * 8;
  const bf16_t* vbase = vc + cache_row0 * d_model + h * D + l16 * 8;
  int key0 = sp * DS_SLICE + wave * (4 * DS_KPG) + grp;   // this 16-lane group: keys key0, key0 + 4, ... (DS_KPG of them) of the current slice
  uint4 kr[DS_KPG], vr[DS_KPG];
#pragma unroll
  for (int i = 0; i < DS_KPG; ++i) {
    const long key = min(key0 + i * 4, max_ctx - 1);
    kr[i] = *reinterpret_cast<const uint4*>(kbase + key * d_model);
    vr[i] = *reinterpret_cast<const uint4*>(vbase + key * d_model);
  }
  const bf16_t* row = qkv + (long)b * ld;
  const uint4 q_raw = *reinterpret_cast<const uint4*>(row + h * D + l16 * 8);
  const uint4 k_raw = *reinterpret_cast<const uint4*>(row + d_model + h * D + l16 * 8);
  const uint4 v_raw = *reinterpret_cast<const uint4*>(row + 2 * d_model + h * D + l16 * 8);
  // cos | sin of the new position: from the row lhrs_decode_advance_cs left for this sequence (no dependence on `pos`), else from the tables
  float4 c0, c1, s0, s1;
  if (cs != nullptr) {
    const float* r = cs + (long)b * 128 + (l16 & 7) * 8;
    c0 = *reinterpret_cast<const float4*>(r); c1 = *reinterpret_cast<const float4*>(r + 4);
    s0 = *reinterpret_cast<const float4*>(r + 64); s1 = *reinterpret_cast<const float4*>(r + 68);
  }
  const int p = pos[b];  // position of the new token; keys 0..p are visible
  if (sp * DS_SLICE > p) return;                        // no visible key in any slice of this workgroup (workgroup-uniform)
  const int nact = min(NS, p / DS_SLICE + 1);           // workgroups of this head that own a visible key
  float q[8], kn[8], vn[8];
  {
    float qa[8], ka[8], qb[8], kb[8];
    unpack8(q_raw, qa); unpack8(k_raw, ka); unpack8(v_raw, vn);
    const int f0 = (l16 & 7) * 8;
    const bool hi = l16 >= 8;
    if (cs == nullptr) {
      c0 = *reinterpret_cast<const float4*>(cos_t + (long)p * HALF + f0); c1 = *reinterpret_cast<const float4*>(cos_t + (long)p * HALF + f0 + 4);
      s0 = *reinterpret_cast<const float4*>(sin_t + (long)p * HALF + f0); s1 = *reinterpret_cast<const float4*>(sin_t + (long)p * HALF + f0 + 4);
    }
    const float cv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      qb[i] = __shfl_xor(qa[i], 8, 64);
      kb[i] = __shfl_xor(ka[i], 8, 64);
      q[i] = bf2f(f2bf(hi ? qa[i] * cv[i] + qb[i] * sv[i] : qa[i] * cv[i] - qb[i] * sv[i])) * scale;
      kn[i] = bf2f(f2bf(hi ? ka[i] * cv[i] + kb[i] * sv[i] : ka[i] * cv[i] - kb[i] * sv[i]));
    }
  }
  float m = -INFINITY, l = 0.f, o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
  for (int base = sp * DS_SLICE; base <= p; base += NS * DS_SLICE) {
    if (base > sp * DS_SLICE) {  // further slices of this workgroup (context > 128 NS)
      key0 = base + wave * (4 * DS_KPG) + grp;
#pragma unroll
      for (int i = 0; i < DS_KPG; ++i) {
        const long key = min(key0 + i * 4, max_ctx - 1);
        kr[i] = *reinterpret_cast<const uint4*>(kbase + key * d_model);
        vr[i] = *reinterpret_cast<const uint4*>(vbase + key * d_model);
      }
    }
    float sc[DS_KPG];
    float mp = -INFINITY;
#pragma unroll
    for (int i = 0; i < DS_KPG; ++i) {
      const int key = key0 + i * 4;
      float kv[8];
      unpack8(kr[i], kv);
      if (key == p) {
#pragma unroll
        for (int e = 0; e < 8; ++e) kv[e] = kn[e];
        *reinterpret_cast<uint4*>(kc + (cache_row0 + p) * d_model + h * D + l16 * 8) = pack8(kn);  // append: this group owns the new key
        *reinterpret_cast<uint4*>(vc + (cache_row0 + p) * d_model + h * D + l16 * 8) = v_raw;
      }
      float dot = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) dot += q[e] * kv[e];
      dot = group16_sum(dot);
      bool ok = key <= p;
      if (ok && kmask != nullptr) ok = kmask[(long)b * ld_kmask + key] != 0;
      sc[i] = ok ? dot : -INFINITY;
      mp = fmaxf(mp, sc[i]);
    }
    mp = fmaxf(mp, __shfl_xor(mp, 16, 64));
    mp = fmaxf(mp, __shfl_xor(mp, 32, 64));
    const float m_new = fmaxf(m, mp);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = __expf(m - m_use);
    l *= alpha;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] *= alpha;
    m = m_new;
#pragma unroll
    for (int i = 0; i < DS_KPG; ++i) {
      if (sc[i] == -INFINITY) continue;
      const float pr = __expf(sc[i] - m_use);
      float vv[8];
      unpack8(vr[i], vv);
      if (key0 + i * 4 == p) {
#pragma unroll
        for (int e = 0; e < 8; ++e) vv[e] = vn[e];
      }
      l += pr;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += pr * vv[e];
    }
  }
  l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64);
#pragma unroll
  for (int e = 0; e < 8; ++e) { o[e] += __shfl_xor(o[e], 16, 64); o[e] += __shfl_xor(o[e], 32, 64); }
  if (lane < 16) {
#pragma unroll
    for (int e = 0; e < 8; ++e) part[wave][4 + l16 * 8 + e] = o[e];
    if (lane == 0) { part[wave][0] = m; part[wave][1] = l; }
  }
  __syncthreads();
  // ---- this workgroup's partial: thread d < 128 holds o[d] relative to the workgroup maximum M
  float M = -INFINITY, L = 0.f, acc = 0.f;
  if (tid < D) {
#pragma unroll
    for (int i = 0; i < DS_WAVES; ++i) M = fmaxf(M, part[i][0]);
#pragma unroll
    for (int i = 0; i < DS_WAVES; ++i) {
      const float mi = part[i][0];
      const float w = mi == -INFINITY ? 0.f : __expf(mi - M);
      L += w * part[i][1];
      acc += w * part[i][4 + tid];
    }
  }
  if (nact == 1) {   // short context: nothing to exchange
    if (tid < D) out[(long)b * ldo + h * D + tid] = f2bf(L > 0.f ? acc / L : 0.f);
    return;
  }
  unsigned* mine = reinterpret_cast<unsigned*>(part_g + ((long)(b * H + h) * NS + sp) * DS_PART);
  if (tid < D) {
    __hip_atomic_store(mine + 4 + tid, __float_as_uint(acc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1: write-through
    if (tid == 0) {
      __hip_atomic_store(mine + 0, __float_as_uint(M), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(mine + 1, __float_as_uint(L), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains its write-through stores ...
  __syncthreads();
  if (tid == 0) s_ticket = __hip_atomic_fetch_add(tickets + b * H + h, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... before ONE lane takes the ticket
  __syncthreads();
  if (s_ticket != nact - 1) return;
  // ---- last arriver: every other partial of this head is complete in memory (sc1 stores drained before each ticket)
  if (tid == 0) __hip_atomic_store(tickets + b * H + h, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid < D) {
    const unsigned* base_p = reinterpret_cast<const unsigned*>(part_g + (long)(b * H + h) * NS * DS_PART);
    float Mx = -INFINITY;
    for (int s2 = 0; s2 < nact; ++s2)
      Mx = fmaxf(Mx, __uint_as_float(__hip_atomic_load(base_p + (long)s2 * DS_PART, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
    float Ls = 0.f, As = 0.f;
    for (int s2 = 0; s2 < nact; ++s2) {
      const unsigned* ps = base_p + (long)s2 * DS_PART;
      const float ms = __uint_as_float(__hip_atomic_load(ps + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      const float ls = __uint_as_float(__hip_atomic_load(ps + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      const float os = __uint_as_float(__hip_atomic_load(ps + 4 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      const float w = ms == -INFINITY ? 0.f : __expf(ms - Mx);
      Ls += w * ls;
      As += w * os;
    }
    out[(long)b * ldo + h * D + tid] = f2bf(Ls > 0.f ? As / Ls : 0.f);
  }
}

}  // namespace

extern "C" int lhrs_decode_emit(const long* next_ids, int* tok32, long* out_ids, int* state, int B, int max_new, void* stream) {
  LHRS_REQUIRE(B >= 1 && B <= 64, "decode_emit: B=%d", B);
  hipLaunchKernelGGL(decode_emit_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, next_ids, tok32, out_ids, state, B, max_new);
  LHRS_CHECK_LAUNCH("decode_emit");
  return 0;
}

static int g_gemv_rpw = 0, g_gemv_unr = 0;  // kernel A/B tests only (lhrs_gemv_set_tuning); 0 = the shape rule below

// y[B, N] = x[B, K] . W[N, K]^T (+ residual[B, N]);  B <= 8, K % 8 == 0
template <int PRO, bool FP8>
static int gemv_chunk(const void* W, long ldw, const float* wscale, const bf16_t* x, long ldx, const bf16_t* norm_w, float eps,
                      const bf16_t* residual, long ldr, void* y, long ldy, int B, int N, int K, int out_f32, hipStream_t s) {
  const dim3 blk(256);
  const size_t sm = (size_t)B * K * 2;
#define GEMV_LAUNCH(NB, RPW, UNR)                                                                                       \
  do {                                                                                                                  \
    if (sm > 65536) (void)hipFuncSetAttribute((const void*)gemv_kernel<NB, PRO, FP8, RPW, UNR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); \
    hipLaunchKernelGGL((gemv_kernel<NB, PRO, FP8, RPW, UNR>), dim3(cdiv(N, 4 * RPW)), blk, sm, s, W, ldw, wscale, x, ldx, norm_w, eps, residual, ldr, y, ldy, N, K, out_f32); \
  } while (0)
  if (B == 1 && !FP8) {
    // HBM needs ~16 MB in flight chip-wide: few-row matrices (o_proj, down_proj: 4096 rows) get one row per wave (1024 blocks) and a
    // 4-chunk K pipeline (down_proj 24.4 -> 20.6 us); the tall ones four rows per wave, two chunks deep (tools/gemv_bench.py sweep)
    int rpw = N <= 4096 ? 1 : 4, unr = N <= 4096 ? 4 : 2;
    if (g_gemv_rpw) { rpw = g_gemv_rpw; unr = g_gemv_unr; }
    const int key = rpw * 10 + unr;
    switch (key) {
      case 11: GEMV_LAUNCH(1, 1, 1); break;
      case 12: GEMV_LAUNCH(1, 1, 2); break;
      case 14: GEMV_LAUNCH(1, 1, 4); break;
      case 18: GEMV_LAUNCH(1, 1, 8); break;
      case 22: GEMV_LAUNCH(1, 2, 2); break;
      case 24: GEMV_LAUNCH(1, 2, 4); break;
      case 42: GEMV_LAUNCH(1, 4, 2); break;
      default: GEMV_LAUNCH(1, 4, 1); break;
    }
    LHRS_CHECK_LAUNCH("gemv");
    return 0;
  }
#define GEMV_CASE(NB) case NB: GEMV_LAUNCH(NB, 4, 1); break;
  switch (B) { GEMV_CASE(1) GEMV_CASE(2) GEMV_CASE(3) GEMV_CASE(4) GEMV_CASE(5) GEMV_CASE(6) GEMV_CASE(7) GEMV_CASE(8) }
#undef GEMV_CASE
#undef GEMV_LAUNCH
  LHRS_CHECK_LAUNCH("gemv");
  return 0;
}

// y[B, N] = pro(x)[B, K] . W[N, K]^T (+ residual[B, N]);  B <= 8.  prologue: 0 none, 1 RMSNorm(norm_w, eps), 2 SwiGLU (x is [B, 2K]).
// w_format 1: W is e4m3 bytes [N, ldw] with per-row scales `wscale` (lhrs_quant_fp8_rows).  Batches exceeding the LDS are split.
// w_format: 0 bf16 rows, 1 e4m3 rows + wscale, 2 bf16 tiles of lhrs_repack_bf16_mfma (batch >= 2 only: the MFMA weight stream).
extern "C" int lhrs_gemv(const void* W, long ldw, const float* wscale, int w_format, const void* x, long ldx, int prologue,
                         const void* norm_w, float eps, const void* residual, long ldr, void* y, long ldy, int B, int N, int K,
                         int out_f32, void* stream) {
  const int w_fp8 = w_format == 1, w_packed = w_format == 2;
  LHRS_REQUIRE(w_format >= 0 && w_format <= 2 && (!w_packed || (B >= 2 && K % 128 == 0)), "gemv: weight format %d (tiles need batch >= 2, K %% 128 == 0)", w_format);
  LHRS_REQUIRE(B >= 1 && B <= 16 && N > 0 && K % 16 == 0 && ldx % 8 == 0, "gemv: B=%d (1..16) N=%d K=%d", B, N, K);
  LHRS_REQUIRE(B <= 8 || (!w_fp8 && K % 128 == 0), "gemv: batches above 8 need bf16 weights and K %% 128 == 0");
  LHRS_REQUIRE(w_fp8 ? (ldw % 16 == 0 && wscale != nullptr) : (w_packed || ldw % 8 == 0), "gemv: weight stride / scales");
  LHRS_REQUIRE(prologue >= 0 && prologue <= 2 && (prologue != 1 || norm_w != nullptr), "gemv: prologue %d", prologue);
  int bmax = (int)((152L * 1024) / ((long)K * 2));
  if (!w_fp8 && prologue == 0 && B >= 2 && K % 128 == 0) bmax = 16;  // the MFMA kernel reads x from L2: no LDS limit
  LHRS_REQUIRE(bmax >= 1, "gemv: one activation vector does not fit LDS (K=%d)", K);
  const long esz = out_f32 ? 4 : 2;
  hipStream_t s = (hipStream_t)stream;
  for (int b0 = 0; b0 < B; b0 += bmax) {
    const int nb = B - b0 < bmax ? B - b0 : bmax;
    const bf16_t* xb = (const bf16_t*)x + b0 * ldx;
    const bf16_t* rb = residual ? (const bf16_t*)residual + b0 * ldr : nullptr;
    void* yb = (char*)y + b0 * ldy * esz;
    int rc;
    LHRS_REQUIRE(!w_packed || nb >= 2, "gemv: a batch chunk of %d row(s) cannot read tiled weights (B=%d, chunks of %d)", nb, B, bmax);
    if (!w_fp8 && nb >= 2 && K % 128 == 0) {  // batched: MFMA weight stream
      const size_t sm = prologue == 0 ? 0 : (size_t)nb * K * 2;
      const int nw = K % 256 == 0 ? 8 : 4;
      const dim3 grid(cdiv(N, 16)), blk(nw * 64);
#define GOM1(P, PK, NW)                                                                                                             \
  do {                                                                                                                             \
    if (sm > 65536) (void)hipFuncSetAttribute((const void*)gemv_mfma_kernel<P, PK, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); \
    hipLaunchKernelGGL((gemv_mfma_kernel<P, PK, NW>), grid, blk, sm, s, (const bf16_t*)W, ldw, xb, ldx, (const bf16_t*)norm_w, eps, rb, ldr, yb, ldy, nb, N, K, out_f32); \
  } while (0)
#define GOM(P)                                                                                                                     \
  do {                                                                                                                             \
    if (w_packed) { if (nw == 8) GOM1(P, true, 8); else GOM1(P, true, 4); }                                                        \
    else { if (nw == 8) GOM1(P, false, 8); else GOM1(P, false, 4); }                                                               \
  } while (0)
      if (prologue == 0) GOM(0); else if (prologue == 1) GOM(1); else GOM(2);
#undef GOM1
#undef GOM
      LHRS_CHECK_LAUNCH("gemv_mfma");
      continue;
    }
#define GO(P, F) rc = gemv_chunk<P, F>(W, ldw, wscale, xb, ldx, (const bf16_t*)norm_w, eps, rb, ldr, yb, ldy, nb, N, K, out_f32, s)
    if (w_fp8) { if (prologue == 0) GO(0, true); else if (prologue == 1) GO(1, true); else GO(2, true); }
    else { if (prologue == 0) GO(0, false); else if (prologue == 1) GO(1, false); else GO(2, false); }
#undef GO
    if (rc) return -1;
  }
  return 0;
}

extern "C" int lhrs_gemv_set_tuning(int rows_per_wave, int chunks_per_iteration) {
  g_gemv_rpw = rows_per_wave;
  g_gemv_unr = chunks_per_iteration;
  return 0;
}

extern "C" int lhrs_gemv_bf16(const void* W, long ldw, const void* x, long ldx, const void* residual, long ldr, void* y, long ldy,
                              int B, int N, int K, int out_f32, void* stream) {
  return lhrs_gemv(W, ldw, nullptr, 0, x, ldx, 0, nullptr, 0.f, residual, ldr, y, ldy, B, N, K, out_f32, stream);
}

extern "C" int lhrs_quant_fp8_rows(const void* W, long ldw, void* W8, long ld8, float* scale, int N, int K, void* stream) {
  LHRS_REQUIRE(N > 0 && K % 16 == 0 && ld8 % 16 == 0 && ldw % 8 == 0, "quant_fp8_rows: N=%d K=%d ldw=%ld ld8=%ld", N, K, ldw, ld8);
  hipLaunchKernelGGL(quant_fp8_rows_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)W, ldw, (uint8_t*)W8, ld8, scale, K);
  LHRS_CHECK_LAUNCH("quant_fp8_rows");
  return 0;
}

extern "C" int lhrs_rope_kv_append(void* qkv, long ld, void* kcache, void* vcache, const float* cos_t, const float* sin_t,
                                   const int* pos, int B, int H, int D, int max_ctx, void* stream) {
  LHRS_REQUIRE(B >= 1 && D % 16 == 0, "rope_kv_append: B=%d D=%d", B, D);
  const int work = 2 * H * (D / 16) + H * D / 8;
  hipLaunchKernelGGL(rope_kv_append_kernel, dim3(cdiv(work, 256), B), dim3(256), 0, (hipStream_t)stream, (bf16_t*)qkv, ld,
                     (bf16_t*)kcache, (bf16_t*)vcache, cos_t, sin_t, pos, H, D, max_ctx);
  LHRS_CHECK_LAUNCH("rope_kv_append");
  return 0;
}

extern "C" int lhrs_decode_advance(int* state, int* desc, int* pos, int B, int max_ctx, int step_inc, void* stream) {
  LHRS_REQUIRE(B >= 1 && B <= 64, "decode_advance: B=%d", B);
  hipLaunchKernelGGL(decode_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, desc, pos, B, max_ctx, step_inc, (const float*)nullptr,
                     (const float*)nullptr, (float*)nullptr);
  LHRS_CHECK_LAUNCH("decode_advance");
  return 0;
}
// the same, and cs[B][128] = the cos | sin rows (head_dim 128: 64 + 64 floats) of the new position for lhrs_decode_attn_split
extern "C" int lhrs_decode_advance_cs(int* state, int* desc, int* pos, const float* cos_t, const float* sin_t, float* cs, int B, int max_ctx,
                                      int step_inc, void* stream) {
  LHRS_REQUIRE(B >= 1 && B <= 64 && cos_t && sin_t && cs, "decode_advance_cs: B=%d", B);
  hipLaunchKernelGGL(decode_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, desc, pos, B, max_ctx, step_inc, cos_t, sin_t, cs);
  LHRS_CHECK_LAUNCH("decode_advance_cs");
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

// One launch per layer and token: RoPE(q, k_new) + KV append + attention over the cache (+ HF attention_mask bytes) -> o [B, H*128].
extern "C" int lhrs_decode_attn(const void* qkv, long ld, void* kcache, void* vcache, const float* cos_t, const float* sin_t,
                                const int* pos, const unsigned char* key_mask, long ld_mask, void* out, long ldo, int B, int H, int D,
                                int max_ctx, float scale, void* stream) {
  LHRS_REQUIRE(D == 128, "decode_attn: head_dim %d (only 128)", D);
  LHRS_REQUIRE(B >= 1 && H >= 1 && max_ctx >= 1 && ld % 8 == 0, "decode_attn: B=%d H=%d max_ctx=%d", B, H, max_ctx);
  hipLaunchKernelGGL(decode_attn_kernel, dim3(H, B), dim3(DA_WAVES * 64), 0, (hipStream_t)stream, (const bf16_t*)qkv, ld, (bf16_t*)kcache,
                     (bf16_t*)vcache, cos_t, sin_t, pos, key_mask, ld_mask, (bf16_t*)out, ldo, H, max_ctx, scale);
  LHRS_CHECK_LAUNCH("decode_attn");
  return 0;
}

// The same with the context split over nsplit workgroups per head (decode_attn_split_kernel: nsplit slices of 128 keys in flight per head,
// 1 <= nsplit <= 16).  part: fp32 [B][H][nsplit][132] partials, tickets: int32 [B][H], ZERO before the first call (the kernel leaves them
// zero); both caller-owned and private to the stream the calls are ordered on.  cs (may be NULL): float [B][128], the cos | sin rows of pos[b]
// written by lhrs_decode_advance_cs - the kernel then takes the rotation from there instead of waiting for pos[b] to index the tables.
extern "C" int lhrs_decode_attn_split(const void* qkv, long ld, void* kcache, void* vcache, const float* cos_t, const float* sin_t,
                                      const int* pos, const unsigned char* key_mask, long ld_mask, void* out, long ldo, int B, int H, int D,
                                      int max_ctx, float scale, int nsplit, float* part, int* tickets, const float* cs, void* stream) {
  LHRS_REQUIRE(D == 128, "decode_attn_split: head_dim %d (only 128)", D);
  LHRS_REQUIRE(B >= 1 && H >= 1 && max_ctx >= 1 && ld % 8 == 0 && nsplit >= 1 && nsplit <= 16 && part && tickets,
               "decode_attn_split: B=%d H=%d max_ctx=%d nsplit=%d", B, H, max_ctx, nsplit);
  hipLaunchKernelGGL(decode_attn_split_kernel, dim3(H * nsplit, B), dim3(DS_WAVES * 64), 0, (hipStream_t)stream, (const bf16_t*)qkv, ld,
                     (bf16_t*)kcache, (bf16_t*)vcache, cos_t, sin_t, pos, key_mask, ld_mask, (bf16_t*)out, ldo, H, max_ctx, scale, nsplit, part,
                     tickets, cs);
  LHRS_CHECK_LAUNCH("decode_attn_split");
  return 0;
}

// y[B, N] = sx[b] * sw[n] * (x8[B, K] . W8[N, K]^T) (+ residual[B, N]): e4m3 weights AND activations (lhrs_quant_fp8_rows /
// lhrs_rmsnorm_fwd_q / lhrs_swiglu_fwd_q) on the block-scaled MFMA; B <= 16, K % 128 == 0.  w_packed: W8 is the tiled copy
// lhrs_repack_fp8_mfma made (ldw unused).
extern "C" int lhrs_gemv_fp8_mfma(const void* W8, long ldw, const float* wscale, const void* x8, long ldx, const float* xscale,
                                  const void* residual, long ldr, void* y, long ldy, int B, int N, int K, int out_f32, int w_packed,
                                  void* stream) {
  LHRS_REQUIRE(B >= 1 && B <= 16 && N > 0 && K >= 128 && K % 128 == 0, "gemv_fp8_mfma: B=%d N=%d K=%d", B, N, K);
  LHRS_REQUIRE((w_packed || ldw % 16 == 0) && ldx % 16 == 0 && wscale && xscale, "gemv_fp8_mfma: strides / scales");
#define GOP(PK)                                                                                                                      \
  hipLaunchKernelGGL((gemv_fp8_mfma_kernel<-1, PK>), dim3(cdiv(N, 16)), dim3(GF_THREADS), 0, (hipStream_t)stream, (const uint8_t*)W8, ldw, \
                     wscale, x8, ldx, xscale, (const bf16_t*)nullptr, 0.f, (const bf16_t*)residual, ldr, y, ldy, B, N, K, out_f32)
  if (w_packed) GOP(true); else GOP(false);
#undef GOP
  LHRS_CHECK_LAUNCH("gemv_fp8_mfma");
  return 0;
}

// the same with bf16 activations: prologue (0 none, 1 RMSNorm(norm_w, eps), 2 SwiGLU with x = [B, 2K]) + per-row e4m3 quantisation of x
// inside the kernel; B <= 2 (every block redoes it: the batch-1 decode step stays at five launches per layer)
extern "C" int lhrs_gemv_fp8_mfma_fused(const void* W8, long ldw, const float* wscale, const void* x, long ldx, int prologue, const void* norm_w,
                                        float eps, const void* residual, long ldr, void* y, long ldy, int B, int N, int K, int out_f32,
                                        int w_packed, void* stream) {
  LHRS_REQUIRE(B >= 1 && B <= 2 && N > 0 && K >= 128 && K % 128 == 0, "gemv_fp8_mfma_fused: B=%d (1..2) N=%d K=%d", B, N, K);
  LHRS_REQUIRE((w_packed || ldw % 16 == 0) && ldx % 8 == 0 && wscale && prologue >= 0 && prologue <= 2 && (prologue != 1 || norm_w),
               "gemv_fp8_mfma_fused: args");
  LHRS_REQUIRE(K <= GF_THREADS * GF_MAXC * 8, "gemv_fp8_mfma_fused: K=%d exceeds the %d values the prologue keeps in registers", K, GF_THREADS * GF_MAXC * 8);
  const size_t sm = (size_t)B * K;
  const dim3 grid(cdiv(N, 16)), blk(GF_THREADS);
  hipStream_t s = (hipStream_t)stream;
#define GOF(P, PK)                                                                                                                    \
  do {                                                                                                                                \
    if (sm > 65536) (void)hipFuncSetAttribute((const void*)gemv_fp8_mfma_kernel<P, PK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); \
    hipLaunchKernelGGL((gemv_fp8_mfma_kernel<P, PK>), grid, blk, sm, s, (const uint8_t*)W8, ldw, wscale, x, ldx, (const float*)nullptr, \
                       (const bf16_t*)norm_w, eps, (const bf16_t*)residual, ldr, y, ldy, B, N, K, out_f32);                          \
  } while (0)
  if (w_packed) { if (prologue == 0) GOF(0, true); else if (prologue == 1) GOF(1, true); else GOF(2, true); }
  else { if (prologue == 0) GOF(0, false); else if (prologue == 1) GOF(1, false); else GOF(2, false); }
#undef GOF
  LHRS_CHECK_LAUNCH("gemv_fp8_mfma_fused");
  return 0;
}

// W [N, ldw] bf16 rows -> out: ceil(N/16) * 16 * K bf16 in the operand order of the batched MFMA GEMV (see repack_bf16_mfma_kernel)
extern "C" int lhrs_repack_bf16_mfma(const void* W, long ldw, void* out, int N, int K, void* stream) {
  LHRS_REQUIRE(N > 0 && K >= 128 && K % 128 == 0 && ldw % 8 == 0, "repack_bf16_mfma: N=%d K=%d ldw=%ld", N, K, ldw);
  const long total = (long)cdiv(N, 16) * (K / 32) * 64;
  hipLaunchKernelGGL(repack_bf16_mfma_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)W, ldw,
                     (bf16_t*)out, N, K);
  LHRS_CHECK_LAUNCH("repack_bf16_mfma");
  return 0;
}

// W8 [N, ldw] e4m3 rows -> out: ceil(N/16) * 16 * K bytes in the operand order of the MFMA GEMV (see repack_fp8_mfma_kernel)
extern "C" int lhrs_repack_fp8_mfma(const void* W8, long ldw, void* out, int N, int K, void* stream) {
  LHRS_REQUIRE(N > 0 && K >= 128 && K % 128 == 0 && ldw % 16 == 0, "repack_fp8_mfma: N=%d K=%d ldw=%ld", N, K, ldw);
  const long total = (long)cdiv(N, 16) * (K / 128) * 128;
  hipLaunchKernelGGL(repack_fp8_mfma_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)W8, ldw,
                     (uint8_t*)out, N, K);
  LHRS_CHECK_LAUNCH("repack_fp8_mfma");
  return 0;
}
