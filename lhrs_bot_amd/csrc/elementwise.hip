// HBM-bound element-wise / layout kernels of the hot path (gfx950).  All bf16 traffic is 16 B per lane.
//   patchify + assemble : HF CLIPVisionEmbeddings (conv 14x14 stride 14 as im2col, class token, position table)
//                         reached from lhrs/models/rgb_vision_modal.py:166-172
//   rope                : HF apply_rotary_pos_emb (rotate_half convention) inside LlamaAttention
//   swiglu fwd / bwd    : HF LlamaMLP  down(silu(gate(x)) * up(x))
//   gelu fwd / bwd      : nn.GELU (erf) in lhrs/models/common_arch.py:286-292
//   colsum              : bias gradients of the projector linears
#include "common.h"

namespace {


// ---- im2col for the 14x14/stride-14 patch conv.  out[(b*GP*GP + py*GP + px)][k], k = c*P*P + i*P + j, zero for k >= 3*P*P
__global__ void patchify_kernel(const float* __restrict__ rgb, bf16_t* __restrict__ out, int B, int img, int P, int KP) {
  const int GP = img / P, K = 3 * P * P;
  const long total = (long)B * GP * GP * KP;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int k = idx % KP;
    const long row = idx / KP;
    float v = 0.f;
    if (k < K) {
      const int p = row % (GP * GP), b = row / (GP * GP);
      const int py = p / GP, px = p % GP;
      const int c = k / (P * P), ij = k % (P * P), i = ij / P, j = ij % P;
      v = rgb[(((long)b * 3 + c) * img + py * P + i) * img + px * P + j];
    }
    out[idx] = f2bf(v);
  }
}

// ---- out[b, 0, :] = cls + pos[0];  out[b, 1+p, :] = patch[b*NP + p, :] + pos[1+p]
__global__ void vit_assemble_kernel(const bf16_t* __restrict__ patch, const bf16_t* __restrict__ cls,
                                    const bf16_t* __restrict__ pos, bf16_t* __restrict__ out, int B, int NP, int dim) {
  const int chunks = dim / 8;
  const long total = (long)B * (NP + 1) * chunks;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = idx % chunks;
    const long row = idx / chunks;
    const int t = row % (NP + 1), b = row / (NP + 1);
    float x[8], p[8];
    const bf16_t* src = (t == 0) ? cls + c * 8 : patch + ((long)b * NP + t - 1) * dim + c * 8;
    unpack8(*reinterpret_cast<const uint4*>(src), x);
    unpack8(*reinterpret_cast<const uint4*>(pos + (long)t * dim + c * 8), p);
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] += p[i];
    *reinterpret_cast<uint4*>(out + row * dim + c * 8) = pack8(x);
  }
}

// ---- RoPE in place on `nheads` consecutive heads of width D starting at x (+ row*ld).
// out[i] = x[i]*cos[i] - x[i+D/2]*sin[i];  out[i+D/2] = x[i+D/2]*cos[i] + x[i]*sin[i];  inverse flips sin.
__global__ void rope_kernel(bf16_t* x, long ld, int rows, int nheads, int D, const float* __restrict__ cos_t,
                            const float* __restrict__ sin_t, const int* __restrict__ pos_ids, int pos_mod, int pos0,
                            float sin_sign) {
  const int half = D / 2, cph = half / 8;  // 16-B chunks per half head
  const long total = (long)rows * nheads * cph;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = idx % cph;
    const long rh = idx / cph;
    const int h = rh % nheads;
    const long row = rh / nheads;
    const int pos = pos_ids ? pos_ids[row] : (int)(row % pos_mod) + pos0;
    bf16_t* p1 = x + row * ld + (long)h * D + c * 8;
    bf16_t* p2 = p1 + half;
    float a[8], b[8], o1[8], o2[8];
    unpack8(*reinterpret_cast<const uint4*>(p1), a);
    unpack8(*reinterpret_cast<const uint4*>(p2), b);
    const float* cs = cos_t + (long)pos * half + c * 8;
    const float* sn = sin_t + (long)pos * half + c * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      rope_pair(a[i], b[i], cs[i], sn[i] * sin_sign, o1[i], o2[i]);
    }
    *reinterpret_cast<uint4*>(p1) = pack8(o1);
    *reinterpret_cast<uint4*>(p2) = pack8(o2);
  }
}

// ---- LoRA dropout: out[m, n] = keep(m * cols + n) ? x[m, n] / (1 - p) : 0
__global__ void dropout_kernel(const bf16_t* __restrict__ x, long ldx, bf16_t* __restrict__ out, long ldo, long rows, int cols, float scale,
                               unsigned seed, unsigned thresh) {
  const int chunks = cols / 8;
  const long total = rows * chunks;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = idx % chunks;
    const long row = idx / chunks;
    float v[8];
    unpack8(*reinterpret_cast<const uint4*>(x + row * ldx + c * 8), v);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = drop_keep(seed, row * cols + c * 8 + i, thresh) ? v[i] * scale : 0.f;
    *reinterpret_cast<uint4*>(out + row * ldo + c * 8) = pack8(v);
  }
}

// ---- SwiGLU: gu[m, 0:F] = gate, gu[m, F:2F] = up
__global__ void swiglu_fwd_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ act, long rows, int F) {
  const int chunks = F / 8;
  const long total = rows * chunks;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = idx % chunks;
    const long row = idx / chunks;
    float g[8], u[8];
    unpack8(*reinterpret_cast<const uint4*>(gu + row * 2 * F + c * 8), g);
    unpack8(*reinterpret_cast<const uint4*>(gu + row * 2 * F + F + c * 8), u);
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = silu(g[i]) * u[i];
    *reinterpret_cast<uint4*>(act + row * F + c * 8) = pack8(g);
  }
}
// dgu may alias gu (each thread reads its g,u before writing the same slots)
__global__ void swiglu_bwd_kernel(const bf16_t* dact, const bf16_t* gu, bf16_t* dgu, long rows, int F) {
  const int chunks = F / 8;
  const long total = rows * chunks;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = idx % chunks;
    const long row = idx / chunks;
    float g[8], u[8], d[8], dg[8], du[8];
    unpack8(*reinterpret_cast<const uint4*>(gu + row * 2 * F + c * 8), g);
    unpack8(*reinterpret_cast<const uint4*>(gu + row * 2 * F + F + c * 8), u);
    unpack8(*reinterpret_cast<const uint4*>(dact + row * F + c * 8), d);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float sg = sigmoid_f(g[i]);
      du[i] = d[i] * g[i] * sg;
      dg[i] = d[i] * u[i] * sg * (1.f + g[i] * (1.f - sg));
    }
    *reinterpret_cast<uint4*>(dgu + row * 2 * F + c * 8) = pack8(dg);
    *reinterpret_cast<uint4*>(dgu + row * 2 * F + F + c * 8) = pack8(du);
  }
}

// ---- unary / binary maps over n (multiple of 8) bf16 elements
enum { OP_GELU = 0, OP_GELU_BWD = 1, OP_ADD = 2, OP_QGELU = 3 };
template <int OP>
__global__ void map_kernel(const bf16_t* a, const bf16_t* b, bf16_t* out, long n8) {
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n8; idx += (long)gridDim.x * blockDim.x) {
    float x[8], y[8];
    unpack8(*reinterpret_cast<const uint4*>(a + idx * 8), x);
    if (OP == OP_GELU_BWD || OP == OP_ADD) unpack8(*reinterpret_cast<const uint4*>(b + idx * 8), y);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == OP_GELU) x[i] = gelu_erf(x[i]);
      else if (OP == OP_QGELU) x[i] = quick_gelu(x[i]);
      else if (OP == OP_GELU_BWD) x[i] = x[i] * gelu_erf_grad(y[i]);  // a = dY, b = pre-activation
      else x[i] = x[i] + y[i];
    }
    *reinterpret_cast<uint4*>(out + idx * 8) = pack8(x);
  }
}

// ---- column sums: partial[split][n] = sum over this split's rows of x[:, n]
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16_t* __restrict__ x, long ld, float* __restrict__ partial,
                                                             int rows, int cols) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rg = threadIdx.x >> 6;
  const int nsplit = gridDim.y;
  const int per = (rows + nsplit - 1) / nsplit;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  float s = 0.f;
  if (col < cols)
    for (int r = r0 + rg; r < r1; r += 4) s += bf2f(x[(long)r * ld + col]);
  red[rg][threadIdx.x & 63] = s;
  __syncthreads();
  if (rg == 0 && col < cols)
    partial[(long)blockIdx.y * cols + col] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int nsplit, int cols,
                                    int accumulate) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cols) return;
  float s = 0.f;
  for (int k = 0; k < nsplit; ++k) s += partial[(long)k * cols + j];
  out[j] = accumulate ? out[j] + s : s;
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = f2bf(in[i]);
}
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = bf2f(in[i]);
}

// plain 2-D transpose with zero padding of the (new) inner dimension: out[c][r] = in[r][c], r in [0, rows_pad)
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ in, long ld_in, bf16_t* __restrict__ out,
                                                        long ld_out, int rows, int cols, int rows_pad) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int rr = r0 + r, c = c0 + tx;
    tile[r][tx] = (rr < rows && c < cols) ? in[(long)rr * ld_in + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r, rr = r0 + tx;
    if (c < cols && rr < rows_pad) out[(long)c * ld_out + rr] = tile[tx][r];
  }
}

// many small transposes in ONE launch (the LoRA operand refresh after an optimizer step: 8 per layer): desc[i] = {src, dst, ld_in, ld_out,
// rows, cols, first_tile}; a block finds its matrix by a linear scan of first_tile (n is a few hundred)
__global__ __launch_bounds__(256) void transpose_batched_kernel(const long* __restrict__ desc, int n) {
  __shared__ bf16_t tile[64][66];
  const int t = blockIdx.x;
  int i = 0;
  while (i + 1 < n && desc[(i + 1) * 7 + 6] <= t) ++i;
  const bf16_t* in = reinterpret_cast<const bf16_t*>(desc[i * 7]);
  bf16_t* out = reinterpret_cast<bf16_t*>(desc[i * 7 + 1]);
  const long ld_in = desc[i * 7 + 2], ld_out = desc[i * 7 + 3];
  const int rows = (int)desc[i * 7 + 4], cols = (int)desc[i * 7 + 5];
  const int lt = t - (int)desc[i * 7 + 6], tr = (rows + 63) / 64;
  const int r0 = (lt % tr) * 64, c0 = (lt / tr) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int rr = r0 + r, c = c0 + tx;
    tile[r][tx] = (rr < rows && c < cols) ? in[(long)rr * ld_in + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r, rr = r0 + tx;
    if (c < cols && rr < rows) out[(long)c * ld_out + rr] = tile[tx][r];
  }
}

inline int grid_for(long work, int block = 256) {
  long g = (work + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" int lhrs_patchify(const float* rgb, void* out, int B, int img, int P, int KP, void* stream) {
  LHRS_REQUIRE(B > 0 && img % P == 0 && KP >= 3 * P * P && KP % 64 == 0, "patchify: B=%d img=%d P=%d KP=%d", B, img, P, KP);
  const long total = (long)B * (img / P) * (img / P) * KP;
  hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, rgb, (bf16_t*)out, B, img, P, KP);
  LHRS_CHECK_LAUNCH("patchify");
  return 0;
}

extern "C" int lhrs_vit_assemble(const void* patch, const void* cls, const void* pos, void* out, int B, int NP, int dim,
                                 void* stream) {
  LHRS_REQUIRE(B > 0 && dim % 8 == 0, "vit_assemble: B=%d dim=%d", B, dim);
  const long total = (long)B * (NP + 1) * (dim / 8);
  hipLaunchKernelGGL(vit_assemble_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)patch,
                     (const bf16_t*)cls, (const bf16_t*)pos, (bf16_t*)out, B, NP, dim);
  LHRS_CHECK_LAUNCH("vit_assemble");
  return 0;
}

extern "C" int lhrs_rope(void* x, long ld, int rows, int nheads, int D, const float* cos_t, const float* sin_t,
                         const int* pos_ids, int pos_mod, int pos0, int inverse, void* stream) {
  LHRS_REQUIRE(rows > 0 && nheads > 0 && D % 16 == 0 && ld % 8 == 0, "rope: rows=%d nheads=%d D=%d ld=%ld", rows, nheads, D, ld);
  LHRS_REQUIRE(pos_ids != nullptr || pos_mod > 0, "rope: need pos_ids or pos_mod");
  const long total = (long)rows * nheads * (D / 16);
  hipLaunchKernelGGL(rope_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, ld, rows, nheads, D,
                     cos_t, sin_t, pos_ids, pos_mod, pos0, inverse ? -1.f : 1.f);
  LHRS_CHECK_LAUNCH("rope");
  return 0;
}

extern "C" int lhrs_swiglu_fwd(const void* gate_up, void* act, long rows, int F, void* stream) {
  LHRS_REQUIRE(rows > 0 && F % 8 == 0, "swiglu_fwd: rows=%ld F=%d", rows, F);
  hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for(rows * (F / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)gate_up, (bf16_t*)act, rows, F);
  LHRS_CHECK_LAUNCH("swiglu_fwd");
  return 0;
}
extern "C" int lhrs_swiglu_bwd(const void* dact, const void* gate_up, void* dgate_up, long rows, int F, void* stream) {
  LHRS_REQUIRE(rows > 0 && F % 8 == 0, "swiglu_bwd: rows=%ld F=%d", rows, F);
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for(rows * (F / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dact, (const bf16_t*)gate_up, (bf16_t*)dgate_up, rows, F);
  LHRS_CHECK_LAUNCH("swiglu_bwd");
  return 0;
}

// op: 0 gelu(a), 1 a * gelu'(b), 2 a + b, 3 quick_gelu(a);  out may alias a
extern "C" int lhrs_map(int op, const void* a, const void* b, void* out, long n, void* stream) {
  LHRS_REQUIRE(n > 0 && n % 8 == 0, "map: n=%ld must be a positive multiple of 8", n);
  const long n8 = n / 8;
  const dim3 g(grid_for(n8)), blk(256);
  hipStream_t s = (hipStream_t)stream;
  switch (op) {
    case OP_GELU: hipLaunchKernelGGL((map_kernel<OP_GELU>), g, blk, 0, s, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n8); break;
    case OP_GELU_BWD: hipLaunchKernelGGL((map_kernel<OP_GELU_BWD>), g, blk, 0, s, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n8); break;
    case OP_ADD: hipLaunchKernelGGL((map_kernel<OP_ADD>), g, blk, 0, s, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n8); break;
    case OP_QGELU: hipLaunchKernelGGL((map_kernel<OP_QGELU>), g, blk, 0, s, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n8); break;
    default: LHRS_FAIL("map: unknown op %d", op);
  }
  LHRS_CHECK_LAUNCH("map");
  return 0;
}

extern "C" int lhrs_colsum_nsplit(int rows) { int n = cdiv(rows, 256); return n < 1 ? 1 : (n > 64 ? 64 : n); }

// partial: lhrs_colsum_nsplit(rows) * cols floats
extern "C" int lhrs_colsum(const void* x, long ld, float* out, float* partial, int rows, int cols, int accumulate,
                           void* stream) {
  LHRS_REQUIRE(rows > 0 && cols > 0 && partial != nullptr, "colsum: rows=%d cols=%d", rows, cols);
  const int ns = lhrs_colsum_nsplit(rows);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(cols, 64), ns), dim3(256), 0, s, (const bf16_t*)x, ld, partial, rows, cols);
  LHRS_CHECK_LAUNCH("colsum_partial");
  hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, s, partial, out, ns, cols, accumulate);
  LHRS_CHECK_LAUNCH("colsum_final");
  return 0;
}

extern "C" int lhrs_cast_f32_to_bf16(const float* in, void* out, long n, void* stream) {
  LHRS_REQUIRE(n > 0, "cast: n=%ld", n);
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)out, n);
  LHRS_CHECK_LAUNCH("cast_f32_to_bf16");
  return 0;
}
extern "C" int lhrs_cast_bf16_to_f32(const void* in, float* out, long n, void* stream) {
  LHRS_REQUIRE(n > 0, "cast: n=%ld", n);
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, out, n);
  LHRS_CHECK_LAUNCH("cast_bf16_to_f32");
  return 0;
}

// out[c][r] = in[r][c]; out rows have stride ld_out >= rows_pad; r in [rows, rows_pad) is written as zero
extern "C" int lhrs_transpose(const void* in, long ld_in, void* out, long ld_out, int rows, int cols, int rows_pad,
                              void* stream) {
  LHRS_REQUIRE(rows > 0 && cols > 0 && rows_pad >= rows && ld_out >= rows_pad, "transpose: rows=%d cols=%d rows_pad=%d ld_out=%ld",
               rows, cols, rows_pad, ld_out);
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(rows_pad, 64), cdiv(cols, 64)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)in, ld_in, (bf16_t*)out, ld_out, rows, cols, rows_pad);
  LHRS_CHECK_LAUNCH("transpose");
  return 0;
}

// desc: device int64 [n][7] = {src, dst, ld_in, ld_out, rows, cols, first_tile}, first_tile ascending; total_tiles = sum of ceil(rows/64)*ceil(cols/64)
extern "C" int lhrs_transpose_batched(const long* desc, int n, int total_tiles, void* stream) {
  LHRS_REQUIRE(desc != nullptr && n > 0 && total_tiles > 0, "transpose_batched: n=%d tiles=%d", n, total_tiles);
  hipLaunchKernelGGL(transpose_batched_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, desc, n);
  LHRS_CHECK_LAUNCH("transpose_batched");
  return 0;
}

// out = dropout(x) with keep probability 1 - p and the counter-based mask of common.h (peft lora_dropout); cols % 8 == 0
extern "C" int lhrs_dropout_bf16(const void* x, long ldx, void* out, long ldo, long rows, int cols, float p, unsigned seed, void* stream) {
  LHRS_REQUIRE(rows > 0 && cols % 8 == 0 && p >= 0.f && p < 1.f && ldx % 8 == 0 && ldo % 8 == 0, "dropout: rows=%ld cols=%d p=%f", rows, cols, p);
  const unsigned thresh = (unsigned)((double)p * 4294967296.0);
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(rows * (cols / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (bf16_t*)out, ldo,
                     rows, cols, 1.f / (1.f - p), seed, thresh);
  LHRS_CHECK_LAUNCH("dropout");
  return 0;
}
