// Token-side kernels of the hot path (gfx950): image-token splice, row gather/scatter and cross-entropy.
//   splice : TextModal.prepare_inputs_for_multimodal, /root/reference lhrs/models/text_modal.py:296-526
//            (IMAGE_TOKEN_INDEX = -200, IGNORE_INDEX = -100: lhrs/models/__init__.py:1-6).  Bit-exact indexing.
//   CE     : shifted CrossEntropyLoss(ignore_index=-100) on logits.float() inside HF LlamaForCausalLM.forward,
//            called from text_modal.py:281-292; mean over the valid targets of the micro-batch.
#include "common.h"

namespace {

constexpr int IMAGE_TOKEN_INDEX = -200;
constexpr long IGNORE_INDEX = -100;

// One block per output row (j, b).  Every block re-derives p = first index of -200 in ids[b, :] (T is small).
__global__ __launch_bounds__(256) void splice_fwd_kernel(const long* __restrict__ ids, const long* __restrict__ labels,
                                                         const uint8_t* __restrict__ mask, const bf16_t* __restrict__ image,
                                                         const bf16_t* __restrict__ embed, bf16_t* __restrict__ out_embeds,
                                                         long* __restrict__ out_labels, uint8_t* __restrict__ out_mask,
                                                         int* __restrict__ img_pos, int T, int NI, int dim, int S, int vocab) {
  __shared__ int s_p;
  const int j = blockIdx.x, b = blockIdx.y;
  const long* row_ids = ids + (long)b * T;
  if (threadIdx.x == 0) s_p = T;
  __syncthreads();
  int local = T;
  for (int t = threadIdx.x; t < T; t += 256)
    if (row_ids[t] == IMAGE_TOKEN_INDEX) local = min(local, t);
  if (local < T) atomicMin(&s_p, local);
  __syncthreads();
  const int p = s_p;  // == T when the sample has no image token
  const bool has_img = p < T;
  const int new_len = has_img ? T - 1 + NI : T;
  const int shift = new_len - T;  // mask is left-extended by `shift` True entries (text_modal.py:511-524)

  // source of this output row
  int src_tok = -1;    // index into ids/labels, or -1
  int src_img = -1;    // row of image[b], or -1
  if (j < new_len) {
    if (!has_img || j < p) src_tok = j;
    else if (j < p + NI) src_img = j - p;
    else src_tok = j - NI + 1;
  }
  if (threadIdx.x == 0) {
    if (out_labels) out_labels[(long)b * S + j] = (src_tok >= 0 && labels) ? labels[(long)b * T + src_tok] : IGNORE_INDEX;
    if (out_mask) {
      uint8_t mv = 0;
      if (j < new_len) mv = (j < shift) ? 1 : (mask ? mask[(long)b * T + (j - shift)] : 1);
      out_mask[(long)b * S + j] = mv;
    }
    if (img_pos && j == 0) img_pos[b] = has_img ? p : -1;
  }
  const bf16_t* src = nullptr;
  if (src_img >= 0) src = image + ((long)b * NI + src_img) * dim;
  else if (src_tok >= 0) {
    long id = row_ids[src_tok];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    src = embed + id * dim;
  }
  bf16_t* dst = out_embeds + ((long)b * S + j) * dim;
  for (int c = threadIdx.x; c < dim / 8; c += 256) {
    const uint4 v = src ? *reinterpret_cast<const uint4*>(src + c * 8) : make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(dst + c * 8) = v;
  }
}

// d_image[b, i, :] = d_embeds[b, p_b + i, :]   (zero when the sample has no image token)
__global__ __launch_bounds__(256) void splice_bwd_kernel(const bf16_t* __restrict__ d_embeds, const int* __restrict__ img_pos,
                                                         bf16_t* __restrict__ d_image, int NI, int dim, int S) {
  const int i = blockIdx.x, b = blockIdx.y;
  const int p = img_pos[b];
  const bf16_t* src = p >= 0 ? d_embeds + ((long)b * S + p + i) * dim : nullptr;
  bf16_t* dst = d_image + ((long)b * NI + i) * dim;
  for (int c = threadIdx.x; c < dim / 8; c += 256) {
    const uint4 v = src ? *reinterpret_cast<const uint4*>(src + c * 8) : make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(dst + c * 8) = v;
  }
}

// General splice (several <image> placeholders per sample; text_modal.py:341-438): the host walked the placeholders (TextModal.splice_plan_host)
// and hands over, per output row, the token index it copies (src_tok >= 0), the row of the flattened image slots [n_slots * NI] it copies
// (src_img >= 0), or neither (right padding -> zeros).  One block per output row.
__global__ __launch_bounds__(256) void splice_map_fwd_kernel(const long* __restrict__ ids, const int* __restrict__ src_tok,
                                                             const int* __restrict__ src_img, const bf16_t* __restrict__ image,
                                                             const bf16_t* __restrict__ embed, bf16_t* __restrict__ out_embeds, int T, int dim,
                                                             int S, int vocab) {
  const int j = blockIdx.x, b = blockIdx.y;
  const int st = src_tok[(long)b * S + j], si = src_img[(long)b * S + j];
  const bf16_t* src = nullptr;
  if (si >= 0) src = image + (long)si * dim;
  else if (st >= 0) {
    long id = ids[(long)b * T + st];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    src = embed + id * dim;
  }
  bf16_t* dst = out_embeds + ((long)b * S + j) * dim;
  for (int c = threadIdx.x; c < dim / 8; c += 256) {
    const uint4 v = src ? *reinterpret_cast<const uint4*>(src + c * 8) : make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(dst + c * 8) = v;
  }
}
// d_image[r, :] = d_embeds[inv[r], :] for every row r of the flattened image slots; inv[r] < 0 (a slot no placeholder took): zeros
__global__ __launch_bounds__(256) void splice_map_bwd_kernel(const bf16_t* __restrict__ d_embeds, const int* __restrict__ inv,
                                                             bf16_t* __restrict__ d_image, int dim) {
  const int r = blockIdx.x;
  const int i = inv[r];
  const bf16_t* src = i >= 0 ? d_embeds + (long)i * dim : nullptr;
  bf16_t* dst = d_image + (long)r * dim;
  for (int c = threadIdx.x; c < dim / 8; c += 256) {
    const uint4 v = src ? *reinterpret_cast<const uint4*>(src + c * 8) : make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(dst + c * 8) = v;
  }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ src, long ld_src, const int* __restrict__ idx,
                                                          bf16_t* __restrict__ dst, long ld_dst, int dim) {
  const int r = blockIdx.x;
  const bf16_t* s = src + (long)idx[r] * ld_src;
  bf16_t* d = dst + (long)r * ld_dst;
  for (int c = threadIdx.x; c < dim / 8; c += 256) *reinterpret_cast<uint4*>(d + c * 8) = *reinterpret_cast<const uint4*>(s + c * 8);
}
__global__ __launch_bounds__(256) void scatter_rows_kernel(const bf16_t* __restrict__ src, long ld_src, const int* __restrict__ idx,
                                                           bf16_t* __restrict__ dst, long ld_dst, int dim) {
  const int r = blockIdx.x;
  const bf16_t* s = src + (long)r * ld_src;
  bf16_t* d = dst + (long)idx[r] * ld_dst;
  for (int c = threadIdx.x; c < dim / 8; c += 256) *reinterpret_cast<uint4*>(d + c * 8) = *reinterpret_cast<const uint4*>(s + c * 8);
}

// One block per row: online (max, sum) pass, then gradient pass.  V = 32000 bf16 = 64 KB per row: L2-resident.
__global__ __launch_bounds__(256) void ce_kernel(const bf16_t* logits, long ld, const int* __restrict__ target,
                                                 float* __restrict__ row_loss, bf16_t* dlogits, long ld_d, int V,
                                                 float grad_scale) {
  __shared__ float red[4];
  const int r = blockIdx.x;
  const bf16_t* x = logits + (long)r * ld;
  const int nch = V / 8;
  const int t = target[r];
  const float xt = bf2f(x[t]);  // read before any aliasing write of pass 2 (barriers of the reductions order it)
  float m = -__builtin_huge_valf(), s = 0.f;
  for (int c = threadIdx.x; c < nch; c += 256) {
    const uint4 u = *reinterpret_cast<const uint4*>(x + c * 8);
    const float v[8] = {bflo(u.x), bfhi(u.x), bflo(u.y), bfhi(u.y), bflo(u.z), bfhi(u.z), bflo(u.w), bfhi(u.w)};
    float cm = v[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) cm = fmaxf(cm, v[i]);
    const float mn = fmaxf(m, cm);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += __expf(v[i] - mn);
    s = s * __expf(m - mn) + acc;
    m = mn;
  }
  const float M = block_max<4>(m, red);
  const float ssum = block_sum<4>(s * __expf(m - M), red);
  const float lse = M + __logf(ssum);
  if (threadIdx.x == 0) row_loss[r] = lse - xt;
  if (!dlogits) return;
  bf16_t* d = dlogits + (long)r * ld_d;
  for (int c = threadIdx.x; c < nch; c += 256) {
    const uint4 u = *reinterpret_cast<const uint4*>(x + c * 8);
    float v[8] = {bflo(u.x), bfhi(u.x), bflo(u.y), bfhi(u.y), bflo(u.z), bfhi(u.z), bflo(u.w), bfhi(u.w)};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float p = __expf(v[i] - lse);
      v[i] = (p - ((c * 8 + i) == t ? 1.f : 0.f)) * grad_scale;
    }
    *reinterpret_cast<uint4*>(d + c * 8) =
        make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
  }
}

// loss = scale * sum(row_loss[0..n)) in a fixed order (single block): deterministic
__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ v, int n, float scale, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += v[i];
  s = block_sum<4>(s, red);
  if (threadIdx.x == 0) *out = s * scale;
}

// greedy pick: index of the first maximum of each fp32 row (HF argmax tie rule: lowest index).  One 16-wave block per row; a thread
// requests eight 16-B pieces before comparing any (the row was just written by other CUs: a dependent-load loop pays one L2 round trip
// per element - 39 us for 32000 logits - instead of one for all of them).
constexpr int AM_THREADS = 1024, AM_UNROLL = 8;
__device__ __forceinline__ void am_take(float v, int i, float& best, int& bi) {
  if (v > best || (v == best && i < bi)) { best = v; bi = i; }
}
__global__ __launch_bounds__(AM_THREADS) void argmax_rows_kernel(const float* __restrict__ x, long ld, long* __restrict__ out, int V, int vec) {
  __shared__ float sv[AM_THREADS / 64];
  __shared__ int si[AM_THREADS / 64];
  const float* row = x + (long)blockIdx.x * ld;
  float best = -__builtin_huge_valf();
  int bi = 0x7fffffff;
  if (vec) {
    const int nv = V / 4;
    for (int c0 = threadIdx.x; c0 < nv; c0 += AM_THREADS * AM_UNROLL) {
      float4 t[AM_UNROLL];
#pragma unroll
      for (int u = 0; u < AM_UNROLL; ++u) {
        const int c = c0 + u * AM_THREADS;
        t[u] = c < nv ? *reinterpret_cast<const float4*>(row + (long)c * 4) : make_float4(best, best, best, best);
      }
#pragma unroll
      for (int u = 0; u < AM_UNROLL; ++u) {
        const int i = (c0 + u * AM_THREADS) * 4;
        if (i < V) { am_take(t[u].x, i, best, bi); am_take(t[u].y, i + 1, best, bi); am_take(t[u].z, i + 2, best, bi); am_take(t[u].w, i + 3, best, bi); }
      }
    }
    for (int i = nv * 4 + threadIdx.x; i < V; i += AM_THREADS) am_take(row[i], i, best, bi);
  } else {
    for (int i = threadIdx.x; i < V; i += AM_THREADS) am_take(row[i], i, best, bi);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    am_take(ov, oi, best, bi);
  }
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < AM_THREADS / 64; ++w) am_take(sv[w], si[w], best, bi);
    out[blockIdx.x] = bi;
  }
}

}  // namespace

extern "C" int lhrs_argmax_rows(const float* x, long ld, long* out, int n, int V, void* stream) {
  LHRS_REQUIRE(n > 0 && V > 0, "argmax_rows: n=%d V=%d", n, V);
  const int vec = ld % 4 == 0 && ((uintptr_t)x & 15) == 0;  // 16-B aligned rows
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(n), dim3(AM_THREADS), 0, (hipStream_t)stream, x, ld, out, V, vec);
  LHRS_CHECK_LAUNCH("argmax_rows");
  return 0;
}

extern "C" int lhrs_splice_fwd(const long* ids, const long* labels, const uint8_t* mask, const void* image,
                               const void* embed, void* out_embeds, long* out_labels, uint8_t* out_mask, int* img_pos,
                               int B, int T, int NI, int dim, int S, int vocab, void* stream) {
  LHRS_REQUIRE(B > 0 && T > 0 && S >= T && dim % 8 == 0, "splice_fwd: B=%d T=%d S=%d dim=%d", B, T, S, dim);
  hipLaunchKernelGGL(splice_fwd_kernel, dim3(S, B), dim3(256), 0, (hipStream_t)stream, ids, labels, mask, (const bf16_t*)image,
                     (const bf16_t*)embed, (bf16_t*)out_embeds, out_labels, out_mask, img_pos, T, NI, dim, S, vocab);
  LHRS_CHECK_LAUNCH("splice_fwd");
  return 0;
}

extern "C" int lhrs_splice_map_fwd(const long* ids, const int* src_tok, const int* src_img, const void* image, const void* embed,
                                   void* out_embeds, int B, int T, int dim, int S, int vocab, void* stream) {
  LHRS_REQUIRE(B > 0 && T > 0 && S > 0 && dim % 8 == 0, "splice_map_fwd: B=%d T=%d S=%d dim=%d", B, T, S, dim);
  hipLaunchKernelGGL(splice_map_fwd_kernel, dim3(S, B), dim3(256), 0, (hipStream_t)stream, ids, src_tok, src_img, (const bf16_t*)image,
                     (const bf16_t*)embed, (bf16_t*)out_embeds, T, dim, S, vocab);
  LHRS_CHECK_LAUNCH("splice_map_fwd");
  return 0;
}
extern "C" int lhrs_splice_map_bwd(const void* d_embeds, const int* inv, void* d_image, int n_rows, int dim, void* stream) {
  LHRS_REQUIRE(n_rows > 0 && dim % 8 == 0, "splice_map_bwd: n_rows=%d dim=%d", n_rows, dim);
  hipLaunchKernelGGL(splice_map_bwd_kernel, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)d_embeds, inv, (bf16_t*)d_image, dim);
  LHRS_CHECK_LAUNCH("splice_map_bwd");
  return 0;
}
extern "C" int lhrs_splice_bwd(const void* d_embeds, const int* img_pos, void* d_image, int B, int NI, int dim, int S,
                               void* stream) {
  LHRS_REQUIRE(B > 0 && NI > 0 && dim % 8 == 0, "splice_bwd: B=%d NI=%d dim=%d", B, NI, dim);
  hipLaunchKernelGGL(splice_bwd_kernel, dim3(NI, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)d_embeds, img_pos,
                     (bf16_t*)d_image, NI, dim, S);
  LHRS_CHECK_LAUNCH("splice_bwd");
  return 0;
}

extern "C" int lhrs_gather_rows(const void* src, long ld_src, const int* idx, void* dst, long ld_dst, int n, int dim,
                                void* stream) {
  LHRS_REQUIRE(n > 0 && dim % 8 == 0, "gather_rows: n=%d dim=%d", n, dim);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, ld_src, idx,
                     (bf16_t*)dst, ld_dst, dim);
  LHRS_CHECK_LAUNCH("gather_rows");
  return 0;
}
extern "C" int lhrs_scatter_rows(const void* src, long ld_src, const int* idx, void* dst, long ld_dst, int n, int dim,
                                 void* stream) {
  LHRS_REQUIRE(n > 0 && dim % 8 == 0, "scatter_rows: n=%d dim=%d", n, dim);
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, ld_src, idx,
                     (bf16_t*)dst, ld_dst, dim);
  LHRS_CHECK_LAUNCH("scatter_rows");
  return 0;
}

// loss_out = mean_r (lse_r - logit_r[target_r]);  dlogits = (softmax - onehot) / n  (may alias logits, may be null)
extern "C" int lhrs_cross_entropy(const void* logits, long ld, const int* target, float* row_loss, float* loss_out,
                                  void* dlogits, long ld_d, int n, int V, void* stream) {
  LHRS_REQUIRE(n > 0 && V % 8 == 0, "cross_entropy: n=%d V=%d", n, V);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_kernel, dim3(n), dim3(256), 0, s, (const bf16_t*)logits, ld, target, row_loss, (bf16_t*)dlogits, ld_d, V,
                     1.f / (float)n);
  LHRS_CHECK_LAUNCH("cross_entropy");
  hipLaunchKernelGGL(sum_rows_kernel, dim3(1), dim3(256), 0, s, row_loss, n, 1.f / (float)n, loss_out);
  LHRS_CHECK_LAUNCH("cross_entropy_sum");
  return 0;
}
