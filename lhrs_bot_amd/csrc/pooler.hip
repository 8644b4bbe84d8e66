// Layout kernels specific to AttnPooler (/root/reference lhrs/models/common_arch.py:134-173).
//   build : t[b] = query (all 144 learned queries);  kv[b] = [q_g0 | img_g0 | q_g1 | img_g1 | q_g2 | img_g2]
//           i.e. cat(sub_token, sub_image) of common_arch.py:160-161 for the three groups, packed per sample so
//           that all groups run in ONE varlen attention launch and ONE GEMM per projection.
//   query_grad : d query = sum_b ( d t0[b] + d kv[b, query rows] )   (fp32)
#include "common.h"

namespace {

struct Groups { int nq[3]; int nimg[3]; };

__global__ __launch_bounds__(256) void pooler_build_kernel(const bf16_t* __restrict__ query, const bf16_t* __restrict__ img,
                                                           bf16_t* __restrict__ t, bf16_t* __restrict__ kv, Groups g, int NQ,
                                                           int NIMG, int KV, int dim) {
  // grid: (NQ + KV, B)
  const int r = blockIdx.x, b = blockIdx.y;
  const bf16_t* src;
  bf16_t* dst;
  if (r < NQ) { src = query + (long)r * dim; dst = t + ((long)b * NQ + r) * dim; }
  else {
    int j = r - NQ, qo = 0, io = 0, grp = 0;
    while (grp < 2 && j >= g.nq[grp] + g.nimg[grp]) { j -= g.nq[grp] + g.nimg[grp]; qo += g.nq[grp]; io += g.nimg[grp]; ++grp; }
    src = (j < g.nq[grp]) ? query + (long)(qo + j) * dim : img + ((long)b * NIMG + io + j - g.nq[grp]) * dim;
    dst = kv + ((long)b * KV + (r - NQ)) * dim;
  }
  for (int c = threadIdx.x; c < dim / 8; c += 256) *reinterpret_cast<uint4*>(dst + c * 8) = *reinterpret_cast<const uint4*>(src + c * 8);
}

__global__ __launch_bounds__(256) void pooler_query_grad_kernel(const bf16_t* __restrict__ dt0, const bf16_t* __restrict__ dkv,
                                                                float* __restrict__ dquery, Groups g, int B, int NQ, int KV,
                                                                int dim, int accumulate) {
  // grid: NQ blocks; thread per column
  const int r = blockIdx.x;
  int grp = 0, j = r, kvo = 0;
  while (grp < 2 && j >= g.nq[grp]) { j -= g.nq[grp]; kvo += g.nq[grp] + g.nimg[grp]; ++grp; }
  const int kvrow = kvo + j;
  for (int c = threadIdx.x; c < dim; c += 256) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
      s += bf2f(dt0[((long)b * NQ + r) * dim + c]);
      if (dkv) s += bf2f(dkv[((long)b * KV + kvrow) * dim + c]);
    }
    dquery[(long)r * dim + c] = accumulate ? dquery[(long)r * dim + c] + s : s;
  }
}

}  // namespace

extern "C" int lhrs_pooler_build(const void* query, const void* img, void* t, void* kv, int B, int nq0, int nq1, int nq2,
                                 int ni0, int ni1, int ni2, int dim, void* stream) {
  LHRS_REQUIRE(B > 0 && dim % 8 == 0, "pooler_build: B=%d dim=%d", B, dim);
  Groups g{{nq0, nq1, nq2}, {ni0, ni1, ni2}};
  const int NQ = nq0 + nq1 + nq2, NIMG = ni0 + ni1 + ni2, KV = NQ + NIMG;
  hipLaunchKernelGGL(pooler_build_kernel, dim3(NQ + KV, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)query,
                     (const bf16_t*)img, (bf16_t*)t, (bf16_t*)kv, g, NQ, NIMG, KV, dim);
  LHRS_CHECK_LAUNCH("pooler_build");
  return 0;
}

extern "C" int lhrs_pooler_query_grad(const void* dt0, const void* dkv, float* dquery, int B, int nq0, int nq1, int nq2,
                                      int ni0, int ni1, int ni2, int dim, int accumulate, void* stream) {
  LHRS_REQUIRE(B > 0 && dim > 0, "pooler_query_grad: B=%d dim=%d", B, dim);
  Groups g{{nq0, nq1, nq2}, {ni0, ni1, ni2}};
  const int NQ = nq0 + nq1 + nq2, KV = NQ + ni0 + ni1 + ni2;
  hipLaunchKernelGGL(pooler_query_grad_kernel, dim3(NQ), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dt0,
                     (const bf16_t*)dkv, dquery, g, B, NQ, KV, dim, accumulate);
  LHRS_CHECK_LAUNCH("pooler_query_grad");
  return 0;
}

// strided block copy (device to device) on the stream: `height` rows of `width_bytes`.  16-byte aligned blocks (every caller on the hot
// path: the ViT taps, the K / V rows of a prefill) go through ONE kernel launch - hipMemcpy2DAsync turns a pitched device copy into one blit
// per row (30 per tap at micro-batch 30: 0.7 ms of copy kernels per training step) - the rest falls back to the runtime copy.
namespace {
__global__ void copy2d_kernel(char* __restrict__ dst, long dpitch, const char* __restrict__ src, long spitch, long w16, long height) {
  const long total = w16 * height;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / w16, c = i - r * w16;
    *reinterpret_cast<uint4*>(dst + r * dpitch + c * 16) = *reinterpret_cast<const uint4*>(src + r * spitch + c * 16);
  }
}
}  // namespace
extern "C" int lhrs_copy_2d(void* dst, long dst_pitch_bytes, const void* src, long src_pitch_bytes, long width_bytes,
                            long height, void* stream) {
  LHRS_REQUIRE(width_bytes > 0 && height > 0, "copy_2d: width=%ld height=%ld", width_bytes, height);
  if (((width_bytes | dst_pitch_bytes | src_pitch_bytes | (long)(size_t)dst | (long)(size_t)src) & 15) == 0) {
    const long total = (width_bytes / 16) * height;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(copy2d_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (char*)dst, dst_pitch_bytes, (const char*)src,
                       src_pitch_bytes, width_bytes / 16, height);
    LHRS_CHECK_LAUNCH("copy_2d");
    return 0;
  }
  hipError_t e = hipMemcpy2DAsync(dst, (size_t)dst_pitch_bytes, src, (size_t)src_pitch_bytes, (size_t)width_bytes,
                                  (size_t)height, hipMemcpyDeviceToDevice, (hipStream_t)stream);
  if (e != hipSuccess) LHRS_FAIL("copy_2d: %s", hipGetErrorString(e));
  return 0;
}

// ---- test aid: fill every CU's LDS with a bit pattern ---------------------------------------------------------------------------------
// LDS is not cleared between kernels: a kernel that reads LDS it never wrote (a padded tail row, a skipped DMA) sees what the PREVIOUS kernel
// on that CU left there - finite leftovers of our own launches on a warm box, anything at all on a fresh one.  One 160 KiB workgroup per CU
// (nothing else fits beside it), several waves deep so that every CU is visited; tests/ and tools/poison_check.py launch it before an operator
// and demand the same result as without it.
namespace {
__global__ void __launch_bounds__(256) poison_lds_kernel(unsigned pattern, int words, unsigned* sink) {
  extern __shared__ unsigned lds_words[];
  for (int i = threadIdx.x; i < words; i += 256) lds_words[i] = pattern;
  __syncthreads();
  if (sink != nullptr && lds_words[(threadIdx.x * 97) % words] != pattern) sink[0] = 1;  // keeps the stores alive
}
}  // namespace
extern "C" int lhrs_debug_poison_lds(unsigned pattern, void* stream) {
  constexpr int kBytes = 160 * 1024;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kBytes);
    if (e != hipSuccess) LHRS_FAIL("debug_poison_lds: %s", hipGetErrorString(e));
    attr_set = true;
  }
  int dev = 0, cus = 256;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  hipLaunchKernelGGL(poison_lds_kernel, dim3((unsigned)(4 * cus)), dim3(256), kBytes, (hipStream_t)stream, pattern, kBytes / 4, (unsigned*)nullptr);
  LHRS_CHECK_LAUNCH("debug_poison_lds");
  return 0;
}
