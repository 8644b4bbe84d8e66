// CLIP image preprocessing on the GPU: uint8 HWC image -> float32 [3, 224, 224] pixel_values, bit-exact with
// what the reference's transform produces - `CLIPImageProcessor.preprocess` (/root/reference lhrs/Dataset/build_transform.py:43-45):
// Pillow BICUBIC resize of the short edge to 224 (libImaging/Resample.c, 8-bit path: two separable passes, 22-bit fixed-point
// coefficients, uint8 rounding after EACH pass), center crop, float32(float64(b)/255), (x - mean) / std.
//
// Integer work, HBM-bound and tiny (<= a few MB per image): three launches per image, nothing on the host but the 768-entry
// normalisation table (exact IEEE arithmetic of the numpy reference):
//   1. coeff_kernel    - per output column / row of the CROP window: Pillow's window bounds and integer coefficients, computed in
//                        double on the device with contraction off (every operation is a correctly rounded IEEE op, same order
//                        as Resample.c, so the integers are identical to the host library's)
//   2. hpass_kernel    - horizontal pass for the 224 crop columns of every input row -> uint8 [H][224][3]
//   3. vpass_kernel    - vertical pass for the 224 crop rows + table lookup -> float32 planar output
// A pass whose size does not change is the identity in Pillow (it is skipped); here it is a one-tap window with coefficient 2^22.
//
// `lhrs_image_preprocess` is the general entry: short edge -> `short_edge`, crop-offset rule and rescale arithmetic selectable, so that
// the evaluation transform of the classification caller (lhrs/Dataset/build_transform.py:27-40: torchvision Resize(256, BICUBIC) ->
// CenterCrop(224) -> ToTensor -> Normalize(ImageNet mean / std), used by build_zero_shot_loader) runs on the same three kernels.
#include "common.h"

namespace {

constexpr int PB = 32 - 8 - 2;  // PRECISION_BITS
constexpr int CROP = 224;

struct Axis { int in_size, out_size, crop0, ksize; };

__device__ double bicubic(double x) {
#pragma clang fp contract(off)
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

// bounds: [2][CROP][2] ints, kk: axis 0 at kk, axis 1 at kk + CROP * ax0.ksize
__global__ void coeff_kernel(Axis ax0, Axis ax1, int* __restrict__ bounds, int* __restrict__ kk) {
#pragma clang fp contract(off)
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * CROP) return;
  const int axis = t / CROP, i = t % CROP;
  const Axis ax = axis ? ax1 : ax0;
  int* k = kk + (axis ? CROP * ax0.ksize : 0) + i * ax.ksize;
  int* b = bounds + (axis * CROP + i) * 2;
  const int xx = ax.crop0 + i;
  if (ax.in_size == ax.out_size) {  // Pillow skips this pass
    b[0] = xx; b[1] = 1;
    k[0] = 1 << PB;
    for (int x = 1; x < ax.ksize; ++x) k[x] = 0;
    return;
  }
  const double scale = (double)ax.in_size / (double)ax.out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 2.0 * filterscale;
  const double center = 0.0 + (xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > ax.in_size) xmax = ax.in_size;
  xmax -= xmin;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) ww += bicubic((x + xmin - center + 0.5) * ss);
  for (int x = 0; x < xmax; ++x) {
    double w = bicubic((x + xmin - center + 0.5) * ss);
    if (ww != 0.0) w /= ww;
    k[x] = w < 0 ? (int)(-0.5 + w * (double)(1 << PB)) : (int)(0.5 + w * (double)(1 << PB));
  }
  for (int x = xmax; x < ax.ksize; ++x) k[x] = 0;
  b[0] = xmin; b[1] = xmax;
}

__device__ __forceinline__ unsigned char clip8(int v) {
  v >>= PB;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// tmp[y][i][c] for every input row y and crop column i
__global__ void hpass_kernel(const unsigned char* __restrict__ img, long stride, int H, const int* __restrict__ bounds,
                             const int* __restrict__ kk, int ksize, unsigned char* __restrict__ tmp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (i >= CROP) return;
  const int xmin = bounds[i * 2], xmax = bounds[i * 2 + 1];
  const int* k = kk + i * ksize;
  const unsigned char* p = img + (long)y * stride + (long)xmin * 3;
  int s0 = 1 << (PB - 1), s1 = s0, s2 = s0;
  for (int x = 0; x < xmax; ++x) {
    const int c = k[x];
    s0 += p[x * 3] * c; s1 += p[x * 3 + 1] * c; s2 += p[x * 3 + 2] * c;
  }
  unsigned char* o = tmp + ((long)y * CROP + i) * 3;
  o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

struct Lut { float v[3][256]; };

__global__ void vpass_kernel(const unsigned char* __restrict__ tmp, const int* __restrict__ bounds, const int* __restrict__ kk,
                             int ksize, Lut lut, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;  // column, row of the crop
  if (i >= CROP) return;
  const int ymin = bounds[j * 2], ymax = bounds[j * 2 + 1];
  const int* k = kk + j * ksize;
  int s0 = 1 << (PB - 1), s1 = s0, s2 = s0;
  for (int y = 0; y < ymax; ++y) {
    const unsigned char* p = tmp + ((long)(ymin + y) * CROP + i) * 3;
    const int c = k[y];
    s0 += p[0] * c; s1 += p[1] * c; s2 += p[2] * c;
  }
  const long o = (long)j * CROP + i;
  out[o] = lut.v[0][clip8(s0)];
  out[o + CROP * CROP] = lut.v[1][clip8(s1)];
  out[o + 2 * CROP * CROP] = lut.v[2][clip8(s2)];
}

// first row / column of the 224 crop window inside a resized axis of n pixels: HF `center_crop` floors (n - 224) / 2, torchvision's
// `CenterCrop` takes Python's round((n - 224) / 2.0) (half to even)
int crop_offset(int n, int crop_round) {
  const int d = n - CROP, k = d / 2;
  return (crop_round == 1 && (d & 1) && (k & 1)) ? k + 1 : k;
}

void plan(int H, int W, int short_edge, int crop_round, Axis& ax, Axis& ay) {
  int nh, nw;  // HF get_resize_output_image_size(shortest_edge, default_to_square = False) == torchvision Resize(int): long = int(size * long / short)
  if (H <= W) { nh = short_edge; nw = (int)((double)short_edge * W / H); } else { nh = (int)((double)short_edge * H / W); nw = short_edge; }
  auto ks = [](int in, int out) {
    double fs = (double)in / (double)out;
    if (fs < 1.0) fs = 1.0;
    return (int)ceil(2.0 * fs) * 2 + 1;
  };
  ax = Axis{W, nw, crop_offset(nw, crop_round), ks(W, nw)};
  ay = Axis{H, nh, crop_offset(nh, crop_round), ks(H, nh)};
}

long ws_bytes(int H, const Axis& ax, const Axis& ay) {
  long ints = 2L * CROP * 2 + (long)CROP * ax.ksize + (long)CROP * ay.ksize;
  return ints * 4 + (long)H * CROP * 3 + 64;
}

}  // namespace

extern "C" long lhrs_image_preprocess_workspace(int H, int W, int short_edge) {
  if (H <= 0 || W <= 0 || short_edge < CROP) return -1;
  Axis ax, ay;
  plan(H, W, short_edge, 0, ax, ay);
  return ws_bytes(H, ax, ay);
}

extern "C" long lhrs_clip_preprocess_workspace(int H, int W) { return lhrs_image_preprocess_workspace(H, W, CROP); }

extern "C" int lhrs_image_preprocess(const unsigned char* img, int H, int W, long row_stride, float* out, void* workspace,
                                     long workspace_bytes, int short_edge, int crop_round, int rescale_mode, const float* mean,
                                     const float* stdv, void* stream) {
  LHRS_REQUIRE(img != nullptr && out != nullptr && workspace != nullptr && mean != nullptr && stdv != nullptr, "image_preprocess: null pointer");
  LHRS_REQUIRE(H > 0 && W > 0 && row_stride >= 3L * W, "image_preprocess: H=%d W=%d row_stride=%ld", H, W, row_stride);
  LHRS_REQUIRE(short_edge >= CROP && (crop_round == 0 || crop_round == 1) && (rescale_mode == 0 || rescale_mode == 1),
               "image_preprocess: short_edge=%d (>= 224) crop_round=%d rescale_mode=%d", short_edge, crop_round, rescale_mode);
  Axis ax, ay;
  plan(H, W, short_edge, crop_round, ax, ay);
  LHRS_REQUIRE(workspace_bytes >= ws_bytes(H, ax, ay), "image_preprocess: workspace %ld < %ld bytes", workspace_bytes, ws_bytes(H, ax, ay));
  Lut lut;
  for (int b = 0; b < 256; ++b) {
    // mode 0: numpy rescale of HF's image processor (float64 product, then float32); mode 1: torchvision ToTensor (float32 division)
    const float r = rescale_mode == 0 ? (float)((double)b * 0.00392156862745098) : (float)b / 255.0f;
    for (int c = 0; c < 3; ++c) lut.v[c][b] = (r - mean[c]) / stdv[c];
  }
  int* bounds = (int*)workspace;
  int* kk = bounds + 2 * CROP * 2;
  unsigned char* tmp = (unsigned char*)(kk + (long)CROP * ax.ksize + (long)CROP * ay.ksize);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(coeff_kernel, dim3(cdiv(2 * CROP, 64)), dim3(64), 0, s, ax, ay, bounds, kk);
  LHRS_CHECK_LAUNCH("clip_preprocess_coeff");
  hipLaunchKernelGGL(hpass_kernel, dim3(1, H), dim3(256), 0, s, img, row_stride, H, bounds, kk, ax.ksize, tmp);
  LHRS_CHECK_LAUNCH("clip_preprocess_h");
  hipLaunchKernelGGL(vpass_kernel, dim3(1, CROP), dim3(256), 0, s, tmp, bounds + CROP * 2, kk + (long)CROP * ax.ksize, ay.ksize, lut, out);
  LHRS_CHECK_LAUNCH("clip_preprocess_v");
  return 0;
}

extern "C" int lhrs_clip_preprocess(const unsigned char* img, int H, int W, long row_stride, float* out, void* workspace,
                                    long workspace_bytes, void* stream) {
  static const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f}, stdv[3] = {0.26862954f, 0.26130258f, 0.27577711f};
  return lhrs_image_preprocess(img, H, W, row_stride, out, workspace, workspace_bytes, CROP, 0, 0, mean, stdv, stream);
}
