// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3, unit scales) operand layout on gfx950: lane (r = lane&31, h = lane>>5) feeds 32
// consecutive k-bytes [32h, 32h+32) of row r for both operands; prints max |D - reference| for D[i][j] = sum_k A[i][k] * B[j][k].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void k(const unsigned char* A, const unsigned char* B, float* D) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = ((const int*)(A + r * 64 + h * 32))[i]; b[i] = ((const int*)(B + r * 64 + h * 32))[i]; }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
  for (int i = 0; i < 16; ++i) D[lane * 16 + i] = c[i];
}

static float e4m3(unsigned char v) {  // OCP e4m3fn
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -x : x;
}

int main() {
  std::vector<unsigned char> A(32 * 64), B(32 * 64);
  srand(1);
  for (auto& v : A) { v = rand() & 0xFF; if ((v & 0x7F) == 0x7F) v &= 0xFE; }
  for (auto& v : B) { v = rand() & 0xFF; if ((v & 0x7F) == 0x7F) v &= 0xFE; }
  unsigned char *dA, *dB; float* dD;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 64 * 16 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  std::vector<float> D(64 * 16);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  double e1 = 0, e2 = 0, mag = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int reg = 0; reg < 16; ++reg) {
      const int col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      double r1 = 0, r2 = 0;
      for (int kk = 0; kk < 64; ++kk) { r1 += (double)e4m3(A[row * 64 + kk]) * e4m3(B[col * 64 + kk]); r2 += (double)e4m3(A[col * 64 + kk]) * e4m3(B[row * 64 + kk]); }
      e1 = fmax(e1, fabs(D[lane * 16 + reg] - r1)); e2 = fmax(e2, fabs(D[lane * 16 + reg] - r2)); mag = fmax(mag, fabs(r1));
    }
  printf("max|D - A[row]B[col]| = %g   max|D - A[col]B[row]| = %g   (max |ref| %g)\n", e1, e2, mag);
  return 0;
}
