// probe: semantics of ds_read_b64_tr_b16 on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  unsigned base = (unsigned)(size_t)((__attribute__((address_space(3))) uint16_t*)lds);
  unsigned addr;
  if (mode == 0) addr = base + l * 8;                       // lane-linear: lane l -> elements 4l..4l+3
  else if (mode == 1) addr = base + ((l & 15) * 64 + (l >> 4) * 8) ;  // 16 rows of 32 elements (64 B), lane group g reads cols 4g..
  else addr = base + ((l >> 2) & 3) * 128 * 2 + (l & 3) * 8 + (l >> 4) * 2048;   // rows = (l>>2)&3 (stride 128 el), col chunk = l&3
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { printf("  lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); }
  }
  return 0;
}
