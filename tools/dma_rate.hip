// micro-benchmark: how fast can ONE workgroup per CU fill its LDS from L2-resident memory?  16 waves per workgroup (the GEMM's shape), every
// "stage" = 64 KiB = four 1-KiB pieces per wave, operands re-read from a per-workgroup 256 KiB window so that they stay in the XCD's L2.
//   mode 0: global_load_lds_dwordx4 (LDS-DMA, the GEMM's path), wait + barrier per stage
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128 (register staged), wait + barrier per stage
//   mode 2: half the pieces by each path
//   mode 3: LDS-DMA, two stages in flight (wait for the older one only), no barrier
// Rows: like the GEMM's DMA, a piece is 8 rows x 128 B at a row stride of `ld` bytes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE>
__global__ __launch_bounds__(1024, 1) void k(const char* __restrict__ src, long win, int ld, int stages, float* out, int share) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 65536];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // workgroups b, b + 8, ... run on one XCD (round-robin dispatch): `share` of them read the same window, as the tiles of a GEMM round share panels
  const char* base = src + (long)((blockIdx.x & 7) + 8 * ((blockIdx.x >> 3) / share)) * win;
  // wave w, piece j: rows (w * 4 + j) * 8 + (lane >> 3), chunk lane & 7; stage s reads k-offset (s % (ld / 128)) * 128
  unsigned off[4];
  for (int j = 0; j < 4; ++j) off[j] = (unsigned)(((wave * 4 + j) * 8 + (lane >> 3)) * ld + (lane & 7) * 16);
  const int nko = ld / 128;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int s = 0; s < stages; ++s) {
    const int ko = (s % nko) * 128;
    char* dst = smem + (s & 1) * 65536 + wave * 4096;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool dma = MODE == 0 || MODE == 3 || (MODE == 2 && (j & 1));
      if (dma) {
        __builtin_amdgcn_global_load_lds((gptr_t)(base + off[j] + ko), (lptr_t)(dst + j * 1024), 16, 0, 0);
      } else {
        const uint4 v = *reinterpret_cast<const uint4*>(base + off[j] + ko);
        *reinterpret_cast<uint4*>(dst + j * 1024 + lane * 16) = v;
      }
    }
    if (MODE == 3) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if ((s & 63) == 63) {  // consume something so that nothing is optimised away
      const uint4 r = *reinterpret_cast<const uint4*>(smem + tid * 16);
      acc.x ^= r.x; acc.y ^= r.y;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[blockIdx.x * 1024 + tid] = (float)(acc.x ^ acc.y);
}

template <int MODE> void run(const char* d, float* o, int ld, int stages, int share) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const long win = 512L * ld;  // 512 rows (A tile + B tile)
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, d, win, ld, 256, o, share);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, d, win, ld, stages, o, share);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 256.0 * stages * 65536;
  printf("mode %d share %2d ld %6d: %7.1f GB/s per CU, %6.2f TB/s chip, %.3f us per 64 KiB stage\n", MODE, share, ld, bytes / 256 / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12,
         ms * 1e3 / stages);
}

int main(int argc, char** argv) {
  const int stages = argc > 1 ? atoi(argv[1]) : 20000;
  char* d; float* o;
  const long bytes = 256L * 512 * 8192;  // 256 windows of 512 rows x 8 KiB
  hipMalloc(&d, bytes); hipMemset(d, 0, bytes); hipMalloc(&o, 256 * 1024 * 4);
  for (int share : {32, 8, 4, 1})
    for (int ld : {512, 8192}) {
      run<0>(d, o, ld, stages, share); run<1>(d, o, ld, stages, share); run<2>(d, o, ld, stages, share); run<3>(d, o, ld, stages, share);
    }
  return 0;
}
