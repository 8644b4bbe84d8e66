// Cost of a device-wide barrier between co-resident workgroups (one per CU) on gfx950, in a few flavours.
// hipcc --offload-arch=gfx950 -O3 tools/probe_gridbar.hip -o tools/probe_gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__device__ __forceinline__ bool grid_barrier(unsigned* cnt, unsigned* flag, unsigned gen, unsigned nblk, unsigned* err) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    long spins = 0;
    if (MODE == 0) {  // release add + acquire poll on the counter
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gen * nblk) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 20000000) { *err = 1; ok = false; break; }
      }
    } else if (MODE == 1) {  // relaxed everywhere (no cache maintenance): the pure counter cost
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen * nblk) {
        if (++spins > 20000000) { *err = 1; ok = false; break; }
      }
    } else if (MODE == 2) {  // last arriver publishes a generation flag; the others poll the flag (loads only); one fence each side
      __atomic_thread_fence(__ATOMIC_RELEASE);  // hip: agent scope by default for thread_fence? use builtin below
      const unsigned prev = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (prev == gen * nblk - 1) __hip_atomic_store(flag, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
          if (++spins > 20000000) { *err = 1; ok = false; break; }
        }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    } else {  // MODE 3: like 2 without any fence
      const unsigned prev = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (prev == gen * nblk - 1) __hip_atomic_store(flag, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
          if (++spins > 20000000) { *err = 1; ok = false; break; }
        }
    }
  }
  __syncthreads();
  return ok;
}

template <int MODE>
__global__ __launch_bounds__(1024) void bar_kernel(unsigned* cnt, unsigned* flag, unsigned* err, int nbar, float* sink, const float* src) {
  float acc = 0.f;
  for (int i = 0; i < nbar; ++i) {
    acc += src[(blockIdx.x * 1024 + threadIdx.x + i * 7) & 0xFFFF];
    if (!grid_barrier<MODE>(cnt, flag, (unsigned)(i + 1), gridDim.x, err)) break;
  }
  if (acc == 12345.f) sink[0] = acc;
}

template <int MODE>
void run(unsigned* cnt, unsigned* flag, unsigned* err, float* sink, float* src, int threads) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float t[2];
  for (int grid : {256, 64}) {
    int k = 0;
    for (int nbar : {161, 321}) {
      hipMemset(cnt, 0, 4); hipMemset(flag, 0, 4); hipMemset(err, 0, 4);
      void* args[] = {&cnt, &flag, &err, &nbar, &sink, &src};
      hipEventRecord(e0);
      hipError_t rc = hipLaunchCooperativeKernel((const void*)bar_kernel<MODE>, dim3(grid), dim3(threads), args, 0, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&t[k++], e0, e1);
      unsigned herr; (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
      if (rc != hipSuccess || herr) printf("  mode %d grid %d: rc=%d err=%u\n", MODE, grid, (int)rc, herr);
    }
    printf("mode %d threads %d grid %d: %.2f us per barrier\n", MODE, threads, grid, (t[1] - t[0]) * 1e3 / 160);
  }
}

int main() {
  unsigned *cnt, *flag, *err; float *sink, *src;
  (void)hipMalloc(&cnt, 256); (void)hipMalloc(&flag, 256); (void)hipMalloc(&err, 4); (void)hipMalloc(&sink, 4); (void)hipMalloc(&src, 65536 * 4);
  (void)hipMemset(src, 0, 65536 * 4);
  for (int threads : {1024, 512}) {
    run<0>(cnt, flag, err, sink, src, threads);
    run<1>(cnt, flag, err, sink, src, threads);
    run<2>(cnt, flag, err, sink, src, threads);
    run<3>(cnt, flag, err, sink, src, threads);
  }
  return 0;
}
