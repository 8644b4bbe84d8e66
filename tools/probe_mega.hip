// Probe for a one-launch-per-token decode step on gfx950: 256 co-resident workgroups run the GEMV phases of a layer back to back,
// separated by device-wide barriers, with the first weight lines of the NEXT phase requested before each barrier.
// hipcc --offload-arch=gfx950 -O3 tools/probe_mega.hip -o tools/probe_mega
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int NWG = 256, WAVES = 16, THREADS = WAVES * 64;

struct Phase { const uint8_t* W; int N, K; const unsigned* x; unsigned* y; };  // W packed [N/16][K/128][2][64][16]; x: K/2 dwords (bf16 pairs)
struct Args { Phase ph[4]; unsigned* cnt; unsigned* err; int nlayers; int mode; };

__device__ __forceinline__ bool grid_barrier(unsigned* cnt, unsigned target, unsigned* err) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > 4000000) { *err = 1; ok = false; break; }
    }
  }
  __syncthreads();
  return ok;
}

// one item = 2 consecutive 128-k steps of one (row group, k slice): 4 x 16 B per lane
struct Item { i32x4 a[4]; };
__device__ __forceinline__ void item_coords(const Phase& p, int it, int wave, int& rg, int& s0, int& ns, int& items_per_unit) {
  const int nsteps = p.K / 128, per = (nsteps + WAVES - 1) / WAVES;   // steps per k slice
  items_per_unit = (per + 1) / 2;
  const int unit = it / items_per_unit, c = it % items_per_unit;
  rg = blockIdx.x + NWG * unit;
  const int sb = min(wave * per, nsteps), se = min(nsteps, sb + per);
  s0 = sb + 2 * c;
  ns = max(0, min(2, se - s0));
}
__device__ __forceinline__ int n_items(const Phase& p) {
  const int nrg = p.N / 16, units = (nrg - (int)blockIdx.x + NWG - 1) / NWG;
  const int nsteps = p.K / 128, per = (nsteps + WAVES - 1) / WAVES;
  return units * ((per + 1) / 2);
}
__device__ __forceinline__ void load_item(const Phase& p, int it, int wave, int lane, Item& d) {
  int rg, s0, ns, ipu;
  item_coords(p, it, wave, rg, s0, ns, ipu);
  const int nsteps = p.K / 128;
  const uint8_t* base = p.W + ((long)rg * nsteps) * 2048 + lane * 16;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const long o = (long)min(s0 + u, nsteps - 1) * 2048;
    if (u < ns || u == 0) {
      d.a[2 * u] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(base + o));
      d.a[2 * u + 1] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(base + o + 1024));
    }
  }
}

__global__ __launch_bounds__(THREADS) void mega_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // x8 [K] e4m3
  __shared__ float part[WAVES][16][17];
  __shared__ float red[WAVES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
  unsigned bar = 0;
  Item cur, nxt;
  load_item(a.ph[0], 0, wave, lane, cur);
  for (int layer = 0; layer < a.nlayers; ++layer) {
    for (int pi = 0; pi < 4; ++pi) {
      const Phase& p = a.ph[pi];
      // ---- prologue: coherent loads of the activation vector, one reduction, e4m3 bytes to LDS
      float q = 0.f;
      const int nd = p.K / 2;
      unsigned xr[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int c = tid + i * THREADS;
        xr[i] = c < nd ? __hip_atomic_load(p.x + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        const float lo = __uint_as_float(xr[i] << 16), hi = __uint_as_float(xr[i] & 0xFFFF0000u);
        q += lo * lo + hi * hi;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
      __syncthreads();
      if (lane == 0) red[wave] = q;
      __syncthreads();
      float tot = 0.f;
#pragma unroll
      for (int i = 0; i < WAVES; ++i) tot += red[i];
      const float rstd = rsqrtf(tot / p.K + 1e-5f);
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int c = tid + i * THREADS;
        if (c < nd) {
          const float lo = __uint_as_float(xr[i] << 16) * rstd, hi = __uint_as_float(xr[i] & 0xFFFF0000u) * rstd;
          const int v = __builtin_amdgcn_cvt_pk_fp8_f32(lo, hi, 0, false);
          reinterpret_cast<unsigned short*>(smem)[c] = (unsigned short)v;
        }
      }
      __syncthreads();
      // ---- stream this phase's items
      const int ni = n_items(p);
      int ipu, rg, s0, ns;
      item_coords(p, 0, wave, rg, s0, ns, ipu);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int it = 0; it < ni; ++it) {
        const bool last = it + 1 == ni;
        if (!last) load_item(p, it + 1, wave, lane, nxt);
        else if (a.mode == 1 && !(layer + 1 == a.nlayers && pi == 3)) load_item(a.ph[(pi + 1) & 3], 0, wave, lane, nxt);  // next phase, across the barrier
        item_coords(p, it, wave, rg, s0, ns, ipu);
        const uint8_t* xp = reinterpret_cast<const uint8_t*>(smem) + fg * 16;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (u < ns) {
            const long o = (long)(s0 + u) * 128;
            const i32x4 xlo = *reinterpret_cast<const i32x4*>(xp + o), xhi = *reinterpret_cast<const i32x4*>(xp + o + 64);
            const i32x8 A = {cur.a[2 * u][0], cur.a[2 * u][1], cur.a[2 * u][2], cur.a[2 * u][3], cur.a[2 * u + 1][0], cur.a[2 * u + 1][1], cur.a[2 * u + 1][2], cur.a[2 * u + 1][3]};
            i32x8 B = {xlo[0], xlo[1], xlo[2], xlo[3], xhi[0], xhi[1], xhi[2], xhi[3]};
            if (fr != 0) B = i32x8{0, 0, 0, 0, 0, 0, 0, 0};
            acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
          }
        }
        if ((it + 1) % ipu == 0) {  // unit finished: fold the k slices of the row group
          if (fr == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) part[wave][fg * 4 + r][0] = acc[r];
          }
          __syncthreads();
          if (tid < 16) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) v += part[w][tid][0];
            const float o = __shfl_xor(v, 1, 64);
            if ((tid & 1) == 0 && rg * 16 + tid < p.N)
              __hip_atomic_store(p.y + (rg * 16 + tid) / 2, (__float_as_uint(v) >> 16) | (__float_as_uint(o) & 0xFFFF0000u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          __syncthreads();
          acc = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        cur = nxt;
      }
      if (a.mode >= 1 || true) {
        ++bar;
        if (!grid_barrier(a.cnt, bar * NWG, a.err)) return;
      }
      if (a.mode == 0 && !(layer + 1 == a.nlayers && pi == 3)) load_item(a.ph[(pi + 1) & 3], 0, wave, lane, cur);  // no overlap: request after the barrier
    }
  }
}

int main() {
  const int Ns[4] = {12288, 4096, 22016, 4096}, Ks[4] = {4096, 4096, 4096, 11008};
  Args a{};
  unsigned* act; (void)hipMalloc(&act, 4 * 32768 * 4); (void)hipMemset(act, 0x3c, 4 * 32768 * 4);
  const int NCOPY = 6;  // rotate weight copies so nothing is served from the 256 MB infinity cache
  std::vector<uint8_t*> Wc[4];
  for (int i = 0; i < 4; ++i)
    for (int c = 0; c < NCOPY; ++c) { uint8_t* w; (void)hipMalloc(&w, (size_t)Ns[i] * Ks[i]); (void)hipMemset(w, 0x38, (size_t)Ns[i] * Ks[i]); Wc[i].push_back(w); }
  (void)hipMalloc(&a.cnt, 4); (void)hipMalloc(&a.err, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute((const void*)mega_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 16384);
  for (int mode : {0, 1, 0, 1}) for (int nl : {1, 3}) {
    float total = 0.f;
    for (int c = 0; c < NCOPY; ++c) {
      for (int i = 0; i < 4; ++i) a.ph[i] = Phase{Wc[i][c], Ns[i], Ks[i], act + ((i + 3) & 3) * 32768, act + i * 32768};
      a.nlayers = nl; a.mode = mode;
      (void)hipMemset(a.cnt, 0, 4); (void)hipMemset(a.err, 0, 4);
      void* args[] = {&a};
      (void)hipEventRecord(e0);
      hipError_t rc = hipLaunchCooperativeKernel((const void*)mega_kernel, dim3(NWG), dim3(THREADS), args, 16384, 0);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      unsigned herr; (void)hipMemcpy(&herr, a.err, 4, hipMemcpyDeviceToHost);
      if (rc != hipSuccess || herr) printf("mode %d: rc=%d err=%u\n", mode, (int)rc, herr);
      if (c > 0) total += ms;
    }
    printf("mode %d, %d layer(s): %.1f us per launch (4 GEMV phases, 215 MB e4m3 per layer)\n", mode, nl, total * 1e3 / (NCOPY - 1));
  }
  return 0;
}
