// Plain bf16 NT products through the vendor library (hipBLASLt), host side only.
//
// Which launches: out[M,N] (bf16) = a[M,K] . b[N,K]^T (+ residual) with no bias, no activation, alpha = 1 and a long k-loop - the decoder's
// down / o projections, the dX products of the backward and the lm_head (HF LlamaDecoderLayer's nn.Linear calls, reached from
// lhrs/models/text_modal.py:133-151).  Everything with a fused epilogue (SwiGLU forward / backward, RoPE, LoRA in the k-loop, bias + GELU, f32
// accumulation) stays on the hand-written kernels of gemm.hip.  Why: same-box A/B at micro-batch 30 (profiles/r04_vendor_ab.txt) - 154.3 / 156.6 -> 167.0 samples/s; the library's hand-scheduled assembly kernel for these shapes (256x256x64 tile, FOUR
// waves of 128x128, stream-K over 256 persistent workgroups) runs the long-k products 14-19 % faster than the 16-wave gemm_nt_256s_kernel.
// gemm_u4.hip is our own kernel of that wave shape (at the library's rate on the longest k-loops, a few per cent behind on K = 4096 ... 11008); which of the
// three runs a given problem is decided by timing them on its first call (gemm.hip: lhrs_gemm_bf16_nt).
//
// No link-time dependency: the entry points are looked up in the hipBLASLt that is already in the process (PyTorch loads its own copy) or, for a
// C caller, in the first libhipblaslt.so.1 the loader finds.  When none is found, or the library has no algorithm for a problem, the caller
// launches the hand-written kernel instead (lhrs_gemm_vendor_status() says which).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>

#include <cstdio>
#include <map>
#include <mutex>
#include <tuple>

extern "C" void lhrs_set_error(const char* msg);

namespace {

struct Api {
  decltype(&hipblasLtCreate) create = nullptr;
  decltype(&hipblasLtMatrixLayoutCreate) layout_create = nullptr;
  decltype(&hipblasLtMatmulDescCreate) desc_create = nullptr;
  decltype(&hipblasLtMatmulDescSetAttribute) desc_set = nullptr;
  decltype(&hipblasLtMatmulPreferenceCreate) pref_create = nullptr;
  decltype(&hipblasLtMatmulPreferenceSetAttribute) pref_set = nullptr;
  decltype(&hipblasLtMatmulPreferenceDestroy) pref_destroy = nullptr;
  decltype(&hipblasLtMatmulAlgoGetHeuristic) heuristic = nullptr;
  decltype(&hipblasLtMatmul) matmul = nullptr;
  int state = 0;  // 0 not tried, 1 resolved, -1 unavailable
  char where[160] = "";
};

Api g_api;
std::mutex g_mu;

bool resolve() {
  if (g_api.state != 0) return g_api.state > 0;
  void* h = dlopen("libhipblaslt.so.1", RTLD_NOW | RTLD_NOLOAD);  // the copy PyTorch brought, if any
  const char* how = "resident libhipblaslt.so.1";
  if (!h) { h = dlopen("libhipblaslt.so.1", RTLD_NOW | RTLD_LOCAL); how = "libhipblaslt.so.1 from the loader path"; }
  if (!h) { h = dlopen("/opt/rocm/lib/libhipblaslt.so.1", RTLD_NOW | RTLD_LOCAL); how = "/opt/rocm/lib/libhipblaslt.so.1"; }
  if (!h) { g_api.state = -1; snprintf(g_api.where, sizeof(g_api.where), "no libhipblaslt.so.1 in the process or on the loader path"); return false; }
#define SYM(field, name) g_api.field = (decltype(g_api.field))dlsym(h, name); if (!g_api.field) { g_api.state = -1; snprintf(g_api.where, sizeof(g_api.where), "%s lacks %s", how, name); return false; }
  SYM(create, "hipblasLtCreate")
  SYM(layout_create, "hipblasLtMatrixLayoutCreate")
  SYM(desc_create, "hipblasLtMatmulDescCreate")
  SYM(desc_set, "hipblasLtMatmulDescSetAttribute")
  SYM(pref_create, "hipblasLtMatmulPreferenceCreate")
  SYM(pref_set, "hipblasLtMatmulPreferenceSetAttribute")
  SYM(pref_destroy, "hipblasLtMatmulPreferenceDestroy")
  SYM(heuristic, "hipblasLtMatmulAlgoGetHeuristic")
  SYM(matmul, "hipblasLtMatmul")
#undef SYM
  g_api.state = 1;
  snprintf(g_api.where, sizeof(g_api.where), "%s", how);
  return true;
}

constexpr int kMaxAlgos = 8;
struct Plan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr, ld = nullptr;
  hipblasLtMatmulAlgo_t algo[kMaxAlgos];
  size_t ws[kMaxAlgos] = {};
  int n = 0, best = 0;   // best: index the launches use - the heuristic's first answer until lhrs_vendor_gemm_tune has timed them
};
using Key = std::tuple<int, int, int, int, int, int, int, int, long>;  // dev, M, N, K, lda, ldb, ldc, ldr (0: no residual), workspace bytes offered
std::map<Key, Plan> g_plans;
hipblasLtHandle_t g_handle[16] = {};

// row-major out[M,N] = a[M,K] . b[N,K]^T  ==  column-major D[N,M] = op_T(b as [K,N]) . (a as [K,M])
Plan build(hipblasLtHandle_t h, int M, int N, int K, int lda, int ldb, int ldc, int ldr, size_t ws_bytes) {
  Plan p;
  const bool res = ldr > 0;
  if (g_api.desc_create(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return p;
  const int32_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
  if (g_api.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)) != HIPBLAS_STATUS_SUCCESS) return p;
  if (g_api.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)) != HIPBLAS_STATUS_SUCCESS) return p;
  if (g_api.layout_create(&p.la, HIP_R_16BF, (uint64_t)K, (uint64_t)N, ldb) != HIPBLAS_STATUS_SUCCESS) return p;   // b: [K, N] column-major, transposed
  if (g_api.layout_create(&p.lb, HIP_R_16BF, (uint64_t)K, (uint64_t)M, lda) != HIPBLAS_STATUS_SUCCESS) return p;   // a: [K, M] column-major
  if (g_api.layout_create(&p.lc, HIP_R_16BF, (uint64_t)N, (uint64_t)M, res ? ldr : ldc) != HIPBLAS_STATUS_SUCCESS) return p;
  if (g_api.layout_create(&p.ld, HIP_R_16BF, (uint64_t)N, (uint64_t)M, ldc) != HIPBLAS_STATUS_SUCCESS) return p;
  hipblasLtMatmulPreference_t pref = nullptr;
  if (g_api.pref_create(&pref) != HIPBLAS_STATUS_SUCCESS) return p;
  const uint64_t wsz = ws_bytes;
  (void)g_api.pref_set(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof(wsz));
  hipblasLtMatmulHeuristicResult_t r[kMaxAlgos];
  int got = 0;
  const hipblasStatus_t st = g_api.heuristic(h, p.desc, p.la, p.lb, p.lc, p.ld, pref, kMaxAlgos, r, &got);
  (void)g_api.pref_destroy(pref);
  if (st != HIPBLAS_STATUS_SUCCESS) return p;
  for (int i = 0; i < got && i < kMaxAlgos; ++i)
    if (r[i].state == HIPBLAS_STATUS_SUCCESS && r[i].workspaceSize <= ws_bytes) { p.algo[p.n] = r[i].algo; p.ws[p.n] = r[i].workspaceSize; p.n++; }
  return p;
}

Plan* plan_for(int M, int N, int K, int lda, int ldb, int ldc, int ldr, bool has_res, void* workspace, long workspace_bytes, int* dev_out) {
  if (!resolve()) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!g_handle[dev] && g_api.create(&g_handle[dev]) != HIPBLAS_STATUS_SUCCESS) { g_handle[dev] = nullptr; return nullptr; }
  const Key key(dev, M, N, K, lda, ldb, ldc, has_res ? ldr : 0, workspace ? workspace_bytes : 0);
  auto it = g_plans.find(key);
  if (it == g_plans.end()) it = g_plans.emplace(key, build(g_handle[dev], M, N, K, lda, ldb, ldc, has_res ? ldr : 0, workspace ? (size_t)workspace_bytes : 0)).first;
  *dev_out = dev;
  return it->second.n > 0 ? &it->second : nullptr;
}

hipblasStatus_t run(int dev, const Plan& p, int i, const void* A, const void* B, void* C, const void* residual, void* workspace, hipStream_t s) {
  const float one = 1.f, zero = 0.f;
  return g_api.matmul(g_handle[dev], p.desc, &one, B, p.la, A, p.lb, residual ? &one : &zero, residual ? residual : C, p.lc, C, p.ld, &p.algo[i],
                      p.ws[i] ? workspace : nullptr, p.ws[i], s);
}

}  // namespace

// 0: launched on `stream`; 1: not taken (library or algorithm unavailable) - the caller launches its own kernel; -1: error (lhrs_last_error)
extern "C" int lhrs_vendor_gemm_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const void* residual, int ldr,
                                   void* workspace, long workspace_bytes, void* stream) {
  std::lock_guard<std::mutex> lock(g_mu);
  int dev = 0;
  Plan* p = plan_for(M, N, K, lda, ldb, ldc, ldr, residual != nullptr, workspace, workspace_bytes, &dev);
  if (p == nullptr) return 1;
  const hipblasStatus_t st = run(dev, *p, p->best, A, B, C, residual, workspace, (hipStream_t)stream);
  if (st != HIPBLAS_STATUS_SUCCESS) {
    char b[160];
    snprintf(b, sizeof(b), "vendor gemm: hipblasLtMatmul failed with status %d (M=%d N=%d K=%d)", (int)st, M, N, K);
    lhrs_set_error(b);
    return -1;
  }
  return 0;
}

// Times every algorithm the heuristic offered for this problem on the caller's operands (1 untimed + `reps` timed launches each, HIP events on
// `stream`, host-synchronous) and keeps the fastest for lhrs_vendor_gemm_nt; *best_us = its time per launch.  Every timing launch writes the
// product's real result to C (A, B and the residual are only read), so C must not alias an operand or the residual (the caller checks).
// 0 tuned, 1 not available.
extern "C" int lhrs_vendor_gemm_tune(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const void* residual, int ldr,
                                     void* workspace, long workspace_bytes, int reps, float* best_us, void* stream) {
  std::lock_guard<std::mutex> lock(g_mu);
  int dev = 0;
  Plan* p = plan_for(M, N, K, lda, ldb, ldc, ldr, residual != nullptr, workspace, workspace_bytes, &dev);
  if (p == nullptr) return 1;
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1;
  float best = 1e30f;
  int best_i = -1;
  for (int i = 0; i < p->n; ++i) {
    if (run(dev, *p, i, A, B, C, residual, workspace, s) != HIPBLAS_STATUS_SUCCESS) continue;
    (void)hipEventRecord(e0, s);
    bool ok = true;
    for (int r = 0; r < reps && ok; ++r) ok = run(dev, *p, i, A, B, C, residual, workspace, s) == HIPBLAS_STATUS_SUCCESS;
    (void)hipEventRecord(e1, s);
    if (hipEventSynchronize(e1) != hipSuccess || !ok) continue;
    float ms = 0;
    if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) continue;
    if (ms / reps < best) { best = ms / reps; best_i = i; }
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (best_i < 0) { p->n = 0; return 1; }
  p->best = best_i;
  *best_us = best * 1e3f;
  return 0;
}

// "" until the first plain product was offered; then where the library came from, or why it is not used
extern "C" const char* lhrs_gemm_vendor_status() { return g_api.where; }
