"""us per launch of the N = 4096 products at micro-batch 32 (M = 8736: 8192 rows on the four-wave kernel + 544 tail rows split-K); LHRS_HIP_LIB selects the library"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import kernels as hk

dev, M, N = "cuda", 8736, 4096
hk.ensure_gemm_workspace(dev)
g = torch.Generator().manual_seed(1)
mk = lambda r, c, s=1.0: (torch.randn(r, c, generator=g) * s).to(dev, torch.bfloat16)
print("library:", os.environ.get("LHRS_HIP_LIB", "(in-tree)"))
for K, res in ((4096, True), (11008, True), (12288, False), (22016, False)):
    a, b = mk(M, K), mk(N, K, 0.02)
    r = mk(M, N) if res else None
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fn = lambda: hk.gemm_nt(a, b, out=out, residual=r)
    best = 1e9
    for _ in range(3):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    ref = a.float() @ b.float().t() + (r.float() if res else 0)
    err = float((out.float() - ref).norm() / ref.norm())
    print(f"  M=8736 N=4096 K={K:5d}{' + residual' if res else '           '}: {best:8.1f} us  {2.0 * M * N * K / best / 1e6:7.1f} TFLOP/s  rel err vs fp32 {err:.2e}", flush=True)
