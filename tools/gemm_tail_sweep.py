"""Kernel choice for the tail rows of a row-split product (M = 8736 = 32 x 273: 8192 rows on 256-row tiles + 544 rows): us per launch by kernel.
   python tools/gemm_tail_sweep.py [M_tail]"""
import os as _os
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import _lib, kernels as hk

lib = _lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 544


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def rnd(*shape, s=1.0):
    return ((torch.rand(*shape, device="cuda") * 2 - 1) * s).to(torch.bfloat16)


for (N, K) in ((4096, 4096), (12288, 4096), (4096, 11008), (4096, 22016), (4096, 12288), (22016, 4096), (11008, 4096)):
    x, w = rnd(M, K), rnd(N, K, s=.03)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fn = lambda: hk.gemm_nt(x, w, out=out)
    row = []
    for name, (pol, mint, bm, st) in {"default": (2, 128, 1, 1), "small": (0, 128, 0, 0), "256s": (2, 1, 0, 0), "144s": (2, 1, 2, 0)}.items():
        lib.lhrs_gemm_set_policy(pol); lib.lhrs_gemm_set_min_tiles(mint); lib.lhrs_gemm_set_bm144(bm); lib.lhrs_gemm_set_tail_split(st)
        t = timeit(fn)
        row.append(f"{name} {t:7.1f} us {2.0 * M * N * K / t / 1e6:6.0f} TF")
    lib.lhrs_gemm_set_policy(2); lib.lhrs_gemm_set_min_tiles(128); lib.lhrs_gemm_set_bm144(1); lib.lhrs_gemm_set_tail_split(1)
    sp = hk.gemm_nt_splitk_f32 if hasattr(hk, "gemm_nt_splitk_f32") else None
    print(f"M={M:5d} N={N:5d} K={K:5d}: " + " | ".join(row))
