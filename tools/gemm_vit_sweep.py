"""Tile-policy sweep for the ViT-sized products (M = B * 257 rows, d = 1024): us per launch for the small-tile kernel, the 144-row and the
256-row persistent kernels.   python tools/gemm_vit_sweep.py [micro-batch]"""
import os as _os
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import _lib, kernels as hk

lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 30


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


for M in (B * 257, B * 144, B * 912, B * 128):
    for (N, K, act) in ((3072, 1024, 0), (1024, 1024, 0), (4096, 1024, 1), (1024, 4096, 0), (4096, 1024, 0)):
        x, w, bias, res = rnd(M, K), rnd(N, K, s=.03), rnd(N), rnd(M, N)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        fn = lambda: hk.gemm_nt(x, w, out=out, bias=bias, act=act, residual=res if act == 0 else None)
        row = []
        for name, (pol, mint, bm) in {"default": (2, 128, 1), "small": (0, 128, 0), "256s": (2, 1, 0), "144s": (2, 1, 2)}.items():
            lib.lhrs_gemm_set_policy(pol); lib.lhrs_gemm_set_min_tiles(mint); lib.lhrs_gemm_set_bm144(bm); lib.lhrs_gemm_set_tail_split(1 if name == "default" else 0)
            t = timeit(fn)
            row.append(f"{name} {t:7.1f} us {2.0 * M * N * K / t / 1e6:6.0f} TF")
        lib.lhrs_gemm_set_policy(2); lib.lhrs_gemm_set_min_tiles(128); lib.lhrs_gemm_set_bm144(1); lib.lhrs_gemm_set_tail_split(1)
        print(f"M={M:6d} N={N:5d} K={K:5d} act={act}: " + " | ".join(row))
