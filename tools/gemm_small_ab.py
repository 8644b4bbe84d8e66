"""Small-tile GEMM policy A/B (ViT / projector shapes at micro-batch B): python tools/gemm_small_ab.py [B=8]"""
import os as _os
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import _lib, kernels as hk
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
g = torch.Generator().manual_seed(0)
rn = lambda *s: (torch.randn(*s, generator=g) * 0.05).to("cuda", torch.bfloat16)
shapes = [(B * 257, 1024, 4096, True), (B * 257, 3072, 1024, False), (B * 257, 1024, 1024, True), (B * 257, 4096, 1024, False),
          (B * 144, 1024, 4096, True), (B * 144, 1024, 1024, False), (B * 144, 4096, 1024, False), (B * 912, 2048, 1024, False), (B * 912, 1024, 2048, False),
          (B * 128, 4096, 32000, False), (B * 128, 4096, 22016, False), (B * 128, 4096, 11008, True), (B * 128, 4096, 4096, True)]
def t(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it
configs = [("default", 256, 128, 1), ("64x64", 1 << 30, 128, 1), ("64x128", 1, 128, 1), ("min256=32 cost", 256, 32, 1), ("min256=32 144-row", 256, 32, 2),
           ("min256=32 256-row", 256, 32, 0)]
print(f"B={B}  us per launch: " + " | ".join(c[0] for c in configs))
for M, N, K, res in shapes:
    a, b = rn(M, K), rn(N, K)
    bias, r = rn(N), (rn(M, N) if res else None)
    row = []
    for _, st, m256, bm in configs:
        lib.lhrs_gemm_set_small_thresh(st); lib.lhrs_gemm_set_min_tiles(m256); lib.lhrs_gemm_set_bm144(bm)
        row.append(t(lambda: hk.gemm_nt(a, b, bias=bias, residual=r)))
    print(f"M={M:5d} N={N:5d} K={K:5d}  " + "  ".join(f"{x:7.1f}" for x in row) + f"   best [{configs[row.index(min(row))][0]}] {min(row):6.1f} us = {2*M*N*K/min(row)/1e6:5.0f} TF")
lib.lhrs_gemm_set_small_thresh(256); lib.lhrs_gemm_set_min_tiles(128); lib.lhrs_gemm_set_bm144(1)
