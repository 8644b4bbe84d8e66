"""Tail-row rule of gemm_launch: the same product with and without the cut, on micro-batch-32 shapes (M = 32 * 273 = 8736)."""
import os as _os
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import _lib, kernels as hk
lib = _lib.load()
for M, N, K in [(8736, 4096, 4096), (8736, 4096, 11008), (8736, 4096, 12288), (8736, 4096, 22016), (8736, 12288, 4096), (4368, 4096, 4096), (10000, 4096, 4096),
                (2184, 4096, 4096), (2184, 4096, 11008), (2184, 4096, 22016), (7710, 1024, 1024), (7710, 1024, 4096), (7710, 3072, 1024), (4320, 1024, 4096), (4320, 4096, 1024)]:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    line = f"M={M} N={N} K={K}:"
    for on in (0, 1):
        lib.lhrs_gemm_set_tail_split(on)
        for _ in range(3): hk.gemm_nt(a, b, out=c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): hk.gemm_nt(a, b, out=c)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        line += f"  split={on}: {us:7.1f} us {2*M*N*K/us/1e6:6.0f} TF/s"
    print(line)
lib.lhrs_gemm_set_tail_split(1)

# the e4m3 kernel and its small-tile sibling
for M, N, K in [(8736, 4096, 4096), (8736, 4096, 11008), (8736, 4096, 12288), (8736, 12288, 4096), (4368, 4096, 4096)]:
    a8, sa = hk.quant_fp8_rows((torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16))
    b8, sb = hk.quant_fp8_rows((torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16))
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    line = f"e4m3 M={M} N={N} K={K}:"
    for on in (0, 1):
        lib.lhrs_gemm_set_tail_split(on)
        for _ in range(3): hk.gemm_fp8_nt(a8, sa, b8, sb, out=c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): hk.gemm_fp8_nt(a8, sa, b8, sb, out=c)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        line += f"  split={on}: {us:7.1f} us {2*M*N*K/us/1e6:6.0f} TF/s"
    print(line)
lib.lhrs_gemm_set_tail_split(1)
