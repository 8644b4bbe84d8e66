"""Does a power-of-two row stride hurt the 256x256 GEMM?  Same product with operands at stride K and K + pad elements."""
import os as _os
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import _lib, kernels as hk
lib = _lib.load()
shapes = [(8190, 12288, 4096), (8190, 4096, 4096), (8190, 22016, 4096), (8190, 4096, 11008), (8190, 4096, 12288)]
for m, n, k in shapes:
    line = f"M={m} N={n} K={k}:"
    for pad in (0, 64, 128, 192):
        A = (torch.rand(m, k + pad, device="cuda") * 2 - 1).to(torch.bfloat16)
        B = (torch.rand(n, k + pad, device="cuda") * 2 - 1).to(torch.bfloat16)
        a, b = A[:, :k], B[:, :k]
        c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        for _ in range(3): hk.gemm_nt(a, b, out=c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): hk.gemm_nt(a, b, out=c)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        line += f"  pad {pad:3d}: {us:7.1f} us {2*m*n*k/us/1e6:6.0f} TF"
    print(line)
