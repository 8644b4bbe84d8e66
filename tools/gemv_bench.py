"""Weight-stream bandwidth of the decode GEMV (batch 1) on the LLaMA-7B shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import _lib, kernels as hk

lib = _lib.load()
shapes = [("qkv", 12288, 4096, hk.PRO_RMSNORM), ("o", 4096, 4096, hk.PRO_NONE), ("gate|up", 22016, 4096, hk.PRO_RMSNORM),
          ("down", 4096, 11008, hk.PRO_SWIGLU), ("lm_head", 32000, 4096, hk.PRO_RMSNORM)]
cfgs = [0]
NL = 8  # distinct weight copies per shape so that nothing is served from the 256 MB infinity cache
for name, N, K, pro in shapes:
    Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(NL)]
    x = torch.randn(1, 2 * K if pro == hk.PRO_SWIGLU else K, device="cuda").to(torch.bfloat16)
    nw = torch.ones(K, device="cuda", dtype=torch.bfloat16)
    y = torch.empty(1, N, device="cuda", dtype=torch.bfloat16)
    line = f"{name:8s} N={N:6d} K={K:6d}:"
    ref = None
    for cfg in cfgs:
        for W in Ws: hk.gemv_fused(W, x, y, K, prologue=pro, norm_w=nw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            for W in Ws: hk.gemv_fused(W, x, y, K, prologue=pro, norm_w=nw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (10 * NL)
        if ref is None: ref = y.clone()
        ok = torch.equal(ref, y)
        line += f"  cfg{cfg}: {us:6.1f} us {N*K*2/us/1e6:5.2f} TB/s{'' if ok else ' (!=)'}"
    print(line)
