"""Weight-stream bandwidth of the decode GEMV (batch 1) on the LLaMA-7B shapes.  Launches are replayed from a captured hipGraph
(as generate() does), so the figures do not include host launch overhead.
    python tools/gemv_bench.py [n_copies] [sweep]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import _lib, kernels as hk

lib = _lib.load()
shapes = [("qkv", 12288, 4096, hk.PRO_RMSNORM), ("o", 4096, 4096, hk.PRO_NONE), ("gate|up", 22016, 4096, hk.PRO_RMSNORM),
          ("down", 4096, 11008, hk.PRO_SWIGLU), ("lm_head", 32000, 4096, hk.PRO_RMSNORM)]
NL = int(sys.argv[1]) if len(sys.argv) > 1 else 8  # distinct weight copies per shape so that nothing is served from the 256 MB infinity cache (1: everything is)
cfgs = [(0, 0)] + ([(1, 1), (1, 2), (1, 4), (1, 8), (2, 2), (2, 4), (4, 1), (4, 2)] if len(sys.argv) > 2 else [])
side = torch.cuda.Stream()
for name, N, K, pro in shapes:
    Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(NL)]
    x = torch.randn(1, 2 * K if pro == hk.PRO_SWIGLU else K, device="cuda").to(torch.bfloat16)
    nw = torch.ones(K, device="cuda", dtype=torch.bfloat16)
    y = torch.empty(1, N, device="cuda", dtype=torch.bfloat16)
    line = f"{name:8s} N={N:6d} K={K:6d}:"
    ref = None
    for cfg in cfgs:
        lib.lhrs_gemv_set_tuning(*cfg)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for W in Ws: hk.gemv_fused(W, x, y, K, prologue=pro, norm_w=nw)
            g = hk.HipGraph(); g.begin()
            for W in Ws: hk.gemv_fused(W, x, y, K, prologue=pro, norm_w=nw)
            g.end()
            g.launch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): g.launch()
            e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (20 * NL)
        if ref is None: ref = y.clone()
        ok = torch.equal(ref, y)
        line += f"  {cfg[0]}x{cfg[1]}: {us:5.1f} us {N*K*2/us/1e6:4.2f} TB/s{'' if ok else ' (!=)'}"
    print(line)
lib.lhrs_gemv_set_tuning(0, 0)
