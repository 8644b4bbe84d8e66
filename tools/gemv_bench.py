"""Weight-stream bandwidth of the decode GEMV (batch 1) on the LLaMA-7B shapes.  Launches are replayed from a captured hipGraph
(as generate() does), so the figures do not include host launch overhead.
    python tools/gemv_bench.py [n_copies] [sweep|fp8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import _lib, kernels as hk

lib = _lib.load()
shapes = [("qkv", 12288, 4096, hk.PRO_RMSNORM), ("o", 4096, 4096, hk.PRO_NONE), ("gate|up", 22016, 4096, hk.PRO_RMSNORM),
          ("down", 4096, 11008, hk.PRO_SWIGLU), ("lm_head", 32000, 4096, hk.PRO_RMSNORM)]
NL = int(sys.argv[1]) if len(sys.argv) > 1 else 8  # distinct weight copies per shape so that nothing is served from the 256 MB infinity cache (1: everything is)
FP8 = len(sys.argv) > 2 and sys.argv[2] == "fp8"  # e4m3 weights + in-kernel activation quantisation on the block-scaled MFMA
cfgs = [(0, 0)] + ([(1, 1), (1, 2), (1, 4), (1, 8), (2, 2), (2, 4), (4, 1), (4, 2)] if len(sys.argv) > 2 and not FP8 else [])
side = torch.cuda.Stream()
for name, N, K, pro in shapes:
    if os.environ.get("GEMV_NOPRO") == "1" and pro == hk.PRO_RMSNORM: pro = hk.PRO_NONE  # cost of the in-kernel prologue
    Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(NL)]
    x = torch.randn(1, 2 * K if pro == hk.PRO_SWIGLU else K, device="cuda").to(torch.bfloat16)
    nw = torch.ones(K, device="cuda", dtype=torch.bfloat16)
    y = torch.empty(1, N, device="cuda", dtype=torch.bfloat16)
    if FP8:
        Ws = [hk.quant_fp8_rows(W) for W in Ws]
        PAD = int(os.environ.get("GEMV_PAD", "0"))  # row stride K + PAD bytes (channel-interleave experiments)
        if PAD:
            Ws = [(torch.cat([w8, torch.zeros(N, PAD, device="cuda", dtype=torch.uint8)], 1)[:, :K], sc) for w8, sc in Ws]
        if os.environ.get("GEMV_PACKED", "1") == "1":
            Ws = [(hk.repack_fp8_mfma(w8), sc) for w8, sc in Ws]
        run = lambda W: hk.gemv_fp8_mfma_fused(W[0], W[1], x, y, K, prologue=pro, norm_w=nw)
    else:
        run = lambda W: hk.gemv_fused(W, x, y, K, prologue=pro, norm_w=nw)
    line = f"{name:8s} N={N:6d} K={K:6d}:"
    ref = None
    for cfg in cfgs:
        lib.lhrs_gemv_set_tuning(*cfg)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for W in Ws: run(W)
            g = hk.HipGraph(); g.begin()
            for W in Ws: run(W)
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
        line += f"  {cfg[0]}x{cfg[1]}: {us:5.1f} us {N*K*(1 if FP8 else 2)/us/1e6:4.2f} TB/s{'' if ok else ' (!=)'}"
    print(line)
lib.lhrs_gemv_set_tuning(0, 0)
