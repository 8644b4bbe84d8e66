"""What the in-kernel prologue of the decode GEMV costs: down projection with the SwiGLU prologue (x = gate|up [1, 2K]) against the same weight stream on a
ready activation (x = act [1, K]); qkv with / without the RMSNorm prologue; bf16 rows and e4m3 MFMA tiles.  Graph-replayed, 8 weight copies.
   python tools/gemv_pro_cost.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import _lib, kernels as hk

lib = _lib.load()
NL = 8
side = torch.cuda.Stream()


def bench(run, Ws):
    with torch.cuda.stream(side):
        for W in Ws:
            run(W)
        g = hk.HipGraph(); g.begin()
        for W in Ws:
            run(W)
        g.end()
        g.launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.launch()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (20 * NL)


for name, N, K in (("down", 4096, 11008), ("qkv", 12288, 4096), ("gate|up", 22016, 4096)):
    Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(NL)]
    W8 = [hk.quant_fp8_rows(W) for W in Ws]
    W8 = [(hk.repack_fp8_mfma(w8), sc) for w8, sc in W8]
    y = torch.empty(1, N, device="cuda", dtype=torch.bfloat16)
    nw = torch.ones(K, device="cuda", dtype=torch.bfloat16)
    pro = hk.PRO_SWIGLU if name == "down" else hk.PRO_RMSNORM
    x_pro = torch.randn(1, 2 * K if pro == hk.PRO_SWIGLU else K, device="cuda").to(torch.bfloat16)
    x_plain = torch.randn(1, K, device="cuda").to(torch.bfloat16)
    t = {}
    t["bf16 pro"] = bench(lambda W: hk.gemv_fused(W, x_pro, y, K, prologue=pro, norm_w=nw), Ws)
    t["bf16 none"] = bench(lambda W: hk.gemv_fused(W, x_plain, y, K, prologue=hk.PRO_NONE), Ws)
    t["fp8 pro"] = bench(lambda W: hk.gemv_fp8_mfma_fused(W[0], W[1], x_pro, y, K, prologue=pro, norm_w=nw), W8)
    t["fp8 none"] = bench(lambda W: hk.gemv_fp8_mfma_fused(W[0], W[1], x_plain, y, K, prologue=hk.PRO_NONE), W8)
    print(f"{name:8s} N={N:6d} K={K:6d}: " + "  ".join(f"{k} {v:5.1f} us" for k, v in t.items()))
