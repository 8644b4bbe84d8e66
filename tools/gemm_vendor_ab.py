"""Per-shape A/B of the plain bf16 NT product: hand-written kernels (gemm.hip) vs the vendor library (vendor.cpp), us per launch in a
back-to-back loop over 4 distinct operand sets (random operands: the power cap is part of the answer).
    python tools/gemm_vendor_ab.py [rows ...]        default rows: 2184 4095 8190 8736 16380"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import kernels as hk

dev = "cuda"
rows = [int(x) for x in sys.argv[1:]] or [2184, 4095, 8190, 8736, 16380]
shapes = [(4096, 4096, "o / d-o"), (4096, 11008, "down"), (4096, 12288, "d-qkv -> dx"), (4096, 22016, "d-gate|up -> dx"), (32000, 4096, "lm_head"), (4096, 32000, "d-logits -> dh")]
hk.ensure_streamk_workspace(torch.device(dev))


def bench(fn, n=12):
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for M in rows:
    for N, K, nm in shapes:
        if nm.startswith("lm_head") or nm.startswith("d-logits"):
            Mr = M * 128 // 273 // 2 * 2      # the supervised rows only
        else:
            Mr = M
        a = [torch.randn(Mr, K, device=dev).bfloat16() for _ in range(4)]
        b = [(torch.randn(N, K, device=dev) * 0.05).bfloat16() for _ in range(4)]
        out = torch.empty(Mr, N, device=dev, dtype=torch.bfloat16)
        res = {}
        for tag, on in (("hand", False), ("vendor", True)):
            hk.gemm_set_vendor(on)
            res[tag] = bench(lambda i: hk.gemm_nt(a[i % 4], b[i % 4], out=out))
        fl = 2.0 * Mr * N * K
        print(f"M={Mr:6d} N={N:6d} K={K:6d} {nm:18s} hand {res['hand']:8.1f} us ({fl / res['hand'] / 1e6:7.1f} TF)   vendor {res['vendor']:8.1f} us "
              f"({fl / res['vendor'] / 1e6:7.1f} TF)   vendor/hand time {res['vendor'] / res['hand']:.3f}", flush=True)
hk.gemm_set_vendor(True)
