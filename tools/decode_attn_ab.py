"""Decode attention: one workgroup per head against the split-context kernel, us per launch at the contexts the bench walks.
   python tools/decode_attn_ab.py"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import kernels as hk
from oracle.lhrs_oracle import rope_tables

dev = "cuda"
B, H, D, max_ctx = 1, 32, 128, 715
d = H * D
cos, sin = (t.to(dev) for t in rope_tables(max_ctx, D))
caches = [(torch.randn(B * max_ctx, d, device=dev).bfloat16(), torch.randn(B * max_ctx, d, device=dev).bfloat16()) for _ in range(32)]   # 32 layers: no L2 reuse between launches
qkv = torch.randn(B, 3 * d, device=dev).bfloat16()
o = torch.empty(B, d, device=dev, dtype=torch.bfloat16)


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / (n * 32)


for ctx in (203, 330, 459, 587, 714):
    pos = torch.full((B,), ctx, dtype=torch.int32, device=dev)
    row = [f"ctx {ctx:4d}: one WG/head {timeit(lambda: [hk.decode_attn(qkv, kc, vc, cos, sin, pos, o, B, H, D, max_ctx, 1 / math.sqrt(D)) for kc, vc in caches]):6.2f} us"]
    for ns in (3, 6, 8, 12):
        part = torch.zeros(B, H, ns, 132, device=dev)
        tk = torch.zeros(B, H, device=dev, dtype=torch.int32)
        t = timeit(lambda: [hk.decode_attn_split(qkv, kc, vc, cos, sin, pos, o, B, H, D, max_ctx, 1 / math.sqrt(D), ns, part, tk) for kc, vc in caches])
        row.append(f"split {ns:2d}: {t:6.2f}")
    print(" | ".join(row))
