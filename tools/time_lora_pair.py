"""us per launch of the products that carry a LoRA pair in their k-loop (stage 3: r = 8 on q, k, v, o -> K2 = 64), four-wave gemm_u4_kernel against the 16-wave
kernels on the same box: python tools/time_lora_pair.py [M]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import kernels as hk

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8190
dev = "cuda"
g = torch.Generator().manual_seed(1)
mk = lambda r, c, s=1.0: (torch.randn(r, c, generator=g) * s).to(dev, torch.bfloat16)
inv = 1.0 / (10000.0 ** (torch.arange(0, 128, 2).float() / 128))
fr = torch.outer(torch.arange(512).float(), inv)
cos, sin = fr.cos().to(dev).contiguous(), fr.sin().to(dev).contiguous()


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, N, K, res, rope in (("o fwd (residual)", 4096, 4096, True, False), ("o dX", 4096, 4096, False, False), ("q|k|v dX", 4096, 12288, False, False),
                              ("q|k|v + RoPE", 12288, 4096, False, True)):
    a, b, a2, b2 = mk(M, K), mk(N, K, 0.02), mk(M, 64, 0.1), mk(N, 64, 0.05)
    r = mk(M, N) if res else None
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    if rope:
        fn = lambda: hk.gemm_rope_fwd(a, b, cos, sin, pos_mod=273, pos0=0, rope_cols=8192, head_dim=128, a2=a2, b2=b2, out=out)
    else:
        fn = lambda: hk.gemm_nt_lora(a, b, a2, b2, residual=r, out=out)
    t = {}
    for on in (False, True, False, True):
        hk.gemm_set_u4(on)
        t.setdefault(on, []).append(timed(fn))
    hk.gemm_set_u4(True)
    fl = 2.0 * M * N * (K + 64)
    print(f"M={M} {name:18s} N={N} K={K}+64: 16-wave {min(t[False]):8.1f} us ({fl / min(t[False]) / 1e6:6.1f} TFLOP/s)   four-wave {min(t[True]):8.1f} us ({fl / min(t[True]) / 1e6:6.1f} TFLOP/s)")
