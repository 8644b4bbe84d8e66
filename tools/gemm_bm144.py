"""A/B of the two persistent tile heights (lhrs_gemm_set_bm144: 0 = 256 rows, 2 = 144 rows) on the eight GEMMs of one LLaMA decoder layer
(forward + activation-gradient backward) at M = B * 273 token rows:  python tools/gemm_bm144.py [B=8] [iters=20]"""
import os as _os
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lhrs_bot_amd import _lib
from lhrs_bot_amd import kernels as hk

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
M, d, ff, hd = B * 273, 4096, 11008, 128
lib = _lib.load()
dev = "cuda"
g = torch.Generator().manual_seed(0)
rn = lambda *s, std=1.0: (torch.randn(*s, generator=g) * std).to(dev, torch.bfloat16)  # noqa: E731
x, dy = rn(M, d), rn(M, d, std=0.1)
xf, dgu_in = rn(M, ff), rn(M, 2 * ff, std=0.1)
wqkv, wo, wgu, wdown = rn(3 * d, d, std=0.02), rn(d, d, std=0.02), rn(2 * ff, d, std=0.02), rn(d, ff, std=0.02)
wdownT, wguT, wqkvT = rn(ff, d, std=0.02), rn(d, 2 * ff, std=0.02), rn(d, 3 * d, std=0.02)
res = rn(M, d)
inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
fr = torch.outer(torch.arange(512).float(), inv)
cos, sin = fr.cos().to(dev).contiguous(), fr.sin().to(dev).contiguous()
gu_saved, _ = hk.gemm_swiglu_fwd(x, wgu, ff)
qkv3 = rn(M, 3 * d, std=0.1)
cases = [
    ("qkv+rope   N=12288 K= 4096", 2.0 * M * 3 * d * d, lambda: hk.gemm_rope_fwd(x, wqkv, cos, sin, pos_mod=273, pos0=0, rope_cols=2 * d, head_dim=hd)),
    ("o +res     N= 4096 K= 4096", 2.0 * M * d * d, lambda: hk.gemm_nt(x, wo, residual=res)),
    ("gu+swiglu  N=22016 K= 4096", 2.0 * M * 2 * ff * d, lambda: hk.gemm_swiglu_fwd(x, wgu, ff)),
    ("down+res   N= 4096 K=11008", 2.0 * M * d * ff, lambda: hk.gemm_nt(xf, wdown, residual=res)),
    ("d-down epi2 N=11008 K= 4096", 2.0 * M * ff * d, lambda: hk.gemm_swiglu_bwd(dy, wdownT, gu_saved.clone(), ff)),
    ("d-gu       N= 4096 K=22016", 2.0 * M * d * 2 * ff, lambda: hk.gemm_nt(dgu_in, wguT)),
    ("d-o        N= 4096 K= 4096", 2.0 * M * d * d, lambda: hk.gemm_nt(dy, wo)),
    ("d-qkv      N= 4096 K=12288", 2.0 * M * d * 3 * d, lambda: hk.gemm_nt(qkv3, wqkvT)),
]


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


tot = {0: 0.0, 2: 0.0, 1: 0.0}
print(f"M = {M} (micro-batch {B}); us per launch, TFLOP/s")
for name, fl, fn in cases:
    row = []
    for mode in (0, 2, 1):
        lib.lhrs_gemm_set_bm144(mode)
        us = timeit(fn)
        if name.startswith("d-down"):
            us -= 0.0  # includes the clone of gu (same in both modes)
        tot[mode] += us
        row.append(f"{us:7.1f} us {fl / us / 1e6:6.0f} TF")
    print(f"{name:28s}  256-row {row[0]}   144-row {row[1]}   policy {row[2]}")
lib.lhrs_gemm_set_bm144(1)
print(f"layer total: 256-row {tot[0]:.0f} us, 144-row {tot[2]:.0f} us, policy {tot[1]:.0f} us")
