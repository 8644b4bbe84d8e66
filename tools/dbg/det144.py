"""Determinism / correctness probe of the 144-row ring kernel: plain and rope GEMMs against the 256-row kernel, repeated."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lhrs_bot_amd import _lib, kernels as hk
lib = _lib.load()
DEV = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2184
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
g = torch.Generator().manual_seed(M)
d, hd = 4096, 128
x = torch.randn(M, d, generator=g).to(DEV, torch.bfloat16)
wbig = (torch.randn(3 * d + 136, d, generator=g) * 0.02).to(DEV, torch.bfloat16)
w = wbig[:3 * d]
inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
fr = torch.outer(torch.arange(512).float(), inv)
cos, sin = fr.cos().to(DEV).contiguous(), fr.sin().to(DEV).contiguous()
lib.lhrs_gemm_set_min_tiles(1); lib.lhrs_gemm_set_tail_split(0)
cases = {"rope N=12288": lambda: hk.gemm_rope_fwd(x, w, cos, sin, pos_mod=273, pos0=0, rope_cols=2 * d, head_dim=hd),
         "plain N=12424": lambda: hk.gemm_nt(x, wbig), "plain N=12288": lambda: hk.gemm_nt(x, w)}
for name, fn in cases.items():
    lib.lhrs_gemm_set_bm144(0)
    ref = fn()
    lib.lhrs_gemm_set_bm144(2)
    nbad = []
    for it in range(reps):
        got = fn()
        bad = got != ref
        n = int(bad.sum())
        nbad.append(n)
        if n and len([b for b in nbad if b]) == 1:
            r, c = bad.nonzero(as_tuple=True)
            print("  first failure: tile rows", sorted(set((r // 144).tolist())), "tile cols", sorted(set((c // 256).tolist())), "rows%144 count", len(set((r % 144).tolist())))
    print(name, "mismatching elements per repetition:", nbad)
