import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lhrs_bot_amd import _lib, kernels as hk
lib = _lib.load()
DEV = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2184
g = torch.Generator().manual_seed(M)
d, ff, hd = 4096, 2048, 128
x = torch.randn(M, d, generator=g).to(DEV, torch.bfloat16)
w = (torch.randn(3 * d + 136, d, generator=g) * 0.02).to(DEV, torch.bfloat16)
wo = (torch.randn(d, d, generator=g) * 0.02).to(DEV, torch.bfloat16)
bias = torch.randn(d, generator=g).to(DEV, torch.bfloat16)
res = torch.randn(M, d, generator=g).to(DEV, torch.bfloat16)
wgu = (torch.randn(2 * ff, d, generator=g) * 0.02).to(DEV, torch.bfloat16)
wdT = (torch.randn(ff, d, generator=g) * 0.02).to(DEV, torch.bfloat16)
dy = (torch.randn(M, d, generator=g) * 0.1).to(DEV, torch.bfloat16)
inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
fr = torch.outer(torch.arange(512).float(), inv)
cos, sin = fr.cos().to(DEV).contiguous(), fr.sin().to(DEV).contiguous()
def rope(ww):
    return hk.gemm_rope_fwd(x, ww, cos, sin, pos_mod=273, pos0=0, rope_cols=2 * d, head_dim=hd)
lib.lhrs_gemm_set_min_tiles(1); lib.lhrs_gemm_set_tail_split(0)
wv = w[:3 * d]
wc = wv.contiguous().clone()
for name, ww in (("view of w[12424]", wv), ("dense copy", wc)):
    lib.lhrs_gemm_set_bm144(0)
    ref = rope(ww)
    unf = hk.gemm_nt(x, ww); plain = unf.clone(); hk.rope_(unf, M, 2 * d // hd, hd, cos, sin, pos_mod=273, pos0=0)
    lib.lhrs_gemm_set_bm144(2)
    got = rope(ww)
    for nm, a in (("256 fused vs unfused", ref), ("144 fused vs unfused", got)):
        bad = a != unf
        print(name, nm, "mismatch", int(bad.sum()))
        if bad.any():
            r, c = bad.nonzero(as_tuple=True)
            print("  rows%144", sorted(set((r % 144).tolist()))[:40], "rows//144", sorted(set((r // 144).tolist()))[:20])
            print("  cols//256", sorted(set((c // 256).tolist()))[:60], "cols%256", sorted(set((c % 256).tolist()))[:20], "...")
