import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lhrs_bot_amd import _lib, kernels as hk
lib = _lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2184
g = torch.Generator().manual_seed(M)
d, hd = 4096, 128
x = torch.randn(M, d, generator=g).to("cuda", torch.bfloat16)
w = (torch.randn(3 * d, d, generator=g) * 0.02).to("cuda", torch.bfloat16)
inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
fr = torch.outer(torch.arange(512).float(), inv)
cos, sin = fr.cos().to("cuda").contiguous(), fr.sin().to("cuda").contiguous()
lib.lhrs_gemm_set_min_tiles(1); lib.lhrs_gemm_set_tail_split(0)
lib.lhrs_gemm_set_bm144(0)
ref = hk.gemm_rope_fwd(x, w, cos, sin, pos_mod=273, pos0=0, rope_cols=2 * d, head_dim=hd)
lib.lhrs_gemm_set_bm144(2)
for it in range(3):
    got = hk.gemm_rope_fwd(x, w, cos, sin, pos_mod=273, pos0=0, rope_cols=2 * d, head_dim=hd)
    bad = (got != ref)
    print("iter", it, "mismatch", int(bad.sum()), "of", bad.numel())
    if bad.any():
        r, c = bad.nonzero(as_tuple=True)
        print(" rows%144:", sorted(set((r % 144).tolist()))[:60])
        print(" rows//144:", sorted(set((r // 144).tolist()))[:20])
        print(" cols%256:", sorted(set((c % 256).tolist()))[:80])
        print(" cols//256:", sorted(set((c // 256).tolist()))[:60])
        print(" first", r[0].item(), c[0].item(), got[r[0], c[0]].item(), ref[r[0], c[0]].item())
