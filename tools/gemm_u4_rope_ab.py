"""q|k|v projection with RoPE in the epilogue: 16-wave kernel vs gemm_u4_kernel<ROPE> - bit identity and us per launch (M = 8190 and 2184, N = 12288, K = 4096)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import _lib, kernels as hk
lib = _lib.load()
d, hd = 4096, 128
inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
fr = torch.outer(torch.arange(512).float(), inv)
cos, sin = fr.cos().cuda().contiguous(), fr.sin().cuda().contiguous()
for M, S, pos0 in ((8190, 273, 0), (2184, 273, 3), (4095, 195, 7)):
    x = [torch.randn(M, d, device="cuda").bfloat16() for _ in range(3)]
    w = [(torch.randn(3 * d, d, device="cuda") * 0.02).bfloat16() for _ in range(3)]
    res = {}
    for on in (0, 1):
        lib.lhrs_gemm_set_u4_rope(on)
        out = hk.gemm_rope_fwd(x[0], w[0], cos, sin, pos_mod=S, pos0=pos0, rope_cols=2 * d, head_dim=hd)
        for i in range(3):
            hk.gemm_rope_fwd(x[i], w[i], cos, sin, pos_mod=S, pos0=pos0, rope_cols=2 * d, head_dim=hd)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(12):
            hk.gemm_rope_fwd(x[i % 3], w[i % 3], cos, sin, pos_mod=S, pos0=pos0, rope_cols=2 * d, head_dim=hd)
        e1.record(); torch.cuda.synchronize()
        res[on] = (out, 1e3 * e0.elapsed_time(e1) / 12)
    print(f"M={M}: 16-wave {res[0][1]:.1f} us, 4-wave {res[1][1]:.1f} us, bit-identical {torch.equal(res[0][0], res[1][0])}, "
          f"max abs diff {(res[0][0].float() - res[1][0].float()).abs().max().item()}", flush=True)
lib.lhrs_gemm_set_u4_rope(0)
