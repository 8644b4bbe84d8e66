"""us per launch of every gemm_u4_kernel instantiation at the decoder's shapes (raw launches) + a bit-identity check against the 16-wave kernels of the same library;
LHRS_HIP_LIB selects the library (kernel A/B on one box): python tools/time_u4_variants.py [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import kernels as hk, _lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 16380
dev, d, ff = "cuda", 4096, 11008
lib = _lib.load()
g = torch.Generator().manual_seed(1)
mk = lambda r, c, s=1.0: (torch.randn(r, c, generator=g) * s).to(dev, torch.bfloat16)
inv = 1.0 / (10000.0 ** (torch.arange(0, 128, 2).float() / 128))
fr = torch.outer(torch.arange(512).float(), inv)
cos, sin = fr.cos().to(dev).contiguous(), fr.sin().to(dev).contiguous()
st = lambda: torch.cuda.current_stream().cuda_stream


def timed(fn, n=20):
    best = 1e9
    for _ in range(3):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


x, dy = mk(M, d), mk(M, d, 0.1)
res = mk(M, d)
w_o, w_qkv, w_gu, w_dT, w_down = mk(d, d, 0.02), mk(3 * d, d, 0.02), mk(2 * ff, d, 0.02), mk(ff, d, 0.02), mk(d, ff, 0.02)
act = mk(M, ff, 0.5)
hk.gemm_set_u4(False)
want_o = hk.gemm_nt(x, w_o, residual=res)
want_rope = hk.gemm_rope_fwd(x, w_qkv, cos, sin, pos_mod=273, pos0=0, rope_cols=2 * d, head_dim=128)
want_gu, want_act = hk.gemm_swiglu_fwd(x, w_gu, ff)
want_dgu = hk.gemm_swiglu_bwd(dy, w_dT, want_gu.clone(), ff)
want_down = hk.gemm_nt(act, w_down)
want_do = hk.gemm_nt(dy, w_o)
hk.gemm_set_u4(True)
out_o, out_qkv, gu, a_out, out_down = torch.empty_like(want_o), torch.empty_like(want_rope), torch.empty_like(want_gu), torch.empty_like(want_act), torch.empty_like(want_down)
dgu = want_gu.clone()
out_o2 = torch.empty_like(want_do)
P = lambda t: t.data_ptr()
runs = {
    "<0,true>  o + residual   K=4096 ": (lambda: lib.lhrs_gemm_u4_nt(P(x), d, P(w_o), d, P(out_o), d, M, d, d, P(res), d, st()), 2.0 * M * d * d, lambda: (out_o.float() - want_o.float()).abs().max().item() < 0.07),
    "<0,false> d-o            K=4096 ": (lambda: lib.lhrs_gemm_u4_nt(P(dy), d, P(w_o), d, P(out_o2), d, M, d, d, None, 0, st()), 2.0 * M * d * d, lambda: torch.equal(out_o2, want_do)),
    "<0,false> down           K=11008": (lambda: lib.lhrs_gemm_u4_nt(P(act), ff, P(w_down), ff, P(out_down), d, M, d, ff, None, 0, st()), 2.0 * M * d * ff, lambda: torch.equal(out_down, want_down)),
    "<3,false> q|k|v + RoPE   K=4096 ": (lambda: lib.lhrs_gemm_u4_rope(P(x), d, P(w_qkv), d, P(out_qkv), 3 * d, M, 3 * d, d, P(cos), P(sin), 273, 0, 2 * d, st()), 2.0 * M * 3 * d * d, lambda: torch.equal(out_qkv, want_rope)),
    "<1,false> gate|up SwiGLU K=4096 ": (lambda: lib.lhrs_gemm_u4_swiglu_fwd(P(x), d, P(w_gu), d, P(gu), 2 * ff, P(a_out), ff, M, ff, d, st()), 2.0 * M * 2 * ff * d, lambda: torch.equal(gu, want_gu) and torch.equal(a_out, want_act)),
    "<2,false> d-down SwiGLU' K=4096 ": (lambda: lib.lhrs_gemm_u4_swiglu_bwd(P(dy), d, P(w_dT), d, P(want_gu), P(dgu), 2 * ff, M, ff, d, st()), 2.0 * M * ff * d, lambda: torch.equal(dgu, want_dgu)),
}
print("library:", os.environ.get("LHRS_HIP_LIB", "(in-tree)"), " M =", M)
for name, (fn, fl, ok) in runs.items():
    assert fn() == 0
    torch.cuda.synchronize()
    good = ok()
    t = timed(fn)
    print(f"  gemm_u4_kernel{name}: {t:8.1f} us  {fl / t / 1e6:7.1f} TFLOP/s  frac {fl / t / 1e6 / 2500:.4f}  result {'ok' if good else 'WRONG'}", flush=True)
