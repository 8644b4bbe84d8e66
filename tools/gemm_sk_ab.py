"""Stream-K tail of the persistent 256x256 GEMM against whole-tile rounds, alternating inside one process on one box (us per launch).
   python tools/gemm_sk_ab.py [micro-batches ...]      default 30 32 8"""
import os as _os
_os.environ.setdefault("LHRS_GEMM_VENDOR", "0")   # these tools measure the hand-written kernels, not the vendor library
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import _lib, kernels as hk

lib = _lib.load()
hk.ensure_streamk_workspace("cuda", force=True)
d, ff = 4096, 11008


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def rnd(*shape, s=1.0):
    return ((torch.rand(*shape, device="cuda") * 2 - 1) * s).to(torch.bfloat16)


for B in [int(a) for a in sys.argv[1:]] or [30, 32, 8]:
    M = B * 273
    x, xf = rnd(M, d), rnd(M, ff)
    wqkv, wo, wgu, wd, wdT, wguT = rnd(3 * d, d, s=.02), rnd(d, d, s=.02), rnd(2 * ff, d, s=.02), rnd(d, ff, s=.02), rnd(ff, d, s=.02), rnd(d, 2 * ff, s=.02)
    lm, xl = rnd(32000, d, s=.02), rnd(B * 128, d)
    gu = rnd(M, 2 * ff)
    dgu = rnd(M, 2 * ff)
    out_qkv = torch.empty(M, 3 * d, device="cuda", dtype=torch.bfloat16)
    cases = [("qkv   N=12288 K=4096 ", lambda: hk.gemm_nt(x, wqkv, out=out_qkv)),
             ("o     N=4096  K=4096 ", lambda: hk.gemm_nt(x, wo)),
             ("gu+swiglu N=22016    ", lambda: hk.gemm_swiglu_fwd(x, wgu, ff)),
             ("down  N=4096  K=11008", lambda: hk.gemm_nt(xf, wd)),
             ("d-down+swiglu' N=11008", lambda: hk.gemm_swiglu_bwd(x, wdT, gu, ff, out=dgu)),
             ("d-gu  N=4096  K=22016", lambda: hk.gemm_nt(dgu, wguT)),
             ("lm_head N=32000      ", lambda: hk.gemm_nt(xl, lm))]
    print(f"--- micro-batch {B}: M = {M}   (us per launch: whole rounds / stream-K tail, two alternations)")
    tot = [0.0, 0.0]
    for name, fn in cases:
        t = []
        for rep in range(2):
            for on in (0, 1):
                lib.lhrs_gemm_set_streamk(on)
                t.append(timeit(fn))
        lib.lhrs_gemm_set_streamk(0)
        off, on = 0.5 * (t[0] + t[2]), 0.5 * (t[1] + t[3])
        tot[0] += off; tot[1] += on
        print(f"{name}: {t[0]:8.1f} {t[2]:8.1f} / {t[1]:8.1f} {t[3]:8.1f}   {100 * (off / on - 1):+5.1f} %")
    print(f"layer sum: {tot[0]:.1f} / {tot[1]:.1f} us  {100 * (tot[0] / tot[1] - 1):+.1f} %")
