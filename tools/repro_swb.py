"""gemm_u4_kernel<2,false> (d-down + SwiGLU') raw launch at one M, against the 16-wave kernel: python tools/repro_swb.py M [ff] [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import kernels as hk, _lib

M = int(sys.argv[1]); ff = int(sys.argv[2]) if len(sys.argv) > 2 else 11008; K = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
dev = "cuda"
g = torch.Generator().manual_seed(1)
mk = lambda r, c, s=1.0: (torch.randn(r, c, generator=g) * s).to(dev, torch.bfloat16)
dy, w, gu0 = mk(M, K, 0.1), mk(ff, K, 0.02), mk(M, 2 * ff)
hk.gemm_set_u4(False)
want = hk.gemm_swiglu_bwd(dy, w, gu0.clone(), ff)
torch.cuda.synchronize()
print("16-wave done", flush=True)
lib = _lib.load()
for i in range(3):
    gu = gu0.clone()
    st = lib.lhrs_gemm_u4_swiglu_bwd(dy.data_ptr(), dy.stride(0), w.data_ptr(), w.stride(0), gu.data_ptr(), gu.data_ptr(), gu.stride(0), M, ff, K, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print("M", M, "ff", ff, "K", K, "launch", i, "status", st, "equal", bool(torch.equal(gu, want)), flush=True)
