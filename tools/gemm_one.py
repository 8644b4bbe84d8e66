import os as _os
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import _lib, kernels as hk
lib = _lib.load()
pol = int(sys.argv[1]); m, n, k = (int(x) for x in sys.argv[2:5])
lib.lhrs_gemm_set_policy(pol)
a = (torch.rand(m, k, device="cuda") * 2 - 1).to(torch.bfloat16)
b = (torch.rand(n, k, device="cuda") * 2 - 1).to(torch.bfloat16)
c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
for _ in range(int(sys.argv[5]) if len(sys.argv) > 5 else 5):
    hk.gemm_nt(a, b, out=c)
torch.cuda.synchronize()
