"""Per-shape throughput of the bf16 NT GEMM on the LLaMA shapes of the stage-1 step (uniform random operands)."""
import ctypes
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import _lib, kernels as hk

lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
M = B * 273
shapes = [(M, 12288, 4096), (M, 4096, 4096), (M, 22016, 4096), (M, 4096, 11008), (M, 11008, 4096), (M, 4096, 22016), (M, 4096, 12288)]
for pol in (3, 2, 3, 2):
    lib.lhrs_gemm_set_policy(pol)
    tot_t = tot_f = 0
    for (m, n, k) in shapes:
        a = (torch.rand(m, k, device="cuda") * 2 - 1).to(torch.bfloat16)
        b = (torch.rand(n, k, device="cuda") * 2 - 1).to(torch.bfloat16)
        c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        for _ in range(3):
            hk.gemm_nt(a, b, out=c)
        torch.cuda.synchronize()
        if pol == 2 and (m, n, k) == shapes[0]:
            lib.lhrs_gemm_set_policy(1)
            ref = hk.gemm_nt(a, b)
            lib.lhrs_gemm_set_policy(2)
            print("policy 3 vs 2 max|diff| =", (ref.float() - c.float()).abs().max().item())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 10
        e0.record()
        for _ in range(it):
            hk.gemm_nt(a, b, out=c)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / it
        tf = 2.0 * m * n * k / (ms * 1e-3) / 1e12
        tot_t += ms; tot_f += 2.0 * m * n * k
        print(f"policy256={pol} M={m} N={n} K={k}: {ms*1e3:8.1f} us  {tf:7.1f} TF")
    print(f"policy256={pol} all shapes: {tot_f / (tot_t * 1e-3) / 1e12:.1f} TF")
