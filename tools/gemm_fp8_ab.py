"""All-shape throughput of the e4m3 NT GEMM (lhrs_gemm_fp8_nt), with a check against the dequantised product on the first shape:
   python tools/gemm_fp8_ab.py 30            # LHRS_HIP_LIB=... selects the build
   GEMM_ZERO=1: all-zero operands (no power limit: the schedule alone); GEMM_LORA=1: with a fused bf16 pair of K2 = 64"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import kernels as hk

B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
M = B * 273
shapes = [(M, 12288, 4096), (M, 4096, 4096), (M, 22016, 4096), (M, 4096, 11008), (M, 11008, 4096), (M, 4096, 22016), (M, 4096, 12288)]
z = 0 if os.environ.get("GEMM_ZERO") == "1" else 1
K2 = 64 if os.environ.get("GEMM_LORA") == "1" else 0
tot_t = tot_f = 0
line = []
for si, (m, n, k) in enumerate(shapes):
    a = (torch.randn(m, k, device="cuda") * z).to(torch.bfloat16)
    b = (torch.randn(n, k, device="cuda") * 0.02 * z).to(torch.bfloat16)
    a8, sa = hk.quant_fp8_rows(a)
    b8, sb = hk.quant_fp8_rows(b)
    kw = {}
    if K2:
        kw = dict(a2=(torch.randn(m, K2, device="cuda") * 0.3).to(torch.bfloat16), b2=(torch.randn(n, K2, device="cuda") * 0.1).to(torch.bfloat16))
    for _ in range(3):
        c = hk.gemm_fp8_nt(a8, sa, b8, sb, **kw)
    if si == 1 and z:
        ref = (a8.view(torch.float8_e4m3fn).float() * sa[:, None]) @ (b8.view(torch.float8_e4m3fn).float() * sb[:, None]).t()
        if K2:
            ref += kw["a2"].float() @ kw["b2"].float().t()
        print("rel err vs dequantised product:", ((c.float() - ref).norm() / ref.norm()).item())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(12):
        hk.gemm_fp8_nt(a8, sa, b8, sb, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 12
    tot_t += ms; tot_f += 2.0 * m * n * (k + K2)
    line.append(f"{2.0 * m * n * (k + K2) / (ms * 1e-3) / 1e12:6.1f}")
print(f"e4m3 zero={1 - z} K2={K2} M={M}: " + " ".join(line) + f" | all {tot_f / (tot_t * 1e-3) / 1e12:.1f} TF")
