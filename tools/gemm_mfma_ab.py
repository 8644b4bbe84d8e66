"""Same-process A/B of the two MFMA shapes of the dominant 256x256 GEMM (lhrs_gemm_set_mfma16): all LLaMA shapes at micro-batch B, random
operands, alternating runs.   python tools/gemm_mfma_ab.py 30"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import _lib, kernels as hk

lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
M = B * 273
shapes = [(M, 12288, 4096), (M, 4096, 4096), (M, 22016, 4096), (M, 4096, 11008), (M, 11008, 4096), (M, 4096, 22016), (M, 4096, 12288)]
ZERO = len(sys.argv) > 2 and sys.argv[2] == "zeros"  # all-zero operands: the chip is not power-limited, what is left is the schedule
ops = []
for (m, n, k) in shapes:
    a = (torch.rand(m, k, device="cuda") * 2 - 1).to(torch.bfloat16) * (0 if ZERO else 1)
    b = (torch.rand(n, k, device="cuda") * 2 - 1).to(torch.bfloat16) * (0 if ZERO else 1)
    ops.append((a, b, torch.empty(m, n, device="cuda", dtype=torch.bfloat16)))
# results of the two kernels on one shape: same products, fp32 accumulation in a different order
lib.lhrs_gemm_set_mfma16(0); c0 = hk.gemm_nt(ops[1][0], ops[1][1]).float()
lib.lhrs_gemm_set_mfma16(1); c1 = hk.gemm_nt(ops[1][0], ops[1][1]).float()
print("max |s - r|:", (c1 - c0).abs().max().item())
for rep in range(3):
    for mode in (0, 1):
        lib.lhrs_gemm_set_mfma16(mode)
        tot_t = tot_f = 0
        line = []
        for (a, b, c), (m, n, k) in zip(ops, shapes):
            for _ in range(2):
                hk.gemm_nt(a, b, out=c)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(12):
                hk.gemm_nt(a, b, out=c)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 12
            tot_t += ms; tot_f += 2.0 * m * n * k
            line.append(f"{2.0 * m * n * k / (ms * 1e-3) / 1e12:6.1f}")
        print(f"{('32x32x16 r', '16x16x32 s')[mode]:12s} M={M}: " + " ".join(line) + f" | all {tot_f / (tot_t * 1e-3) / 1e12:.1f} TF", flush=True)
