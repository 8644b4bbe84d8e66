"""All-shape throughput of the bf16 NT GEMM with the library given by LHRS_HIP_LIB (kernel A/B of build variants on one box):
   for v in base X; do LHRS_HIP_LIB=... python tools/gemm_ab.py 30; done"""
import os as _os
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import _lib, kernels as hk

lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
M = B * 273
shapes = [(M, 12288, 4096), (M, 4096, 4096), (M, 22016, 4096), (M, 4096, 11008), (M, 11008, 4096), (M, 4096, 22016), (M, 4096, 12288)]
tot_t = tot_f = 0
line = []
for (m, n, k) in shapes:
    z = 0 if os.environ.get("GEMM_ZERO") == "1" else 1  # zeros: no power limit, the schedule alone
    pad = int(os.environ.get("GEMM_PAD", "0"))  # extra elements in the leading dimension of both operands (power-of-two row strides vs not)
    pad_a = int(os.environ.get("GEMM_PAD_A", pad)); pad_b = int(os.environ.get("GEMM_PAD_B", pad))
    a = ((torch.rand(m, k + pad_a, device="cuda") * 2 - 1).to(torch.bfloat16) * z)[:, :k]
    b = ((torch.rand(n, k + pad_b, device="cuda") * 2 - 1).to(torch.bfloat16) * z)[:, :k]
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
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
if os.environ.get("GEMM_TORCH") == "1":  # the vendor library on the same operands (torch.mm -> hipBLASLt), for the record; never on the product path
    tt = tf = 0
    tl = []
    for (m, n, k) in shapes:
        z = 0 if os.environ.get("GEMM_ZERO") == "1" else 1
        a = (torch.rand(m, k, device="cuda") * 2 - 1).to(torch.bfloat16) * z
        bt = ((torch.rand(n, k, device="cuda") * 2 - 1).to(torch.bfloat16) * z)
        for _ in range(3):
            torch.mm(a, bt.t())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(12):
            torch.mm(a, bt.t())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 12
        tt += ms; tf += 2.0 * m * n * k
        tl.append(f"{2.0 * m * n * k / (ms * 1e-3) / 1e12:6.1f}")
    print(f"{'torch.mm (hipBLASLt)':28s} M={M}: " + " ".join(tl) + f" | all {tf / (tt * 1e-3) / 1e12:.1f} TF")
print(f"{os.path.basename(_lib.LIB_PATH):28s} M={M}: " + " ".join(line) + f" | all {tot_f / (tot_t * 1e-3) / 1e12:.1f} TF")
