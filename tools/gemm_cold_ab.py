"""Warm vs cold operands of the persistent GEMM: the same launch sequence over ONE set of operands (the Infinity Cache keeps what fits of it between launches:
what tools/gemm_ab.py measures) and over a ring of distinct operand sets larger than the 256 MB cache (what a training step does: every layer has its own
weights and activations).  TFLOP/s per shape.   python tools/gemm_cold_ab.py [micro-batch]"""
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
RING = 12


def rnd(*shape):
    return (torch.rand(*shape, device="cuda") * 2 - 1).to(torch.bfloat16)


for mode in ("warm", "cold", "cold weights only", "cold activations only", "warm"):
    line, tt, tf = [], 0.0, 0.0
    for (m, n, k) in shapes:
        na = RING if mode in ("cold", "cold activations only") else 1
        nb = RING if mode in ("cold", "cold weights only") else 1
        As = [rnd(m, k) for _ in range(na)]
        Bs = [rnd(n, k) for _ in range(nb)]
        Cs = [torch.empty(m, n, device="cuda", dtype=torch.bfloat16) for _ in range(na)]
        for i in range(3):
            hk.gemm_nt(As[i % na], Bs[i % nb], out=Cs[i % na])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(RING):
            hk.gemm_nt(As[i % na], Bs[i % nb], out=Cs[i % na])
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / RING
        tt += ms; tf += 2.0 * m * n * k
        line.append(f"{2.0 * m * n * k / (ms * 1e-3) / 1e12:6.0f}")
        del As, Bs, Cs
    print(f"{mode:22s}: " + " ".join(line) + f" | all {tf / (tt * 1e-3) / 1e12:.1f} TF")
