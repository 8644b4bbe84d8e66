"""Per-shape throughput of the bf16 NT GEMM on the LLaMA shapes of the stage-1 step (uniform random operands): A/B of the 16-wave 256x256
kernel with one workgroup per tile (persistent = 0) vs one persistent workgroup per CU (persistent = 1, default), alternating on the same
box, plus a bit-exactness check between the two (same arithmetic, different schedule).   python tools/gemm_bench.py [micro_batch] [reps]"""
import os as _os
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import _lib, kernels as hk

lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
M = B * 273
shapes = [(M, 12288, 4096), (M, 4096, 4096), (M, 22016, 4096), (M, 4096, 11008), (M, 11008, 4096), (M, 4096, 22016), (M, 4096, 12288)]
ops = {}
for (m, n, k) in shapes:
    a = (torch.rand(m, k, device="cuda") * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(n, k, device="cuda") * 2 - 1).to(torch.bfloat16)
    r = (torch.rand(m, n, device="cuda") * 2 - 1).to(torch.bfloat16)
    ops[(m, n, k)] = (a, b, r, torch.empty(m, n, device="cuda", dtype=torch.bfloat16))
for (m, n, k), (a, b, r, c) in ops.items():
    lib.lhrs_gemm_set_persistent(0)
    c0 = hk.gemm_nt(a, b, residual=r).clone()
    lib.lhrs_gemm_set_persistent(1)
    c1 = hk.gemm_nt(a, b, residual=r)
    ref = (a[:64].float() @ b.float().t() + r[:64].float())
    print(f"M={m} N={n} K={k}: persistent == per-tile: {torch.equal(c0, c1)}; rows 0..63 vs fp32 torch rel {((c1[:64].float() - ref).norm() / ref.norm()).item():.2e}")
for rep in range(reps):
    for pers in (0, 1):
        lib.lhrs_gemm_set_persistent(pers)
        tot_t = tot_f = 0
        line = []
        for (m, n, k), (a, b, r, c) in ops.items():
            for _ in range(3):
                hk.gemm_nt(a, b, out=c)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            it = 10
            e0.record()
            for _ in range(it):
                hk.gemm_nt(a, b, out=c)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / it
            tot_t += ms; tot_f += 2.0 * m * n * k
            line.append(f"{n}x{k}: {ms * 1e3:7.1f} us {2.0 * m * n * k / (ms * 1e-3) / 1e12:6.1f} TF")
        print(f"[rep {rep}] persistent={pers} M={M}: " + " | ".join(line) + f" || all: {tot_f / (tot_t * 1e-3) / 1e12:.1f} TF")
lib.lhrs_gemm_set_persistent(1)
