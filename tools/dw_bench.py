"""The projector's weight-gradient products dW[out, in] = dY^T X at micro-batch B (default 30): time of the two operand transposes and of the
split-K f32 GEMM per shape, and their share of a step.   python tools/dw_bench.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import _lib, kernels as hk

_lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
shapes = [("out_proj   ", B * 144, 4096, 1024, 1), ("mlp.c_proj ", B * 144, 1024, 4096, 6), ("mlp.c_fc   ", B * 144, 4096, 1024, 6),
          ("attn.out   ", B * 144, 1024, 1024, 6), ("in_proj q  ", B * 144, 1024, 1024, 6), ("in_proj kv ", B * 912, 2048, 1024, 6)]


def timeit(fn, it=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


tot_t = tot_g = 0.0
for name, M, out, inn, n in shapes:
    dy = (torch.randn(M, out, device="cuda") * 0.1).to(torch.bfloat16)
    x = torch.randn(M, inn, device="cuda").to(torch.bfloat16)
    g = torch.empty(out, inn, device="cuda")
    Mp = hk.pad64(M)
    t_tr = timeit(lambda: (hk.transpose(dy, rows_pad=Mp), hk.transpose(x, rows_pad=Mp)))
    dyT, xT = hk.transpose(dy, rows_pad=Mp), hk.transpose(x, rows_pad=Mp)
    t_g = timeit(lambda: hk.gemm_nt_splitk_f32(dyT, xT, g))
    fl = 2.0 * M * out * inn
    print(f"{name} tokens {M:6d} -> dW [{out}, {inn}]: transposes {t_tr:6.1f} us, GEMM {t_g:6.1f} us ({fl / t_g / 1e6:6.1f} TF/s), x{n} per step")
    tot_t += n * t_tr
    tot_g += n * t_g
print(f"per step: transposes {tot_t / 1e3:.2f} ms + GEMMs {tot_g / 1e3:.2f} ms")

print("--- lhrs_gemm_tn_f32 (token-major operands, transposing LDS reads, token-split slabs)")
tot = 0.0
for name, M, out, inn, n in shapes:
    dy = (torch.randn(M, out, device="cuda") * 0.1).to(torch.bfloat16)
    x = torch.randn(M, inn, device="cuda").to(torch.bfloat16)
    g = torch.empty(out, inn, device="cuda")
    t = timeit(lambda: hk.gemm_tn_f32(dy, x, g))
    print(f"{name} tokens {M:6d} -> dW [{out}, {inn}]: {t:6.1f} us ({2.0 * M * out * inn / t / 1e6:6.1f} TF/s)")
    tot += n * t
print(f"per step: {tot / 1e3:.2f} ms")

# alternative: the TN kernel of the LoRA path (tn_skinny: both operands transposed by LDS reads, no operand transposes) in chunks of <= 384 output rows
print("--- same products through lhrs_gemm_tn_skinny in chunks of 256 output rows (no operand transposes)")
tot = 0.0
for name, M, out, inn, n in shapes:
    dy = (torch.randn(M, out, device="cuda") * 0.1).to(torch.bfloat16)
    x = torch.randn(M, inn, device="cuda").to(torch.bfloat16)
    g = torch.empty(out, inn, device="cuda")
    gref = torch.empty(out, inn, device="cuda")
    Mp = hk.pad64(M)
    hk.gemm_nt_splitk_f32(hk.transpose(dy, rows_pad=Mp), hk.transpose(x, rows_pad=Mp), gref)

    def run():
        for c0 in range(0, out, 256):
            hk.gemm_tn_skinny(dy[:, c0:c0 + 256], x, g[c0:c0 + 256])
    t = timeit(run)
    err = ((g - gref).norm() / gref.norm()).item()
    print(f"{name} tokens {M:6d} -> dW [{out}, {inn}]: {t:6.1f} us ({2.0 * M * out * inn / t / 1e6:6.1f} TF/s)  rel diff vs transposed path {err:.1e}")
    tot += n * t
print(f"per step: {tot / 1e3:.2f} ms")
