"""Throughput of the causal D=128 attention kernels at the stage-1 shape (B sequences of S=273, 32 heads)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import _lib, kernels as hk

_lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
S = int(sys.argv[2]) if len(sys.argv) > 2 else 273
H, D = 32, 128
d = H * D
M = B * S
qkv = (torch.randn(M, 3 * d, device="cuda") * 0.5).to(torch.bfloat16)
o = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
do = (torch.randn(M, d, device="cuda") * 0.1).to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
LT = hk.pad64(S)
lse = torch.empty(B, H, LT, device="cuda", dtype=torch.float32)
delta = torch.empty_like(lse)
desc = hk.make_desc([(b * S, S, b * S, S) for b in range(B)], "cuda")
sc = 1.0 / math.sqrt(D)
q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]


def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


f_fwd = 4.0 * B * H * S * S * D / 2
t = timeit(lambda: hk.attn_fwd(q, k, v, o, lse, desc, B, H, D, S, S, LT, True, sc))
print(f"fwd   {t:8.1f} us  {f_fwd / t / 1e6:7.1f} TF (causal-counted)")
t = timeit(lambda: hk.attn_delta(o, do, delta, desc, B, H, D, S, LT))
print(f"delta {t:8.1f} us")
t = timeit(lambda: hk.attn_bwd(q, k, v, do, lse, delta, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], desc, B, H, D, S, S, LT, True, sc))
print(f"bwd   {t:8.1f} us  {2.5 * f_fwd / t / 1e6:7.1f} TF (causal-counted)")
def both():
    hk.attn_delta(o, do, delta, desc, B, H, D, S, LT)
    hk.attn_bwd(q, k, v, do, lse, delta, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], desc, B, H, D, S, S, LT, True, sc)
t = timeit(both)
print(f"delta + bwd          {t:8.1f} us")
t = timeit(lambda: hk.attn_bwd_o(q, k, v, do, o, lse, delta, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], desc, B, H, D, S, S, LT, True, sc))
print(f"bwd_o (fused delta)  {t:8.1f} us")
