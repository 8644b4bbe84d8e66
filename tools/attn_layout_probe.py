"""Does the head-interleaved activation layout ([tokens, heads * 128], a head's rows are 256-B pieces at a 24 KiB stride) cost the resident attention forward HBM efficiency?
The same kernel on the same amount of data with every (sequence, head) contiguous (heads as their own 'sequences', H = 1, ld = 128)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import _lib, kernels as hk

_lib.load()
B, S, H, D = 60, 273, 32, 128
sc = 1.0 / math.sqrt(D)
LT = hk.pad64(S)


def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


# (a) the training layout
M = B * S
qkv = (torch.randn(M, 3 * H * D, device="cuda") * 0.5).to(torch.bfloat16)
o = torch.empty(M, H * D, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B, H, LT, device="cuda", dtype=torch.float32)
desc = hk.make_desc([(b * S, S, b * S, S) for b in range(B)], "cuda")
d = H * D
t = timeit(lambda: hk.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, desc, B, H, D, S, S, LT, True, sc))
print(f"interleaved heads (ld 12288): {t:7.1f} us")
# (b) every (sequence, head) contiguous
N = B * H
q2 = (torch.randn(N * S, D, device="cuda") * 0.5).to(torch.bfloat16)
k2, v2 = torch.randn_like(q2), torch.randn_like(q2)
o2 = torch.empty_like(q2)
lse2 = torch.empty(N, 1, LT, device="cuda", dtype=torch.float32)
desc2 = hk.make_desc([(n * S, S, n * S, S) for n in range(N)], "cuda")
t = timeit(lambda: hk.attn_fwd(q2, k2, v2, o2, lse2, desc2, N, 1, D, S, S, LT, True, sc))
print(f"contiguous (sequence, head)  : {t:7.1f} us")
