"""sha256 of the attention kernels' outputs at the stage-1 shape (B sequences of S = 273, 32 heads, seeded inputs): run under two builds (LHRS_HIP_LIB) and compare."""
import hashlib, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import _lib, kernels as hk
_lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
S, H, D = 273, 32, 128
d, M = H * D, B * S
g = torch.Generator().manual_seed(3)
qkv = (torch.randn(M, 3 * d, generator=g) * 0.5).to("cuda", torch.bfloat16)
do = (torch.randn(M, d, generator=g) * 0.1).to("cuda", torch.bfloat16)
o = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
dqkv = torch.empty_like(qkv)
LT = hk.pad64(S)
lse = torch.empty(B, H, LT, device="cuda", dtype=torch.float32)
delta = torch.empty_like(lse)
desc = hk.make_desc([(b * S, S, b * S, S) for b in range(B)], "cuda")
q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
sc = 1 / math.sqrt(D)
for _ in range(3):
    hk.attn_fwd(q, k, v, o, lse, desc, B, H, D, S, S, LT, True, sc)
    hk.attn_bwd_o(q, k, v, do, o, lse, delta, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], desc, B, H, D, S, S, LT, True, sc)
torch.cuda.synchronize()
h = hashlib.sha256()
for t in (o, lse[:, :, :S].contiguous(), dqkv):
    assert bool(torch.isfinite(t.float()).all())
    h.update(t.cpu().numpy().tobytes() if t.dtype != torch.bfloat16 else t.view(torch.int16).cpu().numpy().tobytes())
print("ATTN_SHA", B, h.hexdigest())
