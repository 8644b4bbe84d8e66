"""Phase timing of the resident attention kernels (a build with -DATTN_DIAG: tools/build_variant.sh d attention.hip "-DATTN_DIAG"; run with LHRS_HIP_LIB=.../liblhrs_d.so).
python tools/attn_diag.py B fwd|dq|dkv.  Per wave the kernel stamps wall_clock64() (100 MHz): 0 entry, 1 operands visible, 4 exit; 2 = time in the groups' row loads
(Q / dO / O or K / V), 3 = in the tile loops, 5 = in the epilogues (stores), 6 = tile units.  profiles/r06_attention_phase_analysis.txt is a reading of its output."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import torch
from lhrs_bot_amd import _lib, kernels as hk

L = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
which = sys.argv[2] if len(sys.argv) > 2 else "fwd"
S, H, D = 273, 32, 128
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
q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
sc = 1 / math.sqrt(D)
dbg = torch.zeros(B * H * 8 * 8, device="cuda", dtype=torch.int64)
fn = L.lhrs_attn_set_dbg
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]


def run():
    if which == "fwd":
        hk.attn_fwd(q, k, v, o, lse, desc, B, H, D, S, S, LT, True, sc)
    else:
        hk.attn_bwd_o(q, k, v, do, o, lse, delta, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], desc, B, H, D, S, S, LT, True, sc)


hk.attn_fwd(q, k, v, o, lse, desc, B, H, D, S, S, LT, True, sc)
for _ in range(3):
    run()
torch.cuda.synchronize()
assert fn(dbg.data_ptr(), {"fwd": 0, "dq": 1, "dkv": 2}[which]) == 0
run()
torch.cuda.synchronize()
t = dbg.view(B * H, 8, 8).cpu().double() * 0.01  # us
t0 = t[:, :, 0].min()
wg_start = t[:, :, 0].min(dim=1).values
wg_end = t[:, :, 4].max(dim=1).values
life = wg_end - wg_start
print(f"[{which}] kernel span {float(t[:, :, 4].max() - t0):.1f} us; workgroup life mean {float(life.mean()):.2f} us (min {float(life.min()):.2f}, max {float(life.max()):.2f})")


def row(name, dlt):
    print(f"  {name:42s} mean {float(dlt.mean()):6.2f} us   by wave: " + " ".join(f"{float(dlt[:, w].mean()):5.2f}" for w in range(8)))


row("entry -> operands visible", t[:, :, 1] - t[:, :, 0])
row("groups' row loads (sum)", t[:, :, 2])
row("tile loops (sum)", t[:, :, 3])
row("epilogues (sum)", t[:, :, 5])
units = t[:, :, 6] * 100
row("tile units", units)
print(f"  per tile unit: {float(t[:, :, 3].sum() / units.sum()):.3f} us")
row("wave life", t[:, :, 4] - t[:, :, 0])
first = wg_start < t0 + 3.0
print(f"  first-round workgroups ({int(first.sum())}): life {float(life[first].mean()):.2f} us; later rounds: {float(life[~first].mean()):.2f} us")
