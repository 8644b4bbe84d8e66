"""Per-shape time of every bf16 GEMM launch of one stage-1 step (events around each call; B=30): where the small products go."""
import os as _os
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import kernels as hk
from lhrs_bot_amd.engine import LHRSEngine
from lhrs_bot_amd.unibind import UniBind
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 2
model = UniBind(("rgb", "text"), None, device="cuda", llama_layers=layers).init_random(seed=0)
model.prepare_for_training()
eng = LHRSEngine(model, optimizer="adanp", lr=2e-4, max_grad_norm=0.3)
batch = bench.make_batch(B, 130, torch.device("cuda"), seed=322)
def step():
    out = eng(batch); eng.backward(out["total_loss"]); eng.step()
for _ in range(2): step()
torch.cuda.synchronize()
log = collections.defaultdict(lambda: [0, 0.0])
orig = hk.gemm_nt
def timed(a, b, out=None, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(a, b, out, **kw); e1.record(); torch.cuda.synchronize()
    key = (a.shape[0], b.shape[0], a.shape[1], "f32" if kw.get("out_f32") else "bf16", "bias" if kw.get("bias") is not None else "", "res" if kw.get("residual") is not None else "")
    log[key][0] += 1; log[key][1] += e0.elapsed_time(e1) * 1e3
    return r
hk.gemm_nt = timed
import lhrs_bot_amd.pooler as P, lhrs_bot_amd.vision as V, lhrs_bot_amd.text as T
step()
tot = sum(v[1] for v in log.values())
print(f"gemm_nt launches of one step (B={B}, {layers} LLaMA layers): {tot/1e3:.2f} ms")
for k, (n, us) in sorted(log.items(), key=lambda kv: -kv[1][1]):
    M, N, K = k[:3]
    print(f"M={M:6d} N={N:6d} K={K:6d} {k[3]:4s} {k[4]:4s} {k[5]:3s} x{n:3d}  {us/n:7.1f} us each  {2*M*N*K/(us/n)/1e6:6.0f} TF/s  total {us/1e3:6.2f} ms")
