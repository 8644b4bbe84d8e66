"""Same-box A/B of a library switch on the whole training step: python tools/step_ab.py <setter> <micro_batch> [reps]
e.g. python tools/step_ab.py lhrs_gemm_set_tail_overlap 8      (modes 0 / 1 alternate `reps` times, 12 steps each, median step time)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_batch
from lhrs_bot_amd import _lib
from lhrs_bot_amd.engine import LHRSEngine
from lhrs_bot_amd.unibind import UniBind

setter, B = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
lib = _lib.load()
model = UniBind(("rgb", "text"), None, device="cuda").init_random(seed=0)
model.prepare_for_training()
eng = LHRSEngine(model, optimizer="adanp", lr=2e-4, weight_decay=0.0, max_grad_norm=0.3)
batch = make_batch(B, 130, torch.device("cuda"), seed=0)   # S = 273 like bench.py

def run(n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        out = eng(batch); eng.backward(out["total_loss"]); eng.step(); ev[i + 1].record()
    torch.cuda.synchronize()
    t = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    return t[n // 2]

run(3)
for r in range(reps):
    for mode in (0, 1):
        getattr(lib, setter)(mode)
        run(2)
        ms = run(12)
        print(f"{setter}({mode}): median {ms:8.3f} ms/step = {B / ms * 1e3:7.2f} samples/s", flush=True)
getattr(lib, setter)(1)
