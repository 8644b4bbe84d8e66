"""Decode throughput of generate() (BASELINE configs[4] shape: 1 image, short prompt, greedy, bf16 weights)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd.unibind import UniBind

new = int(sys.argv[1]) if len(sys.argv) > 1 else 64
weights = sys.argv[2] if len(sys.argv) > 2 else "bf16"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
model = UniBind(("rgb", "text"), None, device="cuda", llama_layers=32).init_random(seed=0).eval()
ids = torch.randint(3, 32000, (B, 60)); ids[:, 0] = 1; ids[:, 1] = -200
rgb = torch.randn(B, 3, 224, 224)
model.generate(ids, images=rgb, do_sample=False, max_new_tokens=4, weights=weights)
torch.cuda.synchronize()
t0 = time.perf_counter()
out = model.generate(ids, images=rgb, do_sample=False, max_new_tokens=new, weights=weights)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"[{weights}, batch {B}] {B}x{new} new tokens in {dt:.3f}s = {B*new/dt:.1f} tok/s (incl. ViT+pooler+prefill of {60-1+144} positions); HBM roofline " + ("6.74 GB/token @ 8 TB/s = 1190 tok/s" if weights == "fp8" else "13.5 GB/token @ 8 TB/s = 590 tok/s") + " per sequence")
