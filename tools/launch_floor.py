import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lhrs_bot_amd import _lib, kernels as hk
lib = _lib.load()
dev = "cuda"
state = torch.zeros(4, device=dev, dtype=torch.int32); desc = torch.zeros((1, 8), device=dev, dtype=torch.int32); pos = torch.zeros(1, device=dev, dtype=torch.int32)
side = torch.cuda.Stream()
for n in (50, 200):
    with torch.cuda.stream(side):
        hk.decode_advance(state, desc, pos, 1, 1024, 0)
        g = hk.HipGraph(); g.begin()
        for _ in range(n): hk.decode_advance(state, desc, pos, 1, 1024, 0)
        g.end(); g.launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): g.launch()
        e1.record()
    torch.cuda.synchronize()
    print(f"graph of {n} trivial dependent kernels: {e0.elapsed_time(e1)*1e3/20/n:.2f} us per kernel")
# stream launches (no graph)
with torch.cuda.stream(side):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(2000): hk.decode_advance(state, desc, pos, 1, 1024, 0)
    e1.record()
torch.cuda.synchronize()
print(f"eager stream launches: {e0.elapsed_time(e1)*1e3/2000:.2f} us per kernel")
