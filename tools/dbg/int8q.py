import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lhrs_bot_amd import kernels as hk
from oracle import int8_oracle as I8
g = torch.Generator().manual_seed(1)
w = (torch.randn(4096, 4096, generator=g) * 0.02).to(torch.bfloat16)
wq, ws_ = hk.quant_int8_rows(w.cuda())
cb, scb = I8.quantize_rows_int8(w.float())
bad = (wq.cpu() != cb)
print("mismatches", int(bad.sum()), "of", bad.numel())
r, c = bad.nonzero(as_tuple=True)
for i in range(min(8, len(r))):
    rr, cc = r[i].item(), c[i].item()
    v = w[rr, cc].float().item(); m = scb[rr].item()
    print(rr, cc, "w", v, "absmax", m, "gpu", wq[rr, cc].item(), "cpu", cb[rr, cc].item(), "v*(127/m)", v * (127.0 / m), "fp32:", (torch.tensor(v) * (torch.tensor(127.0) / torch.tensor(m))).item(), "gpu scale*127", ws_[rr].item() * 127)
