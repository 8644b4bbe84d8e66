import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lhrs_bot_amd import kernels as hk
from lhrs_bot_amd.unibind import UniBind
from oracle import int8_oracle as I8, lhrs_oracle as O, params as OP
nl = 1
P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=nl)}
for L in P["llama"]["layers"]:
    for k in ("qkv_w", "o_w", "gu_w", "down_w"):
        L[k] = L[k].to(torch.bfloat16).float()
model = UniBind(("rgb", "text"), None, device="cuda", llama_layers=nl).load_params(P)
model.text.quantize_base(8, "int8")
model.prepare_for_training()
model.text.tail_rows_only = False
g = torch.Generator().manual_seed(11)
B, T = 2, 24
ids = torch.randint(3, 32000, (B, T), generator=g); ids[:, 0], ids[:, 1] = 1, -200
labels = ids.clone(); labels[:, :2] = -100
batch = dict(rgb=torch.randn(B, 3, 224, 224, generator=g), input_ids=ids, labels=labels, attention_mask=ids.ne(0))
out = model(batch)
Pq = dict(P, llama=I8.int8_llama_params(P["llama"]))
col = {}
with torch.no_grad():
    l_int8 = O.unibind_forward(Pq, batch, col).item()
    l_fp32 = O.unibind_forward(P, batch).item()
print("HIP", out["total_loss"].item(), "oracle int8", l_int8, "oracle fp32", l_fp32, "i8 active", model.text._i8(model.text.p["layers"][0], "qkv_w") is not None)
# per-op: feed the oracle's normalised input of layer 0 to both
emb = col["embeds"]
L = P["llama"]["layers"][0]
h = O._rms(emb, L["ln1_w"], 1e-5).reshape(-1, 4096)
hb = h.to(torch.bfloat16)
Ld = model.text.p["layers"][0]
y = hk.int8_linear(hb.cuda(), Ld["qkv_wi8"], Ld["qkv_wi8s"], model.text._i8ws)
want = I8.linear(hb.float(), I8.Int8Weight(L["qkv_w"]))
ref32 = hb.float() @ L["qkv_w"].t()
rel = lambda a, b: ((a.double().cpu() - b.double().cpu()).norm() / b.double().norm()).item()
print("qkv: HIP vs int8 oracle", rel(y.float(), want), " int8 oracle vs fp32", rel(want, ref32), " HIP vs fp32", rel(y.float(), ref32))
hid_h = model.text.last_hidden.float().cpu().reshape(B, -1, 4096)
print("hidden: HIP vs int8 oracle", rel(hid_h, col["hidden"]))
# ---- every op of layer 0 on the oracle's own inputs
import math, torch.nn.functional as F
x = emb.reshape(-1, 4096)
S = emb.shape[1]
def i8(name, inp, res=None):
    y = hk.int8_linear(inp.to(torch.bfloat16).cuda(), Ld[name + "i8"], Ld[name + "i8s"], model.text._i8ws, residual=None if res is None else res.to(torch.bfloat16).cuda())
    want = I8.linear(inp.to(torch.bfloat16).float(), I8.Int8Weight(L[name])) + (0 if res is None else res.to(torch.bfloat16).float())
    print(name, tuple(inp.shape), "HIP vs int8 oracle", rel(y.float(), want), "overflow", model.text._i8ws.overflowed())
    return want
qkv = i8("qkv_w", h)
a = torch.randn(B * S, 4096, generator=g) * 0.05
xm = i8("o_w", a, x)
h2 = O._rms(xm, L["ln2_w"], 1e-5)
gu = i8("gu_w", h2)
act = F.silu(gu[:, :11008]) * gu[:, 11008:]
i8("down_w", act, xm)
# the model's own layer on the oracle embeddings
hid = model.text.forward_hidden(emb.to(torch.bfloat16).cuda(), None, save_ctx=False)
with torch.no_grad():
    want_h = O.llama_hidden(Pq["llama"], emb.to(torch.bfloat16).float(), None)
print("layer: HIP vs int8 oracle", rel(hid.float().reshape(B, S, 4096), want_h))
model.text.base_int8 = False
hid16 = model.text.forward_hidden(emb.to(torch.bfloat16).cuda(), None, save_ctx=False)
print("layer with the dequantised bf16 weights (no int8 product): vs int8 oracle", rel(hid16.float().reshape(B, S, 4096), want_h))
