"""LLM.int8 base of stages 2 / 3 (`bits: 8`; lhrs/models/text_modal.py:91-131 -> bitsandbytes MatMul8bitLt) on the int8 MFMA, against
oracle/int8_oracle.py (the restated algorithm; bitsandbytes itself is not installed: unpinned against the package).  Integer parts -
weight codes, outlier column sets - are compared EXACTLY; products within bf16 rounding of the fp32 oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from lhrs_bot_amd import kernels as hk  # noqa: E402
from lhrs_bot_amd.engine import LHRSEngine  # noqa: E402
from lhrs_bot_amd.unibind import UniBind  # noqa: E402
from oracle import int8_oracle as I8  # noqa: E402
from oracle import lhrs_oracle as O  # noqa: E402
from oracle import params as OP  # noqa: E402

DEV = "cuda"


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.mark.parametrize("M,N,K,n_out,lora", [(300, 4096, 4096, 0, False), (2184, 4096, 4096, 5, False), (1000, 12288, 4096, 3, True),
                                             (777, 4096, 11008, 40, False), (520, 1024, 4096, 200, False), (334, 4096, 11008, 1500, True)])
def test_int8_linear_vs_llm_int8_oracle(M, N, K, n_out, lora):
    g = torch.Generator().manual_seed(M + N + K + n_out)
    x = torch.randn(M, K, generator=g)
    cols = torch.randperm(K, generator=g)[:n_out]
    for c in cols.tolist():                                   # planted outliers: one or a few entries >= 6.0 in each chosen feature column
        rows = torch.randint(0, M, (3,), generator=g)
        x[rows, c] = torch.tensor([7.5, -9.0, 30.0])[: len(rows)] * (1 + torch.rand(3, generator=g))
    x = torch.where((x.abs() >= 6.0) & ~torch.isin(torch.arange(K), cols)[None, :], torch.full_like(x, 5.0), x)   # no accidental outliers
    xb = x.to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16)
    wq, ws_ = hk.quant_int8_rows(w.to(DEV))
    cb, scb = I8.quantize_rows_int8(w.float())
    assert torch.equal(wq.cpu(), cb)                                                   # weight codes: exact
    assert torch.allclose(ws_.cpu() * 127.0, scb, rtol=1e-6)
    assert torch.equal(hk.dequant_int8_rows(wq, ws_).cpu(), I8.dequantize_rows_int8(cb, scb).to(torch.bfloat16))
    ws = hk.Int8Workspace(DEV, kmax=K)
    res = torch.randn(M, N, generator=g).to(torch.bfloat16)
    a2 = b2 = None
    if lora:
        a2 = (torch.randn(M, 64, generator=g) * 0.1).to(torch.bfloat16)
        b2 = (torch.randn(N, 64, generator=g) * 0.05).to(torch.bfloat16)
    y = hk.int8_linear(xb.to(DEV), wq, ws_, ws, residual=res.to(DEV), a2=None if a2 is None else a2.to(DEV), b2=None if b2 is None else b2.to(DEV))
    torch.cuda.synchronize()
    want_cols = sorted(((xb.float().abs() >= 6.0).any(0)).nonzero().flatten().tolist())
    assert ws.last_outlier_columns() == want_cols and int(ws.meta[1]) == (len(want_cols) + 63) // 64 * 64     # outlier column set: exact, any size
    w8 = I8.Int8Weight(w.float())
    want = I8.linear(xb.float(), w8) + res.float()
    if lora:
        want = want + a2.float() @ b2.float().t()
    assert rel(y.float(), want) < 4e-3
    # exactness of the integer product: no outliers, no residual -> y equals the oracle's value rounded once to bf16
    if n_out == 0:
        y0 = hk.int8_linear(xb.to(DEV), wq, ws_, ws)
        exact = I8.linear(xb.float(), w8)
        assert (y0.float().cpu() - exact).abs().max() <= 2 ** -7 * exact.abs().max()
        assert rel(y0.float(), exact) < 3e-3


@pytest.mark.timeout(900)
@pytest.mark.parametrize("with_lora", [False, True])
def test_llm_int8_base_end_to_end_vs_int8_oracle(with_lora):
    """UniBind with the LLM.int8 base (what `bits: 8` of the stage-2/3 YAMLs selects) against the oracle running the SAME int8 arithmetic
    (`int8_llama_params`): loss, d loss / d image, and with adapters every dA / dB; the backward multiplies with the dequantised weights."""
    nl = 2
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=nl)}
    for L in P["llama"]["layers"]:          # the int8 codes are a function of the 16-bit checkpoint weights: both sides quantise the SAME bf16 values
        for k in ("qkv_w", "o_w", "gu_w", "down_w"):
            L[k] = L[k].to(torch.bfloat16).float()
    model = UniBind(("rgb", "text"), None, device=DEV, llama_layers=nl).load_params(P)
    targets = ("q", "k", "v", "o", "gate", "up", "down")
    lora_p = OP.make_lora_params(seed=4, layers=nl, r=16, alpha=32, targets=targets) if with_lora else None
    if with_lora:
        lora = model.enable_lora(r=16, alpha=32, targets=targets, seed=0)
        for l in range(nl):
            for pr in targets:
                lora.set_adapter(l, pr, *lora_p[l][pr])
        lora.refresh()
    model.text.quantize_base(8, "int8")
    assert model.text.base_int8 and not model.text.base8
    model.prepare_for_training(freeze_vision=True, freeze_text=not with_lora, tune_rgb_pooler=True)
    g = torch.Generator().manual_seed(11)
    B, T = 2, 24
    ids = torch.randint(3, 32000, (B, T), generator=g)
    ids[:, 0], ids[:, 1] = 1, -200
    ids[1, 20:] = 0
    labels = ids.clone()
    labels[:, :2] = -100
    labels[ids == 0] = -100
    batch = dict(rgb=torch.randn(B, 3, 224, 224, generator=g), input_ids=ids, labels=labels, attention_mask=ids.ne(0))
    out = model(batch)
    d_image = model.text.backward()
    torch.cuda.synchronize()
    Pq = dict(P, llama=I8.int8_llama_params(P["llama"]))
    if with_lora:
        for L, lo in zip(Pq["llama"]["layers"], lora_p):
            L["lora"] = {"scale": lo["scale"], **{pr: (lo[pr][0].clone().requires_grad_(True), lo[pr][1].clone().requires_grad_(True)) for pr in targets}}
    P["pooler"]["out_proj_b"].requires_grad_(True)
    col = {}
    loss = O.unibind_forward(Pq, batch, col)
    col["image"].retain_grad()
    loss.backward()
    assert abs(out["total_loss"].item() - loss.item()) < 1e-3 * loss.item(), (out["total_loss"].item(), loss.item())
    assert rel(d_image, col["image"].grad) < 5e-2
    if with_lora:
        for l in range(nl):
            for pr in targets:
                dA, dB = model.text.lora.grad_adapter(l, pr)
                Ao, Bo = Pq["llama"]["layers"][l]["lora"][pr]
                assert rel(dA, Ao.grad) < 8e-2 and rel(dB, Bo.grad) < 8e-2, (l, pr)   # bf16 activations through int8 codes: a flipped code is a 1 % step
