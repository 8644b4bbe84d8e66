"""4-bit base storage (`bits: 4`, `quant_type: nf4 | fp4`, `double_quant`; lhrs/models/text_modal.py:91-107 -> bitsandbytes Linear4bit) against
oracle/nf4_oracle.py (the restated algorithm; bitsandbytes itself is not installed: unpinned against the package).  Codes, statistics and the
dequantised bf16 weights are compared EXACTLY; the model on the 4-bit base against the oracle model holding the oracle's dequantised weights."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from lhrs_bot_amd import kernels as hk  # noqa: E402
from lhrs_bot_amd.unibind import UniBind  # noqa: E402
from oracle import lhrs_oracle as O  # noqa: E402
from oracle import nf4_oracle as N4  # noqa: E402
from oracle import params as OP  # noqa: E402

DEV = "cuda"


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def weight(N, K, seed):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(N, K, generator=g) * 0.02
    w[3, 64:128] = 0.0                                   # an all-zero block: 0 * inf in the package's arithmetic
    w[5, 7] = 1.5                                        # a block dominated by one outlier
    w[9] *= 1e-4                                         # small statistics next to large ones (the nested 8-bit table spans 1e-6 .. 1)
    w[11, :64] = torch.tensor(N4.NF4_THR.tolist() + [0.0] * 48 + [1.0])[torch.randperm(64, generator=g)] * 0.25   # values ON the decision thresholds
    return w.to(torch.bfloat16)


@pytest.mark.parametrize("quant_type", ["nf4", "fp4"])
@pytest.mark.parametrize("double_quant", [False, True])
@pytest.mark.parametrize("N,K", [(256, 512), (1000, 4096), (96, 11008)])
def test_quant4_codes_statistics_and_dequantised_weight_bit_exact_vs_oracle(quant_type, double_quant, N, K):
    w = weight(N, K, seed=N + K)
    st = hk.quant4_blocks(w.to(DEV), quant_type, double_quant)
    want = N4.quantize_4bit(w.float().numpy(), quant_type, double_quant)
    assert np.array_equal(st["packed"].cpu().numpy(), want["packed"])
    if double_quant:
        assert st["offset"] == want["offset"]
        assert np.array_equal(st["absmax2"].cpu().numpy(), want["absmax2"])
        assert np.array_equal(st["qabsmax"].cpu().numpy(), want["qabsmax"])
    else:
        assert np.array_equal(st["absmax"].cpu().numpy(), want["absmax"])
    assert np.array_equal(hk.absmax_of(st).cpu().numpy(), N4.absmax_of(want))
    got = hk.dequant4_blocks(st)
    ref = torch.from_numpy(N4.dequantize_4bit(want)).to(torch.bfloat16)
    assert got.shape == w.shape and torch.equal(got.cpu().view(torch.int16), ref.view(torch.int16))
    # the storage is a projection: quantising the dequantised weight reproduces the codes (nf4: thresholds are the level midpoints)
    if quant_type == "nf4" and not double_quant:
        again = hk.quant4_blocks(got, quant_type, False)
        same = (again["packed"] == st["packed"]).float().mean().item()
        assert same > 0.999, same                                     # bf16 rounding of level * absmax can move a value across a threshold, rarely


def test_quant4_rejects_what_the_package_cannot_mean():
    w = weight(64, 128, seed=1).to(DEV)
    with pytest.raises(ValueError, match="quant_type"):
        hk.quant4_blocks(w, "int4", True)
    with pytest.raises(ValueError, match="contiguous"):
        hk.quant4_blocks(w[:, :64], "nf4", True)
    with pytest.raises(RuntimeError, match="multiple of the block size"):
        hk.quant4_blocks(w.flatten()[:96], "nf4", False)


@pytest.mark.parametrize("quant_type,double_quant,with_lora", [("nf4", True, False), ("nf4", True, True), ("fp4", False, False)])
def test_bits4_base_end_to_end_vs_oracle_on_the_dequantised_weights(quant_type, double_quant, with_lora):
    """UniBind with `bits: 4` storage against the oracle holding dequantize_4bit(quantize_4bit(W)) per reference Linear (q, k, v, o, gate, up, down):
    the weights bit-exact, then loss, d loss / d image and (QLoRA) every dA / dB at the bf16 tolerances of the 16-bit path."""
    nl = 2
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=nl)}
    for L in P["llama"]["layers"]:          # the codes are a function of the 16-bit checkpoint weights: both sides quantise the SAME bf16 values
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
    model.text.quantize_base(4, quant_type=quant_type, double_quant=double_quant)
    assert model.text.base4 == (quant_type, double_quant) and not (model.text.base8 or model.text.base_int8)
    parts = {"qkv_w": 3, "o_w": 1, "gu_w": 2, "down_w": 1}
    Pq = dict(P, llama=dict(P["llama"], layers=[dict(L) for L in P["llama"]["layers"]]))
    for li, L in enumerate(Pq["llama"]["layers"]):
        for k, n in parts.items():
            L[k] = N4.fake_quant_weight(L[k], quant_type, double_quant, parts=n)
            got = model.text.p["layers"][li][k]
            assert torch.equal(got.cpu().view(torch.int16), L[k].to(torch.bfloat16).view(torch.int16)), (li, k)      # the weight every product reads
            assert len(model.text.p["layers"][li][k + "q4"]) == n
            assert torch.equal(model.text.p["layers"][li][k + "T"], got.t().contiguous())
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
                assert rel(dA, Ao.grad) < 6e-2 and rel(dB, Bo.grad) < 6e-2, (l, pr)
        # peft re-quantises a merged Linear4bit: W <- Q4(dequant(W) + s B A)
        before = model.text.p["layers"][0]["o_w"].clone()
        model.text.merge_lora()
        assert model.text.lora is None and model.text.base4 == (quant_type, double_quant)
        s = lora_p[0]["scale"]
        A, Bm = lora_p[0]["o"]
        merged = (before.float().cpu() + s * (Bm.float() @ A.float())).to(torch.bfloat16)
        want = N4.fake_quant_weight(merged.float(), quant_type, double_quant).to(torch.bfloat16)
        got = model.text.p["layers"][0]["o_w"].cpu()
        # the device merge multiplies the adapters' bf16 copies and rounds once: a one-ulp difference in a block's largest element moves its 64 values
        assert (got.view(torch.int16) == want.view(torch.int16)).float().mean().item() > 0.95 and rel(got, want) < 3e-2


def test_yaml_bits4_reaches_the_4bit_storage():
    """`bits: 4`, `quant_type`, `double_quant` of the YAML surface (text_modal.py:91-107) select the storage in prepare_for_training."""
    cfg = dict(bits=4, quant_type="fp4", double_quant=False, lora=dict(enable=False), stage=1)
    nl = 1
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=nl)}
    model = UniBind(("rgb", "text"), cfg, device=DEV, llama_layers=nl).load_params(P)
    assert model.bits == 4
    model.prepare_for_training(freeze_vision=True, freeze_text=True, tune_rgb_pooler=True)
    assert model.text.base4 == ("fp4", False)
    st = model.text.p["layers"][0]["qkv_wq4"]
    assert len(st) == 3 and all(s["quant_type"] == "fp4" and "absmax" in s for s in st)
