"""CPU: pin oracle/ (the restatement that travels to the GPU box) against the golden vectors produced by the
REFERENCE's own modules (tests/golden/make_golden.py).  fp32 vs fp32: tolerances are fp32-roundoff level."""
import os

import numpy as np
import pytest
import torch

from oracle import lhrs_oracle as O
from oracle import params as OP
from oracle.optim_oracle import adamw_step_ref

G = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def pooler_inputs(seed):
    g = torch.Generator().manual_seed(int(seed))
    x = torch.randn(2, 768, 1024, generator=g)
    dout = torch.randn(2, 144, 4096, generator=g) * 0.01
    return x, dout


def test_pooler_matches_reference_module():
    z = np.load(os.path.join(G, "pooler.npz"))
    p = OP.make_pooler_params(seed=1)
    leaves = []

    def req(d):
        for k, v in d.items():
            if isinstance(v, list):
                for e in v:
                    req(e)
            else:
                v.requires_grad_(True)
                leaves.append(v)

    req(p)
    x, dout = pooler_inputs(z["input_seed"])
    x.requires_grad_(True)
    out = O.pooler_forward(p, x)
    assert rel(out.detach().half(), torch.from_numpy(z["out"])) < 1e-3  # golden stored as fp16
    assert (out.detach() - torch.from_numpy(z["out"]).float()).abs().max() < 2e-2
    out.backward(dout)
    assert rel(x.grad[:, ::8], torch.from_numpy(z["dx_rows"]).float()) < 2e-3
    ref_sd = OP.pooler_to_ref(p)
    norms = dict(zip(z["grad_names"].tolist(), z["grad_norms"].tolist()))
    # in_proj / fused tensors are views in pooler_to_ref, so gradients live on the engine-layout leaves
    for name, want in norms.items():
        t = ref_sd[name]
        base = t if t.grad is not None or t.is_leaf else t._base
        gsrc = base.grad if base is not None and base.grad is not None else None
        if name == "query":
            got = p["query"].grad.norm().item()
        else:
            assert gsrc is not None, name
            got = gsrc.norm().item()
        assert abs(got - want) <= 2e-4 * max(want, 1e-12), (name, got, want)
    assert rel(p["query"].grad, torch.from_numpy(z["g_query"]).float()) < 2e-3
    assert rel(p["layers"][0]["in_w"].grad[::64, ::64], torch.from_numpy(z["g_l0_in_w_slice"]).float()) < 2e-3
    assert rel(p["layers"][3]["ln1kv_w"].grad, torch.from_numpy(z["g_l3_ln1kv_w"]).float()) < 2e-3


def test_splice_bit_exact_against_reference():
    z = np.load(os.path.join(G, "splice.npz"))
    NI = int(z["n_img_tokens"])
    for name in ("uniform", "ragged_pad", "mixed_noimg", "img_last", "single"):
        ids = torch.from_numpy(z[name + "_ids"]); labels = torch.from_numpy(z[name + "_labels"])
        mask = torch.from_numpy(z[name + "_mask"])
        src, nl, nm = O.splice(ids, labels, mask, NI)
        want_src = torch.from_numpy(z[name + "_src"])
        # padded ids (0) repeat within a row: the golden marks those rows ambiguous (-5); they must copy id 0
        amb = want_src == -5
        assert torch.equal(src[~amb], want_src[~amb]), name
        assert torch.all(torch.gather(ids, 1, src.clamp(min=0))[amb] == 0), name
        assert torch.equal(nl, torch.from_numpy(z[name + "_new_labels"])), name
        assert torch.equal(nm, torch.from_numpy(z[name + "_new_mask"])), name


@pytest.mark.timeout(900)
def test_unibind_end_to_end_matches_reference():
    z = np.load(os.path.join(G, "unibind_e2e.npz"))
    nl = int(z["n_llama_layers"])
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=nl)}
    for L in [P["pooler"]] + P["pooler"]["layers"]:
        for k, v in L.items():
            if torch.is_tensor(v):
                v.requires_grad_(True)
    batch = dict(rgb=torch.from_numpy(z["rgb"]).float(), input_ids=torch.from_numpy(z["input_ids"]),
                 labels=torch.from_numpy(z["labels"]), attention_mask=torch.from_numpy(z["attention_mask"]))
    col = {}
    loss = O.unibind_forward(P, batch, col)
    col["image"].retain_grad()
    loss.backward()
    # rgb was stored as fp16, which is exactly what both sides consumed? no: the reference consumed fp32 -> allow 1e-3
    assert abs(loss.item() - float(z["loss"])) < 2e-3 * float(z["loss"])
    assert rel(col["taps"][:1, ::2].detach(), torch.from_numpy(z["vit_taps"]).float()) < 3e-3
    assert rel(col["image"][:, ::2].detach(), torch.from_numpy(z["image"]).float()) < 3e-3
    assert rel(col["hidden"][:, ::8].detach(), torch.from_numpy(z["hidden_sample"]).float()) < 3e-3
    assert rel(col["image"].grad[:, ::4], torch.from_numpy(z["d_image"])) < 1e-2
    norms = dict(zip(z["grad_names"].tolist(), z["grad_norms"].tolist()))
    assert abs(P["pooler"]["out_proj_b"].grad.norm().item() - norms["out_proj.bias"]) < 1e-2 * norms["out_proj.bias"]
    assert abs(P["pooler"]["query"].grad.norm().item() - norms["query"]) < 1e-2 * norms["query"]
    assert rel(P["pooler"]["out_proj_b"].grad, torch.from_numpy(z["g_out_proj_b"])) < 1e-2


@pytest.mark.timeout(900)
def test_unibind_headline_shape_s273_matches_reference():
    """The HEADLINE sequence (BASELINE configs[1]: T = 130 => S = 273), 2 LLaMA-7B-width layers, reference fixture
    tests/golden/unibind_e2e_s273.npz (make_golden_deep.py s273): loss, final-norm hidden rows, d loss / d image, projector grad norms."""
    z = np.load(os.path.join(G, "unibind_e2e_s273.npz"))
    nl = int(z["n_llama_layers"])
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=nl)}
    for L in [P["pooler"]] + P["pooler"]["layers"]:
        for k, v in L.items():
            if torch.is_tensor(v):
                v.requires_grad_(True)
    ids = torch.from_numpy(z["input_ids"])
    assert ids.shape == (2, 130)
    labels = ids.clone()
    labels[:, :2] = -100
    rgb = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(int(z["rgb_seed"])))
    assert abs(rgb.double().sum().item() - float(z["rgb_checksum"])) < 1e-6
    col = {}
    loss = O.unibind_forward(P, dict(rgb=rgb, input_ids=ids, labels=labels, attention_mask=ids.ne(0)), col)
    col["image"].retain_grad()
    loss.backward()
    assert col["hidden"].shape[1] == 273
    assert abs(loss.item() - float(z["loss"])) < 1e-4 * float(z["loss"])
    assert rel(col["hidden"][:, ::8, ::4].detach(), torch.from_numpy(z["hidden_sample"]).float()) < 2e-3  # fp16 storage
    assert rel(col["image"].grad[:, ::4, ::4], torch.from_numpy(z["d_image"])) < 2e-3
    norms = dict(zip(z["grad_names"].tolist(), z["grad_norms"].tolist()))
    for name, leaf in (("out_proj.bias", P["pooler"]["out_proj_b"]), ("query", P["pooler"]["query"]),
                       ("layers.2.mlp.c_fc.weight", P["pooler"]["layers"][2]["fc_w"])):
        assert abs(leaf.grad.norm().item() - norms[name]) < 2e-3 * norms[name], name
    assert rel(P["pooler"]["out_proj_b"].grad, torch.from_numpy(z["g_out_proj_b"])) < 2e-3


def test_adamw_restatement_matches_torch():
    n = 1000
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(n, generator=g, dtype=torch.float64)
    w = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([w], lr=1e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    st = dict(p=p0.clone(), m=torch.zeros(n, dtype=torch.float64), v=torch.zeros(n, dtype=torch.float64))
    for step in range(1, 6):
        grad = torch.randn(n, generator=g, dtype=torch.float64)
        w.grad = grad.clone()
        opt.step()
        adamw_step_ref(st, grad, step, lr=1e-3, wd=0.1)
        assert (w.detach() - st["p"]).abs().max() < 1e-12


def test_batched_left_padded_generate_matches_reference():
    """§8 f-3: left-padded prompts + attention_mask through splice mask rule -> masked causal LLaMA, teacher-forced steps."""
    Z = np.load(os.path.join(G, "batch_generate.npz"))
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1),
         "llama": OP.make_llama_params(seed=3, layers=int(Z["n_llama_layers"]))}
    rgb = torch.randn(3, 3, 224, 224, generator=torch.Generator().manual_seed(int(Z["rgb_seed"])))
    assert abs(rgb.double().sum().item() - float(Z["rgb_checksum"])) < 1e-6
    ids, mask = torch.from_numpy(Z["input_ids"]), torch.from_numpy(Z["attention_mask"])
    _, _, new_mask = O.splice(ids, None, mask, 144)
    assert np.array_equal(new_mask.numpy(), Z["new_mask"])      # the reference's mask rule on LEFT-padded rows, bit exact
    got = O.generate_logits(P, rgb, ids, torch.from_numpy(Z["forced_tokens"]), attention_mask=mask)
    want = torch.from_numpy(Z["logits"])
    sub = got[:, :, torch.from_numpy(Z["logits_cols"])]
    assert sub.shape == want.shape
    assert ((sub - want).norm() / want.norm()).item() < 2e-4
    clear = torch.from_numpy(Z["top2_margin"]) > 1e-3
    assert torch.equal(got.argmax(-1)[clear], torch.from_numpy(Z["argmax"])[clear])


def test_clip_image_preprocess_bit_exact_against_pillow_and_hf():
    """§8 f-2: Pillow BICUBIC resize (8-bit, two passes) + HF CLIPImageProcessor arithmetic, uint8 and float32 bit-exact."""
    import hashlib
    from image_cases import CASES, make_image
    from oracle import image_oracle as IO
    Z = np.load(os.path.join(G, "clip_preprocess.npz"))
    assert [tuple(c) for c in Z["cases"]] == CASES
    for i, (h, w) in enumerate(CASES):
        crop, pv = IO.clip_preprocess(make_image(100 + i, h, w))
        assert np.array_equal(crop[::9, ::9], Z[f"u8_sample_{i}"]) and np.array_equal(pv[:, ::9, ::9], Z[f"f32_sample_{i}"])
        assert hashlib.sha256(np.ascontiguousarray(crop).tobytes()).hexdigest() == str(Z[f"u8_sha_{i}"])
        assert hashlib.sha256(np.ascontiguousarray(pv).tobytes()).hexdigest() == str(Z[f"f32_sha_{i}"])


def test_llm_int8_restatement_properties():
    """oracle/int8_oracle.py (bitsandbytes LLM.int8, PARITY UNPINNED - bitsandbytes is absent): the algorithm's own invariants.
    (i) without outliers the product is the int8 x int8 one: every output within the vector-wise quantisation bound of the exact
    product; (ii) a column holding an outlier (>= 6.0) is taken out of the int8 path and multiplied in full precision by the DEquantised
    weight column, so scaling that column by 1e3 changes nothing else; (iii) dX = dY . dequant(W) exactly; (iv) rows quantise to
    +-127 at their absmax."""
    from oracle import int8_oracle as I8
    g = torch.Generator().manual_seed(0)
    W = torch.randn(96, 256, generator=g) * 0.02
    x = torch.randn(40, 256, generator=g).clamp_(-5.5, 5.5)
    w8 = I8.Int8Weight(W)
    assert int(w8.cb.abs().max()) == 127 and torch.all(w8.cb.abs().amax(dim=1) == 127)
    wd = I8.dequantize_rows_int8(w8.cb, w8.scb)
    assert (wd - W).abs().max() <= W.abs().amax(dim=1).max() / 254 + 1e-9
    y = I8.linear(x, w8)
    exact = x @ wd.t()
    bound = (x.abs().amax(dim=1)[:, None] / 254) * wd.abs().sum(dim=1)[None, :]      # |x - dq(q(x))| <= absmax/254 per element
    assert torch.all((y - exact).abs() <= bound + 1e-5)
    assert rel(y, x @ W.t()) < 2e-2
    # (ii) outlier column
    x2 = x.clone()
    x2[3, 17] = 9.0
    y2 = I8.linear(x2, w8)
    keep = torch.ones(256, dtype=torch.bool)
    keep[17] = False
    w_rest = I8.Int8Weight(W)
    w_rest.cb, w_rest.scb = w8.cb[:, keep].contiguous(), w8.scb
    want = I8.linear(x2[:, keep], w_rest) + x2[:, 17:18] @ wd[:, 17:18].t()
    assert torch.allclose(y2, want, atol=1e-5)
    # (iii) backward
    xg = x.clone().requires_grad_(True)
    dy = torch.randn(40, 96, generator=g)
    I8.linear(xg, w8).backward(dy)
    assert torch.allclose(xg.grad, dy @ wd, atol=1e-6)


def test_llama_oracle_with_int8_base_stays_close_to_fp32():
    from oracle import int8_oracle as I8
    P = OP.make_llama_params(seed=3, layers=2)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 24, 4096, generator=g)
    with torch.no_grad():
        h32 = O.llama_hidden(P, x, None)
        h8 = O.llama_hidden(I8.int8_llama_params(P), x, None)
    assert 1e-3 < rel(h8, h32) < 8e-2   # measured 4.3e-2 on N(0,1) activations: what vector-wise int8 costs two decoder layers


# ------------------------------------------------------------------------------------------------- LoRA pinned to the reference LLaMA
def lora_case(name):
    """Fixture + the seeded parameters / adapters / batch it was made from (tests/golden/make_golden_lora.py).  Shared with test_lora_gpu.py."""
    z = np.load(os.path.join(G, name))
    nl, r, alpha, targets = int(z["n_llama_layers"]), int(z["r"]), float(z["alpha"]), tuple(str(t) for t in z["targets"])
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=nl)}
    lora = OP.make_lora_params(seed=int(z["adapter_seed"]), layers=nl, r=r, alpha=alpha, targets=targets)
    g = torch.Generator().manual_seed(int(z["batch_seed"]))
    B, T = 2, 20
    ids = torch.randint(3, 32000, (B, T), generator=g)
    ids[:, 0] = 1
    ids[:, 1] = -200
    ids[1, T - 4:] = 0
    labels = ids.clone()
    labels[:, :2] = -100
    labels[ids == 0] = -100
    batch = dict(rgb=torch.randn(B, 3, 224, 224, generator=g), input_ids=ids, labels=labels, attention_mask=ids.ne(0))
    return z, P, lora, batch, targets


@pytest.mark.parametrize("name", ["lora_r8_qkvo.npz", "lora_r128_all.npz"])
def test_lora_restatement_pinned_to_the_reference_llama_with_merged_weights(name):
    """oracle._lora (y = W x + s B A x, UN-merged) against the reference's own LLaMA carrying W' = W + s B A: same loss, hidden states and
    d loss / d image; the oracle's autograd dA, dB against s B^T dW' and s dW' A^T formed from the reference's dW' (SURVEY §8 a7)."""
    z, P, lora, batch, targets = lora_case(name)
    for L, lo in zip(P["llama"]["layers"], lora):
        L["lora"] = {"scale": lo["scale"], **{pr: (lo[pr][0].clone().requires_grad_(True), lo[pr][1].clone().requires_grad_(True)) for pr in targets}}
    P["pooler"]["out_proj_b"].requires_grad_(True)            # makes the image embedding part of the graph (d loss / d image)
    col = {}
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    loss = O.unibind_forward(P, batch, collect=col)
    col["image"].retain_grad()
    loss.backward()
    assert abs(loss.item() - float(z["loss"])) < 2e-5 * float(z["loss"])
    assert rel(col["hidden"][:, ::4], z["hidden_sample"].astype(np.float32)) < 2e-3      # fixture stored in fp16
    assert rel(col["image"].grad[:, ::4], z["d_image"]) < 2e-4
    full = int(z["r"]) <= 16
    for l, L in enumerate(P["llama"]["layers"]):
        for pr in targets:
            A, B = L["lora"][pr]
            assert abs(A.grad.norm().item() - float(z[f"dA_norm.{l}.{pr}"])) < 2e-4 * float(z[f"dA_norm.{l}.{pr}"]), (l, pr)
            assert abs(B.grad.norm().item() - float(z[f"dB_norm.{l}.{pr}"])) < 2e-4 * float(z[f"dB_norm.{l}.{pr}"]), (l, pr)
            assert rel(A.grad if full else A.grad[::16, ::16], z[f"dA.{l}.{pr}"]) < 5e-4, (l, pr)
            assert rel(B.grad if full else B.grad[::16, ::16], z[f"dB.{l}.{pr}"]) < 5e-4, (l, pr)


# ---- 4-bit storage (bits: 4): oracle/nf4_oracle.py against the published constructions ------------------------------------------------------
def test_nf4_table_is_the_published_normal_float_construction():
    """QLoRA Appendix E / bitsandbytes functional.create_normal_map(offset=0.9677083): 8 positive quantiles norm.ppf(linspace(offset, 0.5, 9)[:-1]),
    7 negative ones -norm.ppf(linspace(offset, 0.5, 8)[:-1]), an exact 0, normalised by the largest.  The hard-coded NF4 levels of the oracle
    (and of quant4.hip) must be that table, the decision thresholds its midpoints; FP4: sign + {0, 1/192, 1/6, 1/4, 1/3, 1/2, 2/3, 1}."""
    from scipy.stats import norm
    from oracle import nf4_oracle as N4
    off = 0.9677083
    pos = norm.ppf(torch.linspace(off, 0.5, 9)[:-1].numpy()).tolist()
    neg = (-norm.ppf(torch.linspace(off, 0.5, 8)[:-1].numpy())).tolist()
    table = np.sort(np.asarray(pos + [0.0] + neg, dtype=np.float64))
    table /= table.max()
    assert table.shape == (16,) and np.abs(table - N4.NF4_LEVEL).max() < 2e-6
    assert N4.NF4_LEVEL[7] == 0.0 and N4.NF4_LEVEL[0] == -1.0 and N4.NF4_LEVEL[15] == 1.0
    assert np.abs((N4.NF4_LEVEL[:-1].astype(np.float64) + N4.NF4_LEVEL[1:]) / 2 - N4.NF4_THR).max() < 1e-7
    mags = np.sort(N4.FP4_MAG)
    assert np.allclose(mags, [0, 1 / 192, 1 / 6, 1 / 4, 1 / 3, 1 / 2, 2 / 3, 1], atol=1e-7)
    assert np.allclose((mags[:-1] + mags[1:]) / 2, N4.FP4_THR, atol=1e-6) and np.array_equal(N4.FP4_MAG[N4.FP4_CODE], mags)
    dyn = N4.dynamic_map()
    assert dyn.shape == (256,) and np.all(np.diff(dyn) > 0) and dyn[-1] == 1.0 and 0.0 in dyn
    assert dyn[127] == 0.0 and np.allclose(dyn[0:127][::-1], -dyn[128:255], rtol=1e-6)   # 127 mirrored pairs around 0, plus 0 and the lone 1.0


@pytest.mark.parametrize("quant_type", ["nf4", "fp4"])
@pytest.mark.parametrize("double_quant", [False, True])
def test_nf4_oracle_blocks_round_trip_properties(quant_type, double_quant):
    """Size-independent properties of quantize_4bit / dequantize_4bit: the packing order, +-absmax reproduced exactly, the error bound of the
    coarsest interval, the nested statistics within the 8-bit table's resolution, all-zero blocks."""
    from oracle import nf4_oracle as N4
    rng = np.random.default_rng(5)
    w = torch.from_numpy((rng.standard_normal((64, 1024)) * 0.02).astype(np.float32)).to(torch.bfloat16).float().numpy()
    w[2, 128:192] = 0.0
    st = N4.quantize_4bit(w, quant_type, double_quant)
    assert st["packed"].dtype == np.uint8 and st["packed"].shape == (w.size // 2,)
    x = w.reshape(-1, 64)
    am = np.abs(x).max(1)
    first = N4._q4((x[:, 0] / np.where(am > 0, am, 1)).astype(np.float32), quant_type)
    ok = am > 0
    assert np.array_equal((st["packed"].reshape(-1, 32)[:, 0] >> 4)[ok], first[ok])                 # element 0 of a block: HIGH nibble of byte 0
    a = N4.absmax_of(st)
    if double_quant:
        assert st["qabsmax"].shape == am.shape and st["absmax2"].shape == ((am.size + 255) // 256,)
        assert np.abs(a - am).max() <= 0.008 * np.abs(am - st["offset"]).max() + 1e-12              # coarsest step of the dynamic table: 1.4 % of absmax2 (half of it)
    else:
        assert np.array_equal(a, am)
    d = N4.dequantize_4bit(st)
    assert d.shape == w.shape
    db = d.reshape(-1, 64)
    i = np.abs(x).argmax(1)
    r = np.arange(x.shape[0])
    assert np.allclose(np.abs(db[r, i])[ok], a[ok])                                                         # the block's largest element -> level +-1
    assert np.array_equal(np.sign(db[r, i])[ok], np.sign(x[r, i])[ok])
    half_gap = 0.5 * np.diff(np.sort(N4.NF4_LEVEL if quant_type == "nf4" else N4.FP4_LEVEL)).max()
    assert (np.abs(db - x * (a / np.where(am > 0, am, 1))[:, None])[ok] <= (half_gap + 1e-6) * a[ok, None]).all()
    if not double_quant:
        assert np.all(db[~ok] == 0)                                    # all-zero block: 0 * inf = NaN -> code 0 -> -1 * absmax = -0
    else:
        code0 = -1.0 if quant_type == "nf4" else 0.0                   # ... and with nested statistics absmax comes back as the 8-bit table's rounding error
        assert np.all(db[~ok] == code0 * a[~ok, None])


MULTI_CASES = ("two_each", "two_and_one", "three_none_one", "adjacent_and_last", "four_in_one", "ims_one_each", "ims_ragged")


def test_multi_image_and_im_start_splice_bit_exact_against_reference():
    """The general walk of prepare_inputs_for_multimodal (several placeholders per sample, the batch-wide slot counter, the tune_im_start branch)
    against tests/golden/splice_multi.npz, produced by the reference's own method (tests/golden/make_golden_splice_multi.py)."""
    z = np.load(os.path.join(G, "splice_multi.npz"))
    NI = int(z["n_img_tokens"])
    for name in MULTI_CASES:
        ids = torch.from_numpy(z[name + "_ids"]); labels = torch.from_numpy(z[name + "_labels"]); mask = torch.from_numpy(z[name + "_mask"])
        src, nl, nm, slots = O.splice_multi(ids, labels, mask, NI, tune_im_start=bool(z[name + "_tune_im_start"]))
        assert slots == int(z[name + "_slots"]), name
        want_src = torch.from_numpy(z[name + "_src"])
        amb = want_src == -2 * 10 ** 9                          # repeated id in the row (padding zeros): must copy id 0
        assert torch.equal(src[~amb], want_src[~amb]), name
        assert torch.all(torch.gather(ids, 1, src.clamp(min=0))[amb] == 0), name
        assert torch.equal(nl, torch.from_numpy(z[name + "_new_labels"])), name
        assert torch.equal(nm, torch.from_numpy(z[name + "_new_mask"])), name
    # one placeholder per sample: the general walk is the single-image one with slot = sample index
    z1 = np.load(os.path.join(G, "splice.npz"))
    n1 = int(z1["n_img_tokens"])
    for name in ("uniform", "ragged_pad", "mixed_noimg", "img_last", "single"):
        ids = torch.from_numpy(z1[name + "_ids"]); labels = torch.from_numpy(z1[name + "_labels"]); mask = torch.from_numpy(z1[name + "_mask"])
        src, nl, nm = O.splice(ids, labels, mask, n1)
        srcm, nlm, nmm, slots = O.splice_multi(ids, labels, mask, n1)
        is_img = (src < 0) & (src > -10 ** 8)
        b = torch.arange(ids.shape[0])[:, None].expand_as(src)
        assert slots == ids.shape[0] and torch.equal(nl, nlm) and torch.equal(nm, nmm)
        assert torch.equal(srcm[~is_img], src[~is_img]) and torch.equal(srcm[is_img], (src - b * n1)[is_img])
