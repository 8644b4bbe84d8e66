"""Parity of the HIP hot path against (a) the golden vectors produced by the reference's own modules and (b) the CPU
oracle on the same seeded parameters.  The engine computes in bf16 (fp32 accumulation / statistics); the reference
vectors are fp32.  Tolerances (relative L2 unless stated): activations 2e-2, loss 1e-3 relative, gradients 4e-2.
Token / label / mask indexing is bit-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from lhrs_bot_amd import kernels as hk  # noqa: E402
from lhrs_bot_amd.pooler import AttnPooler  # noqa: E402
from lhrs_bot_amd.text import TextModal  # noqa: E402
from lhrs_bot_amd.unibind import UniBind  # noqa: E402
from oracle import lhrs_oracle as O  # noqa: E402
from oracle import params as OP  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def oracle_pooler_grads(P):
    """name (the reference AttnPooler's state-dict key) -> oracle autograd gradient, for every one of the 87 projector tensors."""
    out = {}
    for name, t in OP.pooler_to_ref(P["pooler"]).items():
        base = t if t.is_leaf else t._base          # only `query` is a view ([None]) of its leaf
        out[name] = (P["pooler"]["query"].grad if name == "query" else base.grad).double()
    return out


def assert_projector_grads_directional(got, want, tol):
    """Every projector gradient by relative L2 of the DIFFERENCE (a norm alone cannot see a transposed or permuted dW)."""
    assert len(got) == 87 and set(got) == set(want)
    bad = [(n, rel(got[n].reshape(want[n].shape), want[n])) for n in got if rel(got[n].reshape(want[n].shape), want[n]) > tol]
    assert not bad, bad


def test_pooler_forward_backward_vs_reference_golden():
    z = np.load(os.path.join(G, "pooler.npz"))
    g = torch.Generator().manual_seed(int(z["input_seed"]))
    x = torch.randn(2, 768, 1024, generator=g)
    dout = torch.randn(2, 144, 4096, generator=g) * 0.01
    pool = AttnPooler(device=DEV)
    pool.load_params(OP.make_pooler_params(seed=1))
    out = pool.forward(x.to(DEV, torch.bfloat16))
    assert rel(out, torch.from_numpy(z["out"]).float()) < 2e-2
    pool.backward(dout.to(DEV, torch.bfloat16))
    torch.cuda.synchronize()
    norms = dict(zip(z["grad_names"].tolist(), z["grad_norms"].tolist()))
    bad = []
    for name, want in norms.items():
        got = pool.g[name].double().norm().item()
        if abs(got - want) > 4e-2 * want:
            bad.append((name, got, want))
    assert not bad, bad
    assert rel(pool.g["query"], torch.from_numpy(z["g_query"]).float()) < 4e-2
    assert rel(pool.g["out_proj.bias"], torch.from_numpy(z["g_out_proj_b"]).float()) < 4e-2
    assert rel(pool.g["layers.0.attn.in_proj_weight"][::64, ::64], torch.from_numpy(z["g_l0_in_w_slice"]).float()) < 4e-2
    assert rel(pool.g["layers.5.mlp.c_fc.weight"][::64, ::64], torch.from_numpy(z["g_l5_fc_w_slice"]).float()) < 4e-2
    assert rel(pool.g["layers.3.ln_1_kv.weight"], torch.from_numpy(z["g_l3_ln1kv_w"]).float()) < 4e-2


def test_splice_bit_exact_vs_reference_golden():
    z = np.load(os.path.join(G, "splice.npz"))
    NI = int(z["n_img_tokens"])
    tm = TextModal(device=DEV, layers=0)
    g = torch.Generator().manual_seed(0)
    embed = torch.randn(32000, 4096, generator=g).to(DEV, torch.bfloat16)
    tm.p = {"embed": embed}
    for name in ("uniform", "ragged_pad", "mixed_noimg", "img_last", "single"):
        ids = torch.from_numpy(z[name + "_ids"]); labels = torch.from_numpy(z[name + "_labels"])
        mask = torch.from_numpy(z[name + "_mask"])
        B = ids.shape[0]
        img = torch.randn(B, NI, 4096, generator=g).to(DEV, torch.bfloat16)
        emb, nl, nm, pos = tm.prepare_inputs_for_multimodal(ids, mask, labels, img)
        assert torch.equal(nl.cpu(), torch.from_numpy(z[name + "_new_labels"])), name
        assert torch.equal(nm.cpu().bool(), torch.from_numpy(z[name + "_new_mask"])), name
        src = torch.from_numpy(z[name + "_src"])
        src_o, _, _ = O.splice(ids, labels, mask, NI)
        want = torch.zeros_like(emb.cpu())
        for b in range(B):
            for j in range(src_o.shape[1]):
                s = int(src_o[b, j])
                if s >= 0:
                    want[b, j] = embed[int(ids[b, s])].cpu()
                elif s > -10 ** 8:
                    want[b, j] = img[b, -s - 1].cpu()
        assert torch.equal(emb.cpu(), want), name  # pure copies: bit-exact
        amb = src == -5
        assert torch.equal(src_o[~amb], src[~amb]), name


@pytest.mark.timeout(1200)
def test_unibind_end_to_end_vs_reference_golden_and_oracle():
    z = np.load(os.path.join(G, "unibind_e2e.npz"))
    nl = int(z["n_llama_layers"])
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=nl)}
    model = UniBind(("rgb", "text"), None, device=DEV, llama_layers=nl).load_params(P)
    model.prepare_for_training()
    model.text.tail_rows_only = False  # this test reads the final-norm hidden state of EVERY position
    batch = dict(rgb=torch.from_numpy(z["rgb"]).float(), input_ids=torch.from_numpy(z["input_ids"]),
                 labels=torch.from_numpy(z["labels"]), attention_mask=torch.from_numpy(z["attention_mask"]))
    taps = model.rgb.encode(batch["rgb"])
    assert rel(taps[:1, ::2], torch.from_numpy(z["vit_taps"]).float()) < 2e-2
    image = model.rgb_pooler.forward(taps, save_ctx=False)
    assert rel(image[:, ::2], torch.from_numpy(z["image"]).float()) < 2e-2
    out = model(batch)
    loss = out["total_loss"].item()
    assert abs(loss - float(z["loss"])) < 1e-3 * float(z["loss"]), (loss, float(z["loss"]))
    hid = model.text.last_hidden.reshape(2, -1, 4096)[:, ::8]          # final-norm hidden rows (reference: forward hook on LlamaModel)
    valid = torch.from_numpy(z["attention_mask"])                      # rows under the right padding are arbitrary on both sides
    S = model.text.last_hidden.shape[0] // 2
    vis = torch.cat([torch.ones(2, S - valid.shape[1], dtype=torch.bool), valid], dim=1)[:, ::8]  # the reference's spliced-mask rule
    assert rel(hid.float().cpu()[vis], torch.from_numpy(z["hidden_sample"]).float()[vis]) < 2e-2
    d_image = model.text.backward()
    assert rel(d_image[:, ::4], torch.from_numpy(z["d_image"])) < 4e-2
    model.rgb_pooler.backward(d_image)
    torch.cuda.synchronize()
    norms = dict(zip(z["grad_names"].tolist(), z["grad_norms"].tolist()))
    bad = [(n, model.rgb_pooler.g[n].double().norm().item(), w) for n, w in norms.items()
           if abs(model.rgb_pooler.g[n].double().norm().item() - w) > 5e-2 * w]
    assert not bad, bad
    assert rel(model.rgb_pooler.g["out_proj.bias"], torch.from_numpy(z["g_out_proj_b"])) < 4e-2
    assert rel(model.rgb_pooler.g["query"][::4], torch.from_numpy(z["g_query"])) < 5e-2


@pytest.mark.timeout(900)
def test_ragged_right_padded_batch_vs_oracle():
    """DataCollatorForSupervisedDataset batches: captions of different length, right-padded with pad_token_id = 0, labels -100 on the
    pads, attention_mask = ids != 0.  Loss and d loss / d image_embedding against oracle autograd; padded rows are invisible."""
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=2)}
    model = UniBind(("rgb", "text"), None, device=DEV, llama_layers=2).load_params(P)
    model.prepare_for_training()
    g = torch.Generator().manual_seed(77)
    B, T = 3, 21
    lens = [21, 13, 6]
    ids = torch.zeros((B, T), dtype=torch.int64)
    for b, n in enumerate(lens):
        ids[b, :n] = torch.randint(3, 32000, (n,), generator=g)
        ids[b, 0], ids[b, 1] = 1, -200
    labels = ids.clone()
    labels[:, :2] = -100
    labels[ids == 0] = -100
    batch = dict(rgb=torch.randn(B, 3, 224, 224, generator=g), input_ids=ids, labels=labels, attention_mask=ids.ne(0))
    loss = model(batch)["total_loss"].item()
    d_image = model.text.backward()
    # oracle: same forward with autograd through the image embedding
    taps = O.vit_forward(P["vit"], batch["rgb"])
    img = O.pooler_forward(P["pooler"], taps).detach().requires_grad_(True)
    src, lab, mask = O.splice(ids, labels, batch["attention_mask"], img.shape[1])
    emb = P["llama"]["embed"]
    tok = emb[torch.gather(ids.clamp(min=0), 1, src.clamp(min=0))]
    img_rows = img[torch.arange(B)[:, None], (-src - 1).clamp(0, img.shape[1] - 1)]
    is_img = (src < 0) & (src > -10 ** 8)
    embeds = torch.where(is_img[..., None], img_rows, tok)
    embeds = torch.where((src <= -10 ** 8)[..., None], torch.zeros_like(embeds), embeds)
    want = O.causal_lm_loss(P["llama"], O.llama_hidden(P["llama"], embeds, mask), lab)
    want.backward()
    assert abs(loss - want.item()) < 1e-3 * want.item(), (loss, want.item())
    assert rel(d_image, img.grad) < 5e-2
    # the value of the padding positions does not matter: junk ids under the mask give the same loss bit for bit
    ids2 = ids.clone()
    ids2[ids == 0] = 777
    batch2 = dict(batch, input_ids=ids2)
    assert model.eval()(batch2)["total_loss"].item() == model(batch)["total_loss"].item()


@pytest.mark.timeout(900)
def test_unibind_eight_layers_vs_reference_golden():
    """Depth check: 8 LLaMA-7B-width layers, reference fixture tests/golden/unibind_e2e_8l.npz (make_golden_deep.py)."""
    z = np.load(os.path.join(G, "unibind_e2e_8l.npz"))
    nl = int(z["n_llama_layers"])
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=nl)}
    model = UniBind(("rgb", "text"), None, device=DEV, llama_layers=nl).load_params(P)
    del P
    model.prepare_for_training()
    ids = torch.from_numpy(z["input_ids"])
    labels = ids.clone()
    labels[:, :2] = -100
    rgb = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(int(z["rgb_seed"])))
    assert abs(rgb.double().sum().item() - float(z["rgb_checksum"])) < 1e-6
    out = model(dict(rgb=rgb, input_ids=ids, labels=labels, attention_mask=ids.ne(0)))
    loss = out["total_loss"].item()
    assert abs(loss - float(z["loss"])) < 1e-3 * float(z["loss"]), (loss, float(z["loss"]))
    d_image = model.text.backward()
    assert rel(d_image[:, ::4, ::4], torch.from_numpy(z["d_image"])) < 6e-2
    model.rgb_pooler.backward(d_image)
    torch.cuda.synchronize()
    norms = dict(zip(z["grad_names"].tolist(), z["grad_norms"].tolist()))
    bad = [(n, model.rgb_pooler.g[n].double().norm().item(), w) for n, w in norms.items()
           if abs(model.rgb_pooler.g[n].double().norm().item() - w) > 6e-2 * w]
    assert not bad, bad
    assert rel(model.rgb_pooler.g["out_proj.bias"], torch.from_numpy(z["g_out_proj_b"])) < 6e-2


@pytest.mark.timeout(900)
def test_unibind_headline_shape_s273_vs_reference_golden():
    """The HEADLINE sequence length (BASELINE configs[1]: T = 130 => S = 273: the LDS-resident attention at its training shape, RoPE
    positions to 272) against the reference fixture tests/golden/unibind_e2e_s273.npz (make_golden_deep.py s273; 2 LLaMA-7B-width layers)."""
    z = np.load(os.path.join(G, "unibind_e2e_s273.npz"))
    nl = int(z["n_llama_layers"])
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=nl)}
    model = UniBind(("rgb", "text"), None, device=DEV, llama_layers=nl).load_params(P)
    model.prepare_for_training()
    model.text.tail_rows_only = False  # this test reads the final-norm hidden state of EVERY position
    ids = torch.from_numpy(z["input_ids"])
    assert ids.shape == (2, 130)
    labels = ids.clone()
    labels[:, :2] = -100
    rgb = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(int(z["rgb_seed"])))
    assert abs(rgb.double().sum().item() - float(z["rgb_checksum"])) < 1e-6
    loss = model(dict(rgb=rgb, input_ids=ids, labels=labels, attention_mask=ids.ne(0)))["total_loss"].item()
    assert abs(loss - float(z["loss"])) < 1e-3 * float(z["loss"]), (loss, float(z["loss"]))
    hid = model.text.last_hidden.reshape(2, 273, 4096)
    assert rel(hid[:, ::8, ::4], torch.from_numpy(z["hidden_sample"]).float()) < 2e-2
    d_image = model.text.backward()
    assert rel(d_image[:, ::4, ::4], torch.from_numpy(z["d_image"])) < 5e-2
    model.rgb_pooler.backward(d_image)
    torch.cuda.synchronize()
    norms = dict(zip(z["grad_names"].tolist(), z["grad_norms"].tolist()))
    bad = [(n, model.rgb_pooler.g[n].double().norm().item(), w) for n, w in norms.items()
           if abs(model.rgb_pooler.g[n].double().norm().item() - w) > 5e-2 * w]
    assert not bad, bad
    assert rel(model.rgb_pooler.g["out_proj.bias"], torch.from_numpy(z["g_out_proj_b"])) < 5e-2


@pytest.mark.timeout(3000)
def test_full_depth_32_layers_s273_vs_oracle_on_host_cpu():
    """The MEASURED workload at full depth: B = 1, S = 273, all 32 LLaMA-2-7B layers (bench.py's model, one sample), HIP path in bf16
    against the fp32 oracle run on the GPU box's HOST cores with the same seeded parameters (27 GB fp32; the oracle is pinned to the
    reference at this sequence length by tests/test_oracle_cpu.py::test_unibind_headline_shape_s273_matches_reference and at depth by the
    8-layer fixture).  Checks bf16 drift over 32 residual layers: loss 1e-3, final-norm hidden 3e-2, d loss / d image 6e-2, every one
    of the 87 projector gradients by rel-L2 of the difference, bounded like d loss / d image."""
    import gc
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    NL, T = 32, 130
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=NL)}
    model = UniBind(("rgb", "text"), None, device=DEV, llama_layers=NL).load_params(P)
    model.prepare_for_training()
    model.text.tail_rows_only = False  # this test reads the final-norm hidden state of EVERY position
    g = torch.Generator().manual_seed(3273)
    ids = torch.randint(3, 32000, (1, T), generator=g)
    ids[:, 0], ids[:, 1] = 1, -200
    labels = ids.clone()
    labels[:, :2] = -100
    batch = dict(rgb=torch.randn(1, 3, 224, 224, generator=g), input_ids=ids, labels=labels, attention_mask=ids.ne(0))
    with hk.gemm_kernel_census() as census:      # which persistent GEMM kernel the shape rules gave each product of this run (deterministic: csrc/gemm.hip)
        loss = model(batch)["total_loss"].item()
        hid = model.text.last_hidden.float().cpu().reshape(1, 273, 4096)
        d_image = model.text.backward()
        model.rgb_pooler.backward(d_image)
    torch.cuda.synchronize()
    got_grads = {n: model.rgb_pooler.g[n].double().cpu() for n, _ in model.rgb_pooler.named_parameters()}
    d_image = d_image.float().cpu()
    # ---- oracle on the host
    leaves = {}
    for k, v in P["pooler"].items():
        if torch.is_tensor(v):
            leaves[k] = v.requires_grad_(True)
    for l, L in enumerate(P["pooler"]["layers"]):
        for k, v in L.items():
            leaves[f"{l}.{k}"] = v.requires_grad_(True)
    col = {}
    want = O.unibind_forward(P, batch, col)
    col["image"].retain_grad()
    want.backward()
    assert abs(loss - want.item()) < 1e-3 * want.item(), (loss, want.item())
    # 32 residual layers forward and 32 backward: bf16 rounding accumulates (2 layers: hidden < 2e-2, d_image < 5e-2).  Yardstick = the SAME
    # oracle code run with torch-CPU bf16 tensors and autograd on the same embeddings (what the reference's own bf16 path does: one
    # rounding per op): the HIP path, which rounds once per fused kernel, must not sit further from fp32 than 1.5x that (measured: hidden 0.0428 vs 0.0349, d_image 0.0767 vs 0.0619)
    Lb = {"layers": [{k: v.bfloat16() for k, v in L.items()} for L in P["llama"]["layers"]], "norm_w": P["llama"]["norm_w"].bfloat16(),
          "lm_head": P["llama"]["lm_head"].bfloat16()}
    eb = col["embeds"].detach().bfloat16().requires_grad_(True)
    hb = O.llama_hidden(Lb, eb, col["mask"])
    O.causal_lm_loss(Lb, hb, col["labels"]).backward()
    del Lb
    pos = int((ids[0] == -200).nonzero()[0])
    yard_h = rel(hb.detach().float(), col["hidden"].detach())
    yard_g = rel(eb.grad[:, pos:pos + 144].float(), col["image"].grad)
    err_h, err_g = rel(hid, col["hidden"].detach()), rel(d_image, col["image"].grad)
    # the yardstick starts from the fp32 oracle's embeddings rounded ONCE; the end-to-end HIP numbers above also carry the bf16 error of the
    # ViT + projector in their 144 image rows.  Like for like: the HIP decoder alone on the SAME bf16-rounded oracle embeddings
    model.text.tail_rows_only = False
    model.text.decode(batch["input_ids"], image_embedding=col["image"].detach().to(DEV, torch.bfloat16), attention_mask=batch["attention_mask"],
                      labels=batch["labels"])
    hid2 = model.text.last_hidden.float().cpu().reshape(1, 273, 4096)
    d_image2 = model.text.backward().float().cpu()
    err_h2, err_g2 = rel(hid2, col["hidden"].detach()), rel(d_image2, col["image"].grad)
    msg = (f"32 layers, S=273: loss HIP {loss:.5f} / fp32 oracle {want.item():.5f}; end to end: hidden rel-L2 HIP {err_h:.4f}, d loss/d image HIP {err_g:.4f}; "
           f"decoder alone on the oracle's embeddings: hidden HIP {err_h2:.4f} vs torch-bf16 {yard_h:.4f}; d loss/d image HIP {err_g2:.4f} vs torch-bf16 {yard_g:.4f}")
    print(msg)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(__file__)), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    open(os.path.join(out_dir, "full_depth_parity.txt"), "w").write(msg + "\npersistent GEMM launches by kernel instantiation (B = 1: M = 273, below the four-wave kernel's "
                                                                   f"shape rule): {census.counts}\n")
    assert err_h < max(3e-2, 1.5 * yard_h), msg
    assert err_g < max(6e-2, 1.5 * yard_g), msg
    assert err_h2 < 1.1 * yard_h and err_g2 < 1.1 * yard_g, msg   # one rounding per fused kernel: not further from fp32 than op-by-op bf16
    # every one of the 87 projector gradients by rel-L2 of the difference (directional), not by norm.  They are linear in d loss / d image,
    # so they inherit its error after 32 + 32 bf16 layers (measured 0.073-0.075 for all 87 at d_image 0.0756): same bound as d_image
    assert_projector_grads_directional(got_grads, oracle_pooler_grads(P), max(6e-2, 1.5 * yard_g))
    del P, model, col
    gc.collect()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("ragged", [False, True])
def test_last_layer_on_supervised_rows_only_equals_every_row(ragged):
    """Training forward / backward with the last decoder layer's post-attention half restricted to the supervised rows (the default when
    they are one contiguous range per sequence) against the same step computing every row as HF does: the loss is the same number (the
    skipped rows never reach it), d loss / d image and all projector gradients agree to fp32 summation-order noise.  ragged: right-padded
    captions of different length (pad rows carry no target: the ranges differ per sequence)."""
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=3)}
    g = torch.Generator().manual_seed(91)
    B, T = 3, 40
    ids = torch.randint(3, 32000, (B, T), generator=g)
    ids[:, 0], ids[:, 1] = 1, -200
    if ragged:
        ids[1, 29:] = 0
        ids[2, 11:] = 0
    labels = ids.clone()
    labels[:, :2] = -100
    labels[ids == 0] = -100
    batch = dict(rgb=torch.randn(B, 3, 224, 224, generator=g), input_ids=ids, labels=labels, attention_mask=ids.ne(0))
    res = {}
    for mode in (True, False):
        model = UniBind(("rgb", "text"), None, device=DEV, llama_layers=3).load_params(P)
        model.prepare_for_training()
        model.text.tail_rows_only = mode
        loss = model(batch)["total_loss"].item()
        n_rows = model.text.last_hidden.shape[0]
        d_image = model.text.backward()
        model.rgb_pooler.backward(d_image)
        torch.cuda.synchronize()
        res[mode] = (loss, n_rows, d_image.float().cpu(), model.rgb_pooler.grad.clone().cpu())
    S = T - 1 + 144
    assert res[False][1] == B * S and res[True][1] == int((labels[:, 1:] != -100).sum())       # compact really ran compact
    assert abs(res[True][0] - res[False][0]) < 1e-5 * res[False][0], (res[True][0], res[False][0])
    # gradients: identical math, but dK / dV of the last layer sum the supervised queries in a different tile grouping (fp32) before the
    # bf16 store - single-ulp (2^-8) flips that two more layers carry along: measured 4e-3, far inside the 2-5e-2 both sit from fp32
    assert rel(res[True][2], res[False][2]) < 8e-3
    assert rel(res[True][3], res[False][3]) < 8e-3
    # a batch whose supervised positions are NOT one range per sequence (multi-turn labels) silently takes the every-row path
    labels2 = labels.clone()
    labels2[:, 10:14] = -100
    model.text.tail_rows_only = True
    model(dict(batch, labels=labels2))
    assert model.text.last_hidden.shape[0] == B * S


@pytest.mark.timeout(3000)
@pytest.mark.parametrize("B", [8, 30, 120, 240])
def test_measured_micro_batches_end_to_end_vs_oracle(B):
    """The micro-batches bench.py measures - 8 (Script/train_stage1.sh:11), 30 (the bench default of rounds 1-4), 120 and 240 (the default since the end of round 6: M = 65520 = 256 tile rows;
    60, the default of rounds 5-6, ran here through round 6: profiles/r06_gpu_suite_final.txt) - at the
    headline sequence length S = 273 with 2 decoder layers, default engine settings (persistent 256x256 GEMM at M = 2184 / 8190 with its
    tail-row rule, fused RoPE / SwiGLU epilogues, last layer on the supervised rows; 60 = the B = 60 line of DESIGN §4.1: 64 tile rows):
    loss, the ViT taps of EVERY sample, d loss / d image and all 87 projector gradients - each by rel-L2 of the difference, so a
    transposed or permuted dW cannot hide behind a matching norm - against oracle autograd (fp32, host cores) on the same seeded parameters."""
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=2)}
    model = UniBind(("rgb", "text"), None, device=DEV, llama_layers=2).load_params(P)
    model.prepare_for_training()
    g = torch.Generator().manual_seed(8000 + B)
    T = 130
    ids = torch.randint(3, 32000, (B, T), generator=g)
    ids[:, 0], ids[:, 1] = 1, -200
    labels = ids.clone()
    labels[:, :2] = -100
    batch = dict(rgb=torch.randn(B, 3, 224, 224, generator=g), input_ids=ids, labels=labels, attention_mask=ids.ne(0))
    with hk.gemm_kernel_census() as census:
        loss = model(batch)["total_loss"].item()
        d_image = model.text.backward()
        model.rgb_pooler.backward(d_image)
    torch.cuda.synchronize()
    # the shipped kernel choice is a shape rule, so it can be asserted: at micro-batch 30 / 60 the decoder products of the full layers run all five instantiations of
    # the four-wave gemm_u4_kernel (plain, plain + residual, RoPE, SwiGLU forward / backward; the compact last layer's MLP - supervised rows only - may fall under the
    # rule's thresholds); at micro-batch 8 (M = 2184) RoPE and SwiGLU' stay on the 16-wave kernel
    u4 = {k: v for k, v in census.counts.items() if k.startswith("gemm_u4_kernel")}
    if B >= 30:   # SwiGLU backward joins at micro-batch 60 only (its rule wants <= 5 % of the last round idle: 5.375 rounds at 30, 10.75 at 60)
        assert len(u4) == (5 if B >= 60 else 4) and all(v > 0 for v in u4.values()), census.counts
    else:
        # M = 2184: RoPE (1.69 rounds) and SwiGLU' (1.51) stay on the 16-wave kernel; gate|up's whole tile rows behind the tail-row cut (2.69 rounds) may take the four-wave one
        assert not any(k.startswith("gemm_u4_kernel<2") or k.startswith("gemm_u4_kernel<3") for k in u4), census.counts
    out_dir = os.path.join(os.path.dirname(os.path.dirname(__file__)), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    open(os.path.join(out_dir, f"parity_micro_batch_{B}_kernels.txt"), "w").write(f"micro-batch {B}, 2 layers: persistent GEMM launches by kernel instantiation: {census.counts}\n")
    got = {n: model.rgb_pooler.g[n].double().cpu() for n, _ in model.rgb_pooler.named_parameters()}
    d_image = d_image.float().cpu()
    for k, v in P["pooler"].items():
        if torch.is_tensor(v):
            v.requires_grad_(True)
    for L in P["pooler"]["layers"]:
        for v in L.values():
            v.requires_grad_(True)
    col = {}
    want = O.unibind_forward(P, batch, col)
    col["image"].retain_grad()
    want.backward()
    assert abs(loss - want.item()) < 1e-3 * want.item(), (loss, want.item())
    assert rel(d_image, col["image"].grad) < 5e-2
    assert rel(model.rgb.encode(batch["rgb"]), col["taps"].detach()) < 2e-2          # all B samples, all 768 tap tokens
    want_grads = oracle_pooler_grads(P)
    assert_projector_grads_directional(got, want_grads, 5e-2)
    # The 5e-2 above is the bf16 error of d loss / d image (two decoder layers backward) carried into every projector gradient - a transposed block of a
    # few per cent in ONE dW could hide under it.  So the projector's own backward (dW = dY^T X by gemm_tn_f32_kernel with fp32 accumulation into the fp32
    # gradient store, dX through the six blocks) is also held on its own: the SAME saved forward, fed the ORACLE's d loss / d image (rounded to bf16 once),
    # every one of the 87 gradients to 2e-2 by rel-L2 of the difference (round 6).
    model.rgb_pooler.grad.zero_()
    model.rgb_pooler.forward(model.rgb.encode(batch["rgb"]), save_ctx=True)
    model.rgb_pooler.backward(col["image"].grad.to(device=DEV, dtype=torch.bfloat16).contiguous())
    torch.cuda.synchronize()
    own = {n: model.rgb_pooler.g[n].double().cpu() for n, _ in model.rgb_pooler.named_parameters()}
    errs = {n: rel(own[n].reshape(want_grads[n].shape), want_grads[n]) for n in own}
    open(os.path.join(out_dir, f"parity_micro_batch_{B}_projector_backward_alone.txt"), "w").write(
        f"micro-batch {B}: projector backward on the oracle's d_image: max rel-L2 over 87 gradients {max(errs.values()):.4f} ({max(errs, key=errs.get)}), median {sorted(errs.values())[43]:.4f}\n")
    assert_projector_grads_directional(own, want_grads, 2e-2)


@pytest.mark.timeout(900)
def test_module_level_layer_entry_points_equal_the_operator_path(monkeypatch):
    """lhrs_llama_layer_forward / lhrs_llama_layer_backward (one library call per decoder layer: include/lhrs_hip.h) against the
    operator-by-operator path of text.py on the same batch: the same launches in the same order, so loss, d loss / d image and every projector
    gradient are bit-identical; the compact last layer (supervised rows only) stays on the operator path in both runs."""
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=3)}
    g = torch.Generator().manual_seed(404)
    B, T = 3, 40
    ids = torch.randint(3, 32000, (B, T), generator=g)
    ids[:, 0], ids[:, 1] = 1, -200
    ids[2, 31:] = 0                                                  # one right-padded sequence: key counts differ per sequence
    labels = ids.clone()
    labels[:, :2] = -100
    labels[ids == 0] = -100
    batch = dict(rgb=torch.randn(B, 3, 224, 224, generator=g), input_ids=ids, labels=labels, attention_mask=ids.ne(0))
    res = {}
    for native in ("1", "0"):
        monkeypatch.setenv("LHRS_NATIVE_LAYER", native)
        model = UniBind(("rgb", "text"), None, device=DEV, llama_layers=3).load_params(P)
        model.prepare_for_training()
        assert model.text._native_layer(None) == (native == "1")
        loss = model(batch)["total_loss"]
        d_image = model.text.backward()
        model.rgb_pooler.backward(d_image)
        torch.cuda.synchronize()
        res[native] = (loss.clone(), d_image.clone(), model.rgb_pooler.grad.clone())
    assert torch.equal(res["1"][0], res["0"][0])
    assert torch.equal(res["1"][1], res["0"][1])
    assert torch.equal(res["1"][2], res["0"][2])
