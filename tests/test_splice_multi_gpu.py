"""Several <image> placeholders per sample (the `while image_token_indices.numel() > 0` walk of TextModal.prepare_inputs_for_multimodal,
lhrs/models/text_modal.py:341-438, with its batch-wide image slot counter): host plan + device copies against tests/golden/splice_multi.npz (the
reference's own method, tests/golden/make_golden_splice_multi.py) - integers and copied rows bit-exact - and a multi-image training step end
to end against the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from lhrs_bot_amd import kernels as hk  # noqa: E402
from lhrs_bot_amd.text import TextModal  # noqa: E402
from lhrs_bot_amd.unibind import UniBind  # noqa: E402
from oracle import lhrs_oracle as O  # noqa: E402
from oracle import params as OP  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"
CASES = ("two_each", "two_and_one", "three_none_one", "adjacent_and_last", "four_in_one")
AMB = -2 * 10 ** 9


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def test_multi_image_splice_bit_exact_vs_reference_golden():
    z = np.load(os.path.join(G, "splice_multi.npz"))
    NI = int(z["n_img_tokens"])
    tm = TextModal(device=DEV, layers=0)
    g = torch.Generator().manual_seed(0)
    embed = torch.randn(32000, 4096, generator=g).to(DEV, torch.bfloat16)
    tm.p = {"embed": embed}
    for name in CASES:
        ids = torch.from_numpy(z[name + "_ids"]); labels = torch.from_numpy(z[name + "_labels"]); mask = torch.from_numpy(z[name + "_mask"])
        n_slots = int(z[name + "_slots"])
        img = torch.randn(n_slots, NI, 4096, generator=g).to(DEV, torch.bfloat16)
        emb, nl, nm, inv = tm.prepare_inputs_for_multimodal(ids, mask, labels, img)
        assert torch.equal(nl.cpu(), torch.from_numpy(z[name + "_new_labels"])), name
        assert torch.equal(nm.cpu().bool(), torch.from_numpy(z[name + "_new_mask"])), name
        want_src = torch.from_numpy(z[name + "_src"])
        # the reference's rows, by what they copy: token index -> embedding row of that id, -(1 + slot * NI + k) -> image row, PAD -> zeros
        flat = img.reshape(-1, 4096).cpu()
        want = torch.zeros_like(emb.cpu())
        B, S = want_src.shape
        for b in range(B):
            for j in range(S):
                s = int(want_src[b, j])
                if s >= 0:
                    want[b, j] = embed[int(ids[b, s])].cpu()
                elif s == AMB:
                    want[b, j] = embed[0].cpu()
                elif s > -10 ** 8:
                    want[b, j] = flat[-s - 1]
        assert torch.equal(emb.cpu(), want), name                           # pure copies: bit-exact
        # backward: every image row's gradient is the row of d_embeds where it was placed; slots nobody took (a placeholder-free sample's) get zeros
        d_emb = torch.randn(B, S, 4096, generator=g).to(DEV, torch.bfloat16)
        d_img = hk.splice_map_bwd(d_emb, inv, n_slots, NI).cpu().reshape(-1, 4096)
        want_d = torch.zeros_like(d_img)
        for b in range(B):
            for j in range(S):
                s = int(want_src[b, j])
                if -10 ** 8 < s < 0:
                    want_d[-s - 1] = d_emb[b, j].cpu()
        assert torch.equal(d_img, want_d), name
    # too few image slots: the reference fails with an IndexError at image_embedding[cur_image_idx]
    ids = torch.from_numpy(z["three_none_one_ids"])
    with pytest.raises(IndexError, match="image slots"):
        tm.prepare_inputs_for_multimodal(ids, ids.ne(0), None, torch.zeros(4, NI, 4096, device=DEV, dtype=torch.bfloat16))


def test_host_plan_equals_the_single_image_device_splice():
    """One placeholder per sample: the general host walk and the device kernel of the common path agree on every integer and every row."""
    z = np.load(os.path.join(G, "splice.npz"))
    NI = int(z["n_img_tokens"])
    g = torch.Generator().manual_seed(1)
    embed = torch.randn(32000, 256, generator=g).to(DEV, torch.bfloat16)
    for name in ("uniform", "ragged_pad", "mixed_noimg", "img_last", "single"):
        ids = torch.from_numpy(z[name + "_ids"]); labels = torch.from_numpy(z[name + "_labels"]); mask = torch.from_numpy(z[name + "_mask"])
        B = ids.shape[0]
        img = torch.randn(B, NI, 256, generator=g).to(DEV, torch.bfloat16)
        plan = TextModal.splice_plan_host(ids, labels, mask, NI)
        assert plan["n_slots"] == B
        emb1, nl, nm, pos = hk.splice_fwd(ids.to(DEV), labels.to(DEV), mask.to(DEV), img, embed, plan["S"])
        emb2 = hk.splice_map_fwd(ids.to(DEV), plan["src_tok"].to(DEV), plan["src_img"].to(DEV), img, embed, plan["S"])
        assert torch.equal(emb1, emb2) and torch.equal(nl.cpu(), plan["labels"]) and torch.equal(nm.cpu(), plan["mask"]), name
        d_emb = torch.randn(B, plan["S"], 256, generator=g).to(DEV, torch.bfloat16)
        assert torch.equal(hk.splice_bwd(d_emb, pos, NI), hk.splice_map_bwd(d_emb, plan["inv"].to(DEV), B, NI)), name


@pytest.mark.timeout(1200)
def test_multi_image_training_step_end_to_end_vs_oracle():
    """A micro-batch whose samples hold 2, 0 and 1 images (4 image slots through ViT + AttnPooler, the placeholder-free sample's slot unused):
    loss, d loss / d image of every slot and the projector gradients against the oracle's autograd."""
    nl = 2
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=nl)}
    model = UniBind(("rgb", "text"), None, device=DEV, llama_layers=nl).load_params(P)
    model.prepare_for_training(freeze_vision=True, freeze_text=True, tune_rgb_pooler=True)
    g = torch.Generator().manual_seed(5)
    B, T = 3, 20
    ids = torch.randint(3, 32000, (B, T), generator=g)
    ids[:, 0] = 1
    ids[0, 1], ids[0, 9] = -200, -200
    ids[2, 4] = -200
    ids[1, 15:] = 0
    labels = ids.clone()
    labels[:, :2] = -100
    labels[(ids == 0) | (ids == -200)] = -100
    batch = dict(rgb=torch.randn(4, 3, 224, 224, generator=g), input_ids=ids, labels=labels, attention_mask=ids.ne(0))
    out = model(batch)
    d_image = model.text.backward()
    torch.cuda.synchronize()
    assert d_image.shape == (4, 144, 4096)
    for L in [P["pooler"]] + P["pooler"]["layers"]:
        for k, v in L.items():
            if torch.is_tensor(v):
                v.requires_grad_(True)
    col = {}
    loss = O.unibind_forward(P, batch, col)
    col["image"].retain_grad()
    loss.backward()
    assert col["embeds"].shape[1] == T + 2 * 143
    assert abs(out["total_loss"].item() - loss.item()) < 1e-3 * loss.item(), (out["total_loss"].item(), loss.item())
    gi = col["image"].grad
    assert float(gi[2].abs().max()) == 0.0 and float(d_image[2].float().abs().max()) == 0.0      # the slot the placeholder-free sample consumed
    for s in (0, 1, 3):
        assert rel(d_image[s], gi[s]) < 5e-2, s
    # the projector sees all four slots: run its backward and compare gradient norms
    model.rgb_pooler.backward(d_image)
    torch.cuda.synchronize()
    want = {"query": P["pooler"]["query"].grad, "out_proj.bias": P["pooler"]["out_proj_b"].grad}
    for name, gw in want.items():
        got = model.rgb_pooler.g[name].double().norm().item()
        assert abs(got - gw.double().norm().item()) < 5e-2 * gw.double().norm().item(), name


@pytest.mark.timeout(900)
def test_generate_with_two_images_in_the_prompt_matches_oracle():
    """UniBind.generate on a prompt with two <image> placeholders (two image slots through ViT + AttnPooler): greedy ids are the argmax of the
    engine's logits, the logits match the oracle's teacher-forced full re-runs."""
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=2)}
    model = UniBind(("rgb", "text"), None, device=DEV, llama_layers=2).load_params(P).eval()
    g = torch.Generator().manual_seed(13)
    T, NEW = 10, 4
    ids = torch.randint(3, 32000, (1, T), generator=g)
    ids[0, 0], ids[0, 1], ids[0, 5] = 1, -200, -200
    rgb = torch.randn(2, 3, 224, 224, generator=g)
    new_ids, logits = model.generate(ids, images=rgb, do_sample=False, max_new_tokens=NEW, return_logits=True)
    assert new_ids.shape == (1, NEW) and torch.equal(new_ids, logits.argmax(-1))
    want = O.generate_logits(P, rgb, ids, new_ids.cpu())
    assert rel(logits, want) < 3e-2
    picked = want.gather(-1, new_ids.cpu()[..., None]).squeeze(-1)
    assert torch.all(want.max(-1).values - picked < 0.15 * want.std(-1))
