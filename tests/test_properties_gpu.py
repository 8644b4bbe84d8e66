"""Size-independent properties of the HIP path at the BASELINE sizes (B = 8, S = 273, LLaMA-2-7B width), where the fp32 CPU
oracle is too slow to be the checker: causality, batch independence, linearity, splice round trip, edge shapes."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from lhrs_bot_amd import kernels as hk  # noqa: E402
from lhrs_bot_amd.text import TextModal  # noqa: E402

DEV = "cuda"
B, S, H, D = 8, 273, 32, 128


def bf(x):
    return x.to(torch.bfloat16)


def run_attn(q, k, v, kv_len=None):
    desc = hk.make_desc([(b * S, S, b * S, S if kv_len is None else kv_len[b], S, 0) for b in range(B)], DEV)
    o = torch.zeros(B * S, H * D, device=DEV, dtype=torch.bfloat16)
    lse = torch.zeros(B, H, hk.pad64(S), device=DEV)
    hk.attn_fwd(q, k, v, o, lse, desc, B, H, D, S, S, hk.pad64(S), True, 1 / math.sqrt(D))
    return o


def test_causal_attention_ignores_the_future_bitwise():
    g = torch.Generator().manual_seed(0)
    q, k, v = (bf(torch.randn(B * S, H * D, generator=g)).to(DEV) for _ in range(3))
    o1 = run_attn(q, k, v)
    t = 100
    k2, v2 = k.clone(), v.clone()
    for b in range(B):  # rewrite every key / value at positions >= t
        k2[b * S + t:(b + 1) * S] = bf(torch.randn(S - t, H * D, generator=g)).to(DEV)
        v2[b * S + t:(b + 1) * S] = 7.0
    o2 = run_attn(q, k2, v2)
    o1v, o2v = o1.view(B, S, -1), o2.view(B, S, -1)
    assert torch.equal(o1v[:, :t], o2v[:, :t]) and not torch.equal(o1v[:, t:], o2v[:, t:])
    # key padding: keys >= kv_len are invisible, whatever they contain
    o3 = run_attn(q, k2, v2, kv_len=[t] * B)
    o4 = run_attn(q, k, v, kv_len=[t] * B)
    assert torch.equal(o3, o4)


def test_llama_layer_is_independent_across_the_batch_and_gemm_is_linear():
    tm = TextModal(device=DEV, layers=1)
    tm.init_random(seed=3)
    g = torch.Generator().manual_seed(1)
    x = bf(torch.randn(B, S, 4096, generator=g)).to(DEV)
    h1 = tm.forward_hidden(x, None, save_ctx=False).view(B, S, -1)
    perm = torch.tensor([3, 0, 7, 1, 2, 6, 5, 4], device=DEV)
    h2 = tm.forward_hidden(x[perm].contiguous(), None, save_ctx=False).view(B, S, -1)
    assert torch.equal(h2, h1[perm])  # swapping samples swaps outputs, bit for bit
    # linearity of the fp32-output GEMM at full size: (a1 + a2) W^T = a1 W^T + a2 W^T up to fp32 accumulation order
    a1 = bf(torch.randn(B * S, 4096, generator=g)).to(DEV)
    a2 = bf(torch.randn(B * S, 4096, generator=g) * 2 ** -4).to(DEV)   # exactly representable sum in bf16? use a coarse grid
    a1 = (a1.float() * 16).round().div(16).to(torch.bfloat16)
    a2 = (a2.float() * 16).round().div(16).to(torch.bfloat16)
    s12 = (a1.float() + a2.float()).to(torch.bfloat16)
    assert torch.equal(s12.float(), a1.float() + a2.float())
    W = tm.p["layers"][0]["gu_w"]
    lhs = hk.gemm_nt(s12, W, out_f32=True)
    rhs = hk.gemm_nt(a1, W, out_f32=True) + hk.gemm_nt(a2, W, out_f32=True)
    assert ((lhs - rhs).norm() / rhs.norm()).item() < 1e-6


def test_splice_round_trip_and_edge_shapes():
    tm = TextModal(device=DEV, layers=0)
    g = torch.Generator().manual_seed(2)
    tm.p = {"embed": bf(torch.randn(32000, 4096, generator=g)).to(DEV)}
    # T = 2: only [BOS, <image>]; image token last; batch of one
    ids = torch.tensor([[1, -200]])
    img = bf(torch.randn(1, 144, 4096, generator=g)).to(DEV)
    emb, lab, msk, pos = tm.prepare_inputs_for_multimodal(ids, ids.ne(0), ids.clone(), img)
    assert emb.shape == (1, 145, 4096) and torch.equal(emb[0, 1:], img[0]) and int(pos[0]) == 1 and bool(msk.all())
    assert torch.equal(lab[0, 1:], torch.full((144,), -100, device=DEV))
    back = hk.splice_bwd(emb, pos, 144)            # the backward slice returns exactly the image rows
    assert torch.equal(back, img)
    # maximum length the tokenizer allows (model_max_length = 2048): S = 2048 - 1 + 144
    T = 2048
    ids = torch.randint(3, 32000, (2, T), generator=g)
    ids[:, 0] = 1
    ids[0, 1] = -200
    ids[1, T - 1] = -200
    img = bf(torch.randn(2, 144, 4096, generator=g)).to(DEV)
    emb, lab, msk, pos = tm.prepare_inputs_for_multimodal(ids, None, None, img)
    assert emb.shape == (2, T + 143, 4096) and pos.tolist() == [1, T - 1]
    assert torch.equal(emb[1, T - 1:], img[1]) and torch.equal(emb[0, 1:145], img[0])
    assert torch.equal(emb[1, :T - 1], tm.p["embed"][ids[1, :T - 1].to(DEV)])


def test_gemm_degenerate_shapes():
    g = torch.Generator().manual_seed(4)
    w = bf(torch.randn(4096, 4096, generator=g) * 0.02).to(DEV)
    for M in (1, 2, 63, 65):
        a = bf(torch.randn(M, 4096, generator=g)).to(DEV)
        out = hk.gemm_nt(a, w, out_f32=True)
        ref = a.float() @ w.float().t()
        assert ((out - ref).norm() / ref.norm()).item() < 1e-5
    with pytest.raises(RuntimeError):
        hk.gemm_nt(torch.zeros(0, 64, device=DEV, dtype=torch.bfloat16), w[:, :64].contiguous())


@pytest.mark.timeout(900)
def test_maximum_sequence_length_end_to_end():
    """model_max_length = 2048 text tokens + 143 image positions = S 2191: the tiled (non LDS-resident) attention kernels, the RoPE
    table beyond position 2048 and the splice at its largest size, forward + backward on a 1-layer model; causality holds at that size."""
    from lhrs_bot_amd.unibind import UniBind
    model = UniBind(("rgb", "text"), None, device="cuda", llama_layers=1).init_random(seed=2)
    model.prepare_for_training()
    g = torch.Generator().manual_seed(1)
    T = 2048
    ids = torch.randint(3, 32000, (1, T), generator=g)
    ids[0, 0], ids[0, 1] = 1, -200
    labels = ids.clone()
    labels[:, :2] = -100
    rgb = torch.randn(1, 3, 224, 224, generator=g)
    out = model(dict(rgb=rgb, input_ids=ids, labels=labels, attention_mask=ids.ne(0)))
    loss = out["total_loss"].item()
    assert 9.0 < loss < 12.5
    model.backward()
    gq = model.rgb_pooler.g["query"]
    assert torch.isfinite(gq).all() and float(gq.abs().sum()) > 0
    # changing the LAST token cannot change the hidden state of any earlier position: compare losses restricted to early labels
    ids2 = ids.clone()
    ids2[0, -1] = 17
    lab_early = labels.clone()
    lab_early[0, 1000:] = -100
    a = model(dict(rgb=rgb, input_ids=ids, labels=lab_early, attention_mask=ids.ne(0)))["total_loss"].item()
    b = model(dict(rgb=rgb, input_ids=ids2, labels=lab_early, attention_mask=ids.ne(0)))["total_loss"].item()
    assert a == b
