"""generate(): KV-cache decoding on the HIP engine vs (a) the engine's own full re-forward and (b) the CPU oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from lhrs_bot_amd import kernels as hk  # noqa: E402
from lhrs_bot_amd.unibind import UniBind  # noqa: E402
from oracle import lhrs_oracle as O  # noqa: E402
from oracle import params as OP  # noqa: E402

DEV = "cuda"


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.mark.timeout(900)
def test_greedy_generate_matches_oracle_and_full_forward():
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=2)}
    model = UniBind(("rgb", "text"), None, device=DEV, llama_layers=2).load_params(P).eval()
    g = torch.Generator().manual_seed(11)
    B, T, NEW = 2, 9, 6
    ids = torch.randint(3, 32000, (B, T), generator=g)
    ids[:, 0] = 1
    ids[:, 1] = -200
    rgb = torch.randn(B, 3, 224, 224, generator=g)
    new_ids, logits = model.generate(ids, images=rgb, do_sample=False, max_new_tokens=NEW, return_logits=True)
    assert new_ids.shape == (B, NEW) and new_ids.dtype == torch.int64
    # greedy: each returned token is the argmax of the logits the engine produced for that step (lowest index on ties)
    assert torch.equal(new_ids, logits.argmax(-1))
    # (b) oracle, teacher-forced with the engine's tokens: logits agree to bf16 tolerance, and the engine's pick is within a
    # small margin of the oracle's best logit (exact argmax equality is not defined for near-ties at bf16 precision)
    want = O.generate_logits(P, rgb, ids, new_ids.cpu())
    assert rel(logits, want) < 3e-2
    picked = want.gather(-1, new_ids.cpu()[..., None]).squeeze(-1)
    assert torch.all(want.max(-1).values - picked < 0.15 * want.std(-1))
    # (a) the cached decode equals a full forward of the engine over [prompt | generated] (same kernels, no cache)
    img = model.encode_image(rgb.to(DEV))
    emb, _, mask, _ = model.text.prepare_inputs_for_multimodal(ids, None, None, img)
    full = torch.cat([emb, model.text.p["embed"][new_ids[:, :-1]]], 1)
    hid = model.text.forward_hidden(full, None, save_ctx=False).view(B, -1, 4096)
    S0 = emb.shape[1]
    lg = hk.gemm_nt(hid[:, S0 - 1:].reshape(-1, 4096).contiguous(), model.text.p["lm_head"], out_f32=True).view(B, NEW, -1)
    assert rel(lg, logits) < 1e-2


def test_text_only_generate_matches_oracle():
    """UniBind.generate(images=None): a text-only turn - plain token embeddings, no splice (text_modal.py:321-339)."""
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=2)}
    model = UniBind(("rgb", "text"), None, device=DEV, llama_layers=2).load_params(P).eval()
    ids = torch.tensor([[1, 50, 600, 7000, 80, 9]])
    new_ids, logits = model.generate(ids, images=None, do_sample=False, max_new_tokens=3, return_logits=True)
    emb = P["llama"]["embed"]
    seq = ids.clone()
    for t in range(3):
        h = O.llama_hidden(P["llama"], emb[seq], None)
        want = torch.nn.functional.linear(h[:, -1], P["llama"]["lm_head"]).float()
        assert rel(logits[:, t], want) < 3e-2
        seq = torch.cat([seq, new_ids[:, t:t + 1].cpu()], 1)
    with pytest.raises(ValueError):
        model.generate(torch.tensor([[1, -200, 5]]), images=None, max_new_tokens=1)


def test_sampling_path_runs_and_respects_eos():
    model = UniBind(("rgb", "text"), None, device=DEV, llama_layers=1).init_random(seed=0).eval()
    ids = torch.tensor([[1, -200, 5, 6, 7]])
    rgb = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(67))   # (seeded: the image used to depend on what ran before in the process)
    torch.manual_seed(0)
    out = model.generate(ids, images=rgb, do_sample=True, temperature=0.4, top_p=0.9, top_k=50, max_new_tokens=5)
    assert out.shape == (1, 5) and int(out.min()) >= 0 and int(out.max()) < 32000
    first = model.generate(ids, images=rgb, do_sample=False, max_new_tokens=4)
    # stop on the first token that has not appeared before it (a random-init one-layer model may repeat its first token: stopping on a
    # repeated token ends the sequence at its FIRST occurrence)
    toks = first[0].tolist()
    j = max(i for i in range(len(toks)) if toks[i] not in toks[:i])
    stop = model.generate(ids, images=rgb, do_sample=False, max_new_tokens=4, eos_token_id=int(toks[j]))
    assert stop.shape[1] == j + 1 and torch.equal(stop[0], first[0, :j + 1])


def test_gemv_and_graph_decode_equal_eager_decode():
    g = torch.Generator().manual_seed(3)
    W = (torch.randn(4096, 11008, generator=g) * 0.02).to(DEV, torch.bfloat16)
    for B in (1, 2, 3, 8, 16):
        x = torch.randn(B, 11008, generator=g).to(DEV, torch.bfloat16)
        res = torch.randn(B, 4096, generator=g).to(DEV, torch.bfloat16)
        y = torch.empty(B, 4096, device=DEV, dtype=torch.bfloat16)
        hk.gemv(W, x, y, residual=res)
        ref = x.float() @ W.float().t() + res.float()
        assert rel(y, ref) < 4e-3
        y32 = torch.empty(B, 4096, device=DEV, dtype=torch.float32)
        hk.gemv(W, x, y32, out_f32=True)
        assert rel(y32, x.float() @ W.float().t()) < 1e-5
    model = UniBind(("rgb", "text"), None, device=DEV, llama_layers=2).init_random(seed=1).eval()
    ids = torch.tensor([[1, -200, 9, 8, 7, 6], [1, -200, 5, 4, 3, 2]])
    rgb = torch.randn(2, 3, 224, 224, generator=g)
    a_ids, a_lg = model.generate(ids, images=rgb, do_sample=False, max_new_tokens=9, return_logits=True, use_graph=True)
    b_ids, b_lg = model.generate(ids, images=rgb, do_sample=False, max_new_tokens=9, return_logits=True, use_graph=False)
    assert torch.equal(a_ids, b_ids) and torch.equal(a_lg, b_lg)   # replayed graph == eager launches, bit for bit


def test_fp8_weight_decode_and_fused_prologues():
    g = torch.Generator().manual_seed(5)
    # fused prologues against their unfused compositions
    W = (torch.randn(4096, 4096, generator=g) * 0.02).to(DEV, torch.bfloat16)
    x = torch.randn(2, 4096, generator=g).to(DEV, torch.bfloat16)
    nw = (1 + 0.1 * torch.randn(4096, generator=g)).to(DEV, torch.bfloat16)
    y = torch.empty(2, 4096, device=DEV, dtype=torch.bfloat16)
    hk.gemv_fused(W, x, y, 4096, prologue=hk.PRO_RMSNORM, norm_w=nw)
    y_ref = torch.empty_like(y)
    hk.gemv(W, hk.rmsnorm_fwd(x, nw), y_ref)
    assert torch.equal(y, y_ref)
    hk.gemv_fused(W, x[:1], y[:1], 4096, prologue=hk.PRO_RMSNORM, norm_w=nw)     # batch 1 = the VALU kernel, batch >= 2 = MFMA
    assert rel(y[:1], y_ref[:1]) < 4e-3
    gu = torch.randn(2, 2 * 11008, generator=g).to(DEV, torch.bfloat16)
    Wd = (torch.randn(4096, 11008, generator=g) * 0.02).to(DEV, torch.bfloat16)
    hk.gemv_fused(Wd, gu, y, 11008, prologue=hk.PRO_SWIGLU)
    hk.gemv(Wd, hk.swiglu_fwd(gu, 11008), y_ref)
    assert torch.equal(y, y_ref)
    # re-tiled bf16 weights (batched decode): the same products in the same order, for every prologue and batch 2 / 16, ragged N
    for Wt, K, pro, xin in ((W, 4096, hk.PRO_RMSNORM, x), (Wd, 11008, hk.PRO_SWIGLU, gu), (W[:4090], 4096, hk.PRO_NONE, x),
                            (W, 4096, hk.PRO_NONE, torch.randn(16, 4096, generator=g).to(DEV, torch.bfloat16))):
        Wp = hk.repack_bf16_mfma(Wt)
        ya = torch.empty(xin.shape[0], Wt.shape[0], device=DEV, dtype=torch.float32)
        yb = torch.empty_like(ya)
        hk.gemv_fused(Wt, xin, ya, K, prologue=pro, norm_w=nw, out_f32=True)
        hk.gemv_fused(Wp, xin, yb, K, prologue=pro, norm_w=nw, out_f32=True)
        assert torch.equal(ya, yb), (K, pro)
    # e4m3 weights: quantisation error only (per-row scaled): ~2^-4 relative per element, averaged down by the dot product
    W8, sc = hk.quant_fp8_rows(W)
    deq = W8.view(torch.float8_e4m3fn).float() * sc[:, None]
    assert rel(deq, W.float()) < 4e-2
    y8 = torch.empty(2, 4096, device=DEV, dtype=torch.float32)
    hk.gemv_fused(W8, x, y8, 4096, wscale=sc, out_f32=True)
    assert rel(y8, x.float() @ deq.t()) < 1e-5
    model = UniBind(("rgb", "text"), None, device=DEV, llama_layers=2).init_random(seed=1).eval()
    ids = torch.tensor([[1, -200, 9, 8, 7, 6]])
    rgb = torch.randn(1, 3, 224, 224, generator=g)
    _, lg_bf = model.generate(ids, images=rgb, do_sample=False, max_new_tokens=5, return_logits=True)
    _, lg_f8 = model.generate(ids, images=rgb, do_sample=False, max_new_tokens=5, return_logits=True, weights="fp8")
    assert rel(lg_f8[:, 0], lg_bf[:, 0]) == 0.0        # the prefill is bf16 in both
    assert rel(lg_f8[:, 1:2], lg_bf[:, 1:2]) < 1.5e-1  # first decoded step: e4m3 weights and activations (same token fed; random weights)
    # the MFMA e4m3 GEMV against its dequantised operands, batch 1 / 5 / 16, K not a multiple of 512
    for Bn in (1, 5, 16):
        xs = torch.randn(Bn, 11008, generator=g).to(DEV, torch.bfloat16)
        Wd = (torch.randn(4096, 11008, generator=g) * 0.02).to(DEV, torch.bfloat16)
        rs = torch.randn(Bn, 4096, generator=g).to(DEV, torch.bfloat16)
        w8, ws = hk.quant_fp8_rows(Wd)
        a8, as_ = hk.quant_fp8_rows(xs)
        out = torch.empty(Bn, 4096, device=DEV, dtype=torch.float32)
        hk.gemv_fp8_mfma(w8, ws, a8, as_, out, residual=rs, out_f32=True)
        ref = (a8.view(torch.float8_e4m3fn).float() * as_[:, None]) @ (w8.view(torch.float8_e4m3fn).float() * ws[:, None]).t() + rs.float()
        assert rel(out, ref) < 1e-4, Bn
        w8p = hk.repack_fp8_mfma(w8)  # tiled operand order of the decode weight stream: same products, same order
        outp = torch.empty_like(out)
        hk.gemv_fp8_mfma(w8p, ws, a8, as_, outp, residual=rs, out_f32=True)
        assert torch.equal(outp, out), Bn
        if Bn <= 2:  # in-kernel quantisation == stand-alone quantisation, bit for bit
            out2 = torch.empty_like(out)
            hk.gemv_fp8_mfma_fused(w8, ws, xs, out2, 11008, residual=rs, out_f32=True)
            assert torch.equal(out2, out)
            hk.gemv_fp8_mfma_fused(w8p, ws, xs, out2.zero_(), 11008, residual=rs, out_f32=True)
            assert torch.equal(out2, out)


@pytest.mark.timeout(900)
def test_batched_left_padded_generate():
    """§8 f-3 (main_vqa.py:205-214): LEFT-padded prompts + attention_mask; the spliced mask hides keys inside the kernels
    (lhrs_attn_fwd_kmask) in the prefill and in every cached step, graph-captured or eager."""
    import os
    import numpy as np
    Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "batch_generate.npz"))
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=2)}
    model = UniBind(("rgb", "text"), None, device=DEV, llama_layers=2).load_params(P).eval()
    rgb = torch.randn(3, 3, 224, 224, generator=torch.Generator().manual_seed(int(Z["rgb_seed"])))
    ids, mask = torch.from_numpy(Z["input_ids"]), torch.from_numpy(Z["attention_mask"])
    NEW = 5
    new_ids, logits = model.generate(ids, images=rgb, attention_mask=mask, do_sample=False, max_new_tokens=NEW, return_logits=True)
    assert new_ids.shape == (3, NEW) and torch.equal(new_ids, logits.argmax(-1))
    # step 0 (the prefill) against the REFERENCE's logits, later steps against the oracle teacher-forced with the engine's tokens
    cols = torch.from_numpy(Z["logits_cols"])
    assert rel(logits[:, 0].cpu()[:, cols], torch.from_numpy(Z["logits"])[:, 0]) < 3e-2
    want = O.generate_logits(P, rgb, ids, new_ids.cpu(), attention_mask=mask)
    assert rel(logits, want) < 3e-2
    picked = want.gather(-1, new_ids.cpu()[..., None]).squeeze(-1)
    assert torch.all(want.max(-1).values - picked < 0.15 * want.std(-1))
    # the mask matters: without it the padded rows change, the unpadded row 0 does not
    _, lg_nomask = model.generate(ids, images=rgb, attention_mask=None, do_sample=False, max_new_tokens=1, return_logits=True)
    assert rel(lg_nomask[0], logits[0, :1]) < 1e-6 and rel(lg_nomask[2], logits[2, :1]) > 1e-3
    # eager launches == replayed graph, bit for bit, with the mask in place
    b_ids, b_lg = model.generate(ids, images=rgb, attention_mask=mask, do_sample=False, max_new_tokens=NEW, return_logits=True, use_graph=False)
    assert torch.equal(b_ids, new_ids) and torch.equal(b_lg, logits)


def test_attention_key_mask_kernel():
    import math
    g = torch.Generator().manual_seed(9)
    H, D, S = 4, 128, 150
    for nq in (S, 1):
        q = torch.randn(2 * nq, H * D, generator=g).to(DEV, torch.bfloat16)
        k = torch.randn(2 * S, H * D, generator=g).to(DEV, torch.bfloat16)
        v = torch.randn(2 * S, H * D, generator=g).to(DEV, torch.bfloat16)
        km = (torch.rand(2, S, generator=g) > 0.3).to(torch.uint8)
        km[:, 0] = 1
        o = torch.empty_like(q)
        coff = S - nq
        desc = hk.make_desc([(b * nq, nq, b * S, S, S, coff) for b in range(2)], DEV)
        hk.attn_fwd(q, k, v, o, None, desc, 2, H, D, nq, S, hk.pad64(nq), True, 1 / math.sqrt(D), key_mask=km.to(DEV))
        qf, kf, vf = (t.float().cpu().view(2, -1, H, D).transpose(1, 2) for t in (q, k, v))
        bias = torch.zeros(2, 1, nq, S)
        bias.masked_fill_(~km.bool()[:, None, None, :], float("-inf"))
        causal = torch.arange(S)[None, :] > (torch.arange(nq)[:, None] + coff)
        bias.masked_fill_(causal[None, None], float("-inf"))
        ref = (torch.softmax(qf @ kf.transpose(-1, -2) / math.sqrt(D) + bias, -1) @ vf).transpose(1, 2).reshape(2 * nq, H * D)
        assert rel(o, ref) < 1e-2


def test_decode_attn_kernel_matches_rope_append_attention():
    """lhrs_decode_attn (one launch) == rope_kv_append + tiled attention over the cache, incl. the append side effect, a key mask,
    contexts that end inside / at the edge of a 512-key pass, and more than one pass."""
    import math
    g = torch.Generator().manual_seed(21)
    B, H, D, max_ctx = 2, 32, 128, 1300
    d = H * D
    cos, sin = (t.to(DEV) for t in __import__("oracle.lhrs_oracle", fromlist=["rope_tables"]).rope_tables(max_ctx, D))
    for ctx, masked in ((0, False), (5, False), (127, True), (300, True), (511, False), (512, True), (640, False), (1299, True)):
        kc = torch.randn(B * max_ctx, d, generator=g).to(DEV, torch.bfloat16)
        vc = torch.randn(B * max_ctx, d, generator=g).to(DEV, torch.bfloat16)
        qkv = torch.randn(B, 3 * d, generator=g).to(DEV, torch.bfloat16)
        km = None
        if masked:
            km = (torch.rand(B, max_ctx, generator=g) > 0.3).to(torch.uint8)
            km[:, 0] = 1
            km = km.to(DEV)
        pos = torch.full((B,), ctx, dtype=torch.int32, device=DEV)
        kc2, vc2, qkv2 = kc.clone(), vc.clone(), qkv.clone()
        o = torch.empty(B, d, device=DEV, dtype=torch.bfloat16)
        hk.decode_attn(qkv, kc, vc, cos, sin, pos, o, B, H, D, max_ctx, 1 / math.sqrt(D), key_mask=km)
        hk.rope_kv_append(qkv2, kc2, vc2, cos, sin, pos, B, H, D, max_ctx)
        desc = hk.make_desc([(b, 1, b * max_ctx, ctx + 1, ctx + 1, ctx) for b in range(B)], DEV)
        o2 = torch.empty_like(o)
        hk.attn_fwd(qkv2[:, :d], kc2, vc2, o2, None, desc, B, H, D, 1, 1 << 30, 64, True, 1 / math.sqrt(D), key_mask=km)
        assert torch.equal(kc, kc2) and torch.equal(vc, vc2), ctx      # the appended rows are bit-identical
        assert rel(o, o2) < 6e-3, (ctx, rel(o, o2))                     # fp32 P.V here vs bf16-rounded P in the MFMA kernel


@pytest.mark.parametrize("nsplit", [2, 6, 11, 16])
def test_decode_attn_split_matches_the_one_workgroup_kernel(nsplit):
    """lhrs_decode_attn_split (context split into 128-key slices over nsplit workgroups per head, partials exchanged behind a ticket,
    last arriver merges) against lhrs_decode_attn on the same inputs: same appended rows bit for bit, same output up to the order of the fp32
    softmax sums, for contexts below one slice (no exchange), at slice edges, beyond nsplit slices (a workgroup walks several), with a key
    mask and per-sequence positions that differ; repeated calls on the same buffers (the tickets return to zero every time)."""
    import math
    g = torch.Generator().manual_seed(77 + nsplit)
    B, H, D, max_ctx = 2, 32, 128, 2100
    d = H * D
    cos, sin = (t.to(DEV) for t in __import__("oracle.lhrs_oracle", fromlist=["rope_tables"]).rope_tables(max_ctx, D))
    part = torch.zeros(B, H, nsplit, 132, device=DEV)
    tickets = torch.zeros(B, H, device=DEV, dtype=torch.int32)
    for ctx, masked in ((0, False), (5, True), (127, False), (128, True), (300, False), (715, True), (1023, False), (1024, True), (2099, False)):
        kc = torch.randn(B * max_ctx, d, generator=g).to(DEV, torch.bfloat16)
        vc = torch.randn(B * max_ctx, d, generator=g).to(DEV, torch.bfloat16)
        qkv = torch.randn(B, 3 * d, generator=g).to(DEV, torch.bfloat16)
        km = None
        if masked:
            km = (torch.rand(B, max_ctx, generator=g) > 0.3).to(torch.uint8)
            km[:, 0] = 1
            km = km.to(DEV)
        pos = torch.tensor([ctx, max(0, ctx - 37)], dtype=torch.int32, device=DEV)      # the two sequences sit at different positions
        kc2, vc2 = kc.clone(), vc.clone()
        o, o2 = torch.empty(B, d, device=DEV, dtype=torch.bfloat16), torch.empty(B, d, device=DEV, dtype=torch.bfloat16)
        hk.decode_attn_split(qkv, kc, vc, cos, sin, pos, o, B, H, D, max_ctx, 1 / math.sqrt(D), nsplit, part, tickets, key_mask=km)
        hk.decode_attn(qkv, kc2, vc2, cos, sin, pos, o2, B, H, D, max_ctx, 1 / math.sqrt(D), key_mask=km)
        torch.cuda.synchronize()
        assert torch.equal(kc, kc2) and torch.equal(vc, vc2), ctx
        assert int(tickets.abs().sum()) == 0, ctx
        assert rel(o, o2) < 2e-3, (ctx, rel(o, o2))
        assert (o.float() - o2.float()).abs().max().item() <= 2.0 ** -6 * o2.float().abs().max().item(), ctx
        # the cos | sin rows handed over by lhrs_decode_advance_cs instead of read from the tables behind pos[b]: the same numbers, bit for bit
        cs = torch.cat([cos[pos.long()], sin[pos.long()]], dim=1).contiguous()
        kc3, vc3, o3 = kc2.clone(), vc2.clone(), torch.empty_like(o)
        hk.decode_attn_split(qkv, kc3, vc3, cos, sin, pos, o3, B, H, D, max_ctx, 1 / math.sqrt(D), nsplit, part, tickets, key_mask=km, cs=cs)
        assert torch.equal(o3, o) and torch.equal(kc3, kc), ctx


@pytest.mark.timeout(600)
def test_decode_attn_split_exchange_under_load_is_stable():
    """The cross-workgroup hand-off of lhrs_decode_attn_split (write-through partials, drained, one ticket per head, last arriver reads with
    sc1 loads) under UNEVEN load and with a warm L1 - the conditions under which a missing release / acquire shows (MI355X_MICROARCH.md,
    inter-workgroup visibility): a side stream keeps the chip busy with GEMMs of varying size while 300 launches walk contexts of 130 - 1500
    keys over the SAME partial buffer (every launch overwrites what the previous one's last arrivers just read).  Every output is compared
    in full with the one-workgroup kernel's, every launch is repeated and must reproduce itself bit for bit, and the tickets must be back
    at zero each time."""
    import math
    g = torch.Generator().manual_seed(5150)
    B, H, D, max_ctx, NS = 2, 32, 128, 1536, 12
    d = H * D
    cos, sin = (t.to(DEV) for t in __import__("oracle.lhrs_oracle", fromlist=["rope_tables"]).rope_tables(max_ctx, D))
    kc0 = torch.randn(B * max_ctx, d, generator=g).to(DEV, torch.bfloat16)
    vc0 = torch.randn(B * max_ctx, d, generator=g).to(DEV, torch.bfloat16)
    part = torch.zeros(B, H, NS, 132, device=DEV)
    tickets = torch.zeros(B, H, device=DEV, dtype=torch.int32)
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=DEV).bfloat16()
    w = torch.randn(4096, 4096, device=DEV).bfloat16()
    outs = [torch.empty(B, d, device=DEV, dtype=torch.bfloat16) for _ in range(3)]
    bad = []
    for it in range(300):
        ctx = 130 + (it * 37) % 1370
        with torch.cuda.stream(side):                              # uneven background load: 1 - 4 GEMMs of 512 - 4096 rows
            for j in range(1 + it % 4):
                hk.gemm_nt(a[: 512 << (j % 4)], w)
        qkv = torch.randn(B, 3 * d, generator=g).to(DEV, torch.bfloat16)
        pos = torch.tensor([ctx, ctx - 1 - it % 100], dtype=torch.int32, device=DEV)
        res = []
        for o in outs[:2]:
            kc, vc = kc0.clone(), vc0.clone()
            hk.decode_attn_split(qkv, kc, vc, cos, sin, pos, o, B, H, D, max_ctx, 1 / math.sqrt(D), NS, part, tickets)
            res.append(o)
        kc, vc = kc0.clone(), vc0.clone()
        hk.decode_attn(qkv, kc, vc, cos, sin, pos, outs[2], B, H, D, max_ctx, 1 / math.sqrt(D))
        torch.cuda.synchronize()
        if not torch.equal(res[0], res[1]) or int(tickets.abs().sum()) != 0 or rel(res[0], outs[2]) > 2e-3 or \
                (res[0].float() - outs[2].float()).abs().max().item() > 2.0 ** -6 * outs[2].float().abs().max().item():
            bad.append((it, ctx, rel(res[0], outs[2]), bool(torch.equal(res[0], res[1])), int(tickets.abs().sum())))
    assert not bad, bad[:5]
