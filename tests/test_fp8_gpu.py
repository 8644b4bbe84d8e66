"""8-bit frozen base weights for stages 2/3: the e4m3 GEMM against its own dequantised operands, and a LoRA
training step on the e4m3 base against the same step on the bf16 base.  bitsandbytes (the reference's LLM.int8) is absent from the
image: the quantisation scheme is this engine's, parity with the reference's int8 arithmetic is UNPINNED."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from lhrs_bot_amd import kernels as hk  # noqa: E402
from lhrs_bot_amd.engine import LHRSEngine  # noqa: E402
from lhrs_bot_amd.unibind import UniBind  # noqa: E402

DEV = "cuda"


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.mark.parametrize("M,N,K", [(8190, 4096, 4096), (300, 512, 256), (4095, 12288, 4096), (8190, 4096, 11008), (257, 1032, 384),
                                   (8736, 4096, 4096), (4368, 4096, 11008), (8714, 4096, 4096)])  # the last three: tail rows on the small-tile kernel
def test_gemm_fp8_matches_dequantised_product(M, N, K):
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.02).to(DEV, torch.bfloat16)
    r = torch.randn(M, N, generator=g).to(DEV, torch.bfloat16)
    x8, sx = hk.quant_fp8_rows(x)
    w8, sw = hk.quant_fp8_rows(w)
    xd = x8.view(torch.float8_e4m3fn).float() * sx[:, None]
    wd = w8.view(torch.float8_e4m3fn).float() * sw[:, None]
    assert rel(xd, x.float()) < 4e-2 and rel(wd, w.float()) < 4e-2          # e4m3 per-row scaling: ~2^-4 per element
    y = hk.gemm_fp8_nt(x8, sx, w8, sw)
    ref = xd @ wd.t()
    assert rel(y, ref) < 3e-3                                                # exact products, fp32 accumulation, bf16 store
    y = hk.gemm_fp8_nt(x8, sx, w8, sw, residual=r, alpha=0.5)
    assert rel(y, 0.5 * ref + r.float()) < 3e-3
    assert rel(ref, x.float() @ w.float().t()) < 4e-2                        # what the 8-bit base costs a single linear
    with pytest.raises(RuntimeError):
        hk.gemm_fp8_nt(x8[:, :K - 64], sx, w8[:, :K - 64], sw)              # K % 128 != 0 is rejected at the ABI
    # fused bf16 pair (LoRA update on an 8-bit base): same launch, same accumulators
    for K2 in (64, 192):
        a2 = (torch.randn(M, K2, generator=g) * 0.3).to(DEV, torch.bfloat16)
        b2 = (torch.randn(N, K2, generator=g) * 0.1).to(DEV, torch.bfloat16)
        y = hk.gemm_fp8_nt(x8, sx, w8, sw, residual=r, a2=a2, b2=b2)
        assert rel(y, ref + a2.float() @ b2.float().t() + r.float()) < 3e-3, K2
    # tail-row rule on / off: the rows that move to the small-tile kernel keep their values (same products, same roundings)
    from lhrs_bot_amd import _lib
    lib = _lib.load()
    try:
        lib.lhrs_gemm_set_tail_split(0)
        whole = hk.gemm_fp8_nt(x8, sx, w8, sw, residual=r, a2=a2, b2=b2)
    finally:
        lib.lhrs_gemm_set_tail_split(1)
    assert rel(y, whole) < 1e-3 and rel(y[-200:], whole[-200:]) < 1e-3


@pytest.mark.timeout(900)
def test_lora_step_on_e4m3_base_tracks_the_bf16_base():
    from bench import make_batch

    def run(bits):
        m = UniBind(("rgb", "text"), None, device=DEV, llama_layers=2).init_random(seed=5)
        m.enable_lora(r=8, alpha=16, targets=("q", "k", "v", "o"), seed=3)
        # B starts at zero in peft: give the adapters a non-trivial state so that dA is not identically zero
        torch.manual_seed(0)
        m.text.lora.master.add_(torch.randn_like(m.text.lora.master) * 0.01)
        hk.cast_f32_to_bf16(m.text.lora.master, m.text.lora.shadow)
        m.text.lora.refresh()
        if bits == 8:
            m.text.quantize_base(8)
        m.prepare_for_training(freeze_text=False, tune_rgb_pooler=False)
        e = LHRSEngine(m, optimizer="adamw", lr=1e-4, weight_decay=0.0, max_grad_norm=1.0)
        b = make_batch(4, 40, torch.device(DEV), seed=9)
        loss = e(b)["total_loss"].item()
        e.backward()
        return loss, m.text.lora.grad.clone()

    l16, g16 = run(16)
    l8, g8 = run(8)
    assert abs(l8 - l16) < 2e-2 * l16, (l8, l16)
    cos = torch.nn.functional.cosine_similarity(g8.flatten().double(), g16.flatten().double(), dim=0).item()
    assert cos > 0.97, cos
    assert 0.8 < (g8.norm() / g16.norm()).item() < 1.25


def test_swiglu_with_fused_quantisation_equals_swiglu_then_quant():
    g = torch.Generator().manual_seed(4)
    for rows, F in ((300, 11008), (64, 512)):
        gu = torch.randn(rows, 2 * F, generator=g).to(DEV, torch.bfloat16)
        dact = (torch.randn(rows, F, generator=g) * 0.1).to(DEV, torch.bfloat16)
        act_ref = hk.swiglu_fwd(gu, F)
        a8_ref, sa_ref = hk.quant_fp8_rows(act_ref)
        act, a8, sa = hk.swiglu_fwd_q(gu, F, want_bf16=True)
        assert torch.equal(act, act_ref) and torch.equal(a8, a8_ref) and torch.equal(sa, sa_ref)
        assert hk.swiglu_fwd_q(gu, F)[0] is None
        dgu_ref = hk.swiglu_bwd(dact, gu, F)
        d8_ref, sd_ref = hk.quant_fp8_rows(dgu_ref)
        gu2 = gu.clone()
        _, d8, sd = hk.swiglu_bwd_q(dact, gu2, F)
        assert torch.equal(gu2, gu) and torch.equal(d8, d8_ref) and torch.equal(sd, sd_ref)
        dgu, d8, sd = hk.swiglu_bwd_q(dact, gu2, F, want_bf16=True)      # in place over gu
        assert dgu.data_ptr() == gu2.data_ptr() and torch.equal(dgu, dgu_ref) and torch.equal(d8, d8_ref)


def test_rmsnorm_with_fused_quantisation_equals_rmsnorm_then_quant():
    g = torch.Generator().manual_seed(8)
    x = torch.randn(300, 4096, generator=g).to(DEV, torch.bfloat16)
    dy = (torch.randn(300, 4096, generator=g) * 0.1).to(DEV, torch.bfloat16)
    add = torch.randn(300, 4096, generator=g).to(DEV, torch.bfloat16)
    w = (1 + 0.1 * torch.randn(4096, generator=g)).to(DEV, torch.bfloat16)
    y_ref = hk.rmsnorm_fwd(x, w)
    y8_ref, s_ref = hk.quant_fp8_rows(y_ref)
    y, (y8, s) = hk.rmsnorm_fwd_q(x, w)
    assert torch.equal(y, y_ref) and torch.equal(y8, y8_ref) and torch.equal(s, s_ref)
    none, (y8b, sb) = hk.rmsnorm_fwd_q(x, w, want_bf16=False)
    assert none is None and torch.equal(y8b, y8_ref) and torch.equal(sb, s_ref)
    dx_ref = hk.rmsnorm_bwd(dy, x, w, None, add=add)
    d8_ref, ds_ref = hk.quant_fp8_rows(dx_ref)
    dx, (d8, ds) = hk.rmsnorm_bwd_q(dy, x, w, None, add=add)
    assert torch.equal(dx, dx_ref) and torch.equal(d8, d8_ref) and torch.equal(ds, ds_ref)


@pytest.mark.timeout(900)
def test_lora_training_on_e4m3_base_follows_the_bf16_loss_curve():
    """25 AdamW steps on a fixed batch (stage-3 shape: LoRA r=8 on q,k,v,o, 2 layers): the e4m3-base run must learn like the bf16-base run."""
    from bench import make_batch

    def curve(bits):
        m = UniBind(("rgb", "text"), None, device=DEV, llama_layers=2).init_random(seed=5)
        m.enable_lora(r=8, alpha=16, targets=("q", "k", "v", "o"), seed=3)
        if bits == 8:
            m.text.quantize_base(8)
        m.prepare_for_training(freeze_text=False, tune_rgb_pooler=False)
        e = LHRSEngine(m, optimizer="adamw", lr=2e-3, weight_decay=0.0, max_grad_norm=1.0)
        b = make_batch(4, 40, torch.device(DEV), seed=9)
        out = []
        for _ in range(25):
            loss = e(b)["total_loss"]
            e.backward()
            e.step()
            out.append(loss.item())
        return out

    c16, c8 = curve(16), curve(8)
    assert c16[-1] < 0.05 * c16[0] and c8[-1] < 0.05 * c8[0], (c16[0], c16[-1], c8[0], c8[-1])   # both fit the batch
    assert all(abs(a - b) < 0.1 * b + 0.02 for a, b in zip(c8, c16)), list(zip(c8, c16))         # along the same curve


@pytest.mark.timeout(900)
def test_e4m3_base_against_the_llm_int8_oracle():
    """§8 f-4: the reference runs stages 2/3 on a bitsandbytes LLM.int8 base (`bits: 8`, text_modal.py:91-131).  This engine's 8-bit base
    is e4m3 on the block-scaled MFMA - a DELIBERATE DEVIATION (no outlier decomposition; weights, activations and gradients all e4m3
    with per-row scales).  oracle/int8_oracle.py restates LLM.int8 (PARITY UNPINNED: bitsandbytes is absent).  Same 2-layer model, same
    LoRA state (r = 8 on q,k,v,o), same batch through (a) the fp32 oracle, (b) the LLM.int8 oracle, (c) the HIP e4m3 path: loss and
    every adapter gradient.  MEASURED on MI355X (round 2): loss fp32 11.2640 / LLM.int8 11.2740 / e4m3 11.2492; adapter-gradient
    rel-L2 vs fp32: LLM.int8 0.046, e4m3 0.143 - e4m3 (3 mantissa bits on weights, activations AND gradients) is ~3x coarser than LLM.int8
    (7-bit vector-wise weights / activations, 16-bit backward through the dequantised weight): that is the price of running the frozen base
    on the 2x-rate block-scaled MFMA, stated here rather than hidden.  The e4m3 loss is one draw of the quantisation noise: last-bit changes in
    the attention kernels (same error against fp64, different roundings) moved it from 11.2492 to 11.2324.  Bars: loss within 6e-3 of fp32 and of
    LLM.int8; gradient rel-L2 vs fp32 <= 0.20 with cosine >= 0.98; LLM.int8 restatement itself <= 0.08."""
    from oracle import int8_oracle as I8
    from oracle import lhrs_oracle as O
    from test_lora_gpu import make
    targets = ("q", "k", "v", "o")
    model, lora, P, batch = make(targets, 8, False)
    model.text.quantize_base(8)
    loss_hip = model(batch)["total_loss"].item()
    model.text.backward(need_input_grad=False)
    torch.cuda.synchronize()
    got = {(l, pr): [t.float().cpu().clone() for t in lora.grad_adapter(l, pr)] for l in range(2) for pr in targets}

    def oracle_run(llama):
        leaves = []
        for L in llama["layers"]:
            for pr in targets:
                for t in L["lora"][pr]:
                    t.grad = None
                    leaves.append(t)
        loss = O.unibind_forward(dict(P, llama=llama), batch)
        loss.backward()
        return loss.item(), {(l, pr): [t.grad.clone() for t in llama["layers"][l]["lora"][pr]] for l in range(2) for pr in targets}

    torch.set_num_threads(32)
    l32, g32 = oracle_run(P["llama"])
    l8, g8 = oracle_run(I8.int8_llama_params(P["llama"]))

    def dist(a, b):
        num = sum(((x.double() - y.double()) ** 2).sum() for k in a for x, y in zip(a[k], b[k]))
        den = sum((y.double() ** 2).sum() for k in b for y in b[k])
        return float((num / den).sqrt())

    d_hip32, d_int32, d_hip_int = dist(got, g32), dist(g8, g32), dist(got, g8)
    print(f"loss fp32 {l32:.4f} int8 {l8:.4f} e4m3 {loss_hip:.4f}; adapter-gradient rel-L2: e4m3 vs fp32 {d_hip32:.4f}, LLM.int8 vs fp32 {d_int32:.4f}, "
          f"e4m3 vs LLM.int8 {d_hip_int:.4f}")
    def cosine(a, b):
        dot = sum((x.double() * y.double()).sum() for k in a for x, y in zip(a[k], b[k]))
        na = sum((x.double() ** 2).sum() for k in a for x in a[k]).sqrt()
        nb = sum((y.double() ** 2).sum() for k in b for y in b[k]).sqrt()
        return float(dot / (na * nb))

    assert abs(loss_hip - l32) < 6e-3 * l32 and abs(loss_hip - l8) < 6e-3 * l8, (loss_hip, l8, l32)
    assert d_int32 < 0.08, d_int32
    assert d_hip32 < 0.20 and cosine(got, g32) > 0.98, (d_hip32, cosine(got, g32))
    assert d_hip_int < 0.22, d_hip_int
