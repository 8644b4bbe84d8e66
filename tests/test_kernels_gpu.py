"""Per-kernel numerics on a real MI355X: each HIP kernel of liblhrs_hip.so against a plain fp32 PyTorch
statement of the same op (tolerances are bf16-level and written next to each check)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from lhrs_bot_amd import kernels as hk  # noqa: E402

DEV = "cuda"


def rel_err(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def bf(x):
    return x.to(torch.bfloat16)


@pytest.fixture(autouse=True)
def _hand_written_gemm_only(request):
    """This file checks the kernels of csrc/gemm.hip first: the plain long-k products stay on the 16-wave kernels here (the product default
    gives them to the four-wave gemm_u4_kernel by a shape rule); the tests of the four-wave kernel switch it on themselves."""
    hk.gemm_set_u4(False)
    yield
    hk.gemm_set_u4(True)


# ------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K,res", [(8190, 4096, 11008, True), (8190, 4096, 22016, False), (2184, 4096, 4096, True), (1000, 1032, 256, False),
                                       (8736, 11008, 4096, False), (300, 264, 320, True), (4095, 4104, 4096, True), (256, 256, 256, False), (8736, 4096, 11008, True)])
def test_gemm_u4_four_wave_kernel_bit_identical_to_the_16_wave_kernel(M, N, K, res):
    """gemm_u4_kernel (csrc/gemm_u4.hip: 128x128 per wave, AGPR accumulators, paced DMA, persistent over tiles): same k order and fp32 accumulation as
    gemm_nt_256s_kernel - bit-identical on full tiles, ragged M / N edges, one to many tiles per workgroup, strided output and residual views; then the
    shape rule of lhrs_gemm_bf16_nt (which of the two kernels the product path runs - twice the same, bit for bit) and the problems the raw launch must decline."""
    g = torch.Generator(device="cpu").manual_seed(M * 3 + N + K)
    a = bf(torch.randn(M, K, generator=g)).to(DEV)
    b = bf(torch.randn(N, K, generator=g) * 0.05).to(DEV)
    r_full = bf(torch.randn(M, N + 8, generator=g)).to(DEV) if res else None
    r = r_full[:, :N] if res else None
    out_u = torch.zeros(M, N + 16, device=DEV, dtype=torch.bfloat16)
    assert hk.gemm_u4_nt(a, b, out_u[:, :N], residual=r)
    out_h = hk.gemm_nt(a, b, residual=r)                                    # fixture: u4 off -> gemm.hip's kernels
    ref = a.float() @ b.float().t() + (r.float() if res else 0.0)
    assert rel_err(out_h, ref) < 4e-3 and rel_err(out_u[:, :N], ref) < 4e-3
    # bit-identical where the persistent 256- / 144-row kernels ran (small problems take other tiles) and no residual is added: this kernel (like the library and
    # the split-K tail) adds the residual to the fp32 sum and rounds once, the staged epilogue of gemm_nt_256s_kernel rounds the sum first
    big = M >= 2048 and N >= 4096 and not res
    assert torch.equal(out_u[:, :N], out_h) if big else rel_err(out_u[:, :N], out_h) < 3e-3
    if res:
        assert (out_u[:, :N].float() - ref).abs().mean() <= (out_h.float() - ref).abs().mean() * 1.001
    assert float(out_u[:, N:].abs().max()) == 0.0
    hk.gemm_set_u4(True)
    out_p = hk.gemm_nt(a, b, residual=r)
    taken = hk.gemm_u4_takes(M, N, K, ldr=(N + 8) if res else 0)
    assert taken == (K >= 4096 and M >= 1024 and N >= 1024 and N % 8 == 0 and 5 * (-(-M // 256)) * (-(-N // 256)) >= 4 * 256)
    from lhrs_bot_amd import _lib
    Mu = _lib.load().lhrs_gemm_u4_main_rows(M, N) if taken else M     # a mostly empty last round (M = 8736, N = 4096: 2.19 rounds) is cut: rows [Mu, M) take small tiles / split-K
    assert Mu == {(8736, 4096): 8192, (4095, 4104): 3840}.get((M, N), M)     # 560 tiles -> 2 rounds + 544 rows; 272 tiles -> 1 round (15 tile rows) + 255 rows
    assert torch.equal(out_p[:Mu], out_u[:Mu, :N]) if taken else torch.equal(out_p, out_h)   # deterministic: the rule, not a timing, names the kernel
    assert rel_err(out_p[Mu:], ref[Mu:]) < 4e-3 if Mu < M else True
    assert torch.equal(hk.gemm_nt(a, b, residual=r), out_p)
    hk.gemm_set_u4(False)
    assert not hk.gemm_u4_nt(a[:, :96] if K > 96 else a, b[:, :96] if K > 96 else b, out_u[:, :N])            # K % 64 != 0 / K < 128: declined


@pytest.mark.parametrize("M,N,K,K2,res", [(8190, 4096, 4096, 64, True), (8190, 4096, 12288, 64, False), (8190, 12288, 4096, 128, False), (8736, 4096, 4096, 64, True),
                                          (4095, 4104, 4096, 192, False), (1000, 1032, 256, 64, True)])
def test_gemm_u4_lora_pair_in_the_k_loop_bit_identical_to_the_16_wave_kernel(M, N, K, K2, res):
    """lhrs_gemm_u4_nt_lora: A2 . B2^T as K2 / 64 more stages of gemm_u4_kernel's k-loop (stage 3's q / k / v / o adapters: peft lora.Linear forward reached from
    lhrs/models/text_modal.py:133-151) - the same k order and fp32 accumulation as the pair in gemm_nt_256s_kernel<.., K2P>: bit-identical without a residual,
    one rounding closer to the fp32 sum with one; then lhrs_gemm_bf16_nt_lora under the shape rule (with the row cut of a mostly empty last round)."""
    from lhrs_bot_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(M + 7 * N + K + K2)
    a = bf(torch.randn(M, K, generator=g)).to(DEV)
    b = bf(torch.randn(N, K, generator=g) * 0.05).to(DEV)
    a2_full = bf(torch.randn(M, K2 + 64, generator=g) * 2.0).to(DEV)
    a2 = a2_full[:, :K2]                                                     # strided pair operand (the adapters' T / U buffers are padded to 64 columns)
    b2 = bf(torch.randn(N, K2, generator=g) * 0.3).to(DEV)
    r = bf(torch.randn(M, N, generator=g)).to(DEV) if res else None
    out_u = torch.zeros(M, N + 16, device=DEV, dtype=torch.bfloat16)
    assert hk.gemm_u4_nt(a, b, out_u[:, :N], residual=r, a2=a2, b2=b2)
    out_h = hk.gemm_nt_lora(a, b, a2, b2, residual=r)                       # fixture: u4 off -> gemm.hip's kernels
    ref = a.float() @ b.float().t() + a2.float() @ b2.float().t() + (r.float() if res else 0.0)
    assert rel_err(out_h, ref) < 4e-3 and rel_err(out_u[:, :N], ref) < 4e-3
    assert rel_err(out_u[:, :N], hk.gemm_nt(a, b, residual=r)) > 1e-2       # the pair is in there
    big = M >= 2048 and N >= 4096 and not res
    Mx = {(4095, 4104): 3840}.get((M, N), M)        # the 16-wave path cuts that product's rows too; its cut rows take base + update as two launches (two roundings)
    assert torch.equal(out_u[:Mx, :N], out_h[:Mx]) if big else True
    assert rel_err(out_u[:, :N], out_h) < 3e-3
    assert float(out_u[:, N:].abs().max()) == 0.0
    hk.gemm_set_u4(True)
    out_p = hk.gemm_nt_lora(a, b, a2, b2, residual=r)
    taken = hk.gemm_u4_takes(M, N, K, ldr=N if res else 0)
    Mu = _lib.load().lhrs_gemm_u4_main_rows(M, N) if taken else M
    assert taken == (K >= 4096 and M >= 1024) and Mu == {(8736, 4096): 8192, (4095, 4104): 3840}.get((M, N), M)
    assert torch.equal(out_p[:Mu], out_u[:Mu, :N]) if taken else torch.equal(out_p, out_h)
    assert rel_err(out_p[Mu:], ref[Mu:]) < 4e-3 if Mu < M else True          # the cut rows: the pair rides on the small-tile kernels
    assert torch.equal(hk.gemm_nt_lora(a, b, a2, b2, residual=r), out_p)
    hk.gemm_set_u4(False)
    assert not hk.gemm_u4_nt(a, b, out_u[:, :N], a2=a2_full[:, :K2 + 32], b2=b2[:, :32].repeat(1, (K2 + 32) // 32))      # K2 % 64 != 0: declined


@pytest.mark.timeout(600)
@pytest.mark.parametrize("M,N,K", [(8190, 4096, 4096), (8190, 4096, 11008), (8190, 4096, 12288), (8190, 4096, 22016), (3822, 32000, 4096)])
def test_gemm_u4_soak_200_launches_under_load_every_result_identical(M, N, K):
    """gemm_u4_kernel waits for its LDS-DMA with a COUNTED `s_waitcnt vmcnt(15)` (stage kt+1 has landed, the 15 younger pieces of stage kt+2 stay in flight;
    csrc/gemm_u4_body.inc: WAITY).  A counted wait is only sound if the pieces of one wave retire in issue order (MI355X_MICROARCH.md: `vmcnt(N)` waits for the
    outstanding - N OLDEST operations) and nothing that may retire out of order shares the counter inside the k-loop (the previous tile's stores are drained by the
    `vmcnt(0)` at the top of every tile).  A premature pass would read a stale LDS stage and produce wrong tiles that come and go with memory load - so: 200
    back-to-back launches per LLaMA shape of the step (o, down, d-qkv, d-gate|up at micro-batch 30; lm_head on the supervised rows) while a side stream keeps the
    memory system busy (pinned host <-> device copies on the DMA engines, device copies between the launches), every result compared ON THE DEVICE with the first
    one, and the first one bit-identical to the 16-wave kernel (which waits vmcnt(0) only)."""
    g = torch.Generator(device="cpu").manual_seed(K + N)
    a = bf(torch.randn(M, K, generator=g)).to(DEV)
    b = bf(torch.randn(N, K, generator=g) * 0.05).to(DEV)
    want = hk.gemm_nt(a, b)                                        # fixture: u4 off -> gemm_nt_256s_kernel
    first = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    assert hk.gemm_u4_nt(a, b, first)
    assert torch.equal(first, want)
    side = torch.cuda.Stream()
    big = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    big2 = torch.empty_like(big)
    host = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    bad = torch.zeros((), device=DEV, dtype=torch.int64)
    for it in range(200):
        with torch.cuda.stream(side):
            big2.copy_(big, non_blocking=True)
            if it % 4 == 0:
                big[: host.numel()].copy_(host, non_blocking=True)
            elif it % 4 == 2:
                host.copy_(big2[: host.numel()], non_blocking=True)
        out.fill_(0)
        assert hk.gemm_u4_nt(a, b, out)
        bad += (out != first).sum()
    torch.cuda.synchronize()
    assert int(bad) == 0, f"{int(bad)} elements differed from the first launch over 200 launches"


@pytest.mark.timeout(900)
def test_gemm_u4_random_shapes_all_epilogues_vs_the_16_wave_kernels():
    """Seeded random shapes through every instantiation of the four-wave kernel's raw launches - ragged M and N (edge tiles take the exposed epilogue, interior tiles
    the in-stream write-out, in every order the tile walk produces: one to nine tiles per workgroup, odd and even stage counts) - against the 16-wave kernels /
    unfused kernel pairs on the same operands: bit-identical wherever no residual joins the sum (same k order), fp32-torch within bf16 rounding everywhere."""
    from lhrs_bot_amd import _lib
    lib, st = _lib.load(), torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(20260929)
    hd = 128
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    fr = torch.outer(torch.arange(700).float(), inv)
    cos, sin = fr.cos().to(DEV).contiguous(), fr.sin().to(DEV).contiguous()
    try:
        lib.lhrs_gemm_set_min_tiles(1)        # the 16-wave kernel from one tile on (the default threshold would send the small cases to other tiles: another k order)
        lib.lhrs_gemm_set_tail_split(0)
        lib.lhrs_gemm_set_bm144(0)
        for case in range(14):
            M = int(rng.integers(300, 9000))
            K = 64 * int(rng.integers(4, 40))
            g = torch.Generator(device="cpu").manual_seed(1000 + case)
            x = bf(torch.randn(M, K, generator=g)).to(DEV)
            kind = case % 4
            if kind == 0:      # plain, with and without a residual, ragged N (multiple of 8)
                N = 8 * int(rng.integers(40, 700))
                w = bf(torch.randn(N, K, generator=g) * 0.05).to(DEV)
                r = bf(torch.randn(M, N, generator=g)).to(DEV)
                out, out_r = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16), torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
                assert hk.gemm_u4_nt(x, w, out) and hk.gemm_u4_nt(x, w, out_r, residual=r)
                assert torch.equal(out, hk.gemm_nt(x, w)), (case, M, N, K)
                assert rel_err(out_r, x.float() @ w.float().t() + r.float()) < 4e-3, (case, M, N, K)
            elif kind == 1:    # SwiGLU forward: ff a multiple of 128
                ff = 128 * int(rng.integers(4, 40))
                w = bf(torch.randn(2 * ff, K, generator=g) * 0.05).to(DEV)
                gu, act = torch.zeros(M, 2 * ff, device=DEV, dtype=torch.bfloat16), torch.zeros(M, ff, device=DEV, dtype=torch.bfloat16)
                assert lib.lhrs_gemm_u4_swiglu_fwd(x.data_ptr(), K, w.data_ptr(), K, gu.data_ptr(), 2 * ff, act.data_ptr(), ff, M, ff, K, st) == 0
                gu_ref = hk.gemm_nt(x, w)
                assert torch.equal(gu, gu_ref) and torch.equal(act, hk.swiglu_fwd(gu_ref, ff)), (case, M, ff, K)
            elif kind == 2:    # SwiGLU backward in place over gate|up: ff a multiple of 8 (ragged last tile column)
                ff = 8 * int(rng.integers(64, 600))
                w = bf(torch.randn(ff, K, generator=g) * 0.05).to(DEV)
                gu = bf(torch.randn(M, 2 * ff, generator=g)).to(DEV)
                want = hk.swiglu_bwd(hk.gemm_nt(x, w), gu, ff)
                assert lib.lhrs_gemm_u4_swiglu_bwd(x.data_ptr(), K, w.data_ptr(), K, gu.data_ptr(), gu.data_ptr(), 2 * ff, M, ff, K, st) == 0
                assert torch.equal(gu, want), (case, M, ff, K)
            else:              # RoPE: heads of 128, rope_cols a multiple of 256, plain columns behind them
                nh = 2 * int(rng.integers(1, 12))
                N = nh * 128 + 8 * int(rng.integers(0, 60))
                S, pos0 = int(rng.integers(16, 400)), int(rng.integers(0, 200))
                w = bf(torch.randn(N, K, generator=g) * 0.05).to(DEV)
                out = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
                assert lib.lhrs_gemm_u4_rope(x.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, cos.data_ptr(), sin.data_ptr(), S, pos0, nh * 128, st) == 0
                ref = hk.gemm_nt(x, w)
                hk.rope_(ref, M, nh, hd, cos, sin, pos_mod=S, pos0=pos0)
                assert torch.equal(out, ref), (case, M, N, K, S, pos0)
    finally:
        lib.lhrs_gemm_set_bm144(1)
        lib.lhrs_gemm_set_tail_split(1)
        lib.lhrs_gemm_set_min_tiles(128)


@pytest.mark.timeout(600)
def test_gemm_u4_fused_epilogues_soak_100_launches_every_result_identical():
    """The write-out of a tile inside the next tile's first stage waits with vmcnt counts that include its own loads and stores (csrc/gemm_u4_flush_*.inc, tools/gen_u4.py):
    100 launches of each variant at micro-batch 30 - plain + residual (o), RoPE (qkv), SwiGLU forward (gate|up), SwiGLU backward (d-down) - with copies on a side
    stream, every result compared on the device with the first launch; the first launch is what the 16-wave kernels give (the fixture's default)."""
    from lhrs_bot_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(99)
    M, d, ff, hd, S = 8190, 4096, 11008, 128, 273
    x = bf(torch.randn(M, d, generator=g)).to(DEV)
    wo = bf(torch.randn(d, d, generator=g) * 0.02).to(DEV)
    wqkv = bf(torch.randn(3 * d, d, generator=g) * 0.02).to(DEV)
    wgu = bf(torch.randn(2 * ff, d, generator=g) * 0.02).to(DEV)
    wdT = bf(torch.randn(ff, d, generator=g) * 0.02).to(DEV)
    res = bf(torch.randn(M, d, generator=g)).to(DEV)
    dy = bf(torch.randn(M, d, generator=g) * 0.1).to(DEV)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    fr = torch.outer(torch.arange(512).float(), inv)
    cos, sin = fr.cos().to(DEV).contiguous(), fr.sin().to(DEV).contiguous()
    want_rope = hk.gemm_rope_fwd(x, wqkv, cos, sin, pos_mod=S, pos0=0, rope_cols=2 * d, head_dim=hd)     # fixture: 16-wave kernels
    want_gu, want_act = hk.gemm_swiglu_fwd(x, wgu, ff)
    want_dgu = hk.gemm_swiglu_bwd(dy, wdT, want_gu.clone(), ff)
    hk.gemm_set_u4(True)
    try:
        first_o = hk.gemm_nt(x, wo, residual=res)
        assert rel_err(first_o, x.float() @ wo.float().t() + res.float()) < 4e-3
        assert torch.equal(hk.gemm_rope_fwd(x, wqkv, cos, sin, pos_mod=S, pos0=0, rope_cols=2 * d, head_dim=hd), want_rope)
        side = torch.cuda.Stream()
        big, big2 = torch.empty(256 << 20, dtype=torch.uint8, device=DEV), torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
        host = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
        bad = torch.zeros(5, device=DEV, dtype=torch.int64)
        for it in range(100):
            with torch.cuda.stream(side):
                big2.copy_(big, non_blocking=True)
                if it % 2 == 0:
                    big[: host.numel()].copy_(host, non_blocking=True)
                else:
                    host.copy_(big2[: host.numel()], non_blocking=True)
            bad[0] += (hk.gemm_nt(x, wo, residual=res) != first_o).sum()
            bad[1] += (hk.gemm_rope_fwd(x, wqkv, cos, sin, pos_mod=S, pos0=0, rope_cols=2 * d, head_dim=hd) != want_rope).sum()
            gu, act = hk.gemm_swiglu_fwd(x, wgu, ff)
            bad[2] += (gu != want_gu).sum()
            bad[3] += (act != want_act).sum()
            # SwiGLU backward: the raw four-wave launch (at this M the operator's shape rule keeps the 16-wave kernel: 5.375 rounds), in place over gate|up
            assert lib.lhrs_gemm_u4_swiglu_bwd(dy.data_ptr(), dy.stride(0), wdT.data_ptr(), wdT.stride(0), gu.data_ptr(), gu.data_ptr(), gu.stride(0), M, ff, d,
                                               torch.cuda.current_stream().cuda_stream) == 0
            bad[4] += (gu != want_dgu).sum()
        torch.cuda.synchronize()
        assert bad.tolist() == [0, 0, 0, 0, 0], bad.tolist()
    finally:
        hk.gemm_set_u4(False)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (257, 1024, 1024), (2184, 4096, 4096), (1000, 12288, 4096),
                                   (273, 4096, 11008), (64, 64, 64), (33, 132, 128), (1152, 1024, 4096),
                                   (4095, 4096, 128), (3000, 4104, 128), (8190, 4096, 4096), (2184, 22016, 4096),
                                   (4095, 4096, 22016)])
def test_gemm_plain(M, N, K):
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N)
    a = bf(torch.randn(M, K, generator=g)).to(DEV)
    b = bf(torch.randn(N, K, generator=g) * 0.05).to(DEV)
    out = hk.gemm_nt(a, b)
    ref = a.float() @ b.float().t()
    assert rel_err(out, ref) < 4e-3  # bf16 output rounding: 2^-9 relative per element


def test_gemm_transpose_detecting():
    # A = "identity-like", asymmetric B: catches swapped operands / transposed C writes
    M = N = K = 128
    a = torch.zeros(M, K)
    a[torch.arange(M), torch.arange(K)] = 1.0
    b = torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251 - 125.0
    out = hk.gemm_nt(bf(a).to(DEV), bf(b).to(DEV), out_f32=True)
    assert torch.equal(out.cpu(), bf(b).float().t().contiguous())


@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_gemm_epilogue(act):
    M, N, Kd = 300, 1024, 512
    g = torch.Generator().manual_seed(act)
    a = bf(torch.randn(M, Kd, generator=g)).to(DEV)
    b = bf(torch.randn(N, Kd, generator=g) * 0.05).to(DEV)
    bias = bf(torch.randn(N, generator=g)).to(DEV)
    res = bf(torch.randn(M, N, generator=g)).to(DEV)
    out = hk.gemm_nt(a, b, bias=bias, residual=res, act=act, alpha=0.5)
    z = 0.5 * (a.float() @ b.float().t()) + bias.float()
    if act == 1:
        z = z * torch.sigmoid(1.702 * z)
    elif act == 2:
        z = F.gelu(z)
    elif act == 3:
        z = F.silu(z)
    ref = z + res.float()
    assert rel_err(out, ref) < 4e-3


@pytest.mark.parametrize("act", [0, 2])
def test_gemm_epilogue_256_tile(act):
    M, N, Kd = 4000, 4096, 256   # 16 x 16 tiles of 256^2 -> the 4-stage-ring kernel
    g = torch.Generator().manual_seed(40 + act)
    a = bf(torch.randn(M, Kd, generator=g)).to(DEV)
    b = bf(torch.randn(N, Kd, generator=g) * 0.05).to(DEV)
    bias = bf(torch.randn(N, generator=g)).to(DEV)
    res = bf(torch.randn(M, N, generator=g)).to(DEV)
    out = hk.gemm_nt(a, b, bias=bias, residual=res, act=act)
    z = a.float() @ b.float().t() + bias.float()
    if act == 2:
        z = F.gelu(z)
    assert rel_err(out, z + res.float()) < 4e-3
    c32 = hk.gemm_nt(a, b, out_f32=True)
    assert rel_err(c32, a.float() @ b.float().t()) < 1e-5


def test_gemm_256_tile_transpose_detecting_and_repeatable():
    M = N = 4096
    Kd = 128
    a = torch.zeros(M, Kd)
    a[torch.arange(M), torch.arange(M) % Kd] = 1.0
    b = (torch.arange(N * Kd, dtype=torch.float32).reshape(N, Kd) % 251) - 125.0
    A, B = bf(a).to(DEV), bf(b).to(DEV)
    ref = (A.float() @ B.float().t())
    first = hk.gemm_nt(A, B, out_f32=True)
    assert torch.equal(first, ref)
    for _ in range(20):  # race screen: the ring pipeline must give bit-identical results every launch
        assert torch.equal(hk.gemm_nt(A, B, out_f32=True), first)


def test_gemm_f32_accumulate_and_strided():
    M, N, Kd = 192, 256, 128
    g = torch.Generator().manual_seed(5)
    abig = bf(torch.randn(M, Kd * 2, generator=g)).to(DEV)
    a = abig[:, Kd:]  # strided view
    b = bf(torch.randn(N, Kd, generator=g)).to(DEV)
    c = torch.ones(M, N, device=DEV, dtype=torch.float32)
    hk.gemm_nt(a, b, out=c, out_f32=True, accumulate=True)
    ref = 1.0 + a.float() @ b.float().t()
    assert rel_err(c, ref) < 1e-5


def test_gemm_rejects_bad_k():
    a = torch.zeros(8, 48, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        hk.gemm_nt(a, a)


def test_gemm_lora_fused_and_fallback():
    g = torch.Generator().manual_seed(77)
    for (M, N, Kd, K2) in [(4095, 4096, 4096, 64), (300, 512, 256, 128)]:   # 256-tile fused path / small two-launch path
        a = bf(torch.randn(M, Kd, generator=g)).to(DEV); b = bf(torch.randn(N, Kd, generator=g) * 0.05).to(DEV)
        a2 = bf(torch.randn(M, K2, generator=g)).to(DEV); b2 = bf(torch.randn(N, K2, generator=g) * 0.05).to(DEV)
        res = bf(torch.randn(M, N, generator=g)).to(DEV)
        out = hk.gemm_nt_lora(a, b, a2, b2, residual=res, alpha=0.5)
        ref = 0.5 * (a.float() @ b.float().t() + a2.float() @ b2.float().t()) + res.float()
        assert rel_err(out, ref) < 5e-3
        o32 = hk.gemm_nt_lora(a, b, a2, b2, out_f32=True)
        assert rel_err(o32, a.float() @ b.float().t() + a2.float() @ b2.float().t()) < 1e-5


@pytest.mark.parametrize("M,N,KP", [(8190, 4096, 64), (1000, 12288, 128), (70, 64, 384)])
def test_gemm_tn_skinny(M, N, KP):
    g = torch.Generator().manual_seed(M + KP)
    pbig = bf(torch.randn(M, KP + 64, generator=g)).to(DEV)
    p = pbig[:, 32:32 + KP] if False else pbig[:, :KP]          # strided rows
    q = bf(torch.randn(M, N, generator=g)).to(DEV)
    out = torch.full((KP, N), 7.0, device=DEV)
    hk.gemm_tn_skinny(p, q, out)
    ref = p.float().t() @ q.float()
    assert rel_err(out, ref) < 1e-5
    hk.gemm_tn_skinny(p, q, out, accumulate=True)
    assert rel_err(out, 2 * ref) < 1e-5
    m = hk.blockdiag_mask(ref.clone(), 16, N // (KP // 16) if N % (KP // 16) == 0 else N, 0b101)
    rr = torch.arange(KP, device=DEV)[:, None] // 16
    cc = torch.arange(N, device=DEV)[None, :] // (N // (KP // 16) if N % (KP // 16) == 0 else N)
    keep = (rr == cc) & (((0b101 >> rr) & 1) == 1)
    assert torch.equal(m, torch.where(keep, ref, torch.zeros_like(ref)))


# ------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("rows,cols", [(7, 1024), (2057, 1024), (100, 512)])
def test_layernorm_fwd_bwd(rows, cols):
    g = torch.Generator().manual_seed(rows)
    x = bf(torch.randn(rows, cols, generator=g) * 2 + 0.5).to(DEV)
    gamma = bf(torch.randn(cols, generator=g)).to(DEV)
    beta = bf(torch.randn(cols, generator=g)).to(DEV)
    dy = bf(torch.randn(rows, cols, generator=g)).to(DEV)
    y, mean, rstd = hk.layernorm_fwd(x, gamma, beta, save_stats=True)
    xr = x.float().requires_grad_(True)
    gr = gamma.float().requires_grad_(True)
    br = beta.float().requires_grad_(True)
    yr = F.layer_norm(xr, (cols,), gr, br, 1e-5)
    assert rel_err(y, yr) < 4e-3
    yr.backward(dy.float())
    dgamma = torch.empty(cols, device=DEV)
    dbeta = torch.empty(cols, device=DEV)
    dx = hk.layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta)
    assert rel_err(dx, xr.grad) < 4e-3
    assert rel_err(dgamma, gr.grad) < 1e-4
    assert rel_err(dbeta, br.grad) < 1e-4


@pytest.mark.parametrize("rows", [5, 2184])
def test_rmsnorm_fwd_bwd(rows):
    cols = 4096
    g = torch.Generator().manual_seed(rows)
    x = bf(torch.randn(rows, cols, generator=g)).to(DEV)
    w = bf(1 + 0.1 * torch.randn(cols, generator=g)).to(DEV)
    dy = bf(torch.randn(rows, cols, generator=g)).to(DEV)
    add = bf(torch.randn(rows, cols, generator=g)).to(DEV)
    y, rstd = hk.rmsnorm_fwd(x, w, save_rstd=True)
    xr = x.float().requires_grad_(True)
    yr = w.float() * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5))
    assert rel_err(y, yr) < 6e-3
    yr.backward(dy.float())
    dx = hk.rmsnorm_bwd(dy, x, w, rstd, add=add)
    assert rel_err(dx, xr.grad + add.float()) < 4e-3
    dx2 = hk.rmsnorm_bwd(dy, x, w, None)
    assert rel_err(dx2, xr.grad) < 4e-3


# ------------------------------------------------------------------------------------------- attention
def ref_attention(q, k, v, causal, kv_len, scale):
    # q [Lq, H, D] ... fp32 reference
    s = torch.einsum("qhd,khd->hqk", q, k) * scale
    Lq, Lk = q.shape[0], k.shape[0]
    mask = torch.zeros(Lq, Lk, dtype=torch.bool, device=q.device)
    mask[:, kv_len:] = True
    if causal:
        mask |= torch.triu(torch.ones(Lq, Lk, dtype=torch.bool, device=q.device), 1)
    s = s.masked_fill(mask, float("-inf"))
    p = torch.softmax(s, -1)
    return torch.einsum("hqk,khd->qhd", p, v)


def run_attention_case(D, H, seqs, causal, same_qkv_buffer):
    """seqs: list of (Lq, Lkv, kv_len_valid)."""
    g = torch.Generator().manual_seed(D + H + len(seqs))
    tq = sum(s[0] for s in seqs)
    tk = sum(s[1] for s in seqs)
    scale = 1.0 / math.sqrt(D)
    q = bf(torch.randn(tq, H * D, generator=g)).to(DEV)
    k = bf(torch.randn(tk, H * D, generator=g)).to(DEV)
    v = bf(torch.randn(tk, H * D, generator=g)).to(DEV)
    do = bf(torch.randn(tq, H * D, generator=g)).to(DEV)
    entries, qo, ko = [], 0, 0
    for (lq, lk, kvl) in seqs:
        entries.append((qo, lq, ko, kvl, lk, 0))
        qo += lq
        ko += lk
    nseq = len(seqs)
    desc = hk.make_desc(entries, DEV)
    max_q = max(s[0] for s in seqs)
    max_kv = max(s[1] for s in seqs)
    LTq, LTkv = hk.pad64(max_q), hk.pad64(max_kv)
    vT = hk.seq_transpose(v, H * D, LTkv, desc, nseq, "kv")   # utility kernel, checked below
    for i, (lq, lk, kvl) in enumerate(seqs):
        koff = sum(s_[1] for s_ in seqs[:i])
        assert torch.equal(vT[i, :, :kvl], v[koff:koff + kvl].t()) and torch.all(vT[i, :, kvl:] == 0)
    o = torch.zeros(tq, H * D, device=DEV, dtype=torch.bfloat16)
    lse = torch.zeros(nseq, H, LTq, device=DEV, dtype=torch.float32)
    hk.attn_fwd(q, k, v, o, lse, desc, nseq, H, D, max_q, max_kv, LTq, causal, scale)
    # backward
    delta = torch.zeros(nseq, H, LTq, device=DEV, dtype=torch.float32)
    hk.attn_delta(o, do, delta, desc, nseq, H, D, max_q, LTq)
    dq = torch.full_like(q, float("nan"))
    dk = torch.full_like(k, float("nan"))
    dv = torch.full_like(v, float("nan"))
    hk.attn_bwd(q, k, v, do, lse, delta, dq, dk, dv, desc, nseq, H, D, max_q, max_kv, LTq, causal, scale)
    torch.cuda.synchronize()
    qo = ko = 0
    for (lq, lk, kvl) in seqs:
        qr = q[qo:qo + lq].float().view(lq, H, D).requires_grad_(True)
        kr = k[ko:ko + lk].float().view(lk, H, D).requires_grad_(True)
        vr = v[ko:ko + lk].float().view(lk, H, D).requires_grad_(True)
        ref = ref_attention(qr, kr, vr, causal, kvl, scale)
        got = o[qo:qo + lq].float().view(lq, H, D)
        assert rel_err(got, ref) < 8e-3, "forward"
        ref.backward(do[qo:qo + lq].float().view(lq, H, D))
        assert rel_err(dq[qo:qo + lq].view(lq, H, D), qr.grad) < 1.5e-2, "dq"
        assert rel_err(dk[ko:ko + lk].view(lk, H, D), kr.grad) < 1.5e-2, "dk"
        assert rel_err(dv[ko:ko + lk].view(lk, H, D), vr.grad) < 1.5e-2, "dv"
        # padded keys get exactly zero gradient
        assert torch.all(dk[ko + kvl:ko + lk] == 0) and torch.all(dv[ko + kvl:ko + lk] == 0)
        qo += lq
        ko += lk


def test_attention_vit_shape():
    run_attention_case(64, 16, [(257, 257, 257)] * 2, causal=False, same_qkv_buffer=False)


def test_attention_pooler_groups():
    run_attention_case(64, 16, [(64, 320, 320), (48, 304, 304), (32, 288, 288)] * 2, causal=False, same_qkv_buffer=False)


def test_attention_llama_causal():
    run_attention_case(128, 32, [(273, 273, 273), (273, 273, 250)], causal=True, same_qkv_buffer=False)


def test_attention_long_sequences_use_tiled_kernels():
    run_attention_case(128, 2, [(700, 700, 650), (330, 330, 330)], causal=True, same_qkv_buffer=False)
    run_attention_case(64, 4, [(100, 900, 900)], causal=False, same_qkv_buffer=False)


@pytest.mark.parametrize("D,H,seqs,causal", [(128, 4, [(273, 273, 273), (273, 273, 250), (65, 65, 3)], True), (64, 4, [(64, 320, 320), (48, 304, 304)], False),
                                              (128, 2, [(700, 700, 650)], True)])
def test_attention_backward_with_fused_delta(D, H, seqs, causal):
    """lhrs_attn_bwd_o (delta = rowsum(dO * O) inside the resident dQ kernel; the delta kernel launched internally for long sequences) against
    lhrs_attn_delta + lhrs_attn_bwd: delta to fp32 summation-order noise, dq / dk / dv to bf16 rounding, with and without the fused inverse RoPE."""
    g = torch.Generator().manual_seed(D + len(seqs))
    tq, tk = sum(s[0] for s in seqs), sum(s[1] for s in seqs)
    scale = 1.0 / math.sqrt(D)
    q, k, v, do = (bf(torch.randn(n, H * D, generator=g)).to(DEV) for n in (tq, tk, tk, tq))
    entries, qo, ko = [], 0, 0
    for (lq, lk, kvl) in seqs:
        entries.append((qo, lq, ko, kvl, lk, 0)); qo += lq; ko += lk
    nseq, desc = len(seqs), hk.make_desc(entries, DEV)
    max_q, max_kv = max(s[0] for s in seqs), max(s[1] for s in seqs)
    LTq = hk.pad64(max_q)
    o = torch.zeros(tq, H * D, device=DEV, dtype=torch.bfloat16)
    lse = torch.zeros(nseq, H, LTq, device=DEV)
    hk.attn_fwd(q, k, v, o, lse, desc, nseq, H, D, max_q, max_kv, LTq, causal, scale)
    S = seqs[0][0]
    ropes = [None]
    if D == 128 and all(s_[0] == s_[1] == S for s_ in seqs):
        pos = torch.arange(S, device=DEV, dtype=torch.float32)[:, None] * (10000.0 ** (-torch.arange(0, D, 2, device=DEV, dtype=torch.float32) / D))[None]
        ropes.append((pos.cos().contiguous(), pos.sin().contiguous(), S, 0))
    for rope in ropes:
        delta_a = torch.zeros(nseq, H, LTq, device=DEV)
        hk.attn_delta(o, do, delta_a, desc, nseq, H, D, max_q, LTq)
        ga = [torch.zeros_like(t) for t in (q, k, v)]
        hk.attn_bwd(q, k, v, do, lse, delta_a, *ga, desc, nseq, H, D, max_q, max_kv, LTq, causal, scale, rope=rope)
        delta_b = torch.full((nseq, H, LTq), float("nan"), device=DEV)
        gb = [torch.zeros_like(t) for t in (q, k, v)]
        hk.attn_bwd_o(q, k, v, do, o, lse, delta_b, *gb, desc, nseq, H, D, max_q, max_kv, LTq, causal, scale, rope=rope)
        for i, (lq, _, _) in enumerate(seqs):
            assert torch.allclose(delta_b[i, :, :lq], delta_a[i, :, :lq], rtol=1e-5, atol=1e-5)
        for x, y, name in zip(gb, ga, ("dq", "dk", "dv")):
            assert rel_err(x, y) < 2e-3, name


def test_attention_small_ragged():
    run_attention_case(128, 2, [(1, 1, 1), (17, 17, 17), (64, 64, 64), (65, 65, 3)], causal=True, same_qkv_buffer=False)


# ------------------------------------------------------------------------------------------- element-wise
def test_rope_roundtrip_and_reference():
    rows, H, D, S = 546, 64, 128, 273
    g = torch.Generator().manual_seed(3)
    x = bf(torch.randn(rows, H * D + 64, generator=g)).to(DEV)  # extra columns must stay untouched
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))
    fr = torch.outer(torch.arange(S).float(), inv)
    cos_t = fr.cos().to(torch.bfloat16).float().to(DEV)
    sin_t = fr.sin().to(torch.bfloat16).float().to(DEV)
    x0 = x.clone()
    hk.rope_(x, rows, H, D, cos_t, sin_t, pos_mod=S)
    xr = x0[:, :H * D].float().view(rows, H, D)
    pos = torch.arange(rows, device=DEV) % S
    c = torch.cat([cos_t, cos_t], -1)[pos][:, None, :]
    s = torch.cat([sin_t, sin_t], -1)[pos][:, None, :]
    rot = torch.cat([-xr[..., D // 2:], xr[..., :D // 2]], -1)
    ref = xr * c + rot * s
    assert rel_err(x[:, :H * D].view(rows, H, D), ref) < 4e-3
    assert torch.equal(x[:, H * D:], x0[:, H * D:])
    hk.rope_(x, rows, H, D, cos_t, sin_t, pos_mod=S, inverse=True)
    # forward then inverse is the identity up to bf16 rounding and cos^2+sin^2 of bf16-rounded tables
    assert rel_err(x[:, :H * D], x0[:, :H * D]) < 1e-2


def test_swiglu_fwd_bwd():
    rows, Fd = 301, 11008
    g = torch.Generator().manual_seed(4)
    gu = bf(torch.randn(rows, 2 * Fd, generator=g)).to(DEV)
    da = bf(torch.randn(rows, Fd, generator=g)).to(DEV)
    act = hk.swiglu_fwd(gu, Fd)
    gr = gu.float().requires_grad_(True)
    ref = F.silu(gr[:, :Fd]) * gr[:, Fd:]
    assert rel_err(act, ref) < 4e-3
    ref.backward(da.float())
    dgu = hk.swiglu_bwd(da, gu, Fd)
    assert rel_err(dgu, gr.grad) < 4e-3


def test_map_colsum_transpose_cast():
    g = torch.Generator().manual_seed(6)
    a = bf(torch.randn(333, 4096, generator=g)).to(DEV)
    b = bf(torch.randn(333, 4096, generator=g)).to(DEV)
    assert rel_err(hk.map_(hk.MAP_GELU, a), F.gelu(a.float())) < 4e-3
    assert rel_err(hk.map_(hk.MAP_ADD, a, b), a.float() + b.float()) < 4e-3
    xr = b.float().requires_grad_(True)
    F.gelu(xr).backward(a.float())
    assert rel_err(hk.map_(hk.MAP_GELU_BWD, a, b), xr.grad) < 4e-3
    out = torch.empty(4096, device=DEV)
    hk.colsum(a, out)
    assert rel_err(out, a.float().sum(0)) < 1e-5
    t = hk.transpose(a[:, :1000], rows_pad=384)
    assert torch.equal(t[:, :333], a[:, :1000].t()) and torch.all(t[:, 333:] == 0)
    f = torch.randn(1000, device=DEV)
    assert torch.equal(hk.cast_f32_to_bf16(f), f.to(torch.bfloat16))


def test_patchify_assemble():
    B = 3
    g = torch.Generator().manual_seed(8)
    rgb = torch.randn(B, 3, 224, 224, generator=g).to(DEV)
    w = torch.randn(1024, 3, 14, 14, generator=g).to(DEV) * 0.02
    patches = hk.patchify(rgb)
    wp = torch.zeros(1024, 640, device=DEV)
    wp[:, :588] = w.reshape(1024, 588)
    emb = hk.gemm_nt(patches, bf(wp))
    ref = F.conv2d(bf(rgb).float(), bf(w).float(), stride=14).flatten(2).transpose(1, 2).reshape(B * 256, 1024)
    assert rel_err(emb, ref) < 4e-3
    cls = bf(torch.randn(1024, generator=g)).to(DEV)
    pos = bf(torch.randn(257, 1024, generator=g)).to(DEV)
    full = hk.vit_assemble(emb, cls, pos, B, 256, 1024).view(B, 257, 1024)
    ref_full = torch.cat([cls.float().expand(B, 1, 1024), emb.float().view(B, 256, 1024)], 1) + pos.float()
    assert rel_err(full, ref_full) < 4e-3


# ------------------------------------------------------------------------------------------- token side
def test_cross_entropy():
    n, V = 77, 32000
    g = torch.Generator().manual_seed(9)
    logits = bf(torch.randn(n, V, generator=g) * 3).to(DEV)
    tgt = torch.randint(0, V, (n,), generator=g).to(DEV)
    lr = logits.float().requires_grad_(True)
    ref = F.cross_entropy(lr, tgt)
    ref.backward()
    loss, dl = hk.cross_entropy(logits.clone(), tgt.to(torch.int32), inplace=True)
    assert abs(loss.item() - ref.item()) < 2e-4 * abs(ref.item())
    assert rel_err(dl, lr.grad) < 5e-3


def test_gather_scatter_rows():
    src = bf(torch.randn(50, 4096)).to(DEV)
    idx = torch.tensor([3, 49, 0, 7], dtype=torch.int32, device=DEV)
    got = hk.gather_rows(src, idx)
    assert torch.equal(got, src[idx.long()])
    dst = torch.zeros_like(src)
    hk.scatter_rows(got, idx, dst)
    assert torch.equal(dst[idx.long()], got)
    keep = torch.ones(50, dtype=torch.bool, device=DEV)
    keep[idx.long()] = False
    assert torch.all(dst[keep] == 0)


# ------------------------------------------------------------------------------------------- optimizer
def test_adan_and_clip_match_restatement():
    from oracle.optim_oracle import adan_step_ref

    n = 10007
    g = torch.Generator().manual_seed(11)
    p = torch.randn(n, generator=g)
    state = dict(p=p.clone().double(), m=torch.zeros(n).double(), v=torch.zeros(n).double(), n=torch.zeros(n).double(), pre=None)
    dp = p.clone().to(DEV)
    dm, dv, dn, dpre = (torch.zeros(n, device=DEV) for _ in range(4))
    shadow = torch.zeros(n, device=DEV, dtype=torch.bfloat16)
    gn = torch.zeros((), device=DEV)
    for step in range(1, 5):
        grad = torch.randn(n, generator=g) * (10.0 if step == 2 else 0.1)
        adan_step_ref(state, grad.double(), step, lr=2e-4, wd=0.02, max_norm=0.3)
        dg = grad.to(DEV)
        hk.sqnorm(dg, gn)
        hk.adan_step(dp, dg, dm, dv, dn, dpre, shadow, step, 2e-4, wd=0.02, gnorm_sq=gn, max_norm=0.3)
        assert (dp.cpu().double() - state["p"]).abs().max() < 1e-6
    assert torch.equal(shadow, dp.to(torch.bfloat16))


@pytest.mark.parametrize("M,lora", [(8190, False), (8190, True), (4095, False), (300, True), (2184, False), (2184, True)])
def test_gemm_fused_swiglu_bit_identical_to_unfused(M, lora):
    """lhrs_gemm_swiglu_fwd / _bwd == GEMM + swiglu kernels, bit for bit (fused 16-wave kernel at M >= 4095, fallback below; M = 2184, the
    reference's micro-batch 8: 774 tiles = 3 rounds + 6 - the tail-row rule sends rows 2048.. through the small-tile GEMM + SwiGLU kernel)."""
    g = torch.Generator().manual_seed(M + lora)
    d, ff, KP = 4096, 11008, 128
    x = torch.randn(M, d, generator=g).to(DEV, torch.bfloat16)
    wgu = (torch.randn(2 * ff, d, generator=g) * 0.02).to(DEV, torch.bfloat16)
    wdT = (torch.randn(ff, d, generator=g) * 0.02).to(DEV, torch.bfloat16)
    dy = (torch.randn(M, d, generator=g) * 0.1).to(DEV, torch.bfloat16)
    a2 = b2 = a2b = b2b = None
    if lora:
        a2 = (torch.randn(M, KP, generator=g) * 0.1).to(DEV, torch.bfloat16)
        b2 = (torch.randn(2 * ff, KP, generator=g) * 0.05).to(DEV, torch.bfloat16)
        a2b = (torch.randn(M, KP, generator=g) * 0.1).to(DEV, torch.bfloat16)
        b2b = (torch.randn(ff, KP, generator=g) * 0.05).to(DEV, torch.bfloat16)
    gu, act = hk.gemm_swiglu_fwd(x, wgu, ff, a2, b2)
    gu_ref = hk.gemm_nt_lora(x, wgu, a2, b2) if lora else hk.gemm_nt(x, wgu)
    act_ref = hk.swiglu_fwd(gu_ref, ff)
    assert torch.equal(gu, gu_ref) and torch.equal(act, act_ref)
    dact_ref = hk.gemm_nt_lora(dy, wdT, a2b, b2b) if lora else hk.gemm_nt(dy, wdT)
    dgu_ref = hk.swiglu_bwd(dact_ref, gu_ref, ff)
    dgu = hk.gemm_swiglu_bwd(dy, wdT, gu, ff, a2b, b2b)          # in place over gu
    assert dgu.data_ptr() == gu.data_ptr() and torch.equal(dgu, dgu_ref)
    if not lora and M >= 1024:
        # the four-wave kernel's SwiGLU epilogues (csrc/gemm_u4.hip: gate column c and up column c in ONE lane - the weight image is [64 gate | 64 up] per wave -, written out
        # inside the next tile's first stage): raw launches, then the operator path under the shape rule - all bit-identical to the unfused kernel pairs
        from lhrs_bot_amd import _lib
        lib, st = _lib.load(), torch.cuda.current_stream().cuda_stream
        gu4, act4 = torch.zeros_like(gu_ref), torch.zeros_like(act_ref)
        assert lib.lhrs_gemm_u4_swiglu_fwd(x.data_ptr(), x.stride(0), wgu.data_ptr(), wgu.stride(0), gu4.data_ptr(), gu4.stride(0), act4.data_ptr(), act4.stride(0),
                                           M, ff, d, st) == 0
        assert torch.equal(gu4, gu_ref) and torch.equal(act4, act_ref)
        assert lib.lhrs_gemm_u4_swiglu_bwd(dy.data_ptr(), dy.stride(0), wdT.data_ptr(), wdT.stride(0), gu4.data_ptr(), gu4.data_ptr(), gu4.stride(0), M, ff, d, st) == 0
        assert torch.equal(gu4, dgu_ref)                              # in place over gate|up
        hk.gemm_set_u4(True)
        taken_f, taken_b = bool(lib.lhrs_gemm_u4_fused_takes(1, M, ff // 128, d, 0)), bool(lib.lhrs_gemm_u4_fused_takes(2, M, -(-ff // 256), d, 0))
        assert taken_f == (M >= 4095) and not taken_b                 # M = 2184 (3.02 rounds) stays on the 16-wave kernels; SwiGLU' (2.69 / 5.375 rounds) needs <= 5 % idle
        assert lib.lhrs_gemm_u4_fused_takes(2, 16380, -(-ff // 256), d, 0) == 1     # micro-batch 60: 10.75 rounds
        gu5, act5 = hk.gemm_swiglu_fwd(x, wgu, ff)
        assert torch.equal(gu5, gu_ref) and torch.equal(act5, act_ref)
        assert torch.equal(hk.gemm_swiglu_bwd(dy, wdT, gu5, ff), dgu_ref)
        hk.gemm_set_u4(False)


@pytest.mark.parametrize("M,N,K", [(8190, 64, 4096), (4095, 128, 11008), (8190, 384, 4096), (2000, 64, 22016), (300, 64, 4096)])
def test_gemm_nt_skinny_split_k(M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
    b = (torch.randn(N, K, generator=g) * 0.05).to(DEV, torch.bfloat16)
    y = hk.gemm_nt_skinny(a, b, alpha=2.0)
    ref = 2.0 * (a.float() @ b.float().t())
    assert y.shape == (M, N) and rel_err(y, ref) < 4e-3


@pytest.mark.parametrize("V,ld", [(32000, 32000), (32003, 32003), (32000, 32004), (257, 260), (5, 5)])
def test_argmax_rows_first_maximum(V, ld):
    """HF greedy pick: the LOWEST index among equal maxima; aligned (16-B vector loads) and unaligned rows, ties across threads."""
    g = torch.Generator().manual_seed(V)
    n = 3
    buf = torch.randn(n, ld, generator=g)
    x = buf[:, :V]
    x[0, [V - 1, V // 2, 3 % V]] = 9.0        # three equal maxima: the first one wins
    x[1, :] = -1.5                             # a constant row: index 0
    x[2, V - 1] = 50.0                         # the very last element
    want = torch.from_numpy(x.numpy().argmax(-1))
    got = hk.argmax_rows(buf.to("cuda")[:, :V])
    assert torch.equal(got.cpu(), want), (got, want)


@pytest.mark.parametrize("M,S,pos0,lora", [(8190, 273, 0, False), (8190, 273, 0, True), (4095, 195, 7, False), (300, 100, 0, True)])
def test_gemm_fused_rope_bit_identical_to_unfused(M, S, pos0, lora):
    """lhrs_gemm_rope_fwd == GEMM (+ LoRA pair) followed by lhrs_rope, bit for bit: q / k heads rotated in the epilogue of the 16-wave
    kernel (M >= 4095), v columns untouched; the small case takes the documented fallback."""
    g = torch.Generator().manual_seed(M + S + lora)
    d, hd, KP = 4096, 128, 64
    x = torch.randn(M, d, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(3 * d, d, generator=g) * 0.02).to(DEV, torch.bfloat16)
    a2 = b2 = None
    if lora:
        a2 = (torch.randn(M, KP, generator=g) * 0.1).to(DEV, torch.bfloat16)
        b2 = (torch.randn(3 * d, KP, generator=g) * 0.05).to(DEV, torch.bfloat16)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    fr = torch.outer(torch.arange(512).float(), inv)
    cos, sin = fr.cos().to(DEV).contiguous(), fr.sin().to(DEV).contiguous()
    got = hk.gemm_rope_fwd(x, w, cos, sin, pos_mod=S, pos0=pos0, rope_cols=2 * d, head_dim=hd, a2=a2, b2=b2)
    if M >= 1024:   # the four-wave kernel's RoPE variant (csrc/gemm_u4.hip: partners d / d + 64 in one lane; the LoRA pair as more stages of its k-loop): raw launch, then the operator path under the shape rule
        from lhrs_bot_amd import _lib
        raw = torch.zeros_like(got)
        st = torch.cuda.current_stream().cuda_stream
        if lora:
            assert _lib.load().lhrs_gemm_u4_rope_lora(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), a2.data_ptr(), a2.stride(0), b2.data_ptr(), b2.stride(0), KP,
                                                      raw.data_ptr(), raw.stride(0), M, 3 * d, d, cos.data_ptr(), sin.data_ptr(), S, pos0, 2 * d, st) == 0
        else:
            assert _lib.load().lhrs_gemm_u4_rope(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), raw.data_ptr(), raw.stride(0), M, 3 * d, d, cos.data_ptr(),
                                                 sin.data_ptr(), S, pos0, 2 * d, st) == 0
        assert torch.equal(raw, got)
        hk.gemm_set_u4(True)
        assert _lib.load().lhrs_gemm_u4_fused_takes(0, M, 3 * d // 256, d, KP if lora else 0) == 1
        with hk.gemm_kernel_census() as census:
            assert torch.equal(hk.gemm_rope_fwd(x, w, cos, sin, pos_mod=S, pos0=pos0, rope_cols=2 * d, head_dim=hd, a2=a2, b2=b2), got)
        assert list(census.counts) == ["gemm_u4_kernel<3, false> RoPE"], census.counts
        hk.gemm_set_u4(False)
    ref = hk.gemm_nt_lora(x, w, a2, b2) if lora else hk.gemm_nt(x, w)
    plain = ref.clone()
    hk.rope_(ref, M, 2 * d // hd, hd, cos, sin, pos_mod=S, pos0=pos0)
    assert torch.equal(got, ref)
    assert torch.equal(got[:, 2 * d:], plain[:, 2 * d:]) and not torch.equal(got[:, :2 * d], plain[:, :2 * d])


@pytest.mark.parametrize("M,N,K,lora", [(8736, 4096, 4096, False), (8736, 4096, 11008, True), (4368, 4096, 4096, False), (8736, 2048, 1024, False)])
def test_gemm_tail_rows_take_the_small_tile_kernel(M, N, K, lora):
    """A nearly empty last round of 256x256 tiles (560 tiles = 2.19 rounds of the 256 CUs) is cut off: whole tile rows on the 16-wave
    kernel, the spill-over rows on the small-tile kernel.  Same result as the single launch (and as the fp32 reference)."""
    from lhrs_bot_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
    b = (torch.randn(N, K, generator=g) * 0.02).to(DEV, torch.bfloat16)
    r = torch.randn(M, N, generator=g).to(DEV, torch.bfloat16)
    a2 = b2 = None
    if lora:
        a2 = (torch.randn(M, 64, generator=g) * 0.1).to(DEV, torch.bfloat16)
        b2 = (torch.randn(N, 64, generator=g) * 0.05).to(DEV, torch.bfloat16)
    run = (lambda: hk.gemm_nt_lora(a, b, a2, b2, residual=r)) if lora else (lambda: hk.gemm_nt(a, b, residual=r))
    try:
        lib.lhrs_gemm_set_tail_split(0)
        whole = run()
    finally:
        lib.lhrs_gemm_set_tail_split(1)
    split = run()
    ref = a.float() @ b.float().t() + r.float() + ((a2.float() @ b2.float().t()) if lora else 0)
    assert rel_err(split.float(), ref) < 1e-2 and rel_err(whole.float(), ref) < 1e-2
    assert rel_err(split.float(), whole.float()) < 2e-3
    f32 = hk.gemm_nt(a, b, out_f32=True)
    assert rel_err(f32, a.float() @ b.float().t()) < 2e-3


@pytest.mark.parametrize("M,N,K", [(2048, 1024, 27392), (1024, 1024, 4352), (4096, 1024, 4352), (1024, 4096, 4352), (256, 128, 640)])
def test_gemm_splitk_f32_weight_gradient_shapes(M, N, K):
    """Long-K products with few output tiles (the projector's dW = dY^T X over the token dimension): split-K slabs + ordered sum."""
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 0.1).to(DEV, torch.bfloat16)
    b = (torch.randn(N, K, generator=g) * 0.1).to(DEV, torch.bfloat16)
    buf = torch.zeros(M + 3, N, device=DEV, dtype=torch.float32)
    out = buf[3:]                                   # a row range of a larger fp32 gradient buffer
    hk.gemm_nt_splitk_f32(a, b, out)
    ref = a.float() @ b.float().t()
    assert rel_err(out, ref) < 1e-5 and float(buf[:3].abs().max()) == 0.0
    again = torch.empty_like(out)
    hk.gemm_nt_splitk_f32(a, b, again)
    assert torch.equal(out, again)                  # fixed summation order: bit-reproducible
    assert rel_err(out, hk.gemm_nt(a, b, out_f32=True)) < 1e-5


def test_host_splice_integers_equal_the_device_splice():
    """TextModal.splice_ints_host (what the training step uses for its host-side bookkeeping) == the integer outputs of lhrs_splice_fwd
    (pinned bit-exact to the reference): ragged right-padded batches, samples without an <image> token, missing mask / labels."""
    from lhrs_bot_amd.text import TextModal
    g = torch.Generator().manual_seed(11)
    NI, dim, V = 7, 64, 500
    embed = torch.randn(V, dim, generator=g).to(DEV, torch.bfloat16)
    for trial in range(12):
        B, T = int(torch.randint(1, 6, (1,), generator=g)), int(torch.randint(2, 19, (1,), generator=g))
        ids = torch.randint(1, V, (B, T), generator=g)
        labels = torch.randint(-1, V, (B, T), generator=g)
        labels[labels < 0] = -100
        lens = torch.randint(1, T + 1, (B,), generator=g)
        mask = (torch.arange(T)[None, :] < lens[:, None])
        with_img = torch.rand(B, generator=g) < (0.0 if trial == 0 else 0.75)
        for b in range(B):
            if with_img[b]:
                ids[b, int(torch.randint(0, int(lens[b]), (1,), generator=g))] = -200
        image = torch.randn(B, NI, dim, generator=g).to(DEV, torch.bfloat16)
        for lab, msk in ((labels, mask), (None, mask), (labels, None)):
            S, has, nl, nm = TextModal.splice_ints_host(ids, lab, msk, NI)
            assert S == (T - 1 + NI if bool(with_img.any()) else T) and torch.equal(has, with_img)
            _, dl, dm, pos = hk.splice_fwd(ids.to(DEV), None if lab is None else lab.to(DEV), None if msk is None else msk.to(DEV), image, embed, S)
            assert torch.equal(dl.cpu(), nl) and torch.equal(dm.cpu(), nm), trial
            assert torch.equal(pos.cpu() >= 0, with_img)


@pytest.mark.parametrize("S,B,H", [(273, 3, 32), (130, 2, 4), (700, 1, 2)])   # resident kernels (twice), tiled kernels + separate pass
def test_attention_backward_with_fused_inverse_rope_is_bit_identical(S, B, H):
    """lhrs_attn_bwd_rope: dq / dk leave the kernel as gradients of the UN-rotated q / k projections (HF LlamaAttention rotates before the
    scores).  The rotation rides in the stores; it must equal lhrs_attn_bwd followed by lhrs_rope(inverse) bit for bit (same bf16
    rounding points), for the LDS-resident kernels of the training shapes and for the tiled fallback of long sequences."""
    D = 128
    d = H * D
    g = torch.Generator().manual_seed(S + B)
    M = B * S
    qkv = bf(torch.randn(M, 3 * d, generator=g) * 0.5).to(DEV)
    do = bf(torch.randn(M, d, generator=g) * 0.1).to(DEV)
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))
    fr = torch.outer(torch.arange(1024).float(), inv)
    cos_t, sin_t = fr.cos().to(torch.bfloat16).float().to(DEV).contiguous(), fr.sin().to(torch.bfloat16).float().to(DEV).contiguous()
    desc = hk.make_desc([(b * S, S, b * S, S if b else S - 7, S, 0) for b in range(B)], DEV)
    LT = hk.pad64(S)
    scale = 1.0 / math.sqrt(D)
    q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    o = torch.zeros(M, d, device=DEV, dtype=torch.bfloat16)
    lse = torch.zeros(B, H, LT, device=DEV, dtype=torch.float32)
    hk.attn_fwd(q, k, v, o, lse, desc, B, H, D, S, S, LT, True, scale)
    delta = torch.zeros(B, H, LT, device=DEV, dtype=torch.float32)
    hk.attn_delta(o, do, delta, desc, B, H, D, S, LT)
    ref = torch.zeros(M, 3 * d, device=DEV, dtype=torch.bfloat16)
    hk.attn_bwd(q, k, v, do, lse, delta, ref[:, :d], ref[:, d:2 * d], ref[:, 2 * d:], desc, B, H, D, S, S, LT, True, scale)
    plain = ref.clone()
    hk.rope_(ref, M, 2 * H, D, cos_t, sin_t, pos_mod=S, inverse=True)
    got = torch.zeros_like(ref)
    hk.attn_bwd(q, k, v, do, lse, delta, got[:, :d], got[:, d:2 * d], got[:, 2 * d:], desc, B, H, D, S, S, LT, True, scale,
                rope=(cos_t, sin_t, S, 0))
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
    assert torch.equal(got[:, 2 * d:], plain[:, 2 * d:]) and not torch.equal(got[:, :2 * d], plain[:, :2 * d])   # dv untouched, dq / dk rotated


@pytest.mark.parametrize("T,Mo,No", [(4320, 1024, 1024), (27360, 2048, 1024), (4320, 4096, 1024), (130, 128, 256), (64, 128, 128), (4321, 1024, 4096)])
def test_gemm_tn_f32_weight_gradient_from_token_major_operands(T, Mo, No):
    """lhrs_gemm_tn_f32: dW[out, in] = dY^T X straight from the token-major operands (transposing LDS reads) vs the fp64 product, and vs the
    round-1 path (transposed copies + NT split-K GEMM) it replaces; partial last 64-token stage (zero fill), strided operands, and a
    transpose-detecting input (out != in, asymmetric values)."""
    g = torch.Generator().manual_seed(T + Mo)
    big = bf(torch.randn(T, Mo + 64, generator=g) * 0.1).to(DEV)
    dy = big[:, :Mo]                                    # row stride Mo + 64: a column slice of a wider buffer
    x = bf(torch.randn(T, No, generator=g)).to(DEV)
    out = torch.full((Mo + 8, No), float("nan"), device=DEV)
    hk.gemm_tn_f32(dy, x, out[4:4 + Mo])
    torch.cuda.synchronize()
    ref = dy.double().t() @ x.double()
    assert rel_err(out[4:4 + Mo], ref) < 2e-6                                  # exact bf16 products, fp32 accumulation
    assert torch.isnan(out[:4]).all() and torch.isnan(out[4 + Mo:]).all()      # nothing outside the target rows is touched
    Mp = hk.pad64(T)
    old = torch.empty(Mo, No, device=DEV)
    hk.gemm_nt_splitk_f32(hk.transpose(dy.contiguous(), rows_pad=Mp), hk.transpose(x, rows_pad=Mp), old)
    assert rel_err(out[4:4 + Mo], old) < 2e-6
    again = torch.empty(Mo, No, device=DEV)
    hk.gemm_tn_f32(dy, x, again)
    assert torch.equal(again, out[4:4 + Mo])                                   # ordered slab sum: bit-reproducible
    with pytest.raises(RuntimeError):
        hk.gemm_tn_f32(dy[:, :Mo - 64], x, torch.empty(Mo - 64, No, device=DEV))   # Mo % 128 != 0 is rejected at the ABI


@pytest.mark.parametrize("M", [2184, 1000, 145, 144, 8190])
def test_gemm_144_row_tiles_bit_identical_to_256_row_tiles(M):
    """gemm_nt_144s_kernel (12 waves, 144 x 256 tiles: 256 tiles instead of 144 at M = 2184, the reference's micro-batch 8) against
    gemm_nt_256s_kernel on the same operands: every epilogue family - plain, bias + activation, residual, fused SwiGLU forward / backward,
    fused RoPE - must agree bit for bit (same k order, same fp32 accumulation per element), ragged last tile rows and columns included."""
    from lhrs_bot_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M)
    d, ff, hd = 4096, 2048, 128
    x = torch.randn(M, d, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(3 * d + 136, d, generator=g) * 0.02).to(DEV, torch.bfloat16)     # N = 12424: ragged last tile column
    wo = (torch.randn(d, d, generator=g) * 0.02).to(DEV, torch.bfloat16)
    bias = torch.randn(d, generator=g).to(DEV, torch.bfloat16)
    res = torch.randn(M, d, generator=g).to(DEV, torch.bfloat16)
    wgu = (torch.randn(2 * ff, d, generator=g) * 0.02).to(DEV, torch.bfloat16)
    wdT = (torch.randn(ff, d, generator=g) * 0.02).to(DEV, torch.bfloat16)
    dy = (torch.randn(M, d, generator=g) * 0.1).to(DEV, torch.bfloat16)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    fr = torch.outer(torch.arange(512).float(), inv)
    cos, sin = fr.cos().to(DEV).contiguous(), fr.sin().to(DEV).contiguous()

    def run():
        out = [hk.gemm_nt(x, w), hk.gemm_nt(x, wo, bias=bias, act=hk.ACT_GELU), hk.gemm_nt(x, wo, residual=res),
               hk.gemm_nt(x, wo, bias=bias, act=hk.ACT_QUICK_GELU, residual=res)]
        gu, act = hk.gemm_swiglu_fwd(x, wgu, ff)
        out += [gu.clone(), act]
        out.append(hk.gemm_swiglu_bwd(dy, wdT, gu, ff).clone())
        out.append(hk.gemm_rope_fwd(x, w[:3 * d], cos, sin, pos_mod=273, pos0=0, rope_cols=2 * d, head_dim=hd))
        return out

    try:
        lib.lhrs_gemm_set_min_tiles(1)       # both tile heights are legal from one tile on (the default threshold would take the small-tile kernel at M <= 1000)
        lib.lhrs_gemm_set_tail_split(0)
        lib.lhrs_gemm_set_bm144(0)
        ref = run()
        lib.lhrs_gemm_set_bm144(2)
        got = run()
    finally:
        lib.lhrs_gemm_set_bm144(1)
        lib.lhrs_gemm_set_tail_split(1)
        lib.lhrs_gemm_set_min_tiles(128)
    for i, (a, b) in enumerate(zip(got, ref)):
        assert torch.equal(a, b), (i, (a.float() - b.float()).abs().max().item())
    assert rel_err(ref[0], x.float() @ w.float().t()) < 4e-3


def test_gemm_tail_rows_split_k_matches_unsplit():
    """M = 8736 (micro-batch 32, BASELINE configs[3]): the 544 rows behind the last whole round of 256-row tiles are a separate launch; for the
    long k-loops (down K = 11008, d-gate|up K = 22016, d-qkv K = 12288) that launch is split-K over the registered workspace (f32 slabs, fixed
    summation order, residual added before the one rounding).  Against the same product with the row split switched off, and fp32 torch."""
    from lhrs_bot_amd import _lib
    lib = _lib.load()
    hk.ensure_gemm_workspace(DEV)
    g = torch.Generator().manual_seed(8736)
    M, N = 8736, 4096
    for K, with_res in ((11008, True), (22016, False), (12288, False), (4096, True)):
        x = (torch.randn(M, K, generator=g) * 0.5).to(DEV, torch.bfloat16)
        w = (torch.randn(N, K, generator=g) * 0.02).to(DEV, torch.bfloat16)
        res = torch.randn(M, N, generator=g).to(DEV, torch.bfloat16) if with_res else None
        got = hk.gemm_nt(x, w, residual=res)
        again = hk.gemm_nt(x, w, residual=res)
        try:
            lib.lhrs_gemm_set_tail_split(0)
            ref = hk.gemm_nt(x, w, residual=res)
        finally:
            lib.lhrs_gemm_set_tail_split(1)
        assert torch.equal(got, again)                                   # deterministic
        assert torch.equal(got[:8192], ref[:8192]), K                    # the rows of the whole rounds: same kernel, same tiles
        # tail rows: another k order, and ONE rounding behind the residual add where the 256-row kernel's epilogue rounds the product first (as the unfused
        # GEMM -> add sequence does): a bf16 ulp on a fraction of the elements
        assert rel_err(got[8192:], ref[8192:]) < 4e-3, (K, rel_err(got[8192:], ref[8192:]))
        want = x[8192:].float() @ w.float().t() + (res[8192:].float() if with_res else 0)
        assert rel_err(got[8192:], want) < 4e-3, K


def test_copy_2d_kernel_and_runtime_fallback():
    """lhrs_copy_2d: pitched device-to-device block copy - one kernel launch for 16-byte aligned blocks (the ViT taps, the K / V rows of a
    prefill), the runtime's pitched copy otherwise; both against torch slicing, bytes outside the block untouched."""
    g = torch.Generator().manual_seed(3)
    src = torch.randint(0, 255, (37, 4096), generator=g, dtype=torch.uint8).to(DEV)
    for (r0, c0, h, w) in ((0, 0, 37, 4096), (3, 32, 30, 2048), (1, 16, 5, 48), (2, 6, 7, 100), (0, 1, 37, 15)):     # the last two: not 16-byte aligned
        dst = torch.full((40, 8192), 7, dtype=torch.uint8, device=DEV)
        hk.copy_2d(dst.data_ptr() + 2 * 8192 + 64, 8192, src.data_ptr() + r0 * 4096 + c0, 4096, w, h)
        torch.cuda.synchronize()
        want = torch.full((40, 8192), 7, dtype=torch.uint8, device=DEV)
        want[2:2 + h, 64:64 + w] = src[r0:r0 + h, c0:c0 + w]
        assert torch.equal(dst, want), (r0, c0, h, w)


