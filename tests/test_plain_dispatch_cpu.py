"""Host-side rules of the plain long-k product dispatch (csrc/gemm.hip: lhrs_gemm_bf16_nt's shape rule; csrc/gemm_u4.hip) that need no GPU:
which problems the four-wave kernel takes - a pure function of the shape, so that runs are bit-reproducible and every data-parallel rank runs the same
kernels - and which problems its raw launch wrappers decline before they touch the device."""
from lhrs_bot_amd import _lib


def test_which_plain_products_take_the_four_wave_kernel_is_a_pure_shape_rule():
    lib = _lib.load()
    takes = lambda *a: lib.lhrs_gemm_u4_takes(*a)
    ok = (8190, 4096, 11008, 11008, 11008, 4096, 0, 0, 0, 0, 0, 1.0)        # M, N, K, lda, ldb, ldc, ldr, bias, act, f32 out, accumulate, alpha
    try:
        lib.lhrs_gemm_set_u4(1)
        assert takes(*ok) == 1
        assert all(takes(*ok) == 1 for _ in range(200))                                       # no state: the answer never changes (the old first-call timing had a 96-problem cap)
        assert takes(8190, 4096, 4096, 4096, 4096, 4104, 4104, 0, 0, 0, 0, 1.0) == 1          # strided output + residual, 16-B rows
        for i, v in ((7, 1), (8, 1), (9, 1), (10, 1), (11, 0.5)):                            # bias, activation, f32 output, accumulate, alpha != 1
            a = list(ok); a[i] = v
            assert takes(*a) == 0, a
        assert takes(1000, 4096, 11008, 11008, 11008, 4096, 0, 0, 0, 0, 0, 1.0) == 0          # M < 1024
        assert takes(8190, 1000, 11008, 11008, 11008, 1000, 0, 0, 0, 0, 0, 1.0) == 0          # N < 1024
        assert takes(8190, 4096, 2048, 2048, 2048, 4096, 0, 0, 0, 0, 0, 1.0) == 0             # short k-loop
        assert takes(8190, 4096, 4128, 4128, 4128, 4096, 0, 0, 0, 0, 0, 1.0) == 0             # K % 64 != 0
        assert takes(8190, 4096, 11008, 11008, 11008, 4100, 0, 0, 0, 0, 0, 1.0) == 0          # output rows not 16-B aligned
        # the tile count decides (>= 80 % of one round of the CUs): the reference's micro-batch 8 (M = 2184: 9 x 16 = 144 tiles) and the projector
        # product (M = 4320, N = 1024: 68 tiles) stay on the 144-row / small-tile kernels; ragged row counts around a boundary flip exactly once
        assert takes(2184, 4096, 11008, 11008, 11008, 4096, 0, 0, 0, 0, 0, 1.0) == 0
        assert takes(4320, 1024, 4096, 4096, 4096, 1024, 0, 0, 0, 0, 0, 1.0) == 0
        assert takes(3840, 4096, 22016, 22016, 22016, 4096, 0, 0, 0, 0, 0, 1.0) == 1          # 240 tiles
        assert takes(3822, 32000, 4096, 4096, 4096, 32000, 0, 0, 0, 0, 0, 1.0) == 1           # lm_head on the supervised rows
        got = [takes(M, 4096, 4096, 4096, 4096, 4096, 0, 0, 0, 0, 0, 1.0) for M in range(1024, 8192, 64)]
        assert got == sorted(got) and got[0] == 0 and got[-1] == 1
        lib.lhrs_gemm_set_u4(0)
        assert takes(*ok) == 0                                                                # kernel A/B switch: the 16-wave kernels everywhere
    finally:
        lib.lhrs_gemm_set_u4(1)


def test_four_wave_kernel_wrappers_decline_before_touching_the_device():
    lib = _lib.load()
    A, B, C = 0x10000000, 0x20000000, 0x30000000                                             # never dereferenced: every call below returns 1 first
    u4 = lambda a, lda, b, ldb, c, ldc, M, N, K, r=None, ldr=0: lib.lhrs_gemm_u4_nt(a, lda, b, ldb, c, ldc, M, N, K, r, ldr, None)
    assert u4(A, 96, B, 96, C, 4096, 2048, 4096, 96) == 1                                     # K % 64 != 0
    assert u4(A, 64, B, 64, C, 4096, 2048, 4096, 64) == 1                                     # K < 128
    assert u4(A, 4100, B, 4096, C, 4096, 2048, 4096, 4096) == 1                               # lda not a multiple of 8
    assert u4(A + 8, 4096, B, 4096, C, 4096, 2048, 4096, 4096) == 1                           # operand not 16-B aligned
    assert u4(A, 4096, B, 4096, C, 4096, 2048, 4098, 4096) == 1                               # N % 4 != 0
    assert u4(A, 4096, B, 4096, C, 2048, 2048, 4096, 4096) == 1                               # ldc < N
    assert u4(A, 32768, B, 32768, C, 4096, 70000, 4096, 32768) == 1                           # operand beyond the 32-bit lane offsets (4 GiB)
    assert u4(A, 4096, B, 4096, C, 4096, 2048, 4096, 4096, 0x40000004, 4096) == 1             # residual not 8-B aligned
    pair = lambda a2=0x70000000, lda2=64, b2=0x78000000, ldb2=64, K2=64, M=2048: lib.lhrs_gemm_u4_nt_lora(A, 4096, B, 4096, a2, lda2, b2, ldb2, K2, C, 4096, M, 4096,
                                                                                                          4096, None, 0, None)
    assert pair(K2=96) == 1 and pair(K2=-64) == 1 and pair(a2=None) == 1 and pair(b2=0x78000008) == 1        # the LoRA pair: K2 % 64, null / misaligned operands,
    assert pair(lda2=60) == 1 and pair(ldb2=32) == 1 and pair(K2=128) == 1                                    # rows shorter than K2 or not 16-B multiples
    rope = lambda rope_cols, cos=0x50000000, sin=0x60000000, pos_mod=273: lib.lhrs_gemm_u4_rope(A, 4096, B, 4096, C, 12288, 2048, 12288, 4096, cos, sin, pos_mod,
                                                                                              0, rope_cols, None)
    assert rope(8192 + 128) == 1 and rope(16384) == 1 and rope(8192, cos=None) == 1 and rope(8192, pos_mod=0) == 1 and rope(8192, sin=0x60000004) == 1
