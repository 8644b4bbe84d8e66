"""Host-side rules of the plain long-k product dispatch (csrc/gemm.hip: lhrs_gemm_bf16_nt's first-call timing; csrc/gemm_u4.hip; csrc/vendor.cpp) that need no GPU:
which problems are timed at all, and which problems the four-wave kernel's wrappers decline before they touch the device."""
import ctypes

from lhrs_bot_amd import _lib


def test_which_plain_products_are_decided_by_first_call_timing():
    lib = _lib.load()
    takes = lambda *a: lib.lhrs_gemm_vendor_takes(*a)
    ok = (8190, 4096, 11008, 11008, 11008, 4096, 0, 0, 0, 0, 0, 1.0)        # M, N, K, lda, ldb, ldc, ldr, bias, act, f32 out, accumulate, alpha
    try:
        lib.lhrs_gemm_set_vendor(1, 4096)
        assert takes(*ok) == 1
        assert takes(8190, 4096, 4096, 4096, 4096, 4104, 4104, 0, 0, 0, 0, 1.0) == 1          # strided output + residual, 16-B rows
        for i, v in ((7, 1), (8, 1), (9, 1), (10, 1), (11, 0.5)):                            # bias, activation, f32 output, accumulate, alpha != 1
            a = list(ok); a[i] = v
            assert takes(*a) == 0, a
        assert takes(1000, 4096, 11008, 11008, 11008, 4096, 0, 0, 0, 0, 0, 1.0) == 0          # M < 1024
        assert takes(8190, 1000, 11008, 11008, 11008, 1000, 0, 0, 0, 0, 0, 1.0) == 0          # N < 1024
        assert takes(8190, 4096, 2048, 2048, 2048, 4096, 0, 0, 0, 0, 0, 1.0) == 0             # short k-loop
        assert takes(8190, 4096, 4128, 4128, 4128, 4096, 0, 0, 0, 0, 0, 1.0) == 0             # K % 64 != 0
        assert takes(8190, 4096, 11008, 11008, 11008, 4100, 0, 0, 0, 0, 0, 1.0) == 0          # output rows not 16-B aligned
        lib.lhrs_gemm_set_vendor(0, 0)
        assert takes(*ok) == 0                                                                # library removed from the candidates
    finally:
        lib.lhrs_gemm_set_vendor(1, 4096)


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
    rope = lambda rope_cols, cos=0x50000000, sin=0x60000000, pos_mod=273: lib.lhrs_gemm_u4_rope(A, 4096, B, 4096, C, 12288, 2048, 12288, 4096, cos, sin, pos_mod,
                                                                                              0, rope_cols, None)
    assert rope(8192 + 128) == 1 and rope(16384) == 1 and rope(8192, cos=None) == 1 and rope(8192, pos_mod=0) == 1 and rope(8192, sin=0x60000004) == 1
    assert isinstance(lib.lhrs_gemm_u4_problems(), int) and lib.lhrs_gemm_u4_problems() >= 0
    st = (ctypes.c_long * 3)()
    assert lib.lhrs_gemm_vendor_stats(ctypes.addressof(st)) == 0 and st[1] + st[2] == st[0]
