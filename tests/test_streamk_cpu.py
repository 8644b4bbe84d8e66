"""Host logic of the stream-K tail of the persistent 256x256 GEMM (csrc/gemm.hip: sk_cut / launch_256s), replayed through the C ABI without
a GPU: for every tail size the ranges of the units must cover every stage of every tail tile exactly once, no work item may be shorter than
the pipeline's minimum, every partial must have exactly one consumer, and a range never starts inside the second operand pair."""
import ctypes

import pytest

from lhrs_bot_amd import _lib

P = 256   # num_cus() without a device


def plan(T, nk, nk2=0, unit=-1):
    out = (ctypes.c_int * 10)()
    rc = _lib.load().lhrs_gemm_streamk_plan(T, nk, nk2, 1, unit, ctypes.addressof(out))
    return rc, list(out)


@pytest.mark.parametrize("nk", [16, 64, 65, 172, 192, 344])
@pytest.mark.parametrize("nk2", [0, 1, 3])
def test_streamk_ranges_cover_every_stage_once(nk, nk2):
    applied = 0
    for T in list(range(1, 256, 7)) + [144, 255, 256 + 96, 5 * 256 + 96, 10 * 256 + 192, 7 * 256 + 83, 560, 512]:
        rc, head = plan(T, nk, nk2)
        tail = T % P
        if tail == 0 or tail / P + 8.0 / nk > 0.92:
            assert rc == -1
            continue
        assert rc == 0 and head[0] == T - tail and head[1] == tail and 1 <= head[2] <= P
        applied += 1
        units = head[2]
        cover = [[0] * nk for _ in range(tail)]
        partials, consumed = 0, 0
        for u in range(units):
            _, o = plan(T, nk, nk2, u)
            i0, s0, n0, n1, part0, add0, add1 = o[3:10]
            if n0 == 0:
                assert n1 == 0
                continue
            assert n0 >= 2 and (n1 == 0 or n1 >= 2) and s0 + n0 <= nk and n1 <= nk       # the pipeline needs >= 2 stages per item
            assert s0 < nk - nk2                                                           # never starts inside the second operand pair
            if n1:
                assert s0 + n0 == nk and i0 + 1 < tail
            for k in range(s0, s0 + n0):
                cover[i0][k] += 1
            for k in range(n1):
                cover[i0 + 1][k] += 1
            partials += part0
            consumed += add0 + add1
            if part0:
                assert add0 == 0
        assert all(c == 1 for row in cover for c in row), (T, nk, nk2)
        assert partials == consumed, (T, nk, nk2, partials, consumed)
    assert applied > 0


def test_streamk_is_declined_for_long_second_pairs_and_short_k():
    assert plan(300, 64, 4)[0] == -1      # r = 128 LoRA on a fused group: 4+ stages of the second pair
    assert plan(300, 8, 0)[0] == -1       # K = 512: the fixed cost of the exchange exceeds what the tail saves
    assert plan(512, 64, 0)[0] == -1      # whole rounds: nothing to split
