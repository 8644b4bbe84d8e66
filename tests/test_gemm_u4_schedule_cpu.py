"""The generated stage bodies of gemm_u4_kernel (lhrs_bot_amd/csrc/gemm_u4_body.inc, gemm_u4_flush_*.inc, gemm_u4_last_*.inc, gemm_u4_agpr.inc): the committed files are
what the generators print, and the schedules keep the invariants the kernel's correctness rests on (csrc/gemm_u4.hip; no GPU needed) - above all that every counted
`s_waitcnt vmcnt(n)` lets exactly the operations issued BEHIND the awaited one stay in flight."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lhrs_bot_amd", "csrc")
GEN = os.path.join(ROOT, "tools")
# file suffix -> generator arguments (what, stores per unit, loads per unit, prefetch depth in units): must match the Makefile comment and U4_DEPTH_* in gemm_u4.hip
FLUSH = {"flush_p": ("flush1", 1, 0, 1), "flush_r": ("flush1", 1, 1, 8), "last_r": ("last1", 1, 1, 8), "flush_f": ("flush2", 3, 0, 1),
         "flush_b": ("flush1", 2, 2, 6), "last_b": ("last1", 2, 2, 6), "flush_o": ("flush2", 2, 4, 3), "last_o": ("last2", 2, 4, 3)}


def run(script, *args):
    return subprocess.run([sys.executable, os.path.join(GEN, script), *[str(a) for a in args]], capture_output=True, text=True, check=True).stdout


def test_committed_includes_are_the_generators_output():
    assert run("gen_u4.py", "body") == open(os.path.join(CSRC, "gemm_u4_body.inc")).read()
    assert run("gen_u4_agpr.py") == open(os.path.join(CSRC, "gemm_u4_agpr.inc")).read()
    for suffix, args in FLUSH.items():
        assert run("gen_u4.py", *args) == open(os.path.join(CSRC, f"gemm_u4_{suffix}.inc")).read(), suffix
    src = open(os.path.join(CSRC, "gemm_u4.hip")).read()
    assert "U4_DEPTH_RES = 8, U4_DEPTH_SWB = 6, U4_DEPTH_ROPE = 3" in src          # the load buffers the kernel declares are the depths the bodies were generated for


def _vm_ops(line, S, L):
    """vector-memory operations a generated line issues, in order"""
    if line.startswith("GLDS("):
        return ["dma"]
    if line.startswith("FL_LOAD("):
        return [("ld", int(re.match(r"FL_LOAD\((\d+),", line).group(1)))] * L
    if line.startswith("FL_OP("):
        return [("st", int(re.match(r"FL_OP\((\d+),", line).group(1)))] * S
    return []


def test_flush_bodies_counted_waits_and_read_before_overwrite():
    """For every write-out variant, replay the `last` body (if the variant loads operands) and the flush body in program order:
    * FL_ACC(u) stands in front of the first MFMA that overwrites one of unit u's accumulator fragments, all first-half MFMAs of the flush body take C = 0 (MFM0),
      the second half accumulates (MFM), and every fragment is read exactly once;
    * FL_WAIT(u, lb, n): n = the number of vector-memory operations issued behind unit u's last load (so a smaller hardware count means that load has retired), and
      unit u's buffer lb = u % depth is not requested again before FL_OP(u) has consumed it;
    * WAITY(n) of the flush body: n = operations issued behind the previous stage's last DMA piece; WAITY of the `last` body: everything issued in that body so far;
    * no count exceeds the 6-bit counter."""
    for fl, la in (("flush_p", None), ("flush_r", "last_r"), ("flush_f", None), ("flush_b", "last_b"), ("flush_o", "last_o")):
        what, S, L, depth = FLUSH[fl]
        paired = what.endswith("2")
        ops = []                                                   # program order over both bodies
        if la is None:
            ops += ["dma"] * 16                                    # the previous stage (an ordinary body): 16 pieces
        else:
            for line in open(os.path.join(CSRC, f"gemm_u4_{la}.inc")).read().splitlines():
                if line.startswith("WAITY("):
                    assert int(re.match(r"WAITY\((\d+)\)", line).group(1)) == min(63, len(ops)), la
                ops += _vm_ops(line, S, L)
            assert sum(1 for o in ops if o == "dma") == 16 and sorted({o[1] for o in ops if o != "dma"}) == list(range(depth)), la
        lines = open(os.path.join(CSRC, f"gemm_u4_{fl}.inc")).read().splitlines()
        read, written, pending_buf, seen_units = set(), set(), {}, []
        for u in range(depth if L else 0):
            pending_buf[u % depth] = u
        n_mfm0 = n_mfm = 0
        for line in lines:
            m = re.match(r"FL_WAIT\((\d+), (\d+), (\d+)\)", line)
            if m:
                u, lb, n = (int(x) for x in m.groups())
                last = max(i for i, o in enumerate(ops) if o == ("ld", u))
                assert n == min(63, len(ops) - 1 - last) and lb == u % depth and pending_buf.get(lb) == u, (fl, line)
            m = re.match(r"FL_ACC\((\d+), (\d+), (\d+), ([\d, ]+)\)", line)
            if m:
                u, mi = int(m.group(1)), int(m.group(3))
                frags = {(mi, int(f)) for f in m.group(4).split(",")}
                assert len(frags) == (4 if paired else 2) and not (frags & read) and not (frags & written), (fl, line)   # read once, before any overwrite
                if paired:
                    assert {f for _, f in frags} == {2 * (u % 2), 2 * (u % 2) + 1, 4 + 2 * (u % 2), 5 + 2 * (u % 2)}          # column c and column c + 64 in one lane
                read |= frags
                seen_units.append(u)
            m = re.match(r"MFM(0?)\(A(\d), B\2, (\d), (\d)\)", line)
            if m:
                half, frag = int(m.group(2)), (int(m.group(3)), int(m.group(4)))
                assert (m.group(1) == "0") == (half == 0), (fl, line)                                                     # first half: C = 0; second half accumulates
                n_mfm0 += half == 0
                n_mfm += half == 1
                if half == 0:
                    assert frag in read, (fl, line, "overwritten before it was read")
                    written.add(frag)
            m = re.match(r"FL_OP\((\d+), (\d+), (\d+),", line)
            if m and L:
                u, lb = int(m.group(1)), int(m.group(3))
                assert pending_buf.pop(lb) == u, (fl, line)
            m = re.match(r"FL_LOAD\((\d+), (\d+),", line)
            if m:
                u, lb = int(m.group(1)), int(m.group(2))
                assert lb == u % depth and lb not in pending_buf, (fl, line, "load buffer requested again before its unit was written out")
                pending_buf[lb] = u
            if line.startswith("WAITY("):
                first_stage_dma = [i for i, o in enumerate(ops) if o == "dma"][:16]
                assert int(re.match(r"WAITY\((\d+)\)", line).group(1)) == min(63, len(ops) - 1 - first_stage_dma[-1]), (fl, line)
            ops += _vm_ops(line, S, L)
        assert n_mfm0 == 64 and n_mfm == 64 and len(read) == 64 and seen_units == list(range(16 if paired else 32)) and not pending_buf, fl


def test_stage_body_invariants():
    lines = open(os.path.join(CSRC, "gemm_u4_body.inc")).read().splitlines()
    mfma = [i for i, l in enumerate(lines) if l.startswith("MFM(")]
    assert len(mfma) == 128
    # every accumulator fragment (mi, ni) gets exactly one MFMA per 32-k half, first on the (A0, B0) registers, then on (A1, B1)
    for half, regs in enumerate(("A0, B0", "A1, B1")):
        seen = set()
        for i in mfma[64 * half: 64 * half + 64]:
            m = re.fullmatch(r"MFM\((A\d, B\d), (\d), (\d)\)", lines[i])
            assert m and m.group(1) == regs
            seen.add((int(m.group(2)), int(m.group(3))))
        assert len(seen) == 64
    slot = lambda i: sum(1 for j in mfma if j <= i) - 1          # index of the last MFMA in front of line i
    barx = next(i for i, l in enumerate(lines) if "BARX" in l)
    waity = next(i for i, l in enumerate(lines) if l.startswith("WAITY("))
    # the 16 fragment reads of this stage's second half come before barrier X and are waited for there (the buffer is free for the DMA only behind it)
    rd1 = [i for i, l in enumerate(lines) if l.startswith("RDQ(A1[") or l.startswith("RDQ(B1[")]
    assert len(rd1) == 16 and max(rd1) < barx and "wait16(A1, B1)" in lines[barx]
    assert sorted(re.match(r"RDQ\((\w\d\[\d\])", lines[i]).group(1) for i in rd1) == sorted(f"{x}1[{k}]" for x in "AB" for k in range(8))
    # 16 DMA pieces, in order, each: m0 update behind one MFMA, the load behind the next one; all behind barrier X; never two pieces closer than 4 MFMAs
    m0p = [i for i, l in enumerate(lines) if l.startswith("M0P(")]
    glds = [i for i, l in enumerate(lines) if l.startswith("GLDS(")]
    assert [int(re.match(r"M0P\((\d+)\)", lines[i]).group(1)) for i in m0p] == list(range(16))
    assert [int(re.match(r"GLDS\((\d+)\)", lines[i]).group(1)) for i in glds] == list(range(16))
    assert all(i > barx for i in m0p) and all(slot(g) == slot(m) + 1 for m, g in zip(m0p, glds))
    assert min(slot(b) - slot(a) for a, b in zip(glds, glds[1:])) >= 4
    # the counted wait at barrier Y lets exactly the pieces of stage kt+2 issued so far stay in flight: everything older (stage kt+1) has landed
    n = int(re.match(r"WAITY\((\d+)\)", lines[waity]).group(1))
    assert n == sum(1 for g in glds if g < waity) and n <= 16
    # the next stage's first 16 fragment reads come behind barrier Y, into the registers the first half no longer needs
    rdn = [i for i, l in enumerate(lines) if l.startswith("RDN(")]
    assert len(rdn) == 16 and min(rdn) > waity and all(re.match(r"RDN\([AB]0\[\d\], [ab]a0, ", lines[i]) for i in rdn)
    assert max(slot(i) for i in rdn) <= 127 and slot(waity) >= 64
