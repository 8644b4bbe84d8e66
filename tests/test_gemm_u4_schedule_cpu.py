"""The generated stage body of gemm_u4_kernel (lhrs_bot_amd/csrc/gemm_u4_body.inc, gemm_u4_agpr.inc): the committed files are what the generators print, and the
schedule keeps the invariants the kernel's correctness rests on (csrc/gemm_u4.hip; no GPU needed)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lhrs_bot_amd", "csrc")
GEN = os.path.join(ROOT, "tools", "gemm_u_proto")


def run(script, *args):
    return subprocess.run([sys.executable, os.path.join(GEN, script), *args], capture_output=True, text=True, check=True).stdout


def test_committed_includes_are_the_generators_output():
    assert run("gen_u5.py", "1", "21", "6", "108", "1") == open(os.path.join(CSRC, "gemm_u4_body.inc")).read()
    assert run("gen_agpr.py") == open(os.path.join(CSRC, "gemm_u4_agpr.inc")).read()


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
