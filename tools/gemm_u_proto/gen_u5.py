"""u5_body.inc: one k-stage (128 MFMA slots per wave) of the 4-wave prototype.
    python gen_u5.py RB0 X E Y R > u5_body.inc
  RB0  the 16 fragment reads of the stage's second 32-k half go behind every RB0-th MFMA from slot 0
  X    behind MFMA X: lgkmcnt(0) + barrier - every wave has read the whole stage, its LDS buffer is free; the 16 DMA pieces of stage kt+2 follow,
  E    one piece every E MFMAs (s_add on m0 behind one MFMA, the load behind the next).  64 pieces of 1 KiB per stage and CU are 1024 cycles of the CU's
       address path at 64 B/clk - half the stage: pieces issued faster than one per 4 MFMAs and wave back up in that path and stall the issuing wave
  Y    behind MFMA Y: vmcnt(pieces issued so far) + barrier - stage kt+1 has landed; its first 16 fragment reads follow, one behind every R-th MFMA
defaults 2 36 4 104 1"""
import sys

a = [int(x) for x in sys.argv[1:]] + [None] * 5
RB0, X, E, Y, R = (a[0] or 2), (a[1] or 36), (a[2] or 4), (a[3] or 104), (a[4] or 1)
reads0 = {i * RB0: i for i in range(16)}                       # slot -> read index (stage kt, half 1)
assert max(reads0) < X
m0p = {X + 1 + p * E: p for p in range(16)}                     # slot -> piece: m0 update
glds = {t + 1: p for t, p in m0p.items()}                       # slot -> piece: the load
assert max(glds) <= 127, max(glds)
reads1 = {Y + 1 + i * R: i for i in range(16)}                  # slot -> read index (stage kt+1, half 0)
assert max(reads1) <= 127, max(reads1)
before = sum(1 for t in glds if t <= Y)                         # pieces of stage kt+2 already issued at the wait: they may stay in flight
out = []
for t in range(128):
    blk, m = divmod(t, 64)
    mi, ni = divmod(m, 8)
    out.append(f"MFM(A{blk}, B{blk}, {mi}, {ni})")
    if t in reads0:
        i = reads0[t]
        arr, idx, ad = ("A1", i, "aa1") if i < 8 else ("B1", i - 8, "ba1")
        out.append(f"RDQ({arr}[{idx}], {ad}, {idx * 2048}); SB")
    if t == X:
        out.append("wait16(A1, B1); BARX")
    if t in glds:
        out.append(f"GLDS({glds[t]}); SB")
    if t in m0p:
        out.append(f"M0P({m0p[t]}); SB")
    if t == Y:
        out.append(f"WAITY({before})")
    if t in reads1:
        i = reads1[t]
        arr, idx, ad = ("A0", i, "aa0") if i < 8 else ("B0", i - 8, "ba0")
        out.append(f"RDN({arr}[{idx}], {ad}, {idx * 2048}); SB")
    if t == 63:
        out.append("wait16(A1, B1);")
print("\n".join(out))
