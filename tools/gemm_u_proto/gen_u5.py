"""u5_body.inc: one k-stage (128 MFMAs per wave) of the 4-wave prototype with the stage-(kt+2) DMA issued as soon as every wave has read stage kt
(barrier X inside the first 64-MFMA block) and the wait for stage kt+1 as late as its first fragment read allows (barrier Y inside the second block).
    python gen_u5.py [X] [Y] [DMA_EVERY] > u5_body.inc        defaults 36 24 2"""
import sys

X = int(sys.argv[1]) if len(sys.argv) > 1 else 36      # after MFMA X of block 0: lgkmcnt(0), barrier, DMA of stage kt+2 starts
Y = int(sys.argv[2]) if len(sys.argv) > 2 else 24      # after MFMA Y of block 1: vmcnt, barrier, fragment reads of stage kt+1 start
E = int(sys.argv[3]) if len(sys.argv) > 3 else 2       # one DMA piece / one fragment read behind every E-th MFMA
out = []
piece = 0
# ---- block 0: MFMAs on (A0, B0); reads half 1 of stage kt into (A1, B1)
rd = 0
for m in range(64):
    mi, ni = divmod(m, 8)
    out.append(f"MFM(A0, B0, {mi}, {ni})")
    if m % 2 == 0 and rd < 16:
        arr, idx, ad = ("A1", rd, "aa1") if rd < 8 else ("B1", rd - 8, "ba1")
        out.append(f"RDQ({arr}[{idx}], {ad}, {idx * 2048}); SB")
        rd += 1
    if m == X:
        assert rd == 16
        out.append("wait16(A1, B1); BARX")
    if m > X and (m - X) % E == 0 and piece < 16:
        out.append(f"ISS({piece}); SB")
        piece += 1
# ---- block 1: MFMAs on (A1, B1); the rest of the DMA; then wait + barrier + reads of stage kt+1 half 0 into (A0, B0)
rd = 0
for m in range(64):
    mi, ni = divmod(m, 8)
    out.append(f"MFM(A1, B1, {mi}, {ni})")
    if m % E == 0 and piece < 16 and m < Y:
        out.append(f"ISS({piece}); SB")
        piece += 1
    if m == Y:
        assert piece == 16, piece
        out.append("WAITY")
    if m > Y and (m - Y) % 2 == 1 and rd < 16:
        arr, idx, ad = ("A0", rd, "aa0") if rd < 8 else ("B0", rd - 8, "ba0")
        out.append(f"RDN({arr}[{idx}], {ad}, {idx * 2048}); SB")
        rd += 1
assert rd == 16, rd
print("\n".join(out))
