"""Generates lhrs_bot_amd/csrc/gemm_256s_sched.inc: the MFMA / fragment-reload schedule of one 64-k stage of gemm_nt_256s_kernel.

A wave's 64x64 sub-tile is 4x4 fragments of v_mfma_f32_16x16x32_bf16; a stage is two k-blocks of 16 MFMAs.  The eight fragment registers
(A0-3: activations, B0-3: weights) are single-buffered: each is re-read for the NEXT k-block right behind its last MFMA of the current one.
The MFMA order inside a block walks the 2x2 quadrants of the fragment grid so that consecutive blocks start on operands that were
released early: block 0 runs Q00 Q01 Q11 Q10, block 1 runs Q01 Q00 Q10 Q11 (Qrc = rows {2r, 2r+1} x columns {2c, 2c+1}), which leaves
every fragment >= 7 MFMA slots between its reload and its next use.  LDS returns data in order, so the wait in front of a fragment's
first use is `s_waitcnt lgkmcnt(N)`, N = reads issued after that fragment's reload (computed here by replaying the schedule).  Block 1
opens with the stage boundary (full wait, vmcnt, s_barrier); the four DMA pieces of the stage after next follow in its first four gaps.
Tried and measured worse (MI355X, all-zero operands, TFLOP/s over the seven LLaMA shapes at M = 8190): the barrier in front of block 1's
first re-read instead of at its top 1294 (vs 1429); the 64 DMA pieces of a stage spread over 16 slots with the wave rows taking turns 1013
/ 1306 (scalar branches in the MFMA stream cost more than the burst they smooth); the DMA pieces behind MFMA slots 0,2,4,6 1298 (vs 1302
random operands), 0,3,6,9 1275, 1,5,9,13 1240, 0,4,8,12 1199; a ring of four 32-k half-stages (every block opens with a boundary, the DMA of
half-stage b + 4 goes out at the boundary of block b and has three block times to land, `vmcnt(4)`) 1313 zeros / 1200 random against 1424 /
1341 on the same box: the pipeline is not waiting for DMA latency, and twice the barriers cost more than the longer prefetch returns.

   python tools/gen_gemm16_sched.py > lhrs_bot_amd/csrc/gemm_256s_sched.inc
"""
Q = {"00": ([0, 1], [0, 1]), "01": ([0, 1], [2, 3]), "11": ([2, 3], [2, 3]), "10": ([2, 3], [0, 1])}
ORDER = (["00", "01", "11", "10"], ["01", "00", "10", "11"])
# (column-major, swap rows, swap columns) per quadrant, found by maximising the smallest reload -> first-use distance (7 slots)
CHOICE = [(1, 0, 0), (1, 0, 0), (1, 0, 0), (1, 0, 0), (1, 1, 0), (1, 0, 0), (1, 0, 0), (1, 1, 0)]


def quad(rows, cols, colmajor, rswap, cswap):
    r = rows[::-1] if rswap else rows
    c = cols[::-1] if cswap else cols
    return [(r[0], c[0]), (r[1], c[0]), (r[0], c[1]), (r[1], c[1])] if colmajor else [(r[0], c[0]), (r[0], c[1]), (r[1], c[0]), (r[1], c[1])]


def blocks():
    out, k = [], 0
    for order in ORDER:
        seq = []
        for q in order:
            seq += quad(*Q[q], *CHOICE[k])
            k += 1
        out.append(seq)
    return out


def frag(f):
    return f"{f[0]}[{f[1]}]"


import os
import sys

DMA_SLOTS = [int(x) for x in os.environ.get("DMA_SLOTS", "0,1,2,3").split(",")]  # MFMA slots of block 1 behind which the four DMA pieces go out


N_DMA, VMCNT, FRAG_STRIDE = 4, 0, 2048  # DMA pieces per wave and stage; pieces that may stay in flight at a boundary; bytes between fragments


def emit_block(name, seq, prev_reads, do_reads, barrier):
    """prev_reads: fragments in the order the PREVIOUS block re-read them (all 8 outstanding, worst case, when this block starts).
    -> (text, reads issued by this block in order)"""
    first, last = {}, {}
    for t, (mi, ni) in enumerate(seq):
        for f in (("A", mi), ("B", ni)):
            first.setdefault(f, t)
            last[f] = t
    outstanding = list(prev_reads)  # oldest first
    issued = []
    lines = [f"#define {name}(aa, ba) \\"]
    dma_pos = 0
    if barrier:
        # the stage boundary: every read of the current stage's buffer is complete (full wait), this wave's DMA pieces of the next stage
        # have landed (vmcnt), the barrier publishes everybody's and frees the current buffer for the stage after next
        cons = ", ".join(f'"+v"({frag(f)})' for f in sorted(set(outstanding)))
        lines.append(f'  asm volatile("s_waitcnt lgkmcnt(0)" : {cons}); asm volatile("s_waitcnt vmcnt({VMCNT})" ::: "memory"); __builtin_amdgcn_s_barrier(); SB \\')
        outstanding = []
    for t, (mi, ni) in enumerate(seq):
        need = [f for f in (("A", mi), ("B", ni)) if first[f] == t and f in outstanding]
        if need:
            pos = max(outstanding.index(f) for f in need)
            n = len(outstanding) - 1 - pos
            cons = ", ".join(f'"+v"({frag(f)})' for f in need)
            lines.append(f'  asm volatile("s_waitcnt lgkmcnt({n})" : {cons}); SB \\')
            outstanding = outstanding[pos + 1:]
        while barrier and dma_pos < N_DMA and DMA_SLOTS[dma_pos] == t - 1 and t == 0:  # slot -1: in front of the block's first MFMA
            lines.append(f"  {{ ISS({dma_pos}) }} SB \\")
            dma_pos += 1
        lines.append(f"  MF({mi}, {ni}) \\")
        dead = [f for f in (("A", mi), ("B", ni)) if last[f] == t]
        if do_reads and dead:
            for f in dead:
                addr = "aa" if f[0] == "A" else "ba"
                lines.append(f"  RDQ({frag(f)}, {addr}, {f[1] * FRAG_STRIDE}); SB \\")
                outstanding.append(f)
                issued.append(f)
        while barrier and dma_pos < N_DMA and t == DMA_SLOTS[dma_pos]:
            lines.append(f"  {{ ISS({dma_pos}) }} SB \\")
            dma_pos += 1
    lines.append("  ;")
    return "\n".join(lines), issued


b0, b1 = blocks()
# steady state: block 1's reload order feeds block 0 and vice versa (two passes to reach the fixed point)
_, r0 = emit_block("x", b0, [], True, False)
_, r1 = emit_block("x", b1, r0, True, True)
t0, r0 = emit_block("S_BLOCK0", b0, r1, True, False)
t1, r1b = emit_block("S_BLOCK1", b1, r0, True, True)
assert r1b == r1
tf, _ = emit_block("S_BLOCK1_FINAL", b1, r0, False, False)
print("// GENERATED by tools/gen_gemm16_sched.py - do not edit.  MF(mi, ni): one MFMA + sched barrier; RDQ(dst, addr, off): ds_read_b128;")
print("// ISS(j): DMA piece j of the stage after next; SB: sched barrier.  (aa, ba): LDS addresses of the NEXT k-block's A / B fragments.")
print(t0 + "\n")
print(t1 + "\n")
print(tf)
