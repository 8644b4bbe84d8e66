"""Generated stage bodies of gemm_u4_kernel (lhrs_bot_amd/csrc/gemm_u4.hip): one k-stage = 128 MFMA slots per wave (two 32-k halves of 8 x 8 fragments).
    python tools/gen_u4.py body   > lhrs_bot_amd/csrc/gemm_u4_body.inc      the steady-state stage (also the last two stages of a launch, through other macro definitions)
    python tools/gen_u4.py flush1 > lhrs_bot_amd/csrc/gemm_u4_flush1.inc    first stage of a tile that also writes the PREVIOUS tile out, one unit = 2 fragments (plain, SwiGLU')
    python tools/gen_u4.py flush2 > lhrs_bot_amd/csrc/gemm_u4_flush2.inc    the same, one unit = 4 fragments (RoPE, SwiGLU: a lane pairs column c with column c + 64)
    python tools/gen_u4.py last1 / last2                                    last stage of a tile whose flush wants operands from memory: the first FL_DEPTH units' loads ride in it

Time line of a stage (docs/design_notes_r03_r04.md: u5; E = one DMA piece per 6 MFMAs):
  slots 0..15   behind every MFMA one of the 16 fragment reads of this stage's second 32-k half
  slot  X = 21  lgkmcnt(0) + barrier: every wave has read the whole stage, its LDS buffer is free; the 16 DMA pieces of stage kt+2 follow (m0 update behind
                one MFMA, the load behind the next)
  slot  Y = 108 counted vmcnt + barrier: stage kt+1 has landed (the younger operations stay in flight); its first 16 fragment reads follow

The flush bodies: the accumulators of the finished tile are read out (FL_ACC) right in front of the first-half MFMAs that overwrite them - those MFMAs take the
constant 0 as their C operand (MFM0), so nothing has to be zeroed - and converted / stored (FL_OP) between the following MFMAs.  Operands a unit needs from memory
(residual rows, gate|up, cos / sin) are requested FL_DEPTH units ahead (FL_LOAD) and waited for with a counted vmcnt (FL_WAIT).  Every vmcnt count in these files is
the number of vector-memory operations issued BEHIND the one waited for, in program order (loads, stores and LDS-DMA share the counter and retire in issue order on
gfx9: MI355X_MICROARCH.md, `vmcnt(N)` waits for the outstanding - N oldest) - so every memory operation of a flush body is UNCONDITIONAL (interior tiles only; edge
tiles are written out by the exposed epilogue in gemm_u4.hip).  S = stores per unit, L = loads per unit, DEPTH = units of load prefetch: plain numbers per file."""
import sys

X, E, Y = 21, 6, 108
reads0 = {i: i for i in range(16)}                              # slot -> read index (stage kt, half 1)
m0p = {X + 1 + p * E: p for p in range(16)}                     # slot -> piece: m0 update
glds = {t + 1: p for t, p in m0p.items()}                       # slot -> piece: the load
reads1 = {Y + 1 + i: i for i in range(16)}                      # slot -> read index (stage kt+1, half 0)
assert max(glds) <= 127 and max(reads1) <= 127
pieces_before_y = sum(1 for t in glds if t <= Y)                # pieces of stage kt+2 already issued at the wait


def common(t, out):
    """what every body hangs behind MFMA slot t"""
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


def tail(t, out):
    if t in reads1:
        i = reads1[t]
        arr, idx, ad = ("A0", i, "aa0") if i < 8 else ("B0", i - 8, "ba0")
        out.append(f"RDN({arr}[{idx}], {ad}, {idx * 2048}); SB")
    if t == 63:
        out.append("wait16(A1, B1);")


def mfm(t, zero_c=False):
    blk, m = divmod(t, 64)
    mi, ni = divmod(m, 8)
    return f"MFM{'0' if zero_c and blk == 0 else ''}(A{blk}, B{blk}, {mi}, {ni})"


def body():
    out = []
    for t in range(128):
        out.append(mfm(t))
        common(t, out)
        if t == Y:
            out.append(f"WAITY({pieces_before_y})")
        tail(t, out)
    return out


def units(paired):
    """-> per unit u: (mi, sub, fragments, slot in front of which FL_ACC goes, slot behind which FL_OP goes); unit order = the order the accumulators are overwritten"""
    us = []
    for u in range(16 if paired else 32):
        if paired:
            mi, q = divmod(u, 2)
            us.append((mi, q, (2 * q, 2 * q + 1, 4 + 2 * q, 5 + 2 * q), 8 * mi + 2 * q, 8 * mi + 2 * q + 3))
        else:
            mi, p = divmod(u, 4)
            us.append((mi, p, (2 * p, 2 * p + 1), 8 * mi + 2 * p, 8 * mi + 2 * p + 1))
    return us


def flush_concrete(paired, S, L, depth):
    """-> (lines of the flush body, lines of the `last` body).  S stores and L loads per unit, loads `depth` units ahead.
    macros: FL_ACC(u, s, mi, frags...)  accumulators of unit u -> temporaries set s (= u % 2)
            FL_OP(u, s, lb, mi, sub)    convert / combine / store unit u (lb = u % depth: its load buffer)
            FL_LOAD(u, lb, mi, sub)     request unit u's operands into load buffer lb
            FL_WAIT(u, lb, n)           s_waitcnt vmcnt(n): unit u's operands have landed"""
    us = units(paired)
    nu = len(us)
    acc_at = {x[3]: u for u, x in enumerate(us)}
    op_at = {x[4]: u for u, x in enumerate(us)}
    ops = []                                                  # program order over BOTH bodies

    def load(u, lines):
        mi, sub = us[u][0], us[u][1]
        lines.append(f"FL_LOAD({u}, {u % depth}, {mi}, {sub}); SB")
        ops.extend([("ld", u)] * L)

    last_lines = []
    if L:
        # the last stage of the finished tile: its DMA pieces fetch the NEXT tile's stage 1; the loads of units 0 .. depth-1 go out in its second half, two MFMAs
        # apart from slot 64 on
        ld_slot = {64 + 2 * i: i for i in range(min(depth, nu))}
        for t in range(128):
            last_lines.append(mfm(t))
            common(t, last_lines)
            if t in glds:
                ops.append(("dma1", glds[t]))
            if t in ld_slot:
                load(ld_slot[t], last_lines)
            if t == Y:
                # stage kt+1 (= the next tile's stage 0) was requested in the previous stage: everything issued in THIS body so far may stay in flight
                last_lines.append(f"WAITY({min(63, len(ops))})")
            tail(t, last_lines)
    else:
        ops.extend(("dma1", p) for p in range(16))            # the previous stage's pieces (stage 1 of this tile): what WAITY of the flush body waits for
    lines = []
    for t in range(128):
        if t in acc_at:
            u = acc_at[t]
            if L:
                pos = max(i for i, o in enumerate(ops) if o == ("ld", u))
                lines.append(f"FL_WAIT({u}, {u % depth}, {min(63, len(ops) - 1 - pos)}); SB")
            lines.append(f"FL_ACC({u}, {u % 2}, {us[u][0]}, {', '.join(str(f) for f in us[u][2])}); SB")
        lines.append(mfm(t, zero_c=True))
        common(t, lines)
        if t in glds:
            ops.append(("dma2", glds[t]))
        if t in op_at:
            u = op_at[t]
            lines.append(f"FL_OP({u}, {u % 2}, {u % depth if L else 0}, {us[u][0]}, {us[u][1]}); SB")
            ops.extend([("st", u)] * S)
            if L and u + depth < nu:
                load(u + depth, lines)
        if t == Y:
            pos = max(i for i, o in enumerate(ops) if o[0] == "dma1")
            lines.append(f"WAITY({min(63, len(ops) - 1 - pos)})")
        tail(t, lines)
    return lines, last_lines


if __name__ == "__main__":
    what = sys.argv[1]
    if what == "body":
        print("\n".join(body()))
    else:
        # flush1|flush2|last1|last2  S L DEPTH
        paired = what.endswith("2")
        S, L, depth = (int(x) for x in sys.argv[2:5])
        fl, la = flush_concrete(paired, S, L, depth)
        print("\n".join(la if what.startswith("last") else fl))
