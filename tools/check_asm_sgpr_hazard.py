"""gfx9 hazard the compiler cannot see: "VALU writes SGPR -> VMEM reads that SGPR: 5 wait states" (MI355X_MICROARCH.md / the CDNA ISA's table of user-inserted wait
states).  The hazard recognizer pads it for its own instructions, but an inline-asm statement is opaque to it: when the register allocator reloads a spilled SGPR pair
(v_readlane_b32) or makes a value uniform (v_readfirstlane_b32) right in front of an asm statement whose VMEM instruction takes that pair as its address, nothing is
padded and the load / store goes out with the OLD register contents (found as a memory access fault of gemm_u4_kernel<2, false> in round 5).
    python tools/check_asm_sgpr_hazard.py file.s [...]      (hipcc -S --cuda-device-only output)
lists every VMEM instruction inside an asm statement whose scalar address operand was written by a VALU instruction fewer than 5 wait states earlier; exit status 1
when there is one.  csrc's asm statements copy their scalar pointer through an SALU move first (no hazard between SALU and VMEM), so the list must be empty."""
import re
import sys

VALU_SGPR_WRITERS = ("v_readlane_b32", "v_readfirstlane_b32", "v_cmp", "v_add_co", "v_sub_co", "v_addc_co", "v_subb_co", "v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale")
VMEM = ("global_load", "global_store", "buffer_load", "buffer_store", "flat_load", "flat_store", "global_atomic", "scratch_")


def sregs(tok):
    m = re.match(r"s\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"s(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(path):
    bad, window, in_asm, fn = [], [], False, "?"
    for ln, raw in enumerate(open(path), 1):
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if t.endswith(":") and not t.startswith("."):
            fn = t[:-1]
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        op, _, rest = t.partition(" ")
        ops = [o.strip() for o in rest.split(";")[0].split(",")]
        if in_asm and op.startswith(VMEM):
            need = set()
            for o in ops:
                need |= sregs(o.split()[0]) if o else set()
            states = 0
            for wop, wregs, wstates in reversed(window):
                if states >= 5:
                    break
                if wregs & need:
                    bad.append((path, ln, fn, t, wop, states))
                    break
                states += wstates
        written = set()
        if op.startswith(VALU_SGPR_WRITERS):
            for o in ops[:2]:                                        # sdst is the first operand (readlane) or the second (carry-out forms)
                written |= sregs(o.split()[0]) if o else set()
        m = re.match(r"s_nop (\d+)", t)
        window.append((op, written, int(m.group(1)) + 1 if m else 1))
        del window[:-12]
    return bad


if __name__ == "__main__":
    bad = [b for p in sys.argv[1:] for b in scan(p)]
    for path, ln, fn, t, wop, states in bad:
        print(f"{path}:{ln}: {fn}: `{t}` reads an SGPR written by {wop} {states} wait state(s) earlier")
    print(f"{len(bad)} unpadded VALU-writes-SGPR -> asm VMEM hazards")
    sys.exit(1 if bad else 0)
