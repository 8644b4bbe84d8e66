"""Second ISA check of the asm-issued register loads (lhrs_bot_amd/csrc/gemm_u4.hip GLD / FL_WAITN): a `global_load_dword*` inside an asm statement writes
VGPRs the compiler believes are defined at once; the data is there only behind a later, COUNTED `s_waitcnt vmcnt(N)` in another asm statement.  Nothing in the language
stops the register allocator from copying or spilling such a register in between - the copy would read stale bits - and a count that does not match the operations
really issued behind the load would release it early.  This scanner replays the in-order retirement rule on the compiled ISA:
  * every VMEM instruction (loads, stores, LDS-DMA; asm or compiler-issued) is one entry of the wave's in-order queue;
  * `s_waitcnt vmcnt(N)` retires every entry that has at least N younger entries behind it (gfx9: loads, stores and LDS-DMA share the counter and retire in order);
  * between an asm register load and the wait that retires it NO instruction may name one of its destination VGPRs (read or write).
The scan is linear in program text per function (the generated flush bodies are straight-line code); a load whose wait sits behind a loop back-edge - the first units
of a tile's write-out are requested in the previous tile's last stage - is left unjudged when the function ends, which the count printed at the end makes visible.
    python tools/check_asm_load_wait.py file.s [...]      exit status 1 when a destination register is touched while its load may still be in flight"""
import re
import sys

VMEM = ("global_load", "global_store", "buffer_load", "buffer_store", "flat_load", "flat_store", "global_atomic", "scratch_load", "scratch_store")


def vregs(tok):
    tok = tok.strip().split()[0] if tok.strip() else ""
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(path):
    """-> (violations, loads judged, loads left pending at a function end)"""
    bad, judged, unjudged = [], 0, 0
    pending = []          # [dest regs, younger VMEM ops, line, text]
    in_asm, fn = False, "?"
    for ln, raw in enumerate(open(path), 1):
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if t.endswith(":") and not t.startswith(".") and not t.startswith(";"):
            fn = t[:-1]
            unjudged += len(pending)
            pending = []
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        code = t.split(";")[0].strip()
        op, _, rest = code.partition(" ")
        ops = [o for o in rest.split(",")] if rest else []
        if op == "s_endpgm":
            unjudged += len(pending)
            pending = []
            continue
        m = re.search(r"vmcnt\((\d+)\)", code) if op == "s_waitcnt" else None
        if m:
            n = int(m.group(1))
            judged += sum(1 for p in pending if p[1] >= n)
            pending = [p for p in pending if p[1] < n]
            continue
        if op == "s_waitcnt" and "vmcnt" not in code and re.match(r"s_waitcnt\s+(0x[0-9a-f]+|\d+)$", code):   # raw immediate form: treat as vmcnt(0) only if the low bits say so
            continue
        touched = set()
        for o in ops:
            touched |= vregs(o)
        is_vmem = op.startswith(VMEM)
        for p in pending:
            if p[0] & touched:
                bad.append((path, ln, fn, code, p[2], p[3], p[1]))
        if is_vmem:
            for p in pending:
                p[1] += 1
            if in_asm and op.startswith("global_load") and "lds" not in op and ops:
                dst = vregs(ops[0])
                if dst:
                    pending.append([dst, 0, ln, code])
    unjudged += len(pending)
    return bad, judged, unjudged


if __name__ == "__main__":
    total_bad = 0
    for p in sys.argv[1:]:
        bad, judged, unjudged = scan(p)
        for path, ln, fn, code, lln, lcode, younger in bad:
            print(f"{path}:{ln}: {fn}: `{code}` touches a register of `{lcode}` (line {lln}) with only {younger} VMEM operation(s) behind it and no wait in between")
        print(f"{p}: {judged} asm register loads retired by a counted wait before any use, {unjudged} left pending at a function end / back-edge, {len(bad)} violations")
        total_bad += len(bad)
    sys.exit(1 if total_bad else 0)
