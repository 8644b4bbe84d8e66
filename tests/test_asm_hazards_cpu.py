"""ISA-level check of the hand-written GEMM kernels (no GPU): "VALU writes SGPR -> VMEM reads that SGPR" needs 5 wait states on gfx9, and the compiler pads that
hazard for its own instructions only - a VMEM instruction inside an asm statement whose scalar address the register allocator reloaded from a spilled lane
(v_readlane_b32) one instruction earlier goes out with the old register contents.  Round 5 met it as a memory access fault of gemm_u4_kernel<2, false> once a change
elsewhere in the kernel made its stage pointers spill.  csrc's asm statements therefore copy every scalar pointer through an SALU move first; this test compiles the
two files that hold such statements to ISA and lets tools/check_asm_sgpr_hazard.py look at every VMEM instruction between ASMSTART / ASMEND."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_asm_sgpr_hazard as chk  # noqa: E402
import check_asm_load_wait as chk_wait  # noqa: E402

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def test_scanner_flags_a_reloaded_pointer_in_front_of_an_asm_vmem(tmp_path):
    bad = tmp_path / "bad.s"
    bad.write_text("k:\n\tv_readlane_b32 s22, v211, 17\n\tv_readlane_b32 s23, v211, 18\n\t;;#ASMSTART\n\tglobal_load_lds_dwordx4 v186, s[22:23]\n\t;;#ASMEND\n")
    ok = tmp_path / "ok.s"
    ok.write_text("k:\n\tv_readlane_b32 s22, v211, 17\n\tv_readlane_b32 s23, v211, 18\n\t;;#ASMSTART\n\ts_mov_b64 s[100:101], s[22:23]\n\tglobal_load_lds_dwordx4 v186, s[100:101]\n\t;;#ASMEND\n")
    padded = tmp_path / "padded.s"
    padded.write_text("k:\n\tv_readfirstlane_b32 s4, v1\n\ts_nop 4\n\t;;#ASMSTART\n\tglobal_store_dwordx4 v2, v[4:7], s[4:5] offset:64\n\t;;#ASMEND\n")
    assert len(chk.scan(str(bad))) == 1 and chk.scan(str(ok)) == [] and chk.scan(str(padded)) == []


def test_load_wait_scanner_replays_in_order_retirement(tmp_path):
    """tools/check_asm_load_wait.py: a register of an asm-issued load may not be named before a counted vmcnt has retired the load (round 6)."""
    load = "\t;;#ASMSTART\n\tglobal_load_dwordx4 v[10:13], v2, s[100:101] offset:64\n\t;;#ASMEND\n"
    store = "\tglobal_store_dwordx4 v[30:31], v[40:43], off\n"
    def run(body):
        f = tmp_path / "k.s"
        f.write_text("k:\n" + body + "\ts_endpgm\n")
        return chk_wait.scan(str(f))
    # two stores behind the load: vmcnt(2) retires it, vmcnt(3) does not
    ok = run(load + store + store + "\ts_waitcnt vmcnt(2)\n\tv_add_f32 v1, v10, v1\n")
    assert ok[0] == [] and ok[1] == 1
    early = run(load + store + store + "\ts_waitcnt vmcnt(3)\n\tv_add_f32 v1, v10, v1\n")
    assert len(early[0]) == 1
    copied = run(load + "\tv_mov_b32 v50, v12\n" + store + "\ts_waitcnt vmcnt(0)\n")       # a register-allocator copy between load and wait
    assert len(copied[0]) == 1
    lds_dma = run("\t;;#ASMSTART\n\tglobal_load_lds_dwordx4 v6, s[100:101]\n\t;;#ASMEND\n\tv_mov_b32 v6, 0\n")   # LDS-DMA has no register destination
    assert lds_dma[0] == []


@pytest.mark.timeout(600)
@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("src", ["gemm_u4.hip", "gemm.hip"])
def test_no_unpadded_valu_sgpr_write_in_front_of_an_asm_vmem(src, tmp_path):
    out = tmp_path / (src + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-w", os.path.join(ROOT, "lhrs_bot_amd", "csrc", src), "-o", str(out)],
                   check=True, capture_output=True, timeout=540)
    text = out.read_text()
    assert text.count("global_load_lds_dwordx4") > 30                      # the asm statements are in there
    bad = chk.scan(str(out))
    assert bad == [], "\n".join(f"{ln}: {t}  <- {wop} {st} wait state(s) earlier" for _, ln, _, t, wop, st in bad)
    # the write-out's asm register loads (GLD) against the counted waits that release them (FL_WAITN): no copy, spill or early use in between
    early, judged, _ = chk_wait.scan(str(out))
    assert early == [], "\n".join(f"{ln}: {code}  <- load at line {lln} ({lcode}), {y} operation(s) behind it" for _, ln, _, code, lln, lcode, y in early)
    if src == "gemm_u4.hip":
        assert judged >= 100, judged   # the residual / SwiGLU' / RoPE write-outs are in there
    ver = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout
    open(os.path.join(ROOT, "gpurun_out", f"asm_checks_{src}.txt"), "w").write(f"{src}: {judged} asm register loads judged, 0 early uses, 0 SGPR hazards; compiler:\n{ver}") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None
