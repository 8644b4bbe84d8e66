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

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def test_scanner_flags_a_reloaded_pointer_in_front_of_an_asm_vmem(tmp_path):
    bad = tmp_path / "bad.s"
    bad.write_text("k:\n\tv_readlane_b32 s22, v211, 17\n\tv_readlane_b32 s23, v211, 18\n\t;;#ASMSTART\n\tglobal_load_lds_dwordx4 v186, s[22:23]\n\t;;#ASMEND\n")
    ok = tmp_path / "ok.s"
    ok.write_text("k:\n\tv_readlane_b32 s22, v211, 17\n\tv_readlane_b32 s23, v211, 18\n\t;;#ASMSTART\n\ts_mov_b64 s[100:101], s[22:23]\n\tglobal_load_lds_dwordx4 v186, s[100:101]\n\t;;#ASMEND\n")
    padded = tmp_path / "padded.s"
    padded.write_text("k:\n\tv_readfirstlane_b32 s4, v1\n\ts_nop 4\n\t;;#ASMSTART\n\tglobal_store_dwordx4 v2, v[4:7], s[4:5] offset:64\n\t;;#ASMEND\n")
    assert len(chk.scan(str(bad))) == 1 and chk.scan(str(ok)) == [] and chk.scan(str(padded)) == []


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
