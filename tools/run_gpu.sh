cd $GRAFT_REPO_ROOT; O=gpurun_out/r6i; mkdir -p $O
C=lhrs_bot_amd/csrc
LHRS_HIP_LIB=$PWD/$C/liblhrs_d.so timeout 200 python tools/attn_diag.py 60 fwd > $O/attn_diag_fwd_base.txt 2>&1; cat $O/attn_diag_fwd_base.txt
timeout 300 python tools/attn_bench.py 60 2>&1 | grep -E "fwd|bwd_o"
