cd $GRAFT_REPO_ROOT; O=gpurun_out/a8; mkdir -p $O
C=lhrs_bot_amd/csrc
for v in base qo hip base hip; do echo "== $v"; LHRS_HIP_LIB=$PWD/$C/liblhrs_$v.so timeout 300 python tools/attn_bench.py 60 2>&1 | grep -E "fwd |bwd_o"; done > $O/bench.txt 2>&1
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -k "attn or attention or rope" > $O/pytest.txt 2>&1
cat $O/bench.txt; tail -5 $O/pytest.txt
