cd $GRAFT_REPO_ROOT; O=gpurun_out/r6k; mkdir -p $O
C=$PWD/lhrs_bot_amd/csrc
LHRS_HIP_LIB=$C/liblhrs_w.so timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_pooler_gpu.py -x -q -k "attn or attention or rope or pooler" > $O/pytest_attn.txt 2>&1; tail -3 $O/pytest_attn.txt
for v in hip w; do LHRS_HIP_LIB=$C/liblhrs_$v.so timeout 300 python tools/attn_checksum.py 60 2>&1 | tail -1; done
for v in hip w hip w; do echo "== $v"; LHRS_HIP_LIB=$C/liblhrs_$v.so timeout 300 python tools/attn_bench.py 120 2>&1 | grep -E "fwd |bwd_o|rror"; done > $O/bench.txt 2>&1; cat $O/bench.txt
for v in hip w; do echo "== $v B=60"; LHRS_HIP_LIB=$C/liblhrs_$v.so timeout 300 python tools/attn_bench.py 60 2>&1 | grep -E "fwd |bwd_o"; done
