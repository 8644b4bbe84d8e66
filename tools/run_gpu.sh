cd $GRAFT_REPO_ROOT; O=gpurun_out/r6a; mkdir -p $O
C=lhrs_bot_amd/csrc
timeout 300 python tools/attn_bench.py 60 > $O/attn_bench.txt 2>&1
for k in fwd dq dkv; do LHRS_HIP_LIB=$PWD/$C/liblhrs_d.so timeout 200 python tools/attn_diag.py 60 $k; done > $O/attn_diag.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-extra --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_prof_line.json 2> $GRAFT_REPO_ROOT/$O/prof_err.txt
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/prof 4 2 r06_baseline_b60 > $O/prof_summary.txt 2>&1
cp profiles/r06_baseline_b60_* $O/ 2>/dev/null
rm -rf $O/prof
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attn or attention or rope" > $O/pytest_attn.txt 2>&1
cat $O/attn_bench.txt $O/attn_diag.txt; tail -3 $O/pytest_attn.txt; head -c 600 $O/bench_line.json
