# round-5 final profile set at the default micro-batch (60): full default line, kernel trace, PMC traffic passes, A/B at micro-batch 30 with the SwiGLU' rule
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e; mkdir -p $O
timeout 900 python bench.py > $O/bench_default_full.json 2> $O/bench_default_full.err
timeout 600 python bench.py --micro-batch 30 --steps 8 --warmup 2 --no-extra --no-cpu-baseline > $O/bench_b30.json 2> $O/bench_b30.err
LHRS_GEMM_U4=0 timeout 600 python bench.py --steps 6 --warmup 2 --no-extra --no-cpu-baseline > $O/bench_b60_16wave_only.json 2> $O/bench_b60_16wave_only.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_b60 -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-extra --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_prof_b60.json 2> $GRAFT_REPO_ROOT/$O/bench_prof_b60.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $GRAFT_REPO_ROOT/$O/pmc_bench_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-extra --no-cpu-baseline > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/prof_b60 4 2 r05_bench_b60 > $O/prof_summary.txt 2>&1
cp profiles/r05_bench_b60_kernel_stats.csv profiles/r05_bench_b60_gemm_launch_summary.json $O/ 2>/dev/null
python tools/pmc_summary.py traffic $O/pmc_bench_FETCH_SIZE $O/pmc_bench_WRITE_SIZE $O/r05_gemm_traffic.json "gemm_u4_kernel<0, false>" > $O/traffic_summary.txt 2>&1
rm -rf $O/prof_b60 $O/pmc_bench_FETCH_SIZE $O/pmc_bench_WRITE_SIZE
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "u4 or fused or gemm" 2>&1 | tail -5 > $O/kernel_tests.txt
ls -la $O; tail -3 $O/kernel_tests.txt
