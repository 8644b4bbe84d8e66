# round-5 closing set after the SGPR-hazard fix and the LoRA pair on the four-wave kernel: driver-like default line, kernel trace at the default micro-batch, full GPU suite
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5m; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default_full.json 2> $O/bench_default_full.err
cut -c1-400 $O/bench_default_full.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_b60 -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-extra --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_prof_b60.json 2> $GRAFT_REPO_ROOT/$O/bench_prof_b60.err
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/prof_b60 4 2 r05_bench_b60 > $O/prof_summary.txt 2>&1
cp profiles/r05_bench_b60_kernel_stats.csv profiles/r05_bench_b60_gemm_launch_summary.json $O/ 2>/dev/null
rm -rf $O/prof_b60
timeout 1100 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1
tail -4 $O/gpu_tests.txt
