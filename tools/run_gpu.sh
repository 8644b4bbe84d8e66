cd $GRAFT_REPO_ROOT; O=gpurun_out/r6final; mkdir -p $O
R=$GRAFT_REPO_ROOT
( time timeout 2400 python -m pytest tests/ -x -q -m gpu ) > $O/gpu_suite.txt 2>&1; tail -4 $O/gpu_suite.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt; head -c 300 $O/bench_line.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/bench.py --steps 4 --warmup 2 --no-extra --no-cpu-baseline > $R/$O/bench_prof_line.json 2> $R/$O/prof_err.txt
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_f -- python $R/bench.py --steps 1 --warmup 1 --no-extra --no-cpu-baseline > /dev/null 2> $R/$O/pmc_f_err.txt
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_w -- python $R/bench.py --steps 1 --warmup 1 --no-extra --no-cpu-baseline > /dev/null 2> $R/$O/pmc_w_err.txt
cd $R
python tools/prof_summary.py $O/prof 4 2 r06_bench_b60 > $O/prof_summary.txt 2>&1
python tools/pmc_summary.py traffic $O/pmc_f $O/pmc_w profiles/r06_gemm_traffic.json "gemm_u4_kernel<0, false>" 60 > $O/pmc_summary.txt 2>&1; tail -3 $O/pmc_summary.txt
cp profiles/r06_bench_b60_* profiles/r06_gemm_traffic.json $O/ 2>/dev/null
rm -rf $O/prof $O/pmc_f $O/pmc_w
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
