cd $GRAFT_REPO_ROOT; O=gpurun_out/r6final240; mkdir -p $O
R=$GRAFT_REPO_ROOT
( time timeout 3000 python -m pytest tests/test_parity_gpu.py -x -q -k "test_measured_micro_batches_end_to_end_vs_oracle and 240" ) > $O/parity240.txt 2>&1; tail -5 $O/parity240.txt; cat gpurun_out/parity_micro_batch_240_*.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt; head -c 250 $O/bench_line.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/bench.py --steps 4 --warmup 2 --no-extra --no-cpu-baseline > $R/$O/bench_prof_line.json 2> $R/$O/prof_err.txt
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_f -- python $R/bench.py --steps 1 --warmup 1 --no-extra --no-cpu-baseline > /dev/null 2> $R/$O/pmc_f_err.txt
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_w -- python $R/bench.py --steps 1 --warmup 1 --no-extra --no-cpu-baseline > /dev/null 2> $R/$O/pmc_w_err.txt
cd $R
python tools/prof_summary.py $O/prof 4 2 r06_bench_b240 > $O/prof_summary.txt 2>&1
python tools/pmc_summary.py traffic $O/pmc_f $O/pmc_w profiles/r06_gemm_traffic_b240.json "gemm_u4_kernel<0, false>" 240 > $O/pmc_summary.txt 2>&1; tail -1 $O/pmc_summary.txt | cut -c1-200
cp profiles/r06_bench_b240_* profiles/r06_gemm_traffic_b240.json $O/ 2>/dev/null
rm -rf $O/prof $O/pmc_f $O/pmc_w
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_line_2.json 2> $O/bench_err2.txt; head -c 250 $O/bench_line_2.json; echo
timeout 1500 python -m pytest tests/test_bench_gpu.py -x -q -k "contract or decode_line" > $O/bench_tests.txt 2>&1; tail -3 $O/bench_tests.txt
