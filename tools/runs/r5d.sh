# round-5 profile set: same-box A/B (four-wave kernel on / off), rocprofv3 kernel trace of the headline run, PMC traffic passes, PMC pass on the new plain kernel, stage-3 lines
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d; mkdir -p $O
timeout 600 python bench.py --steps 8 --warmup 2 --no-extra --no-cpu-baseline > $O/bench_b30_default.json 2> $O/bench_b30_default.err
LHRS_GEMM_U4=0 timeout 600 python bench.py --steps 8 --warmup 2 --no-extra --no-cpu-baseline > $O/bench_b30_16wave_only.json 2> $O/bench_b30_16wave_only.err
timeout 600 python bench.py --steps 8 --warmup 2 --no-extra --no-cpu-baseline > $O/bench_b30_default2.json 2> $O/bench_b30_default2.err
timeout 400 python bench.py --steps 12 --warmup 3 --micro-batch 8 --no-extra --no-cpu-baseline > $O/bench_b8.json 2> $O/bench_b8.err
timeout 600 python bench.py --steps 6 --warmup 2 --micro-batch 60 --no-extra --no-cpu-baseline > $O/bench_b60.json 2> $O/bench_b60.err
timeout 600 python bench.py --steps 8 --warmup 2 --no-extra --no-cpu-baseline > $O/bench_b30_default3.json 2> $O/bench_b30_default3.err
timeout 600 python bench.py --stage 3 --micro-batch 32 --steps 6 --warmup 2 --no-extra --no-cpu-baseline > $O/bench_stage3_b32.json 2> $O/bench_stage3_b32.err
timeout 600 python bench.py --stage 3 --micro-batch 30 --steps 6 --warmup 2 --no-extra --no-cpu-baseline > $O/bench_stage3_b30.json 2> $O/bench_stage3_b30.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_b30 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_prof_b30.json 2> $GRAFT_REPO_ROOT/$O/bench_prof_b30.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $GRAFT_REPO_ROOT/$O/pmc_bench_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-extra --no-cpu-baseline > /dev/null 2>&1
done
for K in 4096 22016; do
  for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $pass | tr ' ' '_' | cut -c1-24)
    timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $pass -d $GRAFT_REPO_ROOT/$O/pmc_u4_K${K}_$tag -- python $GRAFT_REPO_ROOT/tools/gemm_one.py 2 8190 4096 $K 30 > /dev/null 2>&1
  done
done
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/prof_b30 5 2 r05_bench_b30 > $O/prof_summary.txt 2>&1
cp profiles/r05_bench_b30_kernel_stats.csv profiles/r05_bench_b30_gemm_launch_summary.json $O/ 2>/dev/null
python tools/pmc_summary.py traffic $O/pmc_bench_FETCH_SIZE $O/pmc_bench_WRITE_SIZE $O/r05_gemm_traffic.json "gemm_u4_kernel<0, false>" > $O/traffic_summary.txt 2>&1
for K in 4096 22016; do python tools/pmc_summary.py counters $O/pmc_u4_K${K}_SQ_VALU_MFMA_BUSY_CYCLES $O/pmc_u4_K${K}_SQ_ACTIVE_INST_ANY_SQ_BU $O/pmc_u4_K${K}_FETCH_SIZE $O/pmc_u4_K${K}_WRITE_SIZE $O/pmc_u4_new_K$K.csv gemm_u4 > /dev/null 2>&1; done
# keep the merged output small: drop the raw traces
rm -rf $O/prof_b30 $O/pmc_bench_FETCH_SIZE $O/pmc_bench_WRITE_SIZE $O/pmc_u4_K*
ls -la $O
