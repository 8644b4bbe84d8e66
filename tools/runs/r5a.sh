set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r5a/kernels_test.txt
timeout 600 python bench.py --steps 8 --warmup 2 --no-extra --no-cpu-baseline > gpurun_out/r5a/bench_b30.json 2> gpurun_out/r5a/bench_b30.err
timeout 400 python bench.py --steps 12 --warmup 3 --micro-batch 8 --no-extra --no-cpu-baseline > gpurun_out/r5a/bench_b8.json 2> gpurun_out/r5a/bench_b8.err
cd /tmp && export TMPDIR=/tmp
for K in 4096 22016; do
  for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $pass | tr ' ' '_' | cut -c1-24)
    timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $pass -d $GRAFT_REPO_ROOT/gpurun_out/r5a/pmc_u4_K${K}_$tag -- python $GRAFT_REPO_ROOT/tools/gemm_one.py 2 8190 4096 $K 30 > /dev/null 2>&1
  done
done
ls $GRAFT_REPO_ROOT/gpurun_out/r5a
