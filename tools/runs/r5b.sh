set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -m gpu 2>&1 | tail -60 > gpurun_out/r5b/kernels_test.txt
timeout 600 python bench.py --steps 8 --warmup 2 --no-extra --no-cpu-baseline > gpurun_out/r5b/bench_b30.json 2> gpurun_out/r5b/bench_b30.err
tail -5 gpurun_out/r5b/kernels_test.txt; tail -c 600 gpurun_out/r5b/bench_b30.err
