set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -40 > gpurun_out/r5c/gpu_tests.txt
tail -5 gpurun_out/r5c/gpu_tests.txt
