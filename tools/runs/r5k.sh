cd $GRAFT_REPO_ROOT
O=gpurun_out/r5k; mkdir -p $O
for args in "8736" "8190"; do timeout 120 python tools/repro_swb.py $args 2>&1 | grep -v amdgpu.ids | tail -3; done > $O/swb.txt; cat $O/swb.txt
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -k "lora or rope or u4 or tail_rows or swiglu" > $O/kernel_tests.txt 2>&1; tail -3 $O/kernel_tests.txt
timeout 600 python bench.py --stage 3 --micro-batch 32 --no-extra --no-cpu-baseline > $O/stage3_b32.json 2> $O/stage3_b32.err; cut -c1-260 $O/stage3_b32.json
timeout 600 python bench.py --stage 3 --micro-batch 30 --no-extra --no-cpu-baseline > $O/stage3_b30.json 2> $O/stage3_b30.err; cut -c1-260 $O/stage3_b30.json
timeout 900 python -m pytest tests/test_lora_gpu.py -x -q > $O/lora_tests.txt 2>&1; tail -3 $O/lora_tests.txt
