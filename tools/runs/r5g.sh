set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5g; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "lora or rope or u4 or tail_rows" > $O/kernel_tests.txt 2>&1; tail -3 $O/kernel_tests.txt
timeout 300 python tools/time_lora_pair.py 8190 > $O/time_lora_pair.txt 2>&1
timeout 300 python tools/time_lora_pair.py 8736 >> $O/time_lora_pair.txt 2>&1; cat $O/time_lora_pair.txt
timeout 600 python bench.py --stage 3 --micro-batch 30 --no-extra --no-cpu-baseline > $O/stage3_b30.json 2> $O/stage3_b30.err; cat $O/stage3_b30.json | cut -c1-300
timeout 600 python bench.py --stage 3 --micro-batch 32 --no-extra --no-cpu-baseline > $O/stage3_b32.json 2> $O/stage3_b32.err; cat $O/stage3_b32.json | cut -c1-300
timeout 900 python -m pytest tests/test_lora_gpu.py tests/test_parity_gpu.py -x -q > $O/lora_parity_tests.txt 2>&1; tail -3 $O/lora_parity_tests.txt
