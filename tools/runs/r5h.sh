set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h; mkdir -p $O
timeout 200 python tools/repro_pair_8736.py 8736 > $O/repro_8736.txt 2>&1; tail -12 $O/repro_8736.txt
timeout 200 python tools/time_lora_pair.py 8190 > $O/time_lora_pair.txt 2>&1; cat $O/time_lora_pair.txt
