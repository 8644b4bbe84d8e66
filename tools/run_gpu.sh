cd $GRAFT_REPO_ROOT; O=gpurun_out/r6c; mkdir -p $O gpurun_out/soak
timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -k "test_measured_micro_batches_end_to_end_vs_oracle and 8" > $O/pytest_parity8.txt 2>&1
tail -8 $O/pytest_parity8.txt; cat gpurun_out/parity_micro_batch_8_projector_backward_alone.txt
timeout 900 python tools/soak_idle_queue.py trainer 3000 > $O/soak_trainer.txt 2>&1; grep "soak trainer" $O/soak_trainer.txt || tail -5 $O/soak_trainer.txt
timeout 900 python tools/soak_idle_queue.py bench1 2000 > $O/soak_bench1.txt 2>&1; tail -2 $O/soak_bench1.txt
timeout 2400 python tools/soak_idle_queue.py share8 12 40 > $O/soak_share8.txt 2>&1; tail -2 $O/soak_share8.txt
cat gpurun_out/soak/*_log.txt | tail -20
