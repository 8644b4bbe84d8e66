cd $GRAFT_REPO_ROOT; O=gpurun_out/r6soak2; mkdir -p $O gpurun_out/soak
SOAK_LAYERS=32 timeout 1500 python tools/soak_idle_queue.py trainer 2000 > $O/soak_trainer32.txt 2>&1; grep "soak trainer" $O/soak_trainer32.txt || tail -5 $O/soak_trainer32.txt
LHRS_BENCH_IDLE_START_MS=0 timeout 2400 python tools/soak_idle_queue.py share8 30 40 > $O/soak_share8_noidle.txt 2>&1; tail -1 $O/soak_share8_noidle.txt
timeout 2400 python tools/soak_idle_queue.py share8 30 40 > $O/soak_share8_idle.txt 2>&1; tail -1 $O/soak_share8_idle.txt
grep -c "all finite" gpurun_out/soak/share8_log.txt; grep "FAIL" gpurun_out/soak/share8_log.txt | cut -c1-300
