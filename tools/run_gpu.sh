cd $GRAFT_REPO_ROOT; O=gpurun_out/r6relax; mkdir -p $O
C=$PWD/lhrs_bot_amd/csrc
for v in hip relax hip relax; do LHRS_HIP_LIB=$C/liblhrs_$v.so timeout 600 python bench.py --micro-batch 8 --steps 30 --warmup 5 --no-extra --no-cpu-baseline > $O/b8_$v.json 2>$O/err.txt; python - <<P
import json
r=json.loads([l for l in open("$O/b8_$v.json") if l.startswith("{")][-1])
print("$v", r["value"], r["ms_per_step"], r["roofline"]["four_wave_kernel_share_of_gemm_time"], {k.split(">")[0][-20:]:(v["avg_launch_us"], v["frac"]) for k,v in r["roofline"]["variants"].items()})
P
done
