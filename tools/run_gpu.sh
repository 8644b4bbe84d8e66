cd $GRAFT_REPO_ROOT; O=gpurun_out/r6b240; mkdir -p $O
for b in 120 240 120 240; do timeout 900 python bench.py --micro-batch $b --steps 6 --warmup 2 --no-extra --no-cpu-baseline > $O/b$b.json 2>$O/err$b.txt; python - <<P
import json
try:
    r=json.loads(open("$O/b$b.json").read().strip().splitlines()[-1])
    print($b, r["value"], r["ms_per_step"], r["roofline"]["frac"], r["step_mfma_frac"])
except Exception as e:
    print($b, "failed", e, open("$O/err$b.txt").read()[-400:])
P
done
python -c "import torch; print(torch.cuda.mem_get_info())"
