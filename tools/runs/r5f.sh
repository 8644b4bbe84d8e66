set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5f; mkdir -p $O
timeout 600 python bench.py --no-extra --no-cpu-baseline > $O/cold_default.json 2> $O/cold_default.err
timeout 600 python bench.py --no-extra --no-cpu-baseline > $O/warm_default.json 2> $O/warm_default.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/driver_like.json 2> $O/driver_like.err
timeout 900 python bench.py --steps 20 --warmup 5 > $O/driver_like_full.json 2> $O/driver_like_full.err
