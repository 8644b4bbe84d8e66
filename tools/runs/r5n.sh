# larger micro-batches on the 288 GB part (data point; the default stays 60) + the driver's smoke entry on the final tree
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5n; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 400 python bench.py --micro-batch 120 --steps 4 --warmup 2 --no-extra --no-cpu-baseline > $O/bench_b120.json 2> $O/bench_b120.err; cut -c1-230 $O/bench_b120.json; tail -2 $O/bench_b120.err
timeout 300 python bench.py --micro-batch 60 --steps 6 --warmup 2 --no-extra --no-cpu-baseline > $O/bench_b60.json 2> $O/bench_b60.err; cut -c1-230 $O/bench_b60.json
