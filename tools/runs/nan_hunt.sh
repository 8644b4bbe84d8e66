# usage: bash tools/runs/nan_hunt.sh <launches> <tag>   - the 8-ranks-on-one-device plumbing run, looped; keeps the stderr of every failing launch
cd $GRAFT_REPO_ROOT
N=$1; TAG=$2; OUT=gpurun_out/$TAG; mkdir -p $OUT
ok=0; bad=0; t_start=$(date +%s)
for i in $(seq 1 $N); do
  LHRS_SHARE_GPU=1 OMP_NUM_THREADS=2 LHRS_BENCH_HOST_INTS=${HOST_INTS:-0} LHRS_BENCH_TRACE_FINITE=${TRACE_LEVEL:-1} timeout 600 python bench.py --gpus 8 --steps 2 --warmup 1 --llama-layers 1 --micro-batch 2 > $OUT/run_$i.out 2> $OUT/run_$i.err
  rc=$?
  if [ $rc -eq 0 ]; then ok=$((ok+1)); grep -h "finite-trace" $OUT/run_$i.err | grep -v "all finite" >> $OUT/trace_anomalies.txt; rm -f $OUT/run_$i.out; grep -v "Gloo\|amdgpu.ids" $OUT/run_$i.err | grep -c "finite-trace" > /dev/null; rm -f $OUT/run_$i.err
  else bad=$((bad+1)); grep -v "Gloo" $OUT/run_$i.err > $OUT/FAIL_$i.txt; rm -f $OUT/run_$i.err; fi
  echo "launch $i rc=$rc ok=$ok bad=$bad elapsed=$(( $(date +%s) - t_start ))s" >> $OUT/progress.txt
done
echo "launches=$N ok=$ok bad=$bad seconds=$(( $(date +%s) - t_start ))" | tee $OUT/summary.txt
