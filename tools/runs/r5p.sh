# (experiment of 29 Sep; the LHRS_U4_STAGGER switch it sets existed only for this run and was removed with the experiment: profiles/r05_u4_xcd_stagger_ab.txt)
# experiment: start delay per XCD (x * stagger * ~1 us) to take the write-out bursts of the 256 CUs out of lock-step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5p; mkdir -p $O
for st in 0 1 2 4 0; do
  echo "=== LHRS_U4_STAGGER=$st" >> $O/stagger.txt
  LHRS_U4_STAGGER=$st timeout 200 python tools/time_u4_variants.py 16380 2>&1 | grep -v amdgpu.ids >> $O/stagger.txt
done
cat $O/stagger.txt
