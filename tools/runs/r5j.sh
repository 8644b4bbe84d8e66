# (experiments of 29 Sep: variant libraries built into tools/tmp_ab/ for one run, not kept)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5j; mkdir -p $O
for lib in "" tools/tmp_ab/liblhrs_hip_varb.so; do
  for args in "8736" "8704" "8960" "8190" "8736 11008 1024" "8736 4096 4096" "8736 2048 4096"; do
    echo "=== lib=[$lib] args=[$args]" >> $O/swb.txt
    LHRS_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 120 python tools/repro_swb.py $args 2>&1 | grep -v amdgpu.ids | tail -4 >> $O/swb.txt
  done
done
cat $O/swb.txt
