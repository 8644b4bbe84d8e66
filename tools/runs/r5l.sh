# (experiments of 29 Sep: variant libraries built into tools/tmp_ab/ for one run, not kept)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5l; mkdir -p $O
for M in 16380 8190; do
for lib in "" tools/tmp_ab/liblhrs_hip_v1.so tools/tmp_ab/liblhrs_hip_v2.so ""; do
  LHRS_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 300 python tools/time_u4_variants.py $M 2>&1 | grep -v amdgpu.ids >> $O/variants.txt
done
done
cat $O/variants.txt
