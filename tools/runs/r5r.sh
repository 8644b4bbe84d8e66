# (diagnostic of 29 Sep: what a tile's write-out costs - the same kernels with the flush stores removed / with the whole in-stream write-out of the plain instantiation removed; results of those builds are WRONG by construction)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5r; mkdir -p $O
for lib in "" tools/tmp_ab/liblhrs_hip_nostore.so tools/tmp_ab/liblhrs_hip_noflush.so; do
  LHRS_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 200 python tools/time_u4_variants.py 16380 2>&1 | grep -v amdgpu.ids >> $O/writeout_cost.txt
done
cat $O/writeout_cost.txt
