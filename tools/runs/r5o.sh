# (experiment of 29 Sep: gemm_u4_kernel<4,false> split-K tail against the shipped small-tile tail, built as tools/tmp_ab/liblhrs_hip_oldtail.so; code removed afterwards: profiles/r05_tail_rows_four_wave_splitk_ab.txt)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5o; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "splitk or tail_rows or four_wave_kernel_bit_identical or lora_pair" > $O/kernel_tests.txt 2>&1; tail -3 $O/kernel_tests.txt
for lib in "" tools/tmp_ab/liblhrs_hip_oldtail.so ""; do
  LHRS_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 200 python tools/time_tail_rows.py 2>&1 | grep -v amdgpu.ids >> $O/tail_rows.txt
done
cat $O/tail_rows.txt
timeout 600 python bench.py --stage 3 --micro-batch 32 --no-extra --no-cpu-baseline > $O/stage3_b32.json 2> $O/stage3_b32.err; cut -c1-230 $O/stage3_b32.json
LHRS_HIP_LIB=$GRAFT_REPO_ROOT/tools/tmp_ab/liblhrs_hip_oldtail.so timeout 600 python bench.py --stage 3 --micro-batch 32 --no-extra --no-cpu-baseline > $O/stage3_b32_oldtail.json 2> $O/stage3_b32_oldtail.err; cut -c1-230 $O/stage3_b32_oldtail.json
