# PMC counters per gemm_u4_kernel instantiation at micro-batch 60 (raw launches of tools/time_u4_variants.py), separate passes, kernel-trace only
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc $pass -d $GRAFT_REPO_ROOT/$O/pass$i -- python $GRAFT_REPO_ROOT/tools/time_u4_variants.py 16380 > $GRAFT_REPO_ROOT/$O/pass$i.log 2>&1
  tail -2 $GRAFT_REPO_ROOT/$O/pass$i.log
done
cd $GRAFT_REPO_ROOT
for k in "0, true" "0, false" "1, false" "2, false" "3, false"; do
  python tools/pmc_summary.py counters $O/pass1 $O/pass2 $O/pass3 "$O/pmc_u4_$(echo $k | tr -d ' ,').csv" "gemm_u4_kernel<$k>" > /dev/null 2>&1
  echo "== gemm_u4_kernel<$k>"; cat "$O/pmc_u4_$(echo $k | tr -d ' ,').csv"
done
rm -rf $O/pass1 $O/pass2 $O/pass3
