cd $GRAFT_REPO_ROOT; O=gpurun_out/r6suite; mkdir -p $O
( time timeout 3400 python -m pytest tests/ -x -q -m gpu ) > $O/gpu_suite.txt 2>&1; tail -6 $O/gpu_suite.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
