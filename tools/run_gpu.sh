cd $GRAFT_REPO_ROOT; O=gpurun_out/r6e; mkdir -p $O
timeout 900 python -m pytest tests/test_generate_gpu.py -x -q -k "chain" > $O/pytest_chain.txt 2>&1; tail -5 $O/pytest_chain.txt
timeout 600 python bench.py --decode --weights fp8 --new-tokens 512 --steps 2 > $O/decode_fp8_chain.json 2>$O/err1.txt; cut -c1-330 $O/decode_fp8_chain.json
LHRS_DECODE_CHAIN=0 timeout 600 python bench.py --decode --weights fp8 --new-tokens 512 --steps 2 > $O/decode_fp8_nochain.json 2>$O/err2.txt; cut -c1-330 $O/decode_fp8_nochain.json
