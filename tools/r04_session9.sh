#!/bin/bash
# GPU session 9: brick builds with fewer memo levels (registers) +- the second slimming batch +- the non-brick builds' register choices
O=gpurun_out/r04j; mkdir -p $O
timeout -k 10 1100 python3 tools/ab.py --configs 3,5 --reps 2 --out $O/ab_memo.json "product||-" "memo1||build_ab/lib_memo1.so" "memo3||build_ab/lib_memo3.so" "memo3_slim2_roomy||build_ab/lib_memo3_slim2_roomy.so" "memo1_slim2_roomy||build_ab/lib_memo1_slim2_roomy.so" > $O/ab_memo.txt 2>&1; tail -12 $O/ab_memo.txt
TDT_LIB=$PWD/build_ab/lib_memo1_slim2_roomy.so timeout -k 10 600 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -x > $O/parity_memo.txt 2>&1; tail -3 $O/parity_memo.txt
