#!/bin/bash
# GPU session 10: brick builds — memo levels 1 / 2 / 3 with the second slimming batch and the non-brick register choices; parity of the candidates
O=gpurun_out/r04k; mkdir -p $O
for L in memo1_slim2_roomy memo2_slim2_roomy memo3_slim2_roomy; do
  TDT_LIB=$PWD/build_ab/lib_$L.so timeout -k 10 600 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -q -m gpu -x > $O/parity_$L.txt 2>&1; echo $L; tail -2 $O/parity_$L.txt
done
timeout -k 10 1100 python3 tools/ab.py --configs 3,5 --reps 2 --out $O/ab_memo.json "product||-" "memo1_slim2_roomy||build_ab/lib_memo1_slim2_roomy.so" "memo2_slim2_roomy||build_ab/lib_memo2_slim2_roomy.so" "memo2_slim2||build_ab/lib_memo2_slim2.so" "memo3_slim2_roomy||build_ab/lib_memo3_slim2_roomy.so" > $O/ab_memo.txt 2>&1; tail -12 $O/ab_memo.txt
