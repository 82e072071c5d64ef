#!/bin/bash
# GPU session 8: gate on the scalar unit (SLIM3), branchless leaf tail (SLIM4) against the product and SLIM2: parity on the last, A/B
O=gpurun_out/r04h; mkdir -p $O
TDT_LIB=$PWD/build_ab/lib_slim4.so timeout -k 10 900 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_prepass.py tests/test_gpu_variants.py tests/test_gpu_fuzz.py -q -m gpu -x > $O/parity_slim4.txt 2>&1; tail -4 $O/parity_slim4.txt
timeout -k 10 1100 python3 tools/ab.py --configs 2,3,5 --reps 2 --out $O/ab_slim34.json "product||-" "slim2||build_ab/lib_slim2.so" "slim3only||build_ab/lib_slim3only.so" "slim23||build_ab/lib_slim3.so" "slim234||build_ab/lib_slim4.so" > $O/ab_slim34.txt 2>&1; tail -16 $O/ab_slim34.txt
