#!/bin/bash
# GPU session 6: second slimming batch (no +0 adds in the in-octree test of UNIT builds, per-ray cost accounting, in-place leaf box): parity, A/B
O=gpurun_out/r04f; mkdir -p $O
TDT_LIB=$PWD/build_ab/lib_slim2.so timeout -k 10 900 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_prepass.py tests/test_gpu_variants.py tests/test_gpu_fuzz.py -q -m gpu -x > $O/parity_slim2.txt 2>&1; tail -4 $O/parity_slim2.txt
timeout -k 10 1000 python3 tools/ab.py --configs 2,3,5 --reps 2 --out $O/ab_slim2.json "product||-" "slim2||build_ab/lib_slim2.so" "slim2_lazy||build_ab/lib_slim2_lazy.so" > $O/ab_slim2.txt 2>&1; tail -10 $O/ab_slim2.txt
timeout -k 10 200 python3 tools/demo_time.py 100 > $O/demo_product.txt 2>&1; TDT_LIB=$PWD/build_ab/lib_slim2.so timeout -k 10 200 python3 tools/demo_time.py 100 > $O/demo_slim2.txt 2>&1; tail -3 $O/demo_product.txt $O/demo_slim2.txt
