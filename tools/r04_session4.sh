#!/bin/bash
# GPU session 4: traversal-pass slimming variants (mask algebra instead of state compares; no SGPR operands / canonicalising max in the step): parity, then A/B
O=gpurun_out/r04d; mkdir -p $O
TDT_LIB=$PWD/build_ab/lib_slim_rand2.so timeout -k 10 600 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_prepass.py -q -m gpu -x > $O/parity_slim_rand2.txt 2>&1; tail -3 $O/parity_slim_rand2.txt
timeout -k 10 1000 python3 tools/ab.py --configs 2,3,5 --reps 2 --out $O/ab_slim.json "base||-" "masks||build_ab/lib_masks.so" "slim||build_ab/lib_slim.so" "slim_rand2||build_ab/lib_slim_rand2.so" > $O/ab_slim.txt 2>&1; tail -14 $O/ab_slim.txt
