#!/bin/bash
# GPU session 5: the call sites' records in memory instead of registers (exact): parity, then A/B against the slim build
O=gpurun_out/r04e; mkdir -p $O
TDT_LIB=$PWD/build_ab/lib_slim_memcarry.so timeout -k 10 900 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_prepass.py tests/test_gpu_fullsize.py tests/test_gpu_multi.py tests/test_gpu_fuzz.py -q -m gpu -x > $O/parity_memcarry.txt 2>&1; tail -4 $O/parity_memcarry.txt
timeout -k 10 1000 python3 tools/ab.py --configs 2,3,5 --reps 2 --out $O/ab_memcarry.json "slim||build_ab/lib_slim.so" "memcarry||build_ab/lib_memcarry.so" "slim_memcarry||build_ab/lib_slim_memcarry.so" > $O/ab_memcarry.txt 2>&1; tail -10 $O/ab_memcarry.txt
