#!/bin/bash
# GPU session 13: verified fast division in the primary ray; the event threshold's r after the slimming
O=gpurun_out/r04n; mkdir -p $O
TDT_LIB=$PWD/build_ab/lib_fastdiv.so timeout -k 10 600 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_prepass.py tests/test_gpu_fuzz.py -q -m gpu -x > $O/parity_fastdiv.txt 2>&1; tail -3 $O/parity_fastdiv.txt
timeout -k 10 1100 python3 tools/ab.py --configs 2,3,5 --reps 2 --out $O/ab.json "product||-" "fastdiv||build_ab/lib_fastdiv.so" "k_x0.8|TDT_EVENT_K_SCALE=0.8|-" "k_x1.25|TDT_EVENT_K_SCALE=1.25|-" > $O/ab.txt 2>&1; tail -13 $O/ab.txt
