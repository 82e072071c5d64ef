#!/bin/bash
# GPU session 17: one normalize() behind the three material branches — parity subset on the candidate, then the A/B against the same source without it
O=gpurun_out/r04r; mkdir -p $O
TDT_LIB=$PWD/build_ab/libtdtrt_norm.so timeout -k 10 500 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_variants.py -m gpu -x -q > $O/parity_norm.txt 2>&1; rc=$?; tail -3 $O/parity_norm.txt
[ $rc -eq 0 ] && timeout -k 10 600 python3 tools/ab.py --reps 2 --out $O/ab_norm.json "base||build_ab/libtdtrt_base.so" "shared_norm||build_ab/libtdtrt_norm.so" > $O/ab_norm.txt 2>&1; tail -8 $O/ab_norm.txt
