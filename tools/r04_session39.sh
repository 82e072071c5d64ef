#!/bin/bash
# GPU session 39: the material tables' descriptors fetched from the kernel-argument segment where they are used (no spilled SGPRs in most builds): parity subset, A/B, demo
O=gpurun_out/r04al; mkdir -p $O
TDT_LIB=$PWD/build_ab/libtdtrt_cold.so timeout -k 10 400 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_variants.py tests/test_gpu_parity.py -m gpu -x -q > $O/parity.txt 2>&1; rc=$?; tail -2 $O/parity.txt
[ $rc -eq 0 ] && timeout -k 10 600 python3 tools/ab.py --reps 3 --out $O/ab_cold.json "product||-" "cold_args||build_ab/libtdtrt_cold.so" > $O/ab_cold.txt 2>&1; tail -8 $O/ab_cold.txt
for rep in 1 2; do timeout -k 10 100 python3 tools/demo_time.py 200 2>&1 | grep demo | sed "s/^/product: /"; TDT_LIB=$PWD/build_ab/libtdtrt_cold.so timeout -k 10 100 python3 tools/demo_time.py 200 2>&1 | grep demo | sed "s/^/cold: /"; done | tee $O/demo.txt
