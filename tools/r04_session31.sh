#!/bin/bash
# GPU session 31: the stuck-ray cut in the deep builds outside the LDS table that have no bricks: parity subset on the candidate, timing of both
O=gpurun_out/r04ad; mkdir -p $O
TDT_LIB=$PWD/build_ab/libtdtrt_cutnr.so timeout -k 10 500 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_variants.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/parity.txt 2>&1; rc=$?; tail -3 $O/parity.txt
timeout -k 10 200 python3 tools/experiments/time_table_form_deep.py > $O/time_product.txt 2>&1; grep config $O/time_product.txt
TDT_LIB=$PWD/build_ab/libtdtrt_cutnr.so timeout -k 10 200 python3 tools/experiments/time_table_form_deep.py > $O/time_cutnr.txt 2>&1; grep config $O/time_cutnr.txt
