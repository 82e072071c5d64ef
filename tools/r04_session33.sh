#!/bin/bash
# GPU session 33: a long fuzz campaign on the round's final kernels
O=gpurun_out/r04af; mkdir -p $O
timeout -k 10 1000 python3 tools/fuzz_parity.py 900 20261008 > $O/fuzz.txt 2>&1; tail -3 $O/fuzz.txt
