#!/bin/bash
# GPU session 3 of round 4: pipe-model micro-benchmark, the whole -m gpu suite on the round's product build, A/B of carry-traffic variants
O=gpurun_out/r04c; mkdir -p $O
timeout -k 10 200 build_ab/pipe_model > $O/pipe_model.txt 2>&1; echo "micro done"
timeout -k 10 900 python3 -m pytest tests -q -m gpu -x > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
timeout -k 10 900 python3 tools/ab.py --configs 3,5 --reps 2 --out $O/ab_carry.json "base||-" "nocarryio||build_ab/lib_nocarryio.so" "lazyroot||build_ab/lib_lazyroot.so" > $O/ab_carry.txt 2>&1; tail -8 $O/ab_carry.txt
