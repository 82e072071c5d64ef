#!/bin/bash
# GPU session 11: the round's product configuration (slim + slim2 everywhere, three memo levels in the brick builds): whole -m gpu suite; A/B of two small variants
O=gpurun_out/r04l; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -q -m gpu -x > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
timeout -k 10 1000 python3 tools/ab.py --configs 2,3,5 --reps 2 --out $O/ab_small.json "product||-" "full_sh||build_ab/lib_fullsh.so" "brick_rand2||build_ab/lib_brickrand2.so" > $O/ab_small.txt 2>&1; tail -10 $O/ab_small.txt
