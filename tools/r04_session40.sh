#!/bin/bash
# GPU session 40: brick builds with the material descriptors fetched at use (2 spilled SGPRs) and one Rand(hit.xy): the whole -m gpu suite, then the A/B against the build before
O=gpurun_out/r04am; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1; rc=$?; tail -3 $O/gpu_suite.txt
[ $rc -eq 0 ] && timeout -k 10 600 python3 tools/ab.py --reps 3 --out $O/ab_final.json "before||build_ab/libtdtrt_memo2.so" "product||-" > $O/ab_final.txt 2>&1; tail -8 $O/ab_final.txt
