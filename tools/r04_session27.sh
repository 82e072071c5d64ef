#!/bin/bash
# GPU session 27: rays that stop advancing leave the traversal loop at once (brick builds): the whole -m gpu suite on the candidate, then the A/B
O=gpurun_out/r04aa; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1; rc=$?; tail -3 $O/gpu_suite.txt
[ $rc -eq 0 ] && timeout -k 10 900 python3 tools/ab.py --reps 2 --out $O/ab_stuck.json "no_cut||build_ab/libtdtrt_nocut.so" "stuck_cut||-" > $O/ab_stuck.txt 2>&1; tail -8 $O/ab_stuck.txt
