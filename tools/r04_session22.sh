#!/bin/bash
# GPU session 22: per-pixel start / end times of the product's main launch on the 512^3 frame (one-class order, the product's) and its replay
O=gpurun_out/r04w; mkdir -p $O
TDT_NO_HIT_CLASS=1 TDT_LIB=$PWD/build_ab/lib_stats.so timeout -k 10 300 python3 tools/experiments/pixel_times.py 5 $O/times_c5.npz > $O/times_c5.txt 2>&1; cat $O/times_c5.txt | tail -4
ls -la $O
