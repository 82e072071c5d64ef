#!/bin/bash
# GPU session 35: who ends the reference's own frame (demo scene, 1280x720, 4 spp)?  per-pixel start / end times of its history-free launch and its replay
O=gpurun_out/r04ah; mkdir -p $O
TDT_LIB=$PWD/build_ab/lib_stats.so timeout -k 10 200 python3 tools/experiments/pixel_times.py 0 $O/times_c0.npz > $O/times_c0.txt 2>&1; tail -3 $O/times_c0.txt
python3 tools/experiments/pixel_times_report.py $O/times_c0.npz > $O/report_c0.txt 2>&1; cat $O/report_c0.txt
