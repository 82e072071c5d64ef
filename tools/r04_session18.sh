#!/bin/bash
# GPU session 18: which half-rate kinds hide a full-rate instruction (pipe model rows); threshold update every 4th / 8th event pass; per-pixel
# cost dumps (probe range and whole frame) of the bench frame and the 512^3 frame for the hand-out order study
O=gpurun_out/r04s; mkdir -p $O
timeout -k 10 200 build_ab/pipe_model > $O/pipe_model.txt 2>&1; head -32 $O/pipe_model.txt
timeout -k 10 500 python3 tools/ab.py --reps 2 --out $O/ab_th.json "shared_norm||build_ab/libtdtrt_norm.so" "th4||build_ab/libtdtrt_norm_th4.so" "th8||build_ab/libtdtrt_norm_th8.so" > $O/ab_th.txt 2>&1; tail -10 $O/ab_th.txt
timeout -k 10 200 python3 tools/experiments/dump_costs.py 5 $O/costs_c5.npz > $O/dump_c5.txt 2>&1; tail -3 $O/dump_c5.txt
timeout -k 10 200 python3 tools/experiments/dump_costs.py 2 $O/costs_c2.npz > $O/dump_c2.txt 2>&1; tail -3 $O/dump_c2.txt
ls -la $O
