#!/bin/bash
# GPU session 19: product = shared normalize + threshold update every 8th event pass: parity subset; A/B of the stale-record cost bonus
# (2048 / 8192 / 32768 per hit served from a stale leaf record) and of a 16-pass cadence
O=gpurun_out/r04t; mkdir -p $O
timeout -k 10 500 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_variants.py -m gpu -x -q > $O/parity.txt 2>&1; rc=$?; tail -3 $O/parity.txt
[ $rc -eq 0 ] && timeout -k 10 900 python3 tools/ab.py --reps 2 --out $O/ab_stale.json "product||-" "stale2048||build_ab/libtdtrt_stale2048.so" "stale8192||build_ab/libtdtrt_stale8192.so" "stale32768||build_ab/libtdtrt_stale32768.so" "th16||build_ab/libtdtrt_th16.so" > $O/ab_stale.txt 2>&1; tail -16 $O/ab_stale.txt
