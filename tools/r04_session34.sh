#!/bin/bash
# GPU session 34: brick builds after the stuck-ray cut (events are half of their instructions now): one Rand(hit.xy) for metal and dielectric
# lanes; two / four node-memo levels
O=gpurun_out/r04ag; mkdir -p $O
timeout -k 10 900 python3 tools/ab.py --configs 3,5 --reps 3 --out $O/ab_brick2.json "product||-" "brick_shared_rand||build_ab/libtdtrt_brand.so" "memo2||build_ab/libtdtrt_memo2.so" "memo4||build_ab/libtdtrt_memo4.so" > $O/ab_brick2.txt 2>&1; tail -10 $O/ab_brick2.txt
