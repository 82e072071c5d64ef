#!/bin/bash
# GPU session 26: brick builds — one Rand(hit.xy) for metal and dielectric lanes after the shared normalize freed registers; the probe launch
# through the UNIT build (potential only: the probe keeps its own kernel row in profiler statistics)
O=gpurun_out/r04z; mkdir -p $O
timeout -k 10 900 python3 tools/ab.py --configs 3,5 --reps 3 --out $O/ab_brick.json "product||-" "brick_shared_rand||build_ab/libtdtrt_brand.so" "probe_unit|TDT_PROBE_UNIT=1|-" "both|TDT_PROBE_UNIT=1|build_ab/libtdtrt_brand.so" > $O/ab_brick.txt 2>&1; tail -10 $O/ab_brick.txt
