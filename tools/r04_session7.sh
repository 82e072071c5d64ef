#!/bin/bash
# GPU session 7: which of the two carried records costs what (speed proxies: not exact in the stale-record cases)
O=gpurun_out/r04g; mkdir -p $O
timeout -k 10 1000 python3 tools/ab.py --configs 2,3,5 --reps 2 --out $O/ab_lazy.json "product||-" "lazy_root_only||build_ab/lib_lazy2.so" "lazy_both||build_ab/lib_lazy1.so" > $O/ab_lazy.txt 2>&1; tail -10 $O/ab_lazy.txt
