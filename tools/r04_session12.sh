#!/bin/bash
# GPU session 12: depth - levels in the table / brick entries: parity, A/B against the build before it; event-threshold clamp / r after the slimming
O=gpurun_out/r04m; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_variants.py tests/test_gpu_build.py tests/test_gpu_edit_parallel.py -q -m gpu -x > $O/parity.txt 2>&1; tail -3 $O/parity.txt
timeout -k 10 1100 python3 tools/ab.py --configs 2,3,5 --reps 2 --out $O/ab.json "prev||build_ab/lib_prev.so" "sh_all||-" "clamp44|TDT_EVENT_CLAMP=44|-" "clamp48|TDT_EVENT_CLAMP=48|-" "clamp36|TDT_EVENT_CLAMP=36|-" > $O/ab.txt 2>&1; tail -16 $O/ab.txt
