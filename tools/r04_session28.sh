#!/bin/bash
# GPU session 28: the stuck-ray cut on the held-out set (product: brick builds only; candidate: every build of depth >= 7), the demo frame under both,
# the event threshold's r and the probe size re-checked on the bench frames now that the step counts changed
O=gpurun_out/r04ab; mkdir -p $O
timeout -k 10 300 python3 tools/holdout_bench.py $O/holdout_product.json > $O/holdout_product.txt 2>&1; tail -3 $O/holdout_product.txt
TDT_LIB=$PWD/build_ab/libtdtrt_cutdeep.so timeout -k 10 300 python3 tools/holdout_bench.py $O/holdout_cutdeep.json > $O/holdout_cutdeep.txt 2>&1; tail -3 $O/holdout_cutdeep.txt
TDT_LIB=$PWD/build_ab/libtdtrt_nocut.so timeout -k 10 300 python3 tools/holdout_bench.py $O/holdout_nocut.json > $O/holdout_nocut.txt 2>&1; tail -3 $O/holdout_nocut.txt
timeout -k 10 100 python3 tools/demo_time.py 100 > $O/demo_product.txt 2>&1; tail -2 $O/demo_product.txt
TDT_LIB=$PWD/build_ab/libtdtrt_cutdeep.so timeout -k 10 100 python3 tools/demo_time.py 100 > $O/demo_cutdeep.txt 2>&1; tail -2 $O/demo_cutdeep.txt
timeout -k 10 600 python3 tools/ab.py --configs 3,5 --reps 2 --out $O/ab_retune.json "product||-" "k_x0.7|TDT_EVENT_K_SCALE=0.7|-" "k_x1.4|TDT_EVENT_K_SCALE=1.4|-" "probe_div8|TDT_PROBE_DIV=8|-" "probe_div32|TDT_PROBE_DIV=32|-" "one_pass|TDT_NO_TWO_PHASE=1|-" > $O/ab_retune.txt 2>&1; tail -14 $O/ab_retune.txt
