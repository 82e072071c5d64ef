#!/bin/bash
# GPU session 14: fuzz campaign on the round's final build; the held-out timing set
O=gpurun_out/r04o; mkdir -p $O
timeout -k 10 800 python3 tools/fuzz_parity.py 700 20261005 > $O/fuzz.txt 2>&1; tail -4 $O/fuzz.txt
timeout -k 10 350 python3 tools/holdout_bench.py $O/holdout.json > $O/holdout.txt 2>&1; tail -3 $O/holdout.txt
