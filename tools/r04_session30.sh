#!/bin/bash
# GPU session 30: kernel stats, bench lines, pass statistics of the final kernels; then a fuzz campaign
bash tools/profile_r04.sh r04final bench > gpurun_out/r04final.bench.log 2>&1; tail -3 gpurun_out/r04final.bench.log
timeout -k 10 400 python3 tools/fuzz_parity.py 330 20261007 > gpurun_out/r04final/fuzz.txt 2>&1; tail -2 gpurun_out/r04final/fuzz.txt
