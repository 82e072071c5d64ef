#!/bin/bash
# GPU session 42: kernel stats, bench lines, pass statistics of the final kernels; a fuzz campaign
bash tools/profile_r04.sh r04final2 bench > gpurun_out/r04final2.bench.log 2>&1; tail -3 gpurun_out/r04final2.bench.log
timeout -k 10 400 python3 tools/fuzz_parity.py 330 20261009 > gpurun_out/r04final2/fuzz.txt 2>&1; tail -2 gpurun_out/r04final2/fuzz.txt
