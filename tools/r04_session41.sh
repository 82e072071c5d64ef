#!/bin/bash
# GPU session 41: the whole -m gpu suite on the final kernels (brick builds with cold material arguments), then the counter passes
O=gpurun_out/r04an; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1; rc=$?; tail -3 $O/gpu_suite.txt
[ $rc -eq 0 ] && bash tools/profile_r04.sh r04final2 pmc > gpurun_out/r04final2.pmc.log 2>&1; tail -3 gpurun_out/r04final2.pmc.log
