#!/bin/bash
# GPU session 23: the whole -m gpu suite and smoke() on the final kernels (shared normalize, threshold cadence 8), then the fuzz campaign
O=gpurun_out/r04x; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1; rc=$?; tail -3 $O/gpu_suite.txt
[ $rc -eq 0 ] && timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
[ $rc -eq 0 ] && timeout -k 10 420 python3 tools/fuzz_parity.py 360 20261006 > $O/fuzz.txt 2>&1; tail -4 $O/fuzz.txt
