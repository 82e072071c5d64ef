#!/bin/bash
# GPU session 16: the whole -m gpu suite and smoke() on the round's final source
O=gpurun_out/r04q; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1; rc=$?; tail -5 $O/gpu_suite.txt
[ $rc -eq 0 ] && timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
