#!/bin/bash
# GPU session 36: a prior pass (first bounce of the first sample, costs only) in front of low-spp frames without history — the reference's own frame
O=gpurun_out/r04ai; mkdir -p $O
for v in off on; do
  [ $v = on ] && export TDT_PRIOR_PASS=1 || unset TDT_PRIOR_PASS
  for rep in 1 2; do TDT_LIB=$PWD/build_ab/libtdtrt_prior.so timeout -k 10 100 python3 tools/demo_time.py 200 2>&1 | grep demo | sed "s/^/prior $v: /"; done
  TDT_LIB=$PWD/build_ab/libtdtrt_prior.so timeout -k 10 100 python3 tools/demo_time.py 100 1920 1080 4 6 2>&1 | grep demo | sed "s/^/prior $v: /"
  TDT_LIB=$PWD/build_ab/libtdtrt_prior.so timeout -k 10 100 python3 tools/demo_time.py 100 1280 720 8 6 2>&1 | grep demo | sed "s/^/prior $v: /"
done > $O/demo_prior.txt 2>&1; cat $O/demo_prior.txt
TDT_PRIOR_PASS=1 TDT_LIB=$PWD/build_ab/libtdtrt_prior.so timeout -k 10 400 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py -m gpu -x -q > $O/parity_prior.txt 2>&1; tail -2 $O/parity_prior.txt
