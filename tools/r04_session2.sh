#!/bin/bash
# GPU session 2 of round 4: instruction-class micro-benchmark; A/B of the experiment builds (speed proxies included); the turned-camera tests
O=gpurun_out/r04b; mkdir -p $O
timeout -k 10 120 build_ab/valu_rates > $O/valu_rates.txt 2>&1; echo "micro done"
timeout -k 10 600 python3 -m pytest tests/test_gpu_api.py -q -m gpu -k "turned or rolled" -x > $O/turned_tests.txt 2>&1; tail -3 $O/turned_tests.txt
for REP in 1 2; do
for C in 2 3 5; do
  BENCH_ARGS="--config $C --no-target --no-reference-default" STEPS=8 timeout -k 10 900 tools/experiments/ab2.sh "base c$C|X=1|-" "lazyroot c$C|X=1|build_ab/lib_lazyroot.so" "rand2 c$C|X=1|build_ab/lib_rand2.so" \
     "k2proxy40 c$C|X=1|build_ab/lib_k2proxy.so" "k2proxy48 c$C|TDT_EVENT_CLAMP=48|build_ab/lib_k2proxy.so" "k2proxy56 c$C|TDT_EVENT_CLAMP=56|build_ab/lib_k2proxy.so" "k2proxy63 c$C|TDT_EVENT_CLAMP=63|build_ab/lib_k2proxy.so" >> $O/ab.txt 2>&1
  echo "ab c$C rep $REP done"
done; done
cat $O/ab.txt
