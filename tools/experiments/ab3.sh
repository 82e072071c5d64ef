#!/bin/bash
# usage: tools/experiments/ab3.sh  (GPU box): round-2 library vs the current one, bench configs 2 / 3 / 5, interleaved twice
for REP in 1 2; do
for C in 2 3 5; do
  BENCH_ARGS="--config $C --no-target --no-reference-default" STEPS=8 tools/experiments/ab2.sh "r02 c$C|X=1|build_ab/libtdtrt_r02.so" "now c$C|X=1|-"
done; done
