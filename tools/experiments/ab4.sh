#!/bin/bash
# usage: tools/experiments/ab4.sh libA.so libB.so [reps]  (GPU box): two builds, bench configs 2 / 3 / 5, interleaved
A=$1; B=$2; N=${3:-2}
for REP in $(seq $N); do
for C in 2 3 5; do
  BENCH_ARGS="--config $C --no-target --no-reference-default" STEPS=8 tools/experiments/ab2.sh "A c$C|X=1|$A" "B c$C|X=1|$B"
done; done
