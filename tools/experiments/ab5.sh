#!/bin/bash
# usage: tools/ab5.sh name1 name2 ...  (GPU box): product build vs build_ab/lib_<name>.so on bench configs 2 / 3 / 5 (8 steps each)
for C in 2 3 5; do
  ARGS=("product c$C|X=1|-")
  for n in "$@"; do ARGS+=("$n c$C|X=1|build_ab/lib_$n.so"); done
  ARGS+=("product c$C|X=1|-")
  BENCH_ARGS="--config $C --no-target --no-reference-default" STEPS=8 tools/experiments/ab2.sh "${ARGS[@]}"
done
