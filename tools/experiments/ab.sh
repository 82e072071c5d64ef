#!/bin/bash
# usage: tools/ab.sh <lib1.so> <lib2.so> ...   -> bench each variant (steps 5), print value / ms
for L in "$@"; do
  TDT_LIB=$PWD/$L python bench.py --steps 5 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['value'], 'Msamples/s', d['ms_per_step'], 'ms', 'frac', d['roofline']['frac'])"
done
