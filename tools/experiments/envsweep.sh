#!/bin/bash
# usage: tools/envsweep.sh VAR v1 v2 ...   -> bench (BENCH_ARGS) with VAR=v
V=$1; shift
for W in "$@"; do
  env $V=$W python bench.py --steps 5 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V=$W', d['value'], 'Msamples/s', d['ms_per_step'], 'ms', 'frac', d['roofline']['frac'])"
done
