#!/bin/bash
# usage: tools/cw.sh <w1> <w2> ...  -> bench config 2 with TDT_COST_EVENT_W=w
for W in "$@"; do
  TDT_COST_EVENT_W=$W python bench.py --steps 5 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('w=$W', d['value'], 'Msamples/s', d['ms_per_step'], 'ms', 'frac', d['roofline']['frac'])"
done
