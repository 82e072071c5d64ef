#!/bin/bash
# usage: tools/thr.sh "<bench args>" t1 t2 ...
A="$1"; shift
for T in "$@"; do TDT_EVENT_THRESHOLD=$T python bench.py --steps 3 --warmup 1 --no-cpu-baseline $A 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('thr $T', d['value'], 'Msamples/s', d['ms_per_step'], 'ms', 'frac', d['roofline']['frac'])"; done
