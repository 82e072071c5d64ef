#!/bin/bash
# usage: tools/kk.sh "<bench args>" r1 r2 ...  (TDT_EVENT_K = r of the adaptive event threshold model)
A="$1"; shift
for T in "$@"; do TDT_EVENT_K=$T python bench.py --steps 3 --warmup 1 --no-cpu-baseline $A 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('r $T', d['value'], 'Msamples/s', d['ms_per_step'], 'ms', 'frac', d['roofline']['frac'])"; done
