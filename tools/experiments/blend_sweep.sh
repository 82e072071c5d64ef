#!/bin/bash
# usage: tools/blend_sweep.sh  (GPU box): the probe-order estimator of two-phase frames — weight of the tile mean (positive) or tile maximum (negative)
for C in 2 3 5; do for B in 0.5 0.75 1.0 -0.25 -0.5 -1.0; do
  TDT_ORDER_BLEND=$B python3 bench.py --config $C --steps 8 --warmup 2 --no-cpu-baseline --no-strong --no-single-process --no-target --no-reference-default 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config', $C, 'blend', '$B', 'hf', j['config']['history_free_ms'], 'phases', j['roofline']['phases_ms'])"
done; done
