#!/bin/bash
# usage: tools/demo_sweep.sh  (GPU box): the reference's own workload under the scheduler's switches
for T in 40 48 56; do echo "TDT_EVENT_CLAMP=$T"; TDT_EVENT_CLAMP=$T python3 tools/demo_time.py 100 2>&1 | grep demo;  TDT_EVENT_CLAMP=$T python3 tools/demo_time.py 30 1920 1080 16 8 2>&1 | grep demo; 
  for C in 2 3 5; do TDT_EVENT_CLAMP=$T python3 bench.py --config $C --steps 5 --warmup 2 --no-cpu-baseline --no-strong --no-single-process --no-target --no-reference-default 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config', $C, 'hf', j['config']['history_free_ms'], 'replay', j['config']['replay_ms'])"; done
done
