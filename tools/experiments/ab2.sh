#!/bin/bash
# usage: tools/experiments/ab2.sh "<label>|<ENV=..> <ENV=..>|<lib.so or ->" ...   -> bench each variant: history-free ms, replay ms, phases
# BENCH_ARGS selects the config (default: the bench workload)
for V in "$@"; do
  IFS='|' read -r LABEL ENVS LIB <<< "$V"
  if [ "$LIB" != "-" ] && [ -n "$LIB" ]; then export TDT_LIB=$PWD/$LIB; else unset TDT_LIB; fi
  env $ENVS python bench.py --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --no-strong --no-single-process ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('%-28s history-free %8.3f ms   replay %8.3f ms   phases %s' % ('$LABEL', c['history_free_ms'], c['replay_ms'], r.get('phases_ms')))"
done
