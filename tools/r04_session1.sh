#!/bin/bash
# GPU session 1 of round 4: pass statistics of the product kernels (stats build), baseline bench line, wave timeline.
O=gpurun_out/r04a; mkdir -p $O
for C in 2 3 5 0; do for M in fresh replay; do
  TDT_LIB=build_ab/lib_stats.so TDT_STATS_SKIP_PROBE=1 timeout -k 10 300 python3 tools/loss_budget.py collect --config $C --mode $M > $O/stats_c${C}_${M}.json 2> $O/stats_c${C}_${M}.err || exit 1
  echo "stats c$C $M done"
done; done
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err && echo "bench done"
timeout -k 10 200 python3 tools/timeline.py 2 > $O/timeline_c2.txt 2>&1; echo "timeline done"
timeout -k 10 200 python3 tools/phase_probe.py 2 > $O/phase_probe_c2.txt 2>&1; echo "phase probe done"
ls -la $O
