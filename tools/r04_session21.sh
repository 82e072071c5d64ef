#!/bin/bash
# GPU session 21: why does the two-class order lose on the 512^3 frame?  pass statistics (queue-dry time, wave ends, lanes per pass) of the main
# launch under both orders
O=gpurun_out/r04v; mkdir -p $O
for v in one two; do
  [ $v = one ] && export TDT_NO_HIT_CLASS=1 || unset TDT_NO_HIT_CLASS
  TDT_LIB=$PWD/build_ab/lib_stats.so TDT_STATS_SKIP_PROBE=1 timeout -k 10 300 python3 tools/loss_budget.py collect --config 5 --mode fresh > $O/stats_c5_$v.json 2> $O/stats_c5_$v.err || exit 1
done
python3 - <<'PY'
import json
for v in ("one","two"):
    d=json.load(open("gpurun_out/r04v/stats_c5_%s.json"%v))["stats"]
    print(v, "span_ms", d["span_ms"], "dry", d["queue_dry_share_of_span"], "wave ends", d["wave_end_share_of_span_p1_p5_p25_p50_p75_p95"])
    print("   ", {k:d[k] for k in d if k in ("trav_pass","trav_lanes","event_pass","event_lanes","loop_pass","gate_wait_lanes","drained_trav_pass","drained_event_pass","alive_lanes","inside_lanes")})
PY
