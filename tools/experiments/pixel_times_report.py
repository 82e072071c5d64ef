#!/usr/bin/env python3
"""Report on tools/experiments/pixel_times.py's dump: when pixels start, how long they take against their cost, who ends the frame.
usage: python tools/experiments/pixel_times_report.py times.npz"""
import sys
import numpy as np
d = np.load(sys.argv[1])
for mode in ("fresh", "replay"):
    c = d[mode + "_cost"]; t0 = d[mode + "_t0"].astype(np.float64) / 100; t1 = d[mode + "_t1"].astype(np.float64) / 100
    ok = c != 0; cost = (c & 0x7FFFFFFF).astype(np.float64)
    span = t1[ok].max(); dur = t1 - t0
    print(mode, "span %.0f us; pixels %d" % (span, ok.sum()))
    print("  start-time percentiles 50/90/99/100 (share of span):", (np.percentile(t0[ok], [50, 90, 99, 100]) / span).round(3))
    print("  us per cost unit: pct 10/50/90:", np.percentile(dur[ok] / np.maximum(cost[ok], 1), [10, 50, 90]).round(4))
    print("  duration/span pct 50/90/99/99.9/max:", (np.percentile(dur[ok], [50, 90, 99, 99.9, 100]) / span).round(3), " cost pct:", np.percentile(cost[ok], [50, 90, 99, 99.9, 100]).round(0))
    for f in (0.9, 0.75):
        late = ok & (t1 > f * span)
        print("  pixels ending after %.0f%% of the span: %d; their start (share of span) pct 10/50/90: %s; duration/span pct 10/50/90: %s; cost pct 10/50/90 %s"
              % (100 * f, late.sum(), (np.percentile(t0[late], [10, 50, 90]) / span).round(3), (np.percentile(dur[late], [10, 50, 90]) / span).round(3), np.percentile(cost[late], [10, 50, 90]).round(0)))
    r = np.argsort(np.argsort(t0[ok])); print("  spearman(start rank, -cost): %.3f" % np.corrcoef(r, -np.argsort(np.argsort(cost[ok])))[0, 1])
