#!/usr/bin/env python3
"""Hand-out order study on the CPU: what would the main launch of a two-phase frame cost under different sort keys?

Input: tools/experiments/dump_costs.py's per-pixel step / event counts of the probe's sample range and of the whole frame.  Model: L lanes
draw pixels from one queue in key order (most expensive first); a pixel occupies its lane for its cost (7 per step + 64 per event, the
kernel's own units) — no wave coupling, no event gate: only the ORDER's share of the tail.  Reported: makespan / (sum / L) for each key, the
share of the makespan after the queue ran dry, and the heaviest pixel against the ideal makespan.

    python tools/sim/order_sim.py costs.npz [lanes]
"""
import heapq, sys
import numpy as np


def makespan(cost, key, lanes):
    order = np.argsort(-key, kind="stable")
    c = cost[order]
    n = len(c)
    if n <= lanes:
        return float(c.max()), 0.0
    heap = list(c[:lanes].astype(np.float64))
    heapq.heapify(heap)
    for x in c[lanes:]:
        t = heapq.heappop(heap)
        heapq.heappush(heap, t + float(x))
    dry = heap[0]
    return max(heap), dry


def main():
    d = np.load(sys.argv[1])
    lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 256 * 1024
    cost = lambda S, E: 7.0 * S.astype(np.float64) + 64.0 * E.astype(np.float64)
    probe, frame = cost(d["probe_S"], d["probe_E"]), cost(d["frame_S"], d["frame_E"])
    live = d["frame_E"] > 0
    probe, frame = probe[live | True], frame[live | True]
    main_c = np.maximum(frame - probe, 0.0)
    n = len(main_c)
    tile = probe.reshape(-1, 64)
    tmean = np.repeat(tile.mean(1), 64)
    tmax = np.repeat(tile.max(1), 64)
    ideal = main_c.sum() / lanes
    print("pixels %d, lanes %d: %.1f pixels per lane; ideal makespan %.0f units; heaviest pixel %.0f = %.2f of it; p99.9 %.2f, p99 %.2f, median %.3f"
          % (n, lanes, n / lanes, ideal, main_c.max(), main_c.max() / ideal, np.percentile(main_c, 99.9) / ideal, np.percentile(main_c, 99) / ideal, np.median(main_c) / ideal))
    rng = np.random.default_rng(1)
    keys = {
        "exact (replay)": main_c,
        "probe alone": probe,
        "probe 0.5 + tile mean 0.5 (product)": 0.5 * probe + 0.5 * tmean,
        "probe 0.25 + tile mean 0.75": 0.25 * probe + 0.75 * tmean,
        "tile mean alone": tmean,
        "probe 0.5 + tile max 0.5": 0.5 * probe + 0.5 * tmax,
        "max(probe, tile mean)": np.maximum(probe, tmean),
        "image order": -np.arange(n, dtype=np.float64),
        "random": rng.random(n),
    }
    # a wider neighbourhood: the mean over the 3x3 tiles around (needs the slot layout: tiles are 8x8 inside 32x32 groups, group-major)
    for name, k in keys.items():
        m, dry = makespan(main_c, k, lanes)
        print("  %-40s makespan %.3f x ideal, queue dry at %.3f of it" % (name, m / ideal, dry / m))


if __name__ == "__main__":
    main()
