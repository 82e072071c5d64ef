#!/usr/bin/env python3
"""Interleaved A/B of builds and environment switches on the bench frames (GPU box).

    python tools/ab.py [--configs 2,3,5] [--reps 2] [--steps 8] [--out ab.json] "label|ENV=v ENV2=w|path/to/lib.so" ...

Each variant is `label|environment|library` (library `-` = the product build; environment may be empty).  Every (config, rep) runs
all variants back to back, so box-to-box and time drift hit them alike; the table gives the median history-free / replay frame time
per variant and config and its ratio to the FIRST variant.  A variant is one `python bench.py` child (TDT_LIB selects the library at
import time), so a speed proxy that writes garbage pixels can be timed too (the children run with --no-cpu-baseline)."""
import argparse, json, os, statistics, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_variant(env_s, lib, config, steps):
    env = dict(os.environ)
    env.pop("TDT_LIB", None)
    if lib and lib != "-":
        env["TDT_LIB"] = os.path.join(ROOT, lib) if not os.path.isabs(lib) else lib
    for kv in env_s.split():
        k, v = kv.split("=", 1)
        env[k] = v
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "2", "--config", str(config), "--no-cpu-baseline", "--no-strong",
           "--no-single-process", "--no-target", "--no-reference-default"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        return None, p.stderr[-400:]
    d = json.loads(lines[-1])
    return (d["config"]["history_free_ms"], d["config"]["replay_ms"], (d.get("roofline") or {}).get("phases_ms")), None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--configs", default="2,3,5")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--out")
    a = ap.parse_args()
    variants = [v.split("|") + [""] * (3 - len(v.split("|"))) for v in a.variants]
    configs = [int(c) for c in a.configs.split(",")]
    res = {(lab, c): [] for lab, _, _ in variants for c in configs}
    for rep in range(a.reps):
        for c in configs:
            for lab, env_s, lib in variants:
                r, err = run_variant(env_s, lib, c, a.steps)
                if r is None:
                    print(f"[{lab} c{c}] FAILED: {err}", flush=True)
                    continue
                res[(lab, c)].append(r)
                print(f"rep {rep} c{c} {lab:<24} history-free {r[0]:9.3f} ms   replay {r[1]:9.3f} ms   phases {r[2]}", flush=True)
    print()
    table = {}
    for c in configs:
        base = res[(variants[0][0], c)]
        b_hf = statistics.median(x[0] for x in base) if base else None
        b_rp = statistics.median(x[1] for x in base) if base else None
        for lab, _, _ in variants:
            v = res[(lab, c)]
            if not v:
                continue
            hf, rp = statistics.median(x[0] for x in v), statistics.median(x[1] for x in v)
            table[f"{lab}|config{c}"] = {"history_free_ms": hf, "replay_ms": rp, "vs_first_history_free": round(hf / b_hf, 4) if b_hf else None,
                                         "vs_first_replay": round(rp / b_rp, 4) if b_rp else None, "runs": len(v)}
            print(f"config {c}  {lab:<24} history-free {hf:9.3f} ms ({hf / b_hf - 1:+.1%})   replay {rp:9.3f} ms ({rp / b_rp - 1:+.1%})" if b_hf else f"config {c} {lab} {hf} {rp}")
    if a.out:
        json.dump(table, open(a.out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
