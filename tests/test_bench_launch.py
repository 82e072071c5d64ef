"""bench.py's launch path: `python bench.py --gpus N` outside torchrun must start its N ranks itself (the shape of the
driver's N=1 command with N>1), give them torchrun's environment, and pass ONE JSON line through.  CPU: the ranks only
rendezvous (gloo); GPU: the real thing with two ranks sharing the one device."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_rank_environments_look_like_torchrun():
    envs = bench.rank_environments(3, 29999, base={"PATH": "/bin"})
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert all(e["WORLD_SIZE"] == "3" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29999" for e in envs)
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and e["PATH"] == "/bin" for e in envs)


def test_arguments_and_workloads():
    a = bench.parse_args(["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert (a.gpus, a.steps, a.warmup, a.config) == (8, 20, 5, 2)
    assert bench.parse_args([]).gpus == 1
    W, H, spp, bounce, desc, scene_cfg = bench.WORKLOADS[4]                     # BASELINE configs[3]
    assert (W, H, spp, bounce) == (7680, 4320, 64, 8) and "256^3" in desc
    assert bench.WORKLOADS[2][:4] == (1920, 1080, 64, 8)                       # the metric's configuration
    r, w = bench.algorithmic_bytes({"node_loads": 10, "lambertian": 2, "metal": 1, "dielectric": 1, "pixels": 4})
    assert r == 80 + 48 + 28 + 16 + 40 and w == 64


def test_self_launch_spawns_ranks_that_meet():
    """The driver's shape of command: no torchrun, --gpus 3 — three child processes rendezvous over gloo on 127.0.0.1."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--rendezvous-only"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert {k: j[k] for k in ("rendezvous", "rank_sum", "ranks_seen")} == {"rendezvous": 3, "rank_sum": 3.0, "ranks_seen": 3}
    # the fields that make the N>1 strong line self-explaining, computed over the same process group: ranks "traced" 10/20/30 ms
    st = j["strong"]
    assert st["backend"] == "gloo" and st["ranks_seen"] == 3
    assert st["imbalance"] == 1.5 and st["speedup_vs_1gpu"] == 2.0 and st["one_gpu_ms"] == 60.0
    # ... and the N > 1 HEADLINE is strong scaling of one frame: value(N) / value(1) is that same speed-up
    h = j["headline_scaling"]
    assert h["scaling"] == "strong" and h["n_gpus"] == 3 and h["speedup_vs_1gpu"] == st["speedup_vs_1gpu"]
    assert abs(h["value"] / h["one_gpu_value"] - h["speedup_vs_1gpu"]) < 2e-3


def test_headline_summary_is_value_ratio():
    h = bench.headline_summary(25000.0, 6400.0, 4)
    assert h["scaling"] == "strong" and h["speedup_vs_1gpu"] == round(25000.0 / 6400.0, 3) and h["one_gpu_value"] == 6400.0
    assert bench.parse_args(["--gpus", "4"]).scaling is None                   # default: strong at N > 1 (decided from the world size), weak label at N = 1


def test_strong_summary_fields():
    st = bench.strong_summary(100.0, [90.0, 100.0, 95.0, 99.0], 380.0, "rank 0 alone", "nccl", 4)
    assert st["speedup_vs_1gpu"] == 3.8 and st["backend"] == "nccl" and st["ranks_seen"] == 4
    assert abs(st["imbalance"] - 100.0 / 96.0) < 1e-3
    one = bench.strong_summary(617.0, [617.0], 617.0, "this measurement (one rank)", "none (one rank: no collective)", 1)
    assert one["speedup_vs_1gpu"] == 1.0 and one["imbalance"] == 1.0


def test_physical_roofline_is_quoted_from_committed_counters():
    p = bench.physical_roofline("config2_spp64_gpus1")
    assert p and p["bound"] == "valu" and 0 < p["frac"] < 1 and abs(p["frac"] - p["issue_util"] * p["lane_util"]) < 1e-3
    assert p["source"].startswith("profiles/")
    assert bench.physical_roofline("no_such_workload") is None


def test_a_failing_rank_fails_the_launch():
    """Without a GPU the ranks exit with an error: the launcher must report that, not hang and not print a line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert "needs a GPU" in p.stderr


@pytest.mark.gpu
def test_bench_gpus_2_self_launched_on_one_gpu():
    """`python bench.py --gpus 2 --backend gloo --steps 1 --warmup 1` as a fresh child: rc 0 and one JSON line that carries the
    strong-scaled headline (the metric's frame sharded over the ranks) with the weak-scaled side block, the strong 8K block with its per-rank times, and the single-process (multi-device context) block."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "1",
                        "--strong-steps", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["value"] > 0 and j["config"]["image"] == [1920, 1080]      # the metric's own frame, sharded
    assert len(j["config"]["trace_ms_per_rank"]) == 2 and j["config"]["gather_ms"] is not None
    h = j["headline_scaling"]
    assert h["scaling"] == "strong" and h["n_gpus"] == 2 and abs(j["value"] / h["one_gpu_value"] - h["speedup_vs_1gpu"]) < 2e-3
    assert j["weak"]["image"] == [1920, 2160] and j["weak"]["scaling"] == "weak" and j["weak"]["value"] > 0
    s = j["strong"]
    assert s["image"] == [7680, 4320] and s["spp"] == 64 and s["written_pixels"] == 7680 * 4320 and len(s["trace_ms_per_rank"]) == 2
    assert s["backend"] == "gloo" and s["ranks_seen"] == 2 and s["imbalance"] >= 1.0 and s["speedup_vs_1gpu"] > 0 and "rank 0 alone" in s["one_gpu_source"]
    sp = j["single_process"]
    assert "error" not in sp, sp
    assert sp["devices"] == [0, 0] and sp["written_pixels"] == 7680 * 4320 and sp["transport"] == "copy" and sp["rccl_ranks"] == 0
