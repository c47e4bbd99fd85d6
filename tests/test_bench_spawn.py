"""`python bench.py --gpus N` must really start N ranks: the launcher / rendezvous / max-over-ranks skeleton
of bench.py is run here on CPU with gloo and a stub step (ADVCHAIN_BENCH_STUB=1); the GPU step itself is covered by
tests/test_dist_gpu.py and the -m gpu bench test."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, extra_env=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env["ADVCHAIN_BENCH_STUB"] = "1"
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True,
                          text=True, timeout=300)


def test_gpus_2_self_spawns_two_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout          # rank 0 prints ONE JSON line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2
    assert rec["config"]["global_batch"] == 64 and rec["steps"] == 3 and rec["scaling"] == "weak"
    # load imbalance and collective count in the same line
    spread = rec["rank_ms_per_step"]
    assert len(spread["by_rank"]) == 2 and spread["min"] <= spread["max"] and abs(spread["max"] - rec["ms_per_step"]) < 0.5
    assert "all_reduces_per_step" in rec


def test_world_size_mismatch_fails_loudly():
    """A launcher that started 1 rank for --gpus 2 must not print n_gpus: 1."""
    r = _run(["--gpus", "2"], extra_env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_too_few_gpus_fails_loudly():
    """Without the stub the spawn path checks the visible devices first: no GPU here -> rc != 0, no JSON."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "ADVCHAIN_BENCH_STUB")}
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        return
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())
