"""CPU: `python bench.py --gpus N` with WORLD_SIZE unset starts its N ranks itself (BASELINE's metric is quoted at
1/2/4/8 GPUs; the plain command must not silently run one rank).  Driven in --plan-only mode: the ranks rendezvous over
gloo on 127.0.0.1, count themselves with an all-reduce of ones and rank 0 prints the line's skeleton."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=180):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def _line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout                      # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_plain_command_spawns_its_ranks():
    r = _run(["--gpus", "2", "--plan-only"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _line(r.stdout)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["shards"] == [64, 64]


def test_three_ranks_and_equals_form():
    r = _run(["--gpus=3", "--plan-only"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert _line(r.stdout)["ranks_seen"] == 3


def test_mismatch_with_a_launcher_environment_fails_loudly():
    """under torchrun (WORLD_SIZE set) a --gpus that disagrees is an error, never a line with the wrong n_gpus."""
    r = _run(["--gpus", "2", "--plan-only"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "{" not in r.stdout


def test_a_failing_rank_fails_the_launcher():
    r = _run(["--gpus", "2", "--plan-only", "--no-such-flag"])
    assert r.returncode != 0 and "{" not in r.stdout


def test_single_rank_needs_no_launcher():
    r = _run(["--gpus", "1", "--plan-only"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert _line(r.stdout)["n_gpus"] == 1


import pytest


@pytest.mark.gpu
def test_two_rank_bench_rehearsal_on_one_gpu():
    """the whole N > 1 path of bench.py as the plain command runs it - self-launch, torchrun environment, packed-weight
    broadcast, per-rank decode, cfg5 leg on every rank, barrier + max over ranks, ONE line from rank 0 - with two ranks sharing
    the box's GPU (gloo instead of RCCL: the line says so; it is not a scaling measurement)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "1", "--frames", "20", "--no-cpu-baseline"], timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _line(r.stdout)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["scaling"] == "weak"
    assert line["value"] > 0 and "cfg5" in line["legs"] and line["legs"]["cfg5"]["value"] > 0
    if torch.cuda.device_count() < 2:
        assert "rehearsal" in line
