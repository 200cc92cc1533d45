"""GPU: `python bench.py --gpus N` as the driver calls it, exercised end to end at N = 1.

The box has one device, so this is the N > 1 path with one rank: CRESTE_BENCH_FORCE_LAUNCH=1 makes the command re-run itself
under `torch.distributed.run` (rendezvous on 127.0.0.1, RCCL process group, barrier + max-over-ranks around the timed steps,
the data-parallel training legs of BASELINE configs[3] / [4] through DistillTrainer / SSCTrainer / IRLTrainer with the real
gradient exchange -- reference creste/train_ssc.py:342-358, creste/train_pefree.py:261-288, creste/train_traversability.py:400-416),
so that the first 8-GPU run is not the first time this code executes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CRESTE_BENCH_FORCE_LAUNCH="1", **extra)
    return env


def test_bench_self_launch_one_rank_rccl_prints_one_line_with_the_training_legs():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-modes"], env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = lines[0]
    assert "error" not in line and line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1
    assert line["value"] > 0 and abs(line["value"] - 16 * 1e3 / line["ms_per_step"]) < 0.01 * line["value"]
    assert line["host_fed"]["equals_resident"] is True          # H2D inside the timed region: the same costmap, bit for bit
    assert "roofline" in line and "roofline_splat" in line
    dp = line["train_dp"]
    for leg in ("distill", "ssc", "irl_reference", "irl_cf512"):
        assert leg in dp, sorted(dp)
        assert dp[leg]["step_ms"] > 0 and dp[leg]["step_ms_no_collective"] > 0 and dp[leg]["collective_calls"] >= 1, (leg, dp[leg])
        assert dp[leg]["allreduce_bytes"] > 0


def test_bench_self_launch_dead_rank_reports_an_error_line():
    """the only rank exits before its first barrier: no hang, rc != 0, ONE line with an `error` field (from the launching parent)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-modes"], env=_env(CRESTE_BENCH_TEST_DIE="0"), cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode != 0
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and lines[0]["value"] is None and lines[0]["error"], r.stdout[-2000:] + r.stderr[-2000:]
