"""bench.py's own multi-process launcher (`--gpus N` without torch.distributed.run) and its
loud failure when the node has fewer GPUs than asked for.  CPU: the workers run bench.py's stub
engine over gloo (`--stub`), which exercises spawn -> rendezvous -> barrier -> max-over-ranks ->
ONE JSON line; the GPU path proper is covered by the driver's bench runs."""
import json
import os
import subprocess
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
BENCH = os.path.join(REPO, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["MASTER_ADDR"] = "127.0.0.1"
    return env


def test_self_launch_two_ranks_stub():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "5", "--warmup", "2", "--replicas", "8",
                        "--stub"], env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout  # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 5 and j["warmup"] == 2
    assert j["data"] == "stub" and j["scaling"] == "weak"
    # whole-job aggregate: both ranks' units over the max-over-ranks time
    assert abs(j["value"] - 2 * 8 * 16 * 5 / (j["ms_per_step"] * 5e-3)) < 1e-6 * j["value"]


def test_more_gpus_than_present_fails_loudly():
    import torch

    n = torch.cuda.device_count() + 1
    if n < 2:
        n = 2
    p = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "1", "--warmup", "0"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "GPU" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_world_size_mismatch_is_an_error():
    env = dict(_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--stub"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr


def test_eight_ranks_generate_their_own_demands_within_the_start_up_budget():
    """`bench.py --gpus 8` must not time out before it times anything.  Eight ranks (stub engines over
    gloo: no GPU here) each generate the 4096 demands of their shard with the array generator and build their tables --
    what happens on the real node before tsc_create --; the line reports the slowest rank's table build."""
    import time

    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1", "--replicas", "4096", "--stub",
                        "--stub-tables"], env=_env(), capture_output=True, text=True, timeout=900)
    wall = time.time() - t0
    assert p.returncode == 0, p.stderr[-2000:]
    j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 8
    st = j["config"]["startup_s"]["table_build_max_over_ranks"]
    assert 1.0 < st < 120.0, st
    assert wall < 240.0, wall
