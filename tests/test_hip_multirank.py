"""GPU: the N > 1 path of bench.py -- BASELINE config 3's sharding: replicas split over ranks, no collective on the step
path -- rehearsed on ONE card: two ranks, 2048 replicas each, both on GPU 0, gloo for the barrier and the
reductions.  Rank r's replicas must be replicas r * 2048 ... of the unsharded 4096-replica job bit for bit (the flow sets
and the policy's draws follow the GLOBAL replica index), and the one JSON line must aggregate as the contract says:
value = the units of all ranks / the maximum elapsed time.  No scaling number is taken here: that is the driver's."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
BENCH = os.path.join(REPO, "bench.py")


def _run(args, tmp):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["MASTER_ADDR"] = "127.0.0.1"
    env["HIP_VISIBLE_DEVICES"] = "0"
    p = subprocess.run([sys.executable, BENCH, "--no-cpu-baseline", "--no-extras", "--steps", "10", "--warmup", "3"] + args,
                       env=env, capture_output=True, text=True, timeout=900, cwd=tmp)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_equal_the_unsharded_job(tmp_path):
    tmp = str(tmp_path)
    two = _run(["--gpus", "2", "--replicas", "2048", "--one-gpu-rehearsal", "--dump-state", os.path.join(tmp, "two")], tmp)
    one = _run(["--gpus", "1", "--replicas", "4096", "--dump-state", os.path.join(tmp, "one")], tmp)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1 and two["scaling"] == "weak"
    assert two["config"]["replicas_per_gpu"] == 2048 and two["config"]["device_errors"] == 0
    # the contract's aggregate: units of all ranks / max-over-ranks elapsed (ms_per_step is that time per step)
    A, K = two["config"]["agents"], two["steps"]
    assert abs(two["value"] - 2 * 2048 * A * K / (two["ms_per_step"] * K * 1e-3)) < 1e-6 * two["value"]
    whole = np.load(os.path.join(tmp, "one.rank0.npz"))
    for rank in (0, 1):
        part = np.load(os.path.join(tmp, f"two.rank{rank}.npz"))
        lo = rank * 2048
        assert int(part["replica_base"]) == lo
        for key in ("stats", "obs", "mask", "counts"):
            assert np.array_equal(part[key], whole[key][lo:lo + 2048]), (rank, key)
    # the two shards are different work (demands drawn from the global replica index), not two copies of one
    a, b = np.load(os.path.join(tmp, "two.rank0.npz")), np.load(os.path.join(tmp, "two.rank1.npz"))
    assert not np.array_equal(a["counts"], b["counts"])
