"""GPU: full-size runs LINKED TO THE ORACLE.  BASELINE configs at their real replica counts
(4x4 @ 4096 = config 1's metric size, 4x4 @ 8192 = the per-GPU shard of config 3, 16x16 @ 256
replicas), 8 flow sets, a different action stream in every replica, 80+ env steps into congestion.
Replicas {0, 1, 63, 64, 65, R/2, R-1} -- first / last lanes of a wavefront, both sides of a
64-replica block edge, another XCD's blocks -- are re-run on the CPU oracle with exactly their flow
set and the actions the device drew for them, and compared bit for bit: observations, info stats,
f64 rewards, masks, and every vehicle's drivable / position / speed.  Vehicle conservation (K5:
entered == running + finished) is asserted for ALL replicas."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tables(scen, R, set_offset=0, **hip):
    from pytsc_amd.config import Config
    from pytsc_amd.scenario import build_tables

    cfg = Config(scen, signal={"action_space": "phase_selection", "round_robin": False}, hip=hip)
    return build_tables(cfg, n_replicas=R, set_offset=set_offset)


def _run_linked(oracle_lib, scen, R, steps, snap_at, n_sets=8, rate=700.0, min_veh=0, family="auto", **hip):
    import torch

    from pytsc_amd.vec_env import VecEngine

    kw = dict(flow_rate=rate, n_flow_sets=n_sets, **hip)
    sel = sorted({0, 1, 63, 64, 65, R // 2, R - 1} & set(range(R)))
    eng = VecEngine(_tables(scen, R, **kw), family=family)
    orc = oracle_lib.Oracle(_tables(scen, len(sel), **kw))
    orc.reset(sets=np.asarray([r % n_sets for r in sel], np.int32))  # replica -> flow set, as the device maps it
    sel_t = torch.tensor(sel, device="cuda")
    buf, out = eng.observe(), orc.observe()
    assert np.array_equal(buf["mask"][sel_t].cpu().numpy(), out.mask)
    names = ("obs", "mask", "stats", "local_reward64", "global_reward64", "agent_stats", "phase", "state",
             "done", "terminated")
    for s in range(steps):
        acts = eng.random_valid_actions(buf["mask"], seed=77)  # a different stream in every replica
        buf = eng.step(acts)
        out = orc.step(acts[sel_t].cpu().numpy())
        for k in names:
            got = buf[k][sel_t].cpu().numpy()
            assert np.array_equal(got, getattr(out, k)), (k, s, np.argwhere(got != getattr(out, k))[:4])
        if (s + 1) in snap_at:
            for i, r in enumerate(sel):
                a, b = orc.snapshot(i), eng.snapshot(r)
                assert all(np.array_equal(x, y) for x, y in zip(a, b)), (s, r)
    c = eng.replica_counts().cpu().numpy()
    assert np.array_equal(c[:, 0], c[:, 1] + c[:, 2])  # K5, every replica
    assert c[:, 0].min() > 0 and c[:, 2].max() > 0
    # the streams really differ: replicas sharing a flow set have diverged
    assert not torch.equal(buf["obs"][0], buf["obs"][n_sets])
    assert eng.mean_vehicles() > min_veh
    assert eng.errors() == 0 and orc.errors() == 0
    eng.close()
    orc.close()


@pytest.mark.parametrize("family", ["list", "resident"])
def test_4x4_r4096_linked_to_oracle(family, oracle_lib):
    """BASELINE config 1 at the metric's size (4096 replicas), 100 env steps."""
    _run_linked(oracle_lib, "syn_4x4", 4096, 100, snap_at=(50, 100), min_veh=700, family=family)


@pytest.mark.parametrize("family,base", [("resident", 0), ("resident", 4096), ("list", 0), ("list", 4096)])
def test_4x4_r4096_batch_demands_linked_to_oracle(family, base, oracle_lib):
    """The driver-run workload's OWN inputs (VERDICT r4 "what's weak" 2): the tables built exactly as bench.py builds
    them -- `bench.workload_tables`: 4096 demands from the array generator `poisson_flow_sets`, one per replica, demand
    g = base + replica (base 4096 = rank 1's shard of a multi-GPU job) -- on both kernel families; replicas {0, 1, 63, 64,
    65, R/2, R-1} against oracle runs of exactly those demands and the actions the device drew, 100 env steps: every
    output of every step, every vehicle at steps 50 and 100, K5 for all replicas."""
    import os
    import sys

    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from pytsc_amd.vec_env import VecEngine

    R = 4096
    sel = [0, 1, 63, 64, 65, R // 2, R - 1]
    tb = bench.workload_tables(4, 4, R, base, 700.0, R)
    assert tb.distinct_demands == R and tb.flows.n_sets == R
    tb.cfg.replica_base = base  # (what bench.worker sets: the policy's draws follow the global replica index)
    eng = VecEngine(tb, family=family)
    # the oracle's tables: the same demands base + 0 ... base + R - 1 generated again (set s <-> demand base + s),
    # seven replicas, each told its set
    to = bench.workload_tables(4, 4, len(sel), base, 700.0, R)
    for s_ in sel:  # the two builds hold the same demands
        assert np.array_equal(tb.fc.sets[s_].veh_start, to.fc.sets[s_].veh_start)
    orc = oracle_lib.Oracle(to)
    orc.reset(sets=np.asarray(sel, np.int32))
    sel_t = torch.tensor(sel, device="cuda")
    buf, out = eng.observe(), orc.observe()
    assert np.array_equal(buf["mask"][sel_t].cpu().numpy(), out.mask)
    names = ("obs", "mask", "stats", "local_reward64", "global_reward64", "agent_stats", "phase", "state", "done", "terminated")
    for s in range(100):
        acts = eng.random_valid_actions(buf["mask"], seed=1234)
        buf = eng.step(acts)
        out = orc.step(acts[sel_t].cpu().numpy())
        for k in names:
            got = buf[k][sel_t].cpu().numpy()
            assert np.array_equal(got, getattr(out, k)), (k, s, np.argwhere(got != getattr(out, k))[:4])
        if (s + 1) in (50, 100):
            for i, r in enumerate(sel):
                a, b = orc.snapshot(i), eng.snapshot(r)
                assert all(np.array_equal(x, y) for x, y in zip(a, b)), (s, r)
    c = eng.replica_counts().cpu().numpy()
    assert np.array_equal(c[:, 0], c[:, 1] + c[:, 2])  # K5, every replica
    assert c[:, 0].min() > 0 and c[:, 2].max() > 0
    # the demands really are distinct: vehicles entered differ between neighbouring replicas
    assert len(np.unique(c[:64, 0])) > 8
    assert eng.mean_vehicles() > 700
    assert eng.errors() == 0 and orc.errors() == 0
    assert eng.family == family
    eng.close()
    orc.close()


@pytest.mark.parametrize("family", ["list", "resident"])
def test_4x4_r8192_linked_to_oracle(family, oracle_lib):
    """The per-GPU shard of BASELINE config 3 (65536 replicas over 8 GPUs = 8192 each)."""
    _run_linked(oracle_lib, "syn_4x4", 8192, 80, snap_at=(80,), min_veh=600, family=family)


def test_4x4_whole_simulated_hour_linked_to_oracle(oracle_lib):
    """The reference's episode (pytsc/__init__.py:52-62: 720 env steps = one simulated hour) from the empty network to
    `terminated`, 512 replicas on the resident family (the instantiation the bench runs): seven replicas against the
    oracle at EVERY step -- through the filling of the network to ~1 900 vehicles per replica, where the record capacity,
    the stage slots and the 16-bit times of that family are used most."""
    _run_linked(oracle_lib, "syn_4x4", 512, 720, snap_at=(360, 720), min_veh=1500, family="resident")


def test_16x16_r256_linked_to_oracle(oracle_lib):
    """BASELINE config 2's network (256 agents) with 256 replicas: four 64-replica blocks."""
    _run_linked(oracle_lib, "syn_16x16", 256, 80, snap_at=(80,), min_veh=2000, family="list", sim_length=900)


@pytest.mark.parametrize("family", ["list", "resident"])
def test_visibility_12_matches_oracle(family, oracle_lib):
    """A visibility other than the default 10 (window predicate bits beyond 32: the 64-bit word
    of k_measure / k_agent), 3x3 grid, 6 replicas with different action streams."""
    import torch

    from pytsc_amd.config import Config
    from pytsc_amd.scenario import build_tables
    from pytsc_amd.vec_env import VecEngine

    def tables(R):
        cfg = Config("syn_3x3", signal={"action_space": "phase_selection", "round_robin": False, "visibility": 12},
                     hip=dict(flow_rate=1200.0, n_flow_sets=2, sim_length=600))
        return build_tables(cfg, n_replicas=R)

    R = 6
    eng, orc = VecEngine(tables(R), family=family), oracle_lib.Oracle(tables(R))
    assert eng.obs_size == 16 * (9 + 12) + 20
    buf, out = eng.observe(), orc.observe()
    for s in range(90):
        acts = eng.random_valid_actions(buf["mask"], seed=5)
        buf = eng.step(acts)
        out = orc.step(acts.cpu().numpy())
        assert np.array_equal(buf["obs"].cpu().numpy(), out.obs), s
        assert np.array_equal(buf["stats"].cpu().numpy(), out.stats), s
    assert np.array_equal(eng.lane_window_counts().cpu().numpy(), orc.lane_window_counts())
    assert (out.obs[..., :-20] == 1).sum() > 50  # moving vehicles did show up in the windows
    assert eng.errors() == 0
    eng.close()
