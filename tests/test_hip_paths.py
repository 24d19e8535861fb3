"""GPU: every data-dependent path of the resident kernel `kr_step`, against the oracle, in the product library.

The kernel has three paths that ordinary traffic reaches rarely or never:
  * two-phase hand-offs   -- taken when two hand-offs of a sub-step touch the same list end (else leavers relink themselves);
  * the measuring walk    -- taken after a vehicle left its list from behind another one (else lanes are FIFO and the
                             position of a record follows from its entry number);
  * the serial cross scan -- taken by a vehicle whose (vehicle, cross) tasks do not fit the on-chip task list;
plus the stall of a replica that runs out of records while the host is running ahead.  tsc_config_t carries runtime knobs
that force each of them (resident_force, resident_task_capacity, resident_margin) and tsc_path_counts tells which ran.
Every test compares with the ORACLE (which has none of these paths), not with the default path of the kernel.
Oracle spec: oracle/tsc_oracle.c (engine_step, update_location); reference call sites: pytsc/backends/cityflow/simulator.py:80-89.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tables(scen, R, **hip):
    from pytsc_amd.config import Config
    from pytsc_amd.scenario import build_tables

    cfg = Config(scen, signal={"action_space": "phase_selection", "round_robin": False}, hip=hip)
    return build_tables(cfg, n_replicas=R)


def _run_against_oracle(oracle_lib, scen, R, steps, knobs, kw, seed, snap_every=10, hold_actions=True):
    """Congested run with replica-specific action streams; observations, stats, f64 rewards every step and every
    vehicle's drivable / position / speed / id every `snap_every` steps must equal the oracle's."""
    import torch

    from pytsc_amd.vec_env import VecEngine

    eng = VecEngine(_tables(scen, R, **kw, **knobs), family="resident")
    orc = oracle_lib.Oracle(_tables(scen, R, **kw))
    A = eng.A
    rng = np.random.default_rng(seed)
    out = orc.observe()
    hold = np.zeros((R, A), dtype=np.int32)
    for s in range(steps):
        m = out.mask
        pick = (rng.random(m.shape) * m).argmax(-1).astype(np.int32)
        if hold_actions:  # replica r changes its phases every 2 + 3 r steps: from quick cycling to very long reds
            for r in range(R):
                if s % (2 + 3 * r) == 0:
                    hold[r] = pick[r]
            acts = np.where(np.take_along_axis(m, hold[..., None], -1)[..., 0] > 0, hold, pick).astype(np.int32)
        else:
            acts = pick
        out = orc.step(acts)
        buf = eng.step(torch.tensor(acts, device="cuda"))
        for k in ("obs", "state", "mask", "stats", "local_reward64", "agent_stats", "phase"):
            assert np.array_equal(buf[k].cpu().numpy(), getattr(out, k)), (s, k)
        if s % snap_every == snap_every - 1:
            for r in range(R):
                a, b = orc.snapshot(r), eng.snapshot(r)
                assert all(np.array_equal(x, y) for x, y in zip(a, b)), (s, r)
    counts = eng.path_counts()
    assert eng.errors() == 0 and orc.errors() == 0
    mean_veh = eng.mean_vehicles()
    eng.close()
    return counts, mean_veh


CONGESTED = {
    "syn_3x3": dict(flow_rate=1500.0, n_flow_sets=2, sim_length=900),
    "syn_4x4": dict(flow_rate=1100.0, n_flow_sets=2, sim_length=900),
}


@pytest.mark.parametrize("scen", ["syn_3x3", "syn_4x4"])
def test_forced_two_phase_handoffs_match_oracle(scen, oracle_lib):
    """resident_force bit 0: every sub-step unlinks per source list and appends per target list in order."""
    R, steps = 5, 120
    c, v = _run_against_oracle(oracle_lib, scen, R, steps, dict(resident_force=1), CONGESTED[scen], seed=31)
    assert c["two_phase"] == R * steps * 5, c  # every (replica, sub-step)
    assert c["walk"] == c["dirty"] and c["serial_scan"] == 0, c
    assert v > 150


@pytest.mark.parametrize("scen", ["syn_3x3", "syn_4x4"])
def test_forced_list_walk_matches_oracle(scen, oracle_lib):
    """resident_force bit 1: the measure phase numbers the records by walking the lists."""
    R, steps = 5, 120
    c, v = _run_against_oracle(oracle_lib, scen, R, steps, dict(resident_force=2), CONGESTED[scen], seed=32)
    assert c["walk"] == R * steps, c
    assert c["two_phase"] == c["conflict"] and c["serial_scan"] == 0, c
    assert v > 150


@pytest.mark.parametrize("scen", ["syn_3x3", "syn_4x4"])
def test_tiny_task_list_serial_cross_scan_matches_oracle(scen, oracle_lib):
    """resident_task_capacity 32: most sub-steps of a busy grid need more (vehicle, cross) tasks than that, so the
    vehicles beyond scan their crosses one after the other inside stage A2; a capacity of 1 sends everybody there."""
    R, steps = 4, 100
    c, v = _run_against_oracle(oracle_lib, scen, R, steps, dict(resident_task_capacity=32), CONGESTED[scen], seed=33)
    assert c["serial_scan"] > 100, c
    c1, _ = _run_against_oracle(oracle_lib, scen, 2, 60, dict(resident_task_capacity=1), CONGESTED[scen], seed=34)
    assert c1["serial_scan"] > c["serial_scan"] / 4, (c, c1)


def test_all_fallbacks_together_match_oracle(oracle_lib):
    R, steps = 4, 100
    c, _ = _run_against_oracle(oracle_lib, "syn_3x3", R, steps, dict(resident_force=3, resident_task_capacity=16),
                               CONGESTED["syn_3x3"], seed=35)
    assert c["two_phase"] == R * steps * 5 and c["walk"] == R * steps and c["serial_scan"] > 100, c


def test_natural_triggers_of_the_two_phase_path(oracle_lib):
    """No knob: on a busy grid two vehicles do reach one list end in the same sub-step now and then (two lane-links of
    an intersection end on the same lane; a lane-link takes a vehicle while its head leaves) -- the kernel notices it
    when the decisions are committed and that sub-step takes the two-phase path by itself."""
    R, steps = 6, 150
    c, v = _run_against_oracle(oracle_lib, "syn_3x3", R, steps, {}, CONGESTED["syn_3x3"], seed=11)
    assert c["conflict"] > 0 and c["two_phase"] == c["conflict"], c
    assert c["walk"] == c["dirty"], c
    assert v > 150


def test_overlap_leaver_marks_the_replica_dirty(oracle_lib):
    """The natural trigger of the measuring walk: a vehicle that leaves its lane-link from BEHIND another one.  Lane-links
    are a few metres long and a vehicle may enter one while its predecessor still stands on it -- with three vehicle
    templates of very different speeds (a slow truck ahead, a car behind it turning into a faster movement is not
    possible on one link, but a leader that stops beyond the end of a short link while its follower's position wraps
    onto the next drivable is) the walk is reached without any knob; whatever happens, results equal the oracle and the
    counters tell what ran."""
    import torch

    from pytsc_amd.config import Config
    from pytsc_amd.flows import DEFAULT_VEHICLE, poisson_flows
    from pytsc_amd.grid import generate_grid
    from pytsc_amd.roadnet import RoadNet
    from pytsc_amd.scenario import build_tables
    from pytsc_amd.vec_env import VecEngine

    net = generate_grid(2, 3)
    rn = RoadNet(net)
    van = dict(DEFAULT_VEHICLE, length=7.0, maxPosAcc=1.5, usualPosAcc=1.5, maxSpeed=9.5, minGap=3.0)
    truck = dict(DEFAULT_VEHICLE, length=12.0, maxPosAcc=1.0, usualPosAcc=1.0, maxNegAcc=3.5, usualNegAcc=3.0,
                 maxSpeed=7.5, minGap=3.5, headwayTime=2.0)
    sets = []
    for seed in (0, 1):
        entries = poisson_flows(rn, 1400.0, horizon=700, seed=seed)
        rng = np.random.default_rng(100 + seed)
        for e in entries:
            k = rng.integers(0, 10)
            e["vehicle"] = truck if k < 2 else (van if k < 5 else dict(DEFAULT_VEHICLE))
        sets.append(entries)

    def tables(R, **knobs):
        cfg = Config("syn_2x3", signal={"action_space": "phase_selection", "round_robin": False},
                     hip=dict({"roadnet_json": net, "flow_entries": sets, "sim_length": 700}, **knobs))
        return build_tables(cfg, n_replicas=R)

    R = 6
    eng = VecEngine(tables(R), family="resident")
    orc = oracle_lib.Oracle(tables(R))
    rng = np.random.default_rng(6)
    out = orc.observe()
    for s in range(130):
        m = out.mask
        acts = (rng.random(m.shape) * m).argmax(-1).astype(np.int32)
        out = orc.step(acts)
        buf = eng.step(torch.tensor(acts, device="cuda"))
        assert np.array_equal(buf["obs"].cpu().numpy(), out.obs), s
        assert np.array_equal(buf["stats"].cpu().numpy(), out.stats), s
        assert np.array_equal(buf["agent_stats"].cpu().numpy(), out.agent_stats), s
        if s % 10 == 9:
            for r in range(R):
                assert all(np.array_equal(x, y) for x, y in zip(orc.snapshot(r), eng.snapshot(r))), (s, r)
    c = eng.path_counts()
    print("path counts, mixed templates:", c)
    assert c["walk"] == c["dirty"] and c["two_phase"] == c["conflict"], c
    assert eng.errors() == 0 and orc.errors() == 0
    eng.close()


@pytest.mark.parametrize("how", ["state", "replay"])
def test_async_caller_across_the_migration_threshold(oracle_lib, how, monkeypatch):
    """A caller that never waits: 190 env steps enqueued back to back (actions precomputed, outputs copied aside on the
    stream) on an AUTO handle whose replicas outgrow the resident family on the way.  The handle holds the caller back
    to 16 steps ahead of the overflow flag, migrates (round 5: by writing the blobs' state across; `replay`: by replaying
    its log, PYTSC_AMD_MIGRATE=replay), and every step's outputs equal the oracle's;
    no device error, no stalled replica (a margin of 256 records outlasts 16 steps of this demand)."""
    import torch

    from pytsc_amd.vec_env import VecEngine

    kw = dict(flow_rate=1500.0, n_flow_sets=2, sim_length=1100, episode_limit=400)
    R, steps = 6, 190
    monkeypatch.setenv("PYTSC_AMD_MIGRATE", how)
    orc = oracle_lib.Oracle(_tables("syn_3x3", R, **kw))
    rng = np.random.default_rng(21)
    out = orc.observe()
    acts_all, exp = [], []
    for s in range(steps):
        m = out.mask
        acts = (rng.random(m.shape) * m).argmax(-1).astype(np.int32)
        out = orc.step(acts)
        acts_all.append(acts)
        exp.append((out.obs.copy(), out.stats.copy(), out.mask.copy()))
    eng = VecEngine(_tables("syn_3x3", R, resident_capacity=1472, resident_spill_limit=-1, **kw), family="auto")  # (no spill pool: whole-handle migration)
    assert eng.family == "resident"
    acts_dev = torch.tensor(np.stack(acts_all), device="cuda")
    torch.cuda.synchronize()
    got = []
    for s in range(steps):  # no host synchronisation in this loop
        buf = eng.step(acts_dev[s])
        got.append((buf["obs"].clone(), buf["stats"].clone(), buf["mask"].clone()))
    torch.cuda.synchronize()
    assert eng.family == "list"  # it did migrate
    for s in range(steps):
        for g, e, name in zip(got[s], exp[s], ("obs", "stats", "mask")):
            assert np.array_equal(g.cpu().numpy(), e), (s, name)
    for r in range(R):
        assert all(np.array_equal(x, y) for x, y in zip(orc.snapshot(r), eng.snapshot(r))), r
    c = eng.path_counts()
    assert c["stalled"] == 0, c
    assert eng.errors() == 0 and orc.errors() == 0
    eng.close()


def test_replica_that_runs_out_of_records_stalls_and_the_migration_repairs_it(oracle_lib, capfd):
    """resident_margin 1: the flag is raised only when a replica is all but full, so with a caller 16 steps ahead some
    replica does run out of records.  It stalls (keeps the state and outputs of its last completed step, no error), the
    others carry on; once the host has seen the flag the replay on the list family repairs everything: from the
    migration on every output and every vehicle equals the oracle again."""
    import torch

    from pytsc_amd.vec_env import VecEngine

    kw = dict(flow_rate=1800.0, n_flow_sets=2, sim_length=1500, episode_limit=500)
    R, steps = 6, 270  # (the grid has to hold more than 1472 vehicles + the stage scratch of its heavy ones)
    orc = oracle_lib.Oracle(_tables("syn_3x3", R, **kw))
    rng = np.random.default_rng(22)
    out = orc.observe()
    acts_all, exp = [], []
    for s in range(steps):
        m = out.mask
        acts = (rng.random(m.shape) * m).argmax(-1).astype(np.int32)
        out = orc.step(acts)
        acts_all.append(acts)
        exp.append(out.obs.copy())
    eng = VecEngine(_tables("syn_3x3", R, resident_capacity=1472, resident_margin=1, resident_spill_limit=-1, **kw), family="auto")
    acts_dev = torch.tensor(np.stack(acts_all), device="cuda")
    torch.cuda.synchronize()
    got, fam, valid = [], [], []
    for s in range(steps):
        buf = eng.step(acts_dev[s])
        got.append(buf["obs"].clone())
        valid.append(buf["valid"].clone())
        fam.append(eng.family)  # (host-side field: no synchronisation)
    torch.cuda.synchronize()
    assert fam[-1] == "list"
    first_list = fam.index("list")
    c = eng.path_counts()
    assert c["stalled"] >= 1, c
    assert "state written across" not in capfd.readouterr().err  # (a stalled replica's blob is steps old: the log is replayed)
    # The caller is told which rows are stale (the reference never returns a stale observation,
    # pytsc/__init__.py:178-182): a replica's rows equal the oracle's wherever valid == 1 -- every replica, every step --,
    # valid == 0 exactly for the stalled replicas, only inside the run-ahead window before the migration, and a row that
    # differs from the oracle's is always marked
    n_stale_rows = 0
    for s in range(steps):
        g, v = got[s].cpu().numpy(), valid[s].cpu().numpy()
        for r in range(R):
            same = np.array_equal(g[r], exp[s][r])
            assert same or v[r] == 0, (s, r)
            if v[r] == 0:
                n_stale_rows += 1
                assert s < first_list and first_list - s <= 17, (s, r, first_list)
    assert n_stale_rows >= 1
    assert all(int(v.min()) == 1 for v in valid[first_list:])  # (the list family never stalls)
    for r in range(R):
        assert all(np.array_equal(x, y) for x, y in zip(orc.snapshot(r), eng.snapshot(r))), r
    assert eng.errors() == 0 and orc.errors() == 0
    eng.close()


def test_forced_resident_handle_that_runs_out_of_records_reports_an_error(oracle_lib):
    """a handle FORCED to the resident family has nothing to migrate to.  When one of its replicas runs
    out of records -- by the spawn that finds none, or by `live + heavy items > capacity` in the stall check behind the
    classification -- it stalls for good: tsc_errors counts it, valid says which replica, the others stay exact."""
    import torch

    from pytsc_amd.vec_env import VecEngine

    kw = dict(flow_rate=1800.0, n_flow_sets=2, sim_length=1500, episode_limit=500)
    R, steps = 4, 270
    orc = oracle_lib.Oracle(_tables("syn_3x3", R, **kw))
    rng = np.random.default_rng(22)
    out = orc.observe()
    eng = VecEngine(_tables("syn_3x3", R, resident_capacity=1472, **kw), family="resident")
    stalled = np.zeros(R, bool)
    for s in range(steps):
        m = out.mask
        acts = (rng.random(m.shape) * m).argmax(-1).astype(np.int32)
        out = orc.step(acts)
        buf = eng.step(torch.tensor(acts, device="cuda"))
        v = buf["valid"].cpu().numpy().astype(bool)
        assert not (stalled & v).any()  # (a stalled replica of such a handle never comes back)
        stalled |= ~v
        g = buf["obs"].cpu().numpy()
        for r in range(R):
            if v[r]:
                assert np.array_equal(g[r], out.obs[r]), (s, r)
    assert stalled.any(), stalled
    assert eng.path_counts()["stalled"] == int(stalled.sum())
    assert eng.errors() >= int(stalled.sum())
    eng.close()


def test_forced_paths_at_full_size_equal_the_default_paths():
    """BASELINE config 1 size (4x4 @ 4096, 8 flow sets, per-replica action streams): a handle with every knob set
    (two-phase hand-offs and the walk forced, a task list of 16) produces, for every one of the 4096 replicas, the
    outputs of a default handle -- whose replicas tests/test_hip_fullsize_oracle.py links to the oracle at this size."""
    import torch

    import bench
    from pytsc_amd.vec_env import VecEngine

    R = 4096
    want = ("obs", "mask", "stats", "local_reward64", "phase")
    a = VecEngine(bench.workload_tables(4, 4, R, 0, 700.0, 8), want=want, family="resident")
    tb = bench.workload_tables(4, 4, R, 0, 700.0, 8)
    tb.cfg.resident_force = 3
    tb.cfg.resident_task_capacity = 16
    b = VecEngine(tb, want=want, family="resident")
    acts = torch.empty((R, 16), device="cuda", dtype=torch.int32)
    ba, bb = a.observe(), b.observe()
    for s in range(70):
        a.random_valid_actions(ba["mask"], seed=99, out=acts)
        ba = a.step(acts)
        bb = b.step(acts)
        if s % 10 == 9:
            for k in want:
                assert torch.equal(ba[k], bb[k]), (s, k)
    for r in (0, 1, 63, 2048, R - 1):
        assert all(np.array_equal(x, y) for x, y in zip(a.snapshot(r), b.snapshot(r))), r
    ca, cb = a.path_counts(), b.path_counts()
    assert cb["two_phase"] == R * 70 * 5 and cb["walk"] == R * 71 and cb["serial_scan"] > 1000, cb  # (71: + the observe())
    assert ca["walk"] == ca["dirty"] and ca["serial_scan"] == 0 and ca["two_phase"] == ca["conflict"] > 0, ca
    assert a.errors() == 0 and b.errors() == 0
    a.close()
    b.close()


@pytest.mark.parametrize("scenario", ["grid", "jinan"])
def test_the_three_instantiations_of_kr_step_agree_at_full_size(scenario, monkeypatch):
    """BASELINE config 1 (4x4 @ 4096) and config 4 (Jinan, 2048 of its 8192 replicas): the handle the bench runs launches
    kr_step with the network's layout AND its vehicle folded in at compile time (resident_alloc picks that instantiation when
    the template row is the known vehicle's, bit for bit: the trip generators' default vehicle / the shipped real-world flow
    files' one); with PYTSC_AMD_NO_DEFVEH the layout-specialised instantiation reads the same constants from the device
    table, with PYTSC_AMD_NO_SPEC the generic one reads the layout too.  Same outputs for every replica, same vehicles."""
    import torch

    import bench
    from pytsc_amd.vec_env import VecEngine

    R = 4096 if scenario == "grid" else 2048
    want = ("obs", "mask", "stats", "local_reward64", "phase")
    engines = []
    for env in ({}, {"PYTSC_AMD_NO_DEFVEH": "1"}, {"PYTSC_AMD_NO_SPEC": "1"}):
        for k in ("PYTSC_AMD_NO_DEFVEH", "PYTSC_AMD_NO_SPEC"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        engines.append(VecEngine(bench.workload_tables(4, 4, R, 0, 700.0, 8, scenario=scenario), want=want, family="resident"))
    for k in ("PYTSC_AMD_NO_DEFVEH", "PYTSC_AMD_NO_SPEC"):
        monkeypatch.delenv(k, raising=False)
    a = engines[0]
    assert [e.kernel_variant for e in engines] == ["layout + default vehicle", "layout", "generic"]
    A = a.tables.n_agents
    acts = torch.empty((R, A), device="cuda", dtype=torch.int32)
    bufs = [e.observe() for e in engines]
    for s in range(80):
        a.random_valid_actions(bufs[0]["mask"], seed=41, out=acts)
        bufs = [e.step(acts) for e in engines]
        if s % 10 == 9:
            for k in want:
                assert torch.equal(bufs[0][k], bufs[1][k]) and torch.equal(bufs[0][k], bufs[2][k]), (s, k)
    for r in (0, 1, 63, R // 2, R - 1):
        sa = a.snapshot(r)
        for e in engines[1:]:
            assert all(np.array_equal(x, y) for x, y in zip(sa, e.snapshot(r))), r
    assert a.mean_vehicles() > 300
    for e in engines:
        assert e.errors() == 0
        e.close()
