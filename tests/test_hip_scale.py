"""GPU: the other BASELINE configs at reduced replica counts against the oracle
(16x16 grid = occupancy stress, multi flow sets with sequential restarts), and
size-independent properties at the full replica count."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tables(scen, R, **hip):
    from pytsc_amd.config import Config
    from pytsc_amd.scenario import build_tables

    cfg = Config(scen, signal={"action_space": "phase_selection", "round_robin": False}, hip=hip)
    return build_tables(cfg, n_replicas=R)


def test_16x16_grid_matches_oracle(oracle_lib):
    """BASELINE config 2 shape (16x16, 256 agents) at R = 2, 40 env steps."""
    import torch

    from pytsc_amd.vec_env import VecEngine

    kw = dict(flow_rate=700.0, n_flow_sets=1, sim_length=600)
    eng = VecEngine(_tables("syn_16x16", 2, **kw))
    orc = oracle_lib.Oracle(_tables("syn_16x16", 1, **kw))
    rng = np.random.default_rng(3)
    out = orc.observe()
    for s in range(40):
        m = out.mask[0]
        acts = (rng.random(m.shape) * m).argmax(-1).astype(np.int32)
        out = orc.step(acts[None])
        buf = eng.step(torch.tensor(np.repeat(acts[None], 2, 0), dtype=torch.int32, device="cuda"))
        for r in (0, 1):
            assert np.array_equal(buf["obs"][r].cpu().numpy(), out.obs[0]), s
            assert np.array_equal(buf["stats"][r].cpu().numpy(), out.stats[0]), s
            assert np.array_equal(buf["local_reward64"][r].cpu().numpy(), out.local_reward64[0]), s
    a, b = orc.snapshot(0), eng.snapshot(1)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert eng.errors() == 0
    eng.close()


@pytest.mark.parametrize("family", ["list", "resident"])
def test_flow_sets_and_sequential_restart_match_oracle(family, oracle_lib):
    """3 flow sets, sequential advance on restart (backends/cityflow/config.py:63-76),
    short simulated 'hour' so that restarts happen inside the test."""
    import torch

    from pytsc_amd.vec_env import VecEngine

    kw = dict(flow_rate=800.0, n_flow_sets=3, sim_length=100, episode_limit=50,
              flow_rate_type="sequential")
    R = 5
    eng = VecEngine(_tables("syn_2x2", R, **kw), family=family)
    orc = oracle_lib.Oracle(_tables("syn_2x2", R, **kw))
    rng = np.random.default_rng(4)
    out = orc.observe()
    for s in range(70):
        m = out.mask
        acts = (rng.random(m.shape) * m).argmax(-1).astype(np.int32)
        out = orc.step(acts)
        buf = eng.step(torch.tensor(acts, device="cuda"))
        assert np.array_equal(buf["obs"].cpu().numpy(), out.obs), s
        assert np.array_equal(buf["stats"].cpu().numpy(), out.stats), s
        assert np.array_equal(buf["terminated"].cpu().numpy(), out.terminated), s
        if out.terminated.any():
            assert (s + 1) % 20 == 0
            mask = out.terminated.copy()
            if s < 20:
                assert mask.all()
                mask[1] = 0  # leave one replica running past its simulated hour
            orc.reset(mask)
            eng.reset(torch.tensor(mask, device="cuda"))
            out = orc.observe()
            buf = eng.observe()
            assert np.array_equal(buf["obs"].cpu().numpy(), out.obs)
    assert eng.errors() == 0 and orc.errors() == 0
    eng.close()


@pytest.mark.parametrize("family", ["list", "resident"])
def test_full_size_properties_4x4_r4096(family):
    """BASELINE config 1 size: replicas sharing a flow set stay bit-identical under
    identical actions (K6), vehicles are conserved (K5), no device error."""
    import torch

    from pytsc_amd.vec_env import VecEngine

    R = 4096
    eng = VecEngine(_tables("syn_4x4", R, flow_rate=700.0, n_flow_sets=4),
                    want=("obs", "mask", "stats", "global_reward"), family=family)
    g = torch.Generator(device="cuda")
    g.manual_seed(0)
    buf = eng.observe()
    for s in range(60):
        u = torch.rand((4, 16, 16), device="cuda", generator=g).clamp_min(1e-6)
        # the same action for every replica of a flow set
        acts = (u.repeat(R // 4, 1, 1) * buf["mask"]).argmax(-1).to(torch.int32)
        buf = eng.step(acts)
    obs = buf["obs"]
    for k in range(4):
        assert torch.equal(obs[k::4], obs[k].expand_as(obs[k::4]))
        assert torch.equal(buf["stats"][k::4], buf["stats"][k].expand_as(buf["stats"][k::4]))
    assert not torch.equal(obs[0], obs[1])
    mean_veh = eng.mean_vehicles()
    assert 100 < mean_veh < 3000
    c = eng.replica_counts().cpu().numpy()  # K5: entered == running + finished, every replica
    assert np.array_equal(c[:, 0], c[:, 1] + c[:, 2]) and c[:, 0].min() > 0
    assert eng.errors() == 0
    eng.close()


@pytest.mark.parametrize("family", ["list", "resident"])
def test_congested_queues_match_oracle(family, oracle_lib):
    """Heavy demand with long red phases: long standing queues, spill-back onto
    lane-links, discharge waves -- the states in which k_enum settles vehicles without
    arithmetic (stopped queue members, heads at a red light) and skips standing blocks.
    Replicas get different action streams; vehicle lists are compared slot by slot."""
    import torch

    from pytsc_amd.vec_env import VecEngine

    kw = dict(flow_rate=1500.0, n_flow_sets=2, sim_length=900)
    R = 6
    eng = VecEngine(_tables("syn_3x3", R, **kw), family=family)
    orc = oracle_lib.Oracle(_tables("syn_3x3", R, **kw))
    rng = np.random.default_rng(11)
    out = orc.observe()
    hold = np.zeros((R, 9), dtype=np.int32)
    for s in range(150):
        m = out.mask
        # replica r changes its phases every 2 + 3 r steps: from quick cycling to very long reds
        pick = (rng.random(m.shape) * m).argmax(-1).astype(np.int32)
        for r in range(R):
            if s % (2 + 3 * r) == 0:
                hold[r] = pick[r]
        acts = np.where(np.take_along_axis(m, hold[..., None], -1)[..., 0] > 0, hold, pick).astype(np.int32)
        out = orc.step(acts)
        buf = eng.step(torch.tensor(acts, device="cuda"))
        assert np.array_equal(buf["obs"].cpu().numpy(), out.obs), s
        assert np.array_equal(buf["stats"].cpu().numpy(), out.stats), s
        assert np.array_equal(buf["local_reward64"].cpu().numpy(), out.local_reward64), s
        if s % 10 == 9:
            for r in range(R):
                a, b = orc.snapshot(r), eng.snapshot(r)
                assert all(np.array_equal(x, y) for x, y in zip(a, b)), (s, r)
    assert eng.mean_vehicles() > 150  # the network did fill up
    assert eng.errors() == 0 and orc.errors() == 0
    eng.close()


@pytest.mark.parametrize("family", ["list", "resident"])
def test_mixed_vehicle_templates_match_oracle(family, oracle_lib):
    """Three vehicle templates (car, van, slow truck) in one demand: the multi-template
    instantiations of k_enum / k_exec / k_commit (per-vehicle parameter look-ups, leader
    lengths in gaps, approach distances) under congestion, against the oracle."""
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
        entries = poisson_flows(rn, 1300.0, horizon=700, seed=seed)
        rng = np.random.default_rng(100 + seed)
        for e in entries:
            k = rng.integers(0, 10)
            e["vehicle"] = truck if k == 0 else (van if k < 4 else dict(DEFAULT_VEHICLE))
        sets.append(entries)

    def tables(R):
        cfg = Config("syn_2x3", signal={"action_space": "phase_selection", "round_robin": False},
                     hip={"roadnet_json": net, "flow_entries": sets, "sim_length": 700})
        return build_tables(cfg, n_replicas=R)

    R = 4
    eng = VecEngine(tables(R), family=family)
    orc = oracle_lib.Oracle(tables(R))
    rng = np.random.default_rng(5)
    out = orc.observe()
    hold = np.zeros((R, 6), dtype=np.int32)
    for s in range(120):
        m = out.mask
        pick = (rng.random(m.shape) * m).argmax(-1).astype(np.int32)
        for r in range(R):
            if s % (2 + 4 * r) == 0:
                hold[r] = pick[r]
        acts = np.where(np.take_along_axis(m, hold[..., None], -1)[..., 0] > 0, hold, pick).astype(np.int32)
        out = orc.step(acts)
        buf = eng.step(torch.tensor(acts, device="cuda"))
        assert np.array_equal(buf["obs"].cpu().numpy(), out.obs), s
        assert np.array_equal(buf["stats"].cpu().numpy(), out.stats), s
        if s % 10 == 9:
            for r in range(R):
                a, b = orc.snapshot(r), eng.snapshot(r)
                assert all(np.array_equal(x, y) for x, y in zip(a, b)), (s, r)
    assert eng.mean_vehicles() > 100
    assert eng.errors() == 0 and orc.errors() == 0
    eng.close()


def _identical_replicas_stay_identical(eng, R, n_sets, A, n_act, steps):
    """K6 at full size: replicas that share a flow set and receive the same actions hold
    bit-identical observations and stats; different flow sets differ."""
    import torch

    g = torch.Generator(device="cuda")
    g.manual_seed(0)
    buf = eng.observe()
    for s in range(steps):
        u = torch.rand((n_sets, A, n_act), device="cuda", generator=g).clamp_min(1e-6)
        acts = (u.repeat(R // n_sets, 1, 1) * buf["mask"]).argmax(-1).to(torch.int32)
        buf = eng.step(acts)
    for k in range(n_sets):
        assert torch.equal(buf["obs"][k::n_sets], buf["obs"][k].expand_as(buf["obs"][k::n_sets]))
        assert torch.equal(buf["stats"][k::n_sets], buf["stats"][k].expand_as(buf["stats"][k::n_sets]))
    assert not torch.equal(buf["obs"][0], buf["obs"][1])
    assert eng.errors() == 0
    return buf


def test_full_size_properties_16x16_r4096():
    """BASELINE config 2 size (256 agents x 4096 replicas on one GPU)."""
    from pytsc_amd.vec_env import VecEngine

    R = 4096
    eng = VecEngine(_tables("syn_16x16", R, flow_rate=700.0, n_flow_sets=2, sim_length=600),
                    want=("obs", "mask", "stats", "global_reward"))
    _identical_replicas_stay_identical(eng, R, 2, 256, eng.n_actions, 25)
    assert eng.mean_vehicles() > 500
    eng.close()


@pytest.mark.parametrize("family", ["list", "resident"])
def test_full_size_properties_jinan_r8192(family):
    """BASELINE config 4 size: the Jinan 3x4 network with its real flow file, 8192 replicas
    (one flow set: every replica must stay identical to replica 0 under identical actions,
    and differ from it under different ones)."""
    import torch

    from fixtures import load_case
    from pytsc_amd.vec_env import VecEngine

    R = 8192
    _, _, _, _, tables = load_case("jinan_3_4_random", n_replicas=R)
    eng = VecEngine(tables, want=("obs", "mask", "stats"), family=family)
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    buf = eng.observe()
    A, n_act = buf["mask"].shape[1], buf["mask"].shape[2]
    for s in range(40):
        u = torch.rand((2, A, n_act), device="cuda", generator=g).clamp_min(1e-6)
        # even replicas follow one action stream, odd replicas another
        acts = (u.repeat(R // 2, 1, 1) * buf["mask"]).argmax(-1).to(torch.int32)
        buf = eng.step(acts)
    for k in range(2):
        assert torch.equal(buf["obs"][k::2], buf["obs"][k].expand_as(buf["obs"][k::2]))
        assert torch.equal(buf["stats"][k::2], buf["stats"][k].expand_as(buf["stats"][k::2]))
    assert not torch.equal(buf["stats"][0], buf["stats"][1])
    assert eng.mean_vehicles() > 100 and eng.errors() == 0
    eng.close()


@pytest.mark.parametrize("how", ["state", "replay"])
def test_resident_handle_migrates_to_the_list_family_before_it_overflows(oracle_lib, how, monkeypatch, capfd):
    """kernel_family AUTO: when a replica of the resident family comes within 256 vehicle records of its capacity the
    handle moves itself to the list family -- `state` (round 5, the default): no replica has stalled, so the state of every
    blob is written across (kr_to_list); `replay` (PYTSC_AMD_MIGRATE=replay; what a handle with a stalled replica still
    does): by replaying its logged resets / actions (deterministic engine) -- and carries on there.  Heavy demand on a 3x3 grid with a capacity of 1472 records, partial restarts in between:
    every step before, across and after the migration equals the oracle, and no device error is ever raised."""
    import torch

    from pytsc_amd.vec_env import VecEngine

    kw = dict(flow_rate=1500.0, n_flow_sets=2, sim_length=900, episode_limit=40)
    R = 6
    monkeypatch.setenv("PYTSC_AMD_MIGRATE", how)
    eng = VecEngine(_tables("syn_3x3", R, resident_capacity=1472, resident_spill_limit=-1, **kw), family="auto")  # (no spill pool: whole-handle migration)
    assert eng.family == "resident"
    orc = oracle_lib.Oracle(_tables("syn_3x3", R, **kw))
    rng = np.random.default_rng(21)
    out = orc.observe()
    buf = eng.observe()
    migrated_at = None
    for s in range(170):
        m = out.mask
        acts = (rng.random(m.shape) * m).argmax(-1).astype(np.int32)
        out = orc.step(acts)
        buf = eng.step(torch.tensor(acts, device="cuda"))
        assert np.array_equal(buf["obs"].cpu().numpy(), out.obs), s
        assert np.array_equal(buf["stats"].cpu().numpy(), out.stats), s
        assert np.array_equal(buf["local_reward64"].cpu().numpy(), out.local_reward64), s
        if migrated_at is None and eng.lib.tsc_kernel_family(eng.h) == 1:
            migrated_at = s
        if s == 25:  # a partial restart of two replicas while still resident: replayed by the migration
            mask = np.zeros(R, np.uint8)
            mask[[1, 4]] = 1
            orc.reset(mask)
            eng.reset(torch.tensor(mask, device="cuda"))
            out = orc.observe()
            buf = eng.observe()
            assert np.array_equal(buf["obs"].cpu().numpy(), out.obs)
    assert migrated_at is not None and 25 < migrated_at < 160, migrated_at
    assert ("state written across" in capfd.readouterr().err) == (how == "state")
    for r in range(R):
        a, b = orc.snapshot(r), eng.snapshot(r)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), r
    assert max(len(orc.snapshot(r)[0]) for r in range(R)) > 1100  # the run did need the room
    assert eng.errors() == 0 and orc.errors() == 0
    eng.close()


@pytest.mark.parametrize("how", ["state", "replay"])
def test_migrated_handle_returns_to_the_resident_family_at_a_full_restart(oracle_lib, how, monkeypatch):
    """kernel_family AUTO, `sequential` flow sets: run resident -> (capacity) -> list family -> full restart -> resident
    again -> (capacity) -> list family a second time.  The second replay starts at that full restart, not at
    tsc_create: all it needs of the past are the restart counters (the flow set of a replica is derived from them),
    kept from before the restart.  A restart mask that selects nobody is a no-op, one that selects everybody is a full
    restart.  Every step equals the oracle, snapshots included (they grow the handle's scratch: once a bug here)."""
    import torch

    from pytsc_amd.vec_env import VecEngine

    kw = dict(flow_rate=1500.0, n_flow_sets=3, sim_length=900, episode_limit=40, flow_rate_type="sequential")
    R = 4
    monkeypatch.setenv("PYTSC_AMD_MIGRATE", how)
    eng = VecEngine(_tables("syn_3x3", R, resident_capacity=1472, resident_spill_limit=-1, **kw), family="auto")  # (no spill pool: whole-handle migration)
    orc = oracle_lib.Oracle(_tables("syn_3x3", R, **kw))
    fam = lambda: eng.lib.tsc_kernel_family(eng.h)  # 1: list, 2: resident
    assert fam() == 2
    rng = np.random.default_rng(5)

    def run(n_steps, partial_at=None):
        out, buf = orc.observe(), eng.observe()
        assert np.array_equal(buf["obs"].cpu().numpy(), out.obs)
        trace = []
        for s in range(n_steps):
            m = out.mask
            acts = (rng.random(m.shape) * m).argmax(-1).astype(np.int32)
            out = orc.step(acts)
            buf = eng.step(torch.tensor(acts, device="cuda"))
            assert np.array_equal(buf["obs"].cpu().numpy(), out.obs), s
            assert np.array_equal(buf["stats"].cpu().numpy(), out.stats), s
            trace.append(fam())
            if s == partial_at:
                mask = np.zeros(R, np.uint8)
                mask[2] = 1
                orc.reset(mask)
                eng.reset(torch.tensor(mask, device="cuda"))
                out, buf = orc.observe(), eng.observe()
                assert np.array_equal(buf["obs"].cpu().numpy(), out.obs)
        return trace

    t1 = run(150, partial_at=20)
    assert t1[0] == 2 and t1[-1] == 1, t1  # migrated on the way
    for r in range(R):
        assert all(np.array_equal(x, y) for x, y in zip(orc.snapshot(r), eng.snapshot(r))), r
    eng.reset(torch.zeros(R, dtype=torch.uint8, device="cuda"))  # nobody: no-op
    assert fam() == 1
    orc.reset(np.ones(R, np.uint8))
    eng.reset(torch.ones(R, dtype=torch.uint8, device="cuda"))  # everybody: a full restart
    assert fam() == 2
    t2 = run(150, partial_at=30)
    assert t2[0] == 2 and t2[-1] == 1, t2  # and a second migration, replayed from the full restart
    orc.reset(None)
    eng.reset(None)
    assert fam() == 2
    t3 = run(10)
    assert set(t3) == {2}
    for r in range(R):
        assert all(np.array_equal(x, y) for x, y in zip(orc.snapshot(r), eng.snapshot(r))), r
    assert eng.errors() == 0 and orc.errors() == 0
    eng.close()
