"""GPU: per-replica promotion to the spill pool (include/pytsc_amd.h, TSC_FAMILY_AUTO): a replica of an
`auto` handle that outgrows the LDS-sized record capacity moves -- alone -- into a pool slot with room for every drivable
bumper to bumper and is stepped from there by the same kernel with its working image in global memory; the other replicas
stay where they are; a replica that had stalled before the host noticed is caught up from the ring of the last steps'
actions.  Every output of every step against the ORACLE (the reference never returns a stale observation,
pytsc/__init__.py:178-182: rows a caller may not use are marked valid == 0)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tables(scen, R, **hip):
    from pytsc_amd.config import Config
    from pytsc_amd.scenario import build_tables

    cfg = Config(scen, signal={"action_space": "phase_selection", "round_robin": False}, hip=hip)
    return build_tables(cfg, n_replicas=R)


KW = dict(flow_rate=1500.0, n_flow_sets=2, sim_length=1100, episode_limit=400)


def _oracle_trace(oracle_lib, R, steps, seed, kw=KW):
    orc = oracle_lib.Oracle(_tables("syn_3x3", R, **kw))
    rng = np.random.default_rng(seed)
    out = orc.observe()
    acts_all, exp = [], []
    for s in range(steps):
        m = out.mask
        acts = (rng.random(m.shape) * m).argmax(-1).astype(np.int32)
        out = orc.step(acts)
        acts_all.append(acts)
        exp.append((out.obs.copy(), out.stats.copy(), out.mask.copy(), out.local_reward64.copy()))
    return orc, np.stack(acts_all), exp


def test_replicas_are_promoted_one_by_one_and_every_step_stays_exact(oracle_lib):
    import torch

    from pytsc_amd.vec_env import VecEngine

    R, steps = 6, 200
    orc, acts, exp = _oracle_trace(oracle_lib, R, steps, 21)
    eng = VecEngine(_tables("syn_3x3", R, resident_capacity=1472, **KW), family="auto")
    assert eng.spill_counts()["limit"] == 16 and eng.spill_counts()["capacity"] > 1472
    live = []
    for s in range(steps):
        buf = eng.step(torch.tensor(acts[s], device="cuda"))
        assert eng.family == "resident"  # nobody drags the handle to the list family
        for g, e, name in zip((buf["obs"], buf["stats"], buf["mask"], buf["local_reward64"]), exp[s], ("obs", "stats", "mask", "reward")):
            assert np.array_equal(g.cpu().numpy(), e), (s, name)
        assert bool(buf["valid"].all())
        live.append(eng.spill_counts()["live"])
        if s % 40 == 39:  # the helpers read a promoted replica from its pool slot
            cnt, spd, _ = eng.lane_detail()
            for r in range(R):
                d, _, _, _ = eng.snapshot(r)
                assert np.array_equal(np.bincount(d[d < eng.tables.n_lanes], minlength=eng.tables.n_lanes), cnt[r, :, 0].cpu().numpy())
            c3 = eng.replica_counts().cpu().numpy()
            assert np.array_equal(c3[:, 0], c3[:, 1] + c3[:, 2])  # K5: entered == running + finished, in the pool too
    assert live[-1] >= 1 and live[0] == 0 and sorted(live) == live and len(set(live)) >= 2, live  # one by one, not all at once
    for r in range(R):
        assert all(np.array_equal(x, y) for x, y in zip(orc.snapshot(r), eng.snapshot(r))), r
    c = eng.path_counts()
    assert c["stalled"] == 0 and eng.errors() == 0 and orc.errors() == 0, c
    eng.close()


def test_a_replica_that_stalls_is_promoted_and_caught_up(oracle_lib):
    """resident_margin 1 and a caller that never waits: some replica runs out of records before the host looks.  It stalls
    (valid == 0, rows of its last completed step), is promoted when the host gets to it and replays the steps it missed from
    the action ring: from then on exact again -- and so are, throughout, all rows marked valid."""
    import torch

    from pytsc_amd.vec_env import VecEngine

    kw = dict(flow_rate=1800.0, n_flow_sets=2, sim_length=1500, episode_limit=500)
    R, steps = 6, 270
    orc, acts, exp = _oracle_trace(oracle_lib, R, steps, 22, kw)
    eng = VecEngine(_tables("syn_3x3", R, resident_capacity=1472, resident_margin=1, **kw), family="auto")
    acts_dev = torch.tensor(acts, device="cuda")
    torch.cuda.synchronize()
    got, valid = [], []
    for s in range(steps):  # no host synchronisation in this loop
        buf = eng.step(acts_dev[s])
        got.append(buf["obs"].clone())
        valid.append(buf["valid"].clone())
    torch.cuda.synchronize()
    assert eng.family == "resident"
    n_stale = 0
    run = np.zeros(R, int)
    for s in range(steps):
        g, v = got[s].cpu().numpy(), valid[s].cpu().numpy()
        for r in range(R):
            assert np.array_equal(g[r], exp[s][0][r]) or v[r] == 0, (s, r)
            run[r] = run[r] + 1 if v[r] == 0 else 0
            assert run[r] <= 17, (s, r)  # never longer than the run-ahead window
            n_stale += v[r] == 0
    assert n_stale >= 1
    assert bool(valid[-1].all())
    c = eng.path_counts()
    assert c["stalled"] >= 1 and eng.spill_counts()["live"] >= c["stalled"], (c, eng.spill_counts())
    for r in range(R):
        assert all(np.array_equal(x, y) for x, y in zip(orc.snapshot(r), eng.snapshot(r))), r
    assert eng.errors() == 0 and orc.errors() == 0
    eng.close()


@pytest.mark.parametrize("how", ["state", "replay"])
def test_beyond_the_pool_limit_the_handle_moves_to_the_list_family(oracle_lib, how, monkeypatch, capfd):
    """(round 5) `state`: the blobs -- the replicas' own and the pool's two -- are written across into the list family's
    arrays (kr_to_list); `replay`: the log is replayed, as in rounds 3-4 (PYTSC_AMD_MIGRATE=replay)."""
    import torch

    monkeypatch.setenv("PYTSC_AMD_MIGRATE", how)

    from pytsc_amd.vec_env import VecEngine

    R, steps = 6, 200
    orc, acts, exp = _oracle_trace(oracle_lib, R, steps, 21)
    eng = VecEngine(_tables("syn_3x3", R, resident_capacity=1472, resident_spill_limit=2, **KW), family="auto")
    fam = []
    for s in range(steps):
        buf = eng.step(torch.tensor(acts[s], device="cuda"))
        fam.append(eng.family)
        assert np.array_equal(buf["obs"].cpu().numpy(), exp[s][0]) and np.array_equal(buf["stats"].cpu().numpy(), exp[s][1]), s
    assert fam[0] == "resident" and fam[-1] == "list"
    assert ("state written across" in capfd.readouterr().err) == (how == "state")
    for r in range(R):
        assert all(np.array_equal(x, y) for x, y in zip(orc.snapshot(r), eng.snapshot(r))), r
    assert eng.errors() == 0
    eng.close()


def test_a_promoted_replica_that_restarts_goes_back_to_its_own_blob(oracle_lib):
    import torch

    from pytsc_amd.vec_env import VecEngine

    R = 4
    eng = VecEngine(_tables("syn_3x3", R, resident_capacity=1472, **KW), family="auto")
    orc = oracle_lib.Oracle(_tables("syn_3x3", R, **KW))
    rng = np.random.default_rng(8)
    out = orc.observe()
    restarted = False
    max_live = 0
    for s in range(230):
        m = out.mask
        acts = (rng.random(m.shape) * m).argmax(-1).astype(np.int32)
        out = orc.step(acts)
        buf = eng.step(torch.tensor(acts, device="cuda"))
        assert np.array_equal(buf["obs"].cpu().numpy(), out.obs), s
        live = eng.spill_counts()["live"]
        max_live = max(max_live, live)
        if live >= 2 and not restarted:
            mask = np.zeros(R, np.uint8)
            # (the oracle knows which replicas hold the most vehicles: those are the promoted ones)
            big = np.argsort([-len(orc.snapshot(r)[0]) for r in range(R)])[:1]
            mask[big] = 1
            orc.reset(mask)
            eng.reset(torch.tensor(mask))
            out = orc.observe()
            buf = eng.observe()
            assert np.array_equal(buf["obs"].cpu().numpy(), out.obs)
            assert eng.spill_counts()["live"] == live - 1
            restarted = True
    assert restarted
    # (the slot the restart gave back was handed out again -- the pool holds as many slots as it ever held replicas at once)
    assert eng.spill_counts()["live"] >= 2 and eng.spill_counts()["slots"] == max_live, (eng.spill_counts(), max_live)
    for r in range(R):
        assert all(np.array_equal(x, y) for x, y in zip(orc.snapshot(r), eng.snapshot(r))), r
    assert eng.errors() == 0
    eng.close()


def test_a_promoted_replica_that_is_parked_gives_its_slot_back(oracle_lib):
    """tsc_park of a replica that lives in the spill pool -- the pool's launch is keyed on the slot table
    only and went on stepping it (rows and valid_dev written by two launches).  Now the parked replica's slot is given back
    (valid == 0, its rows untouched), the others stay exact; after a restart of the parked replica it is an ordinary replica
    again (exact from the empty network on)."""
    import torch

    from pytsc_amd.vec_env import VecEngine

    R = 4
    eng = VecEngine(_tables("syn_3x3", R, resident_capacity=1472, **KW), family="auto")
    orc = oracle_lib.Oracle(_tables("syn_3x3", R, **KW))
    rng = np.random.default_rng(8)
    out = orc.observe()
    parked_r, parked_at, slots_then = None, None, None
    frozen = None
    for s in range(260):
        m = out.mask
        acts = (rng.random(m.shape) * m).argmax(-1).astype(np.int32)
        out = orc.step(acts)
        buf = eng.step(torch.tensor(acts, device="cuda"))
        live = eng.spill_counts()["live"]
        if parked_r is None:
            assert np.array_equal(buf["obs"].cpu().numpy(), out.obs), s
            if live >= 1:
                parked_r = int(np.argmax([len(orc.snapshot(r)[0]) for r in range(R)]))  # the promoted one holds the most vehicles
                mask = np.zeros(R, np.uint8)
                mask[parked_r] = 1
                eng.park(torch.tensor(mask))
                assert eng.spill_counts()["live"] == live - 1
                slots_then = eng.spill_counts()["slots"]
                frozen = buf["obs"][parked_r].clone()
                parked_at = s
        else:
            others = [r for r in range(R) if r != parked_r]
            if s < parked_at + 40:  # parked: not stepped, not written, the caller is told
                assert int(buf["valid"][parked_r]) == 0 and torch.equal(buf["obs"][parked_r], frozen), s
                assert bool(buf["valid"][others].all())
                assert np.array_equal(buf["obs"].cpu().numpy()[others], out.obs[others]), s
            if s == parked_at + 40:  # started again from an empty network -- in the oracle too
                mask = np.zeros(R, np.uint8)
                mask[parked_r] = 1
                orc.reset(mask)
                eng.reset(torch.tensor(mask))
                out = orc.observe()
                buf = eng.observe()
                assert np.array_equal(buf["obs"].cpu().numpy(), out.obs) and bool(buf["valid"].all())
            if s > parked_at + 40:
                # (a handle that has parked replicas has no stand-by any more -- include/pytsc_amd.h, tsc_park --: another
                #  replica of this congested run may run out of its 1472 records from here on and stall, valid == 0; the
                #  restarted one is far from that, and every row marked valid is the oracle's)
                v = buf["valid"].cpu().numpy().astype(bool)
                assert v[parked_r] and np.array_equal(buf["obs"].cpu().numpy()[v], out.obs[v]), s
            if s > parked_at + 70:
                break
    assert parked_r is not None and orc.errors() == 0
    eng.close()
