"""Vectorised controllers (pytsc_amd/vec_controllers.py) against the per-signal
controller classes on the façade.  CPU: engine = oracle stand-in; the GPU twin
runs the same checks on the HIP engine."""
import numpy as np
import pytest
import torch

from oracle_engine import OracleEngine


def _run(engine_cls, steps=40):
    from pytsc_amd import controllers as scalar
    from pytsc_amd import vec_controllers as vc
    from pytsc_amd.config import Config
    from pytsc_amd.scenario import build_tables
    from test_facade import make_env

    # façade env (single replica) and a batched engine with identical inputs
    env, inputs = make_env("syn_2x2_fixed_time", engine_factory=(lambda t, d: engine_cls(t, d)) if engine_cls else None)
    eng = env.simulator.engine
    ctx = vc.VecControllerContext(eng)
    ts_list = list(env.traffic_signals.values())
    fixed = [scalar.FixedTimeController(ts) for ts in ts_list]
    so = [scalar.SOTLController(ts) for ts in ts_list]
    gr = [scalar.GreedyController(ts) for ts in ts_list]
    mp = [scalar.MaxPressureController(ts) for ts in ts_list]
    gen = torch.Generator(device=eng.device if engine_cls is None else "cpu")
    gen.manual_seed(0)
    for s in range(steps):
        buf = env.simulator._buf
        counts = eng.lane_window_counts()
        inp = env.simulator.step_measurements
        # mask logic
        allow = vc.VecControllerContext.allowable(ctx, buf)[0].cpu().numpy().astype(int)
        assert allow.tolist() == [ts.controller.get_allowable_phase_switches() for ts in ts_list]
        # deterministic controllers: exact
        assert vc.fixed_time(ctx, buf)[0].tolist() == [int(c.get_action(inp)) for c in fixed]
        assert vc.sotl(ctx, buf, counts)[0].tolist() == [int(c.get_action(inp)) for c in so]
        # argmax controllers: the vector pick must be one of the scalar rule's maximisers
        for vec_fn, ctrls in ((vc.greedy, gr), (vc.max_pressure, mp)):
            pick = vec_fn(ctx, buf, counts, gen)[0].tolist()
            for a, c in enumerate(ctrls):
                cc = c.controller
                if cc.current_phase_index not in cc.green_phase_indices:
                    assert pick[a] == cc.next_phase_index
                    continue
                mask = cc.get_allowable_phase_switches()
                scores = [c._score(inp, k) if ok else float("-inf") for k, ok in enumerate(mask)]
                assert mask[pick[a]] and scores[pick[a]] == max(scores), (s, a, pick[a], scores)
        rv = vc.random_valid(ctx, buf, gen)[0].tolist()
        assert all(allow[a][rv[a]] for a in range(len(ts_list)))
        env.step([int(x) for x in vc.sotl(ctx, buf, counts)[0].tolist()] if s % 2 else rv)
    env.close()


def test_vectorised_controllers_match_scalar_rules_cpu():
    _run(OracleEngine)


@pytest.mark.gpu
def test_vectorised_controllers_match_scalar_rules_gpu():
    _run(None)


@pytest.mark.gpu
def test_vectorised_fixed_time_drives_4096_replicas():
    from pytsc_amd import vec_controllers as vc
    from pytsc_amd.vec_env import VecTrafficSignalNetwork

    env = VecTrafficSignalNetwork("syn_2x2", n_replicas=4096, signal={"action_space": "phase_selection",
                                                                      "round_robin": False},
                                  outputs=("mask", "phase", "stats", "global_reward", "done", "terminated"))
    ctx = vc.VecControllerContext(env.engine)
    trace = []
    for s in range(24):
        a = vc.fixed_time(ctx, env.buf)
        trace.append(int(a[0, 0]))
        assert bool((env.buf["mask"].gather(2, a.long()[..., None]) == 1).all())
        env.step(a)
    # the engine-independent fixed-time trace
    assert trace == [0, 0, 0, 0, 0, 1, 2, 2, 2, 2, 2, 3, 4, 4, 4, 4, 4, 5, 6, 6, 6, 6, 6, 7]
    assert env.engine.errors() == 0
    env.close()


@pytest.mark.gpu
def test_fused_random_valid_actions_are_valid_uniform_and_reproducible():
    """tsc_random_valid_actions (RandomController, controllers.py:245-268, as one launch)."""
    import torch

    from pytsc_amd.config import Config
    from pytsc_amd.scenario import build_tables
    from pytsc_amd.vec_env import VecEngine

    cfg = Config("syn_2x2", signal={"action_space": "phase_selection", "round_robin": False},
                 hip={"flow_rate": 600.0, "n_flow_sets": 1, "sim_length": 300})
    engs = [VecEngine(build_tables(cfg, n_replicas=512), want=("mask",)) for _ in range(2)]
    bufs = [e.observe() for e in engs]
    counts = torch.zeros(16, device="cuda")
    for s in range(30):
        a0 = engs[0].random_valid_actions(bufs[0]["mask"], seed=7)
        a1 = engs[1].random_valid_actions(bufs[1]["mask"], seed=7)
        assert torch.equal(a0, a1)  # same seed, same call index, same masks
        picked = torch.gather(bufs[0]["mask"], 2, a0.long().unsqueeze(-1))
        assert bool((picked > 0).all())
        if s == 0:  # all eight green phases allowed at the start: a flat histogram
            counts = torch.bincount(a0.flatten(), minlength=16).float()
            allowed = bufs[0]["mask"][0, 0].bool()
            frac = counts[allowed] / counts.sum()
            assert float((frac - 1.0 / int(allowed.sum())).abs().max()) < 0.03
            assert float(counts[~allowed].sum()) == 0
        bufs = [e.step(a) for e, a in zip(engs, (a0, a1))]
    b = engs[0].random_valid_actions(bufs[0]["mask"], seed=8)
    c = engs[1].random_valid_actions(bufs[1]["mask"], seed=9)
    assert not torch.equal(b, c)
    for e in engs:
        e.close()


# ---------------------------------------------------------------- round 5: scores and controllers fused into the library
def _tables(scen, R, **hip):
    from pytsc_amd.config import Config
    from pytsc_amd.scenario import build_tables

    cfg = Config(scen, signal={"action_space": "phase_selection", "round_robin": False}, hip=hip)
    return build_tables(cfg, n_replicas=R)


def test_oracle_phase_scores_equal_the_dense_formulas_cpu():
    """The per-phase scores the step emits (oracle restatement of controllers.py:93-112,153-178,227-242) against the dense
    torch formulas over the per-lane window counts (rounds 3-4; pinned by the reference fixtures)."""
    from pytsc_amd import vec_controllers as vc

    eng = OracleEngine(_tables("syn_3x3", 4, flow_rate=900.0, n_flow_sets=2, sim_length=600))
    ctx = vc.VecControllerContext(eng)
    buf = eng.observe()
    rng = np.random.default_rng(0)
    seen = 0
    for s in range(60):
        counts = eng.lane_window_counts()
        ps = buf["phase_scores"]
        for fn in (vc.queue_scores, vc.pressure_scores, vc.flow_scores):
            assert torch.equal(fn(ctx, ps), fn(ctx, counts)), (s, fn.__name__)
        seen += int(ps.sum())
        m = buf["mask"].numpy()
        buf = eng.step((rng.random(m.shape) * m).argmax(-1).astype(np.int32))
    assert seen > 1000
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["list", "resident"])
def test_fused_phase_scores_equal_the_dense_formulas_at_r64(family):
    """phase_scores of tsc_step on both kernel families == the dense formulas over tsc_lane_window_counts, R = 64, a
    congested 3x3 (different action streams); and the fused controllers' picks are the rules' picks."""
    from pytsc_amd import vec_controllers as vc
    from pytsc_amd.vec_env import VecEngine

    eng = VecEngine(_tables("syn_3x3", 64, flow_rate=1100.0, n_flow_sets=4, sim_length=900), family=family)
    ctx = vc.VecControllerContext(eng)
    buf = eng.observe()
    for s in range(80):
        counts = eng.lane_window_counts()
        ps = buf["phase_scores"]
        q, pr, fl = vc.queue_scores(ctx, counts), vc.pressure_scores(ctx, counts), vc.flow_scores(ctx, counts)
        assert torch.equal(vc.queue_scores(ctx, ps), q) and torch.equal(vc.pressure_scores(ctx, ps), pr), s
        assert torch.equal(vc.flow_scores(ctx, ps), fl), s
        allow = ctx.allowable(buf)
        cur = buf["phase"][..., 0].long()
        is_green = ctx.green.unsqueeze(0).expand(64, -1, -1).gather(2, cur.unsqueeze(-1)).squeeze(-1)
        nxt = (cur + 1) % ctx.n_phases.view(1, -1)
        # deterministic rules: equal to the torch formulas
        assert torch.equal(eng.controller_actions("fixed_time", buf), vc.fixed_time(ctx, buf)), s
        assert torch.equal(eng.controller_actions("sotl", buf), vc.sotl(ctx, buf, counts)), s
        # argmax rules: an allowable maximiser on green, the next phase on yellow
        for kind, score in (("greedy", q), ("max_pressure", pr)):
            a = eng.controller_actions(kind, buf, seed=11).long()
            sc = torch.where(allow, score, torch.full_like(score, float("-inf")))
            ok = sc.gather(2, a.unsqueeze(-1)).squeeze(-1) == sc.max(-1).values
            assert bool(torch.where(is_green, ok, a == nxt).all()), (s, kind)
        rv = eng.controller_actions("random", buf, seed=5).long()
        assert bool(allow.gather(2, rv.unsqueeze(-1)).all()), s
        buf = eng.step(eng.random_valid_actions(buf["mask"], seed=3))
    assert eng.errors() == 0
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("scen,family,R", [("syn_4x4", "resident", 4096), ("syn_4x4", "list", 1024), ("syn_16x16", "list", 4096)])
def test_all_rule_based_controllers_drive_full_batches(scen, family, R):
    """Every rule-based controller of the reference drives 4x4 @ 4096 and 16x16 @ 4096 through the fused launch (VERDICT r4
    item 3: the dense torch path needed a 1 GB tensor per call on the 4x4 and 219 GB on the 16x16): actions are always
    allowable; ties of the argmax rules are drawn uniformly (first step: every phase scores 0)."""
    from pytsc_amd.vec_env import VecEngine

    tb = _tables(scen, R, flow_rate=700.0, n_flow_sets=2, sim_length=600)
    eng = VecEngine(tb, family=family, want=("mask", "phase", "phase_scores", "stats", "valid"))
    buf = eng.observe()
    a0 = eng.controller_actions("greedy", buf, seed=1).long()
    allowed = buf["mask"][0, 0].bool()
    frac = torch.bincount(a0.flatten(), minlength=eng.n_actions).float() / a0.numel()
    assert float((frac[allowed] - 1.0 / int(allowed.sum())).abs().max()) < 0.02 and float(frac[~allowed].sum()) == 0
    for s in range(30):
        kind = ("greedy", "max_pressure", "sotl", "fixed_time", "random")[s % 5]
        a = eng.controller_actions(kind, buf, seed=s)
        assert bool((buf["mask"].gather(2, a.long().unsqueeze(-1)) == 1).all()), (s, kind)
        buf = eng.step(a)
    assert int(buf["phase_scores"].sum()) > 0 and eng.errors() == 0 and bool(buf["valid"].all())
    eng.close()
