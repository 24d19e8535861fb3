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
