"""The "next" rows of SURVEY.md §8(f) against data recorded from the REFERENCE's own classes
(tests/golden/gen_golden.py, cases syn_*_ctrl_* and syn_2x2_epymarl):

* N1  vectorised rule-based controllers (pytsc_amd/vec_controllers.py) vs the reference's GreedyController /
      MaxPressureController / SOTLController (pytsc/controllers/controllers.py:57-242): allowable-switch flags and
      the three per-phase scores every step, the actions the reference's controller took;
* N3  the EPyMARL wrapper's step / reset tuples (pytsc/wrappers/epymarl.py:93-111), single env and batched;
* N4  the optional per-agent / per-lane info keys, density_map and its maximum spanning tree
      (pytsc/backends/cityflow/metrics.py:170-260).

Every check runs on the CPU with the oracle engine standing in (the fixtures pin the host logic) and, marked gpu,
on the HIP engine through the C-ABI.
"""
import os

import numpy as np
import pytest
import torch

from fixtures import load_case
from oracle_engine import OracleEngine

CTRL_CASES = {"syn_2x2_ctrl_sotl": "sotl", "syn_2x2_ctrl_greedy": "greedy", "syn_1x3_ctrl_max_pressure": "max_pressure"}


def _engine(tables, gpu, family="auto"):
    if gpu:
        from pytsc_amd.vec_env import VecEngine

        return VecEngine(tables, family=family)
    return OracleEngine(tables, 0)


def _run_controllers(case, gpu, R=3, family="auto"):
    from pytsc_amd import vec_controllers as vc

    inputs, exp, ref, cfg, tables = load_case(case, n_replicas=R)
    eng = _engine(tables, gpu, family)
    ctx = vc.VecControllerContext(eng)
    buf = eng.observe()
    P = exp["x_allow"].shape[2]
    gen = torch.Generator(device=eng.device)
    gen.manual_seed(0)
    n_checked_ties = 0
    for s in range(inputs["steps"]):
        counts = eng.lane_window_counts()
        allow = ctx.allowable(buf)
        q, pr, fl = vc.queue_scores(ctx, counts), vc.pressure_scores(ctx, counts), vc.flow_scores(ctx, counts)
        # round 5: the same three scores as the STEP computes them (tsc_step_out_t::phase_scores_dev: oracle on the CPU, the
        # signal phase of kr_step / k_agent on the GPU) against the reference's controllers' own numbers
        ps = buf["phase_scores"]
        fq, fp, ff = vc.queue_scores(ctx, ps), vc.pressure_scores(ctx, ps), vc.flow_scores(ctx, ps)
        for r in (0, R - 1):
            assert np.array_equal(fq[r].cpu().numpy()[:, :P], exp["x_score_queue"][s]), (s, "fused queue")
            assert np.array_equal(fp[r].cpu().numpy()[:, :P], exp["x_score_pressure"][s]), (s, "fused pressure")
            assert np.array_equal(ff[r].cpu().numpy()[:, :P], exp["x_score_flow"][s]), (s, "fused flow")
        assert int(ps[:, :, P:].abs().sum()) == 0  # phases the signals do not have: 0
        if gpu:  # ... and the fused controller launch (tsc_controller_actions) on those scores
            kind_ = CTRL_CASES[case]
            fa = eng.controller_actions(kind_, buf, seed=s)[0].cpu().numpy()
            if kind_ == "sotl":
                assert fa.tolist() == exp["actions"][s].tolist(), (s, "fused sotl")
            else:
                sc_ = (fq if kind_ == "greedy" else fp)[0]
                sc_ = torch.where(allow[0], sc_, torch.full_like(sc_, float("-inf"))).cpu().numpy()
                cur_ = buf["phase"][0, :, 0].cpu().numpy()
                for a in range(tables.n_agents):
                    if bool(ctx.green[a, cur_[a]]):
                        assert sc_[a, fa[a]] == sc_[a].max(), (s, a, "fused argmax")
                    else:
                        assert fa[a] == (cur_[a] + 1) % int(ctx.n_phases[a])
        for r in (0, R - 1):
            assert np.array_equal(allow[r].cpu().numpy().astype(np.int8)[:, :P], exp["x_allow"][s]), (s, "allow")
            assert np.array_equal(q[r].cpu().numpy()[:, :P], exp["x_score_queue"][s]), (s, "queue")
            assert np.array_equal(pr[r].cpu().numpy()[:, :P], exp["x_score_pressure"][s]), (s, "pressure")
            assert np.array_equal(fl[r].cpu().numpy()[:, :P], exp["x_score_flow"][s]), (s, "flow")
        ref_act = exp["actions"][s]
        kind = CTRL_CASES[case]
        if kind == "sotl":  # deterministic: the very actions of the reference's controller
            assert vc.sotl(ctx, buf, counts)[0].cpu().tolist() == ref_act.tolist(), s
        else:
            fn, score = (vc.greedy, q) if kind == "greedy" else (vc.max_pressure, pr)
            pick = fn(ctx, buf, counts, gen)[0].cpu().numpy()
            sc = torch.where(allow[0], score[0], torch.full_like(score[0], float("-inf"))).cpu().numpy()
            cur = buf["phase"][0, :, 0].cpu().numpy()
            for a in range(tables.n_agents):
                green = bool(ctx.green[a, cur[a]])
                if not green:  # yellow: both take the next phase
                    assert pick[a] == ref_act[a] == (cur[a] + 1) % int(ctx.n_phases[a])
                else:  # both picks are maximisers of the same scores (ties are drawn at random on either side)
                    assert sc[a, ref_act[a]] == sc[a].max() and sc[a, pick[a]] == sc[a].max(), (s, a)
                    n_checked_ties += int((sc[a] == sc[a].max()).sum() > 1)
        buf = eng.step(torch.tensor(np.repeat(ref_act[None], R, 0), dtype=torch.int32, device=eng.device))
        # the env under the reference's controller is the recorded one
        assert np.array_equal(buf["obs"][0].cpu().numpy(), exp["obs"][s].astype(np.float32)), s
    assert eng.errors() == 0
    eng.close()


@pytest.mark.parametrize("case", sorted(CTRL_CASES))
def test_vec_controllers_match_reference_controllers_cpu(case):
    _run_controllers(case, gpu=False)


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["list", "resident"])
@pytest.mark.parametrize("case", sorted(CTRL_CASES))
def test_vec_controllers_match_reference_controllers_gpu(case, family):
    _run_controllers(case, gpu=True, family=family)


# ---------------------------------------------------------------- N4: optional info keys, density map, MST
def _run_metrics(engine_factory):
    from test_facade import make_env

    case = "syn_2x2_ctrl_sotl"
    env, inputs = make_env(case, **({"engine_factory": engine_factory} if engine_factory else {"engine_factory": None}))
    _, exp, ref, _, _ = load_case(case)
    ts_list = list(env.traffic_signals.values())
    lanes = ref["lanes"]
    agent_keys = ("phase", "n_queued", "mean_speed", "mean_delay", "density", "pressure")
    lane_keys = ("n_vehicles", "n_queued", "mean_speed", "occupancy")
    dm_steps = exp["x_dm_steps"].tolist()
    for s in range(inputs["steps"]):
        _, done, info = env.step([int(a) for a in exp["actions"][s]])
        got_a = np.array([[float(info[f"{ts.id}__{k}"]) for k in agent_keys] for ts in ts_list])
        got_l = np.array([[float(info[f"{l}__{k}"]) for k in lane_keys] for l in lanes])
        np.testing.assert_allclose(got_a, exp["x_opt_agent"][s], rtol=1e-9, atol=1e-12, err_msg=f"agent keys step {s}")
        np.testing.assert_allclose(got_l, exp["x_opt_lane"][s], rtol=1e-9, atol=1e-12, err_msg=f"lane keys step {s}")
        assert len(info) == 8 + 2 + len(agent_keys) * len(ts_list) + len(lane_keys) * len(lanes)
        if s in dm_steps:
            k = dm_steps.index(s)
            np.testing.assert_allclose(env.metrics.density_map, exp["x_density_map"][k], rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(np.asarray(env.metrics.mst), exp["x_mst"][k], rtol=1e-9, atol=1e-12)
        if done:
            env.restart()
    env.close()


def test_optional_info_keys_density_map_and_mst_cpu():
    from test_facade import oracle_engine_factory

    _run_metrics(oracle_engine_factory)


@pytest.mark.gpu
def test_optional_info_keys_density_map_and_mst_gpu():
    _run_metrics(None)


# ---------------------------------------------------------------- N3: wrapper tuples
def _run_wrapper(engine_factory):
    import gzip
    import json
    import os

    from fixtures import GOLDEN
    from pytsc_amd.flows import expand_flows
    from pytsc_amd.grid import generate_grid
    from pytsc_amd.wrappers.epymarl import EPyMARLTrafficSignalNetwork

    case = "syn_2x2_epymarl"
    with gzip.open(os.path.join(GOLDEN, case, "inputs.json.gz"), "rt") as f:
        inputs = json.load(f)
    _, exp, ref, _, _ = load_case(case)
    net = generate_grid(*inputs["roadnet"]["grid"])
    kw = dict(hip={"roadnet_json": net, "flow_entries": [expand_flows(inputs["flows"])]},
              signal=dict(inputs["signal"]), misc=dict(inputs["misc"]))
    if engine_factory is not None:
        kw["engine_factory"] = engine_factory
    w = EPyMARLTrafficSignalNetwork(map_name=case, simulator_backend="hip", **kw)
    assert w.get_env_info()["n_agents"] == len(ref["agent_order"])
    assert w.get_obs() == exp["obs0"].astype(int).tolist()
    n_reset = 0
    for s in range(inputs["steps"]):
        obs, rew, done, trunc, info = w.step([int(a) for a in exp["actions"][s]])
        assert trunc is False and bool(done) == bool(exp["x_wrap_done"][s])
        assert rew == pytest.approx(float(exp["x_wrap_reward"][s]), rel=1e-9, abs=1e-9), s
        assert obs == exp["obs"][s].astype(int).tolist(), s
        assert w.get_avail_actions() == exp["mask"][s].astype(int).tolist()
        np.testing.assert_allclose(w.get_local_rewards(), exp["local_reward"][s], rtol=1e-9, atol=1e-9)
        for k, key in enumerate(inputs["info_keys"]):
            assert info[key] == pytest.approx(exp["info"][s][k], rel=1e-9, abs=1e-9), (s, key)
        if done:
            r_obs, r_state = w.reset()
            assert np.array_equal(np.asarray(r_obs, np.int64), exp["x_wrap_reset_obs"][n_reset].astype(np.int64))
            n_reset += 1
    assert n_reset == len(exp["x_wrap_reset_obs"])
    w.close()


def test_epymarl_wrapper_tuples_match_reference_cpu():
    from test_facade import oracle_engine_factory

    _run_wrapper(oracle_engine_factory)


@pytest.mark.gpu
def test_epymarl_wrapper_tuples_match_reference_gpu():
    _run_wrapper(None)


@pytest.mark.gpu
def test_vec_epymarl_wrapper_matches_reference_tuples():
    """The batched wrapper (pytsc_amd/wrappers/vec.py): every replica, fed the reference's actions, returns the
    reference wrapper's reward / done / observations."""
    import gzip
    import json
    import os

    from fixtures import GOLDEN
    from pytsc_amd.flows import expand_flows
    from pytsc_amd.grid import generate_grid
    from pytsc_amd.wrappers.vec import VecEPyMARLTrafficSignalNetwork

    case = "syn_2x2_epymarl"
    with gzip.open(os.path.join(GOLDEN, case, "inputs.json.gz"), "rt") as f:
        inputs = json.load(f)
    _, exp, ref, _, _ = load_case(case)
    R = 5
    w = VecEPyMARLTrafficSignalNetwork(
        map_name=case, n_replicas=R,
        hip={"roadnet_json": generate_grid(*inputs["roadnet"]["grid"]), "flow_entries": [expand_flows(inputs["flows"])]},
        signal=dict(inputs["signal"]), misc=dict(inputs["misc"]))
    for s in range(inputs["steps"]):
        acts = torch.tensor(np.repeat(exp["actions"][s][None], R, 0), dtype=torch.int32, device="cuda")
        obs, rew, done, trunc, info = w.step(acts)
        assert obs.shape == (R, len(ref["agent_order"]), ref["obs_size"])
        for r in (0, R - 1):
            assert np.array_equal(obs[r].cpu().numpy(), exp["obs"][s].astype(np.float32)), s
            assert float(rew[r]) == pytest.approx(float(exp["x_wrap_reward"][s]), rel=1e-6, abs=1e-6)
            assert bool(done[r]) == bool(exp["x_wrap_done"][s])
        if bool(done.any()):
            w.reset()
    w.close()


# ---------------------------------------------------------------- N2: a generator-written flow file through the HIP path
@pytest.mark.gpu
@pytest.mark.parametrize("family", ["list", "resident"])
def test_trip_generator_file_runs_on_hip_like_the_oracle(family, tmp_path, oracle_lib):
    """A flow file written by the restated Gaussian inter-arrival generator (tests/golden/trips pins its trips
    against the reference's generator), read back from disk, compiled and stepped for 60 env steps: HIP == oracle,
    bit for bit."""
    import pytsc_amd.trip_generator as tg
    from pytsc_amd.config import Config
    from pytsc_amd.flows import load_flow_file
    from pytsc_amd.scenario import build_tables
    from pytsc_amd.vec_env import VecEngine

    gen = tg.CityFlowTripGenerator("syn_2x2", 0, 720, 5.0, 0.8, seed=0)
    path = gen.generate_flows(str(tmp_path))
    assert os.path.basename(path) == "syn_2x2__gaussian_144_flows.json"
    entries = load_flow_file(path)
    assert len(entries) > 300

    def tables(R):
        cfg = Config("syn_2x2", signal={"action_space": "phase_selection", "round_robin": False},
                     hip={"flow_entries": [entries], "sim_length": 720})
        return build_tables(cfg, n_replicas=R)

    R = 3
    eng, orc = VecEngine(tables(R), family=family), oracle_lib.Oracle(tables(R))
    rng = np.random.default_rng(9)
    out = orc.observe()
    for s in range(60):
        m = out.mask
        acts = (rng.random(m.shape) * m).argmax(-1).astype(np.int32)
        out = orc.step(acts)
        buf = eng.step(torch.tensor(acts, device="cuda"))
        assert np.array_equal(buf["obs"].cpu().numpy(), out.obs), s
        assert np.array_equal(buf["stats"].cpu().numpy(), out.stats), s
    for r in range(R):
        assert all(np.array_equal(x, y) for x, y in zip(orc.snapshot(r), eng.snapshot(r)))
    assert eng.errors() == 0 and orc.errors() == 0
    eng.close()
    orc.close()


# ---------------------------------------------------------------- N4: observation noise, lane dropout (numpy stream parity)
def _run_noisy_obs(case, engine_factory):
    import random

    from test_facade import make_env

    _, exp, ref, _, _ = load_case(case)
    random.seed(0)
    np.random.seed(0)  # as tests/golden/gen_golden.py seeds the reference
    env, inputs = make_env(case, engine_factory=engine_factory)
    assert env.get_observation_size() == ref["obs_size"]
    assert env.get_observations() == exp["obs0"].astype(int).tolist()
    for s in range(inputs["steps"]):
        _, done, _ = env.step([int(a) for a in exp["actions"][s]])
        assert env.get_observations() == exp["obs"][s].astype(int).tolist(), f"obs step {s}"
        assert env.get_state() == exp["state"][s].astype(int).tolist(), f"state step {s}"
        if done:
            env.restart()
    env.close()


@pytest.mark.parametrize("case", ["syn_2x2_obs_noise", "syn_2x2_lane_dropout"])
def test_observation_noise_and_lane_dropout_match_reference_cpu(case):
    from test_facade import oracle_engine_factory

    exp = load_case(case)[1]
    if case == "syn_2x2_obs_noise":  # the fixture is not vacuous: noise changed some entries of the plain observation
        plain = load_case("syn_2x2_fixed_time")[1]
        assert exp["obs"].shape[2] == plain["obs"].shape[2]
    _run_noisy_obs(case, oracle_engine_factory)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["syn_2x2_obs_noise", "syn_2x2_lane_dropout"])
def test_observation_noise_and_lane_dropout_match_reference_gpu(case):
    _run_noisy_obs(case, None)
