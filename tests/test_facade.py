"""Host-side façade (pytsc_amd.TrafficSignalNetwork, backends registry, wrappers,
controllers) driven through the reference's list-of-lists API and compared with
the golden fixtures.  CPU: the engine underneath is the oracle stand-in
(tests/oracle_engine.py); the GPU twin is tests/test_hip_facade.py."""
import gzip
import json
import os

import numpy as np
import pytest

from fixtures import GOLDEN, load_case
from oracle_engine import oracle_engine_factory


def make_env(case, engine_factory=oracle_engine_factory, **extra):
    from pytsc_amd import TrafficSignalNetwork
    from pytsc_amd.flows import expand_flows
    from pytsc_amd.grid import generate_grid

    with gzip.open(os.path.join(GOLDEN, case, "inputs.json.gz"), "rt") as f:
        inputs = json.load(f)
    net = generate_grid(*inputs["roadnet"]["grid"]) if "grid" in inputs["roadnet"] else inputs["roadnet"]["net"]
    env = TrafficSignalNetwork(
        case, "cityflow", engine_factory=engine_factory,
        hip={"roadnet_json": net, "flow_entries": [expand_flows(inputs["flows"])]},
        signal=dict(inputs["signal"]), misc=dict(inputs["misc"]), **extra)
    return env, inputs


def run_and_compare(env, inputs, exp, n_steps):
    assert env.get_observations() == exp["obs0"].astype(int).tolist()
    assert env.get_action_mask() == exp["mask0"].astype(int).tolist()
    for s in range(n_steps):
        greward, done, info = env.step([int(a) for a in exp["actions"][s]])
        assert env.get_observations() == exp["obs"][s].astype(int).tolist(), f"obs step {s}"
        assert env.get_state() == exp["state"][s].astype(int).tolist(), f"state step {s}"
        assert env.get_action_mask() == exp["mask"][s].astype(int).tolist(), f"mask step {s}"
        np.testing.assert_allclose(env.get_rewards(), exp["local_reward"][s], rtol=1e-9, atol=1e-9)
        assert greward == pytest.approx(exp["global_reward"][s], rel=1e-9, abs=1e-9)
        assert bool(done) == bool(exp["done"][s])
        for k, key in enumerate(inputs["info_keys"]):
            assert info[key] == pytest.approx(exp["info"][s][k], rel=1e-9, abs=1e-9), (s, key)
        assert set(info) == set(inputs["info_keys"]) | {"episode_count", "episode_limit"}
        if done:
            env.restart()


@pytest.mark.parametrize("case", ["syn_2x2_fixed_time", "syn_1x3_switch_rr", "syn_2x2_pressure_2hop"])
def test_facade_list_api_matches_reference(case):
    env, inputs = make_env(case)
    _, exp, ref, _, _ = load_case(case)
    assert env.n_agents == len(ref["agent_order"])
    assert env.get_observation_size() == ref["obs_size"]
    assert env.get_state_size() == ref["state_size"]
    assert env.get_action_size() == ref["action_size"]
    assert env.episode_limit == ref["episode_limit"]
    run_and_compare(env, inputs, exp, 60)
    env.close()


def test_position_matrices_and_agent_stats_match_reference():
    case = "syn_2x2_fixed_time"
    env, inputs = make_env(case)
    _, exp, ref, _, _ = load_case(case)
    pm_steps = list(exp["posmat_steps"])
    for s in range(pm_steps[2] + 1):
        env.step([int(a) for a in exp["actions"][s]])
        if s in pm_steps:
            lanes = env.simulator.step_measurements["lane"]
            e = exp["posmat"][pm_steps.index(s)]
            for k, lane in enumerate(env.parsed_network.lanes):
                row = e[k][~np.isnan(e[k])]
                np.testing.assert_allclose(lanes[lane]["position_matrix"], row, rtol=0, atol=1e-12)
                assert lanes[lane]["n_vehicles"] == exp["lane_n"][s][k]
                assert lanes[lane]["n_queued"] == exp["lane_q"][s][k]
                assert lanes[lane]["occupancy"] == pytest.approx(exp["lane_occ"][s][k], abs=1e-12)
            for a, ts in enumerate(env.traffic_signals.values()):
                got = [ts.n_queued, ts.occupancy, ts.mean_speed, ts.mean_delay, ts.outgoing_occupancy,
                       ts.pressure, ts.time_on_phase, ts.sim_step]
                np.testing.assert_allclose(got, exp["agent_stats"][s][a], rtol=1e-9, atol=1e-12)
    env.close()


def test_fixed_time_controller_reproduces_reference_trace():
    """BASELINE config 0: 2x2 grid, fixed-time controller.  The recorded actions
    come from the reference's FixedTimeController (controllers.py:26-54)."""
    from pytsc_amd.controllers import CONTROLLERS

    case = "syn_2x2_fixed_time"
    env, inputs = make_env(case)
    _, exp, _, _, _ = load_case(case)
    ctrl = [CONTROLLERS["fixed_time"](ts) for ts in env.traffic_signals.values()]
    for s in range(120):
        acts = [int(c.get_action(env.simulator.step_measurements)) for c in ctrl]
        assert acts == exp["actions"][s].tolist(), f"fixed-time trace diverged at step {s}"
        env.step(acts)
    # engine-independent golden trace of one agent
    assert exp["actions"][:24, 0].tolist() == [0, 0, 0, 0, 0, 1, 2, 2, 2, 2, 2, 3, 4, 4, 4, 4, 4, 5, 6, 6, 6, 6, 6, 7]
    env.close()


@pytest.mark.skipif(not os.path.isdir("/root/reference/pytsc"), reason="reference checkout not present")
def test_reference_controllers_drop_in_unchanged():
    """The reference's own rule-based controller classes, imported unmodified,
    run against the façade objects (north star: controllers drop in unchanged)."""
    import sys

    stubs = os.path.join(GOLDEN, "ref_stubs")
    sys.dont_write_bytecode = True
    os.environ.setdefault("SUMO_HOME", "/tmp/none")
    for p in ("/root/reference", stubs):
        if p not in sys.path:
            sys.path.insert(0, p)
    from pytsc.controllers.controllers import (FixedTimeController, GreedyController,
                                               MaxPressureController, SOTLController)

    import pytsc_amd.controllers as mine

    env, inputs = make_env("syn_2x2_fixed_time")
    ts_list = list(env.traffic_signals.values())
    pairs = [
        ([FixedTimeController(ts) for ts in ts_list], [mine.FixedTimeController(ts) for ts in ts_list]),
        ([SOTLController(ts) for ts in ts_list], [mine.SOTLController(ts) for ts in ts_list]),
    ]
    rnd = [(GreedyController, mine.GreedyController), (MaxPressureController, mine.MaxPressureController)]
    for s in range(40):
        inp = env.simulator.step_measurements
        for ref_c, my_c in pairs:
            assert [int(c.get_action(inp)) for c in ref_c] == [int(c.get_action(inp)) for c in my_c]
        for ref_cls, my_cls in rnd:
            for ts in ts_list:
                np.random.seed(s)
                a = int(ref_cls(ts).get_action(inp))
                np.random.seed(s)
                assert a == int(my_cls(ts).get_action(inp))
        acts = [int(c.get_action(inp)) for c in pairs[1][1]]  # SOTL drives
        env.step(acts)
    env.close()


def test_wrappers_keep_the_smac_api():
    from pytsc_amd.flows import expand_flows
    from pytsc_amd.grid import generate_grid
    from pytsc_amd.wrappers.epymarl import EPyMARLTrafficSignalNetwork
    from pytsc_amd.wrappers.pymarl import PyMARLTrafficSignalNetwork

    with gzip.open(os.path.join(GOLDEN, "syn_2x2_fixed_time", "inputs.json.gz"), "rt") as f:
        inputs = json.load(f)
    _, exp, _, _, _ = load_case("syn_2x2_fixed_time")
    kw = dict(engine_factory=oracle_engine_factory,
              hip={"roadnet_json": generate_grid(2, 2), "flow_entries": [expand_flows(inputs["flows"])]},
              signal=dict(inputs["signal"]))
    env = EPyMARLTrafficSignalNetwork("syn_2x2", "hip", **kw)
    info = env.get_env_info()
    assert info["n_agents"] == 4 and info["n_actions"] == 16 and info["obs_shape"] == 324
    assert info["state_shape"] == 212 and info["episode_limit"] == 72
    assert info["obs_info"] == {"lane_obs_dim": 19, "max_n_controlled_lanes": 16, "max_phases": 20}
    assert np.asarray(info["lane_features"]).shape == (4, 16, 9)
    obs, state = env.reset()
    assert np.asarray(obs).shape == (4, 324) and np.asarray(state).shape == (4, 212)
    acts = [int(a) for a in exp["actions"][0]]
    obs, reward, done, truncated, step_info = env.step(acts)
    assert obs == exp["obs"][0].astype(int).tolist()
    assert reward == pytest.approx(exp["global_reward"][0] / 4)
    assert truncated is False and done is False
    assert env.get_avail_actions() == exp["mask"][0].astype(int).tolist()
    env.close()
    env2 = PyMARLTrafficSignalNetwork("syn_2x2", "hip", **kw)
    r, d, i = env2.step(acts)
    assert r == pytest.approx(exp["global_reward"][0])
    env2.close()


def test_domain_randomized_wrapper_pads_to_the_largest_map():
    """pytsc/wrappers/epymarl.py:112-336: fixed interface over maps of different
    size; a new map is drawn when the simulated hour is over."""
    import random

    from pytsc_amd.flows import poisson_flows
    from pytsc_amd.grid import generate_grid
    from pytsc_amd.roadnet import RoadNet
    from pytsc_amd.wrappers.epymarl import DomainRandomizedEPyMARLTrafficSignalNetwork

    maps = {}
    for name, (r, c) in {"syn_1x2": (1, 2), "syn_2x2": (2, 2)}.items():
        net = generate_grid(r, c)
        maps[name] = {"hip": {"roadnet_json": net,
                              "flow_entries": [poisson_flows(RoadNet(net), 500.0, horizon=200, seed=1)],
                              "sim_length": 60, "episode_limit": 4}}
    env = DomainRandomizedEPyMARLTrafficSignalNetwork(
        list(maps), "hip", map_kwargs=maps, rng=random.Random(5), engine_factory=oracle_engine_factory,
        signal={"action_space": "phase_selection", "round_robin": False})
    assert env.max_n_agents == 4
    seen = set()
    for episode in range(12):
        n_before = env.current_n_agents  # reset() reads obs/state BEFORE it may draw a new map (as the reference)
        obs, state = env.reset()
        seen.add(env.current_map)
        n = env.current_n_agents
        info = env.get_env_info()
        assert info["n_agents"] == 4 and np.asarray(info["adjacency_matrix"]).shape == (4, 4)
        assert np.asarray(obs).shape == (4, env.get_obs_size()) and np.asarray(state).shape == (4, env.get_state_size())
        assert all(row == [-1] * env.get_obs_size() for row in obs[n_before:])
        for _ in range(4):
            mask = env.get_avail_actions()
            assert len(mask) == 4 and all(row == [1] * env.get_total_actions() for row in mask[n:])
            acts = [int(np.argmax(row)) for row in mask]  # 4 actions, the surplus ones are ignored
            obs, reward, done, truncated, _ = env.step(acts)
            assert len(obs) == 4 and len(env.get_local_rewards()) == 4 and env.get_local_rewards()[n:] == [0] * (4 - n)
        assert done
    assert seen == set(maps)  # both maps were drawn
    env.close()


@pytest.mark.parametrize("case", ["syn_2x2_selection_rr", "syn_1x3_switch_rr"])
def test_trad_controller_actions_entry_point(case):
    """action_space.get_trad_controller_actions (reference common/actions.py:64-84,190-211): the facade's entry point for
    the rule-based controllers.  phase_selection: the controllers' phase picks themselves, "random" a mask-weighted draw
    of Python's `random`; phase_switch: 1 where the pick differs from the current phase index, else 0."""
    import random

    env, inputs = make_env(case)
    switch = inputs["signal"]["action_space"] == "phase_switch"
    for s in range(25):
        for ctrl in ("greedy", "max_pressure", "sotl", "fixed_time"):
            np.random.seed(7 * s + 1)  # (greedy / max_pressure break ties with numpy's global stream)
            picks = [ts.get_controller_action(ctrl) for ts in env.traffic_signals.values()]
            np.random.seed(7 * s + 1)
            acts = env.action_space.get_trad_controller_actions(ctrl)
            cur = [ts.controller.program.current_phase_index for ts in env.traffic_signals.values()]
            assert acts == ([int(p != c) for p, c in zip(picks, cur)] if switch else picks), (s, ctrl)
        random.seed(100 + s)
        acts = env.action_space.get_trad_controller_actions("random")
        mask = env.get_action_mask()
        random.seed(100 + s)
        for i, ts in enumerate(env.traffic_signals.values()):
            pick = random.choices([0, 1] if switch else range(ts.controller.n_phases), weights=mask[i])[0]
            want = int(ts.controller.program.current_phase_index != pick) if switch else pick
            assert acts[i] == want, (s, i)
        env.step(env.action_space.get_trad_controller_actions("sotl" if s % 2 else "greedy"))
    env.close()
