"""Generate the golden fixtures under tests/golden/<case>/ by running the
REFERENCE's own Python layer (imported from /root/reference, never copied) on
top of the CPU oracle's engine through an Engine-compatible shim
(tests/golden/ref_stubs/cityflow.py).

What this pins: everything above the `cityflow.Engine` boundary — network
parser tables, lane measurements, per-agent stats, observations, states, masks,
rewards, info dicts, phase state machines, the fixed-time controller trace.
What it cannot pin: the engine's own dynamics (CityFlow is absent).

Run (only in the build container, /root/reference must exist):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py [case ...]
Each fixture = inputs.json.gz (roadnet or grid size, compact flows, config
overrides) + expected.npz (arrays) + parser.json (parser tables).
"""
import gzip
import json
import logging
import os
import random
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
REF_SCEN = os.path.join(REF, "pytsc", "scenarios")

sys.dont_write_bytecode = True
os.environ.setdefault("SUMO_HOME", "/tmp/none")
sys.path[:0] = [os.path.join(HERE, "ref_stubs"), REPO, REF]

from pytsc_amd.flows import compact_flows, load_flow_file, poisson_flows  # noqa: E402
from pytsc_amd.grid import generate_grid  # noqa: E402
from pytsc_amd.roadnet import RoadNet  # noqa: E402

CASES = {
    # BASELINE config 0: 2x2 grid, fixed-time controller (green 25 s)
    "syn_2x2_fixed_time": dict(
        roadnet=("grid", 2, 2), flows=("poisson", 700.0, 0), steps=240,
        signal={"action_space": "phase_selection", "round_robin": False},
        policy="fixed_time",
    ),
    "syn_3x3_random": dict(
        roadnet=("grid", 3, 3), flows=("file", "cityflow/syn_3x3/1__syn_3x3__gaussian_600_flows.json"),
        steps=144, signal={"action_space": "phase_selection", "round_robin": False},
        policy="random",
    ),
    "syn_1x3_switch_rr": dict(  # default action space phase_switch + round robin
        roadnet=("grid", 1, 3), flows=("poisson", 600.0, 1), steps=144,
        signal={"action_space": "phase_switch", "round_robin": True},
        policy="random",
    ),
    "syn_2x2_selection_rr": dict(
        roadnet=("grid", 2, 2), flows=("poisson", 500.0, 2), steps=100,
        signal={"action_space": "phase_selection", "round_robin": True},
        policy="random",
    ),
    "syn_2x2_pressure_2hop": dict(
        roadnet=("grid", 2, 2), flows=("poisson", 700.0, 3), steps=100,
        signal={"action_space": "phase_selection", "round_robin": False,
                "reward_function": "max_pressure"},
        misc={"max_hops": 2},
        policy="random",
    ),
    "jinan_3_4_random": dict(
        roadnet=("file", "cityflow/jinan_3_4/roadnet_3_4.json"),
        flows=("file", "cityflow/jinan_3_4/anon_3_4_jinan_real.json"),
        steps=144, signal={"action_space": "phase_selection", "round_robin": False},
        policy="random",
    ),
    "syn_4x4_random": dict(  # BASELINE config 1 shape at R=1
        roadnet=("grid", 4, 4), flows=("poisson", 700.0, 0), steps=144,
        signal={"action_space": "phase_selection", "round_robin": False},
        policy="random",
    ),
    # ---- parser overrides: `phase_sequence` (backends/cityflow/network_parser.py:52-57; the
    #      shipped new_york_arterial scenario sets exactly this list) and `neighbors` (common/network_parser.py:19-35)
    "syn_2x2_phase_sequence": dict(
        roadnet=("grid", 2, 2), flows=("poisson", 600.0, 4), steps=80,
        signal={"action_space": "phase_switch", "round_robin": True},
        cityflow={"phase_sequence": [1, 0, 3, 0, 2, 0, 4, 0]},
        policy="random",
    ),
    "syn_2x3_neighbors_override": dict(
        roadnet=("grid", 2, 3), flows=("poisson", 600.0, 5), steps=80,
        signal={"action_space": "phase_selection", "round_robin": False},
        misc={"max_hops": 2},
        # east-west chains only (the road graph would also link the two rows), and not symmetric on purpose
        network={"neighbors": {
            "intersection_1_1": ["intersection_2_1"], "intersection_2_1": ["intersection_1_1", "intersection_3_1"],
            "intersection_3_1": ["intersection_2_1"], "intersection_1_2": ["intersection_2_2"],
            "intersection_2_2": ["intersection_3_2"], "intersection_3_2": ["intersection_2_2", "intersection_3_1"]}},
        policy="random",
    ),
    # ---- "next" rows: the reference's rule-based controllers drive the env
    # (pytsc/controllers/controllers.py:57-242); every step also records, for every signal and phase, the
    # allowable-switch flags and the three scores the controllers rank phases by (stopped vehicles, pressure,
    # flow), the optional per-agent / per-lane info keys, and every 6th step density_map and its max spanning tree
    # (backends/cityflow/metrics.py:170-260); the EPyMARL wrapper's step / reset tuples (wrappers/epymarl.py:93-111).
    "syn_2x2_ctrl_sotl": dict(
        roadnet=("grid", 2, 2), flows=("poisson", 800.0, 5), steps=120,
        signal={"action_space": "phase_selection", "round_robin": False},
        misc={"return_agent_stats": True, "return_lane_stats": True},
        policy="sotl", extras=True,
    ),
    "syn_2x2_ctrl_greedy": dict(
        roadnet=("grid", 2, 2), flows=("poisson", 800.0, 6), steps=100,
        signal={"action_space": "phase_selection", "round_robin": False},
        policy="greedy", extras=True,
    ),
    "syn_1x3_ctrl_max_pressure": dict(
        roadnet=("grid", 1, 3), flows=("poisson", 800.0, 7), steps=100,
        signal={"action_space": "phase_selection", "round_robin": False},
        policy="max_pressure", extras=True,
    ),
    # N4: observation noise (numpy stream parity: one normal draw per surviving bin) and LaneFeatures lane dropout
    "syn_2x2_obs_noise": dict(
        roadnet=("grid", 2, 2), flows=("poisson", 700.0, 9), steps=60,
        signal={"action_space": "phase_selection", "round_robin": False, "obs_noise_std": 0.35},
        policy="random",
    ),
    "syn_2x2_lane_dropout": dict(
        roadnet=("grid", 2, 2), flows=("poisson", 700.0, 10), steps=60,
        signal={"action_space": "phase_selection", "round_robin": False, "observation_space": "lane_features",
                "obs_dropout_prob": 0.25},
        policy="random",
    ),
    "syn_2x2_epymarl": dict(
        roadnet=("grid", 2, 2), flows=("poisson", 700.0, 8), steps=80,
        signal={"action_space": "phase_selection", "round_robin": False},
        policy="random", wrapper="epymarl",
    ),
}


def _make_scenario_dir(root, name, case):
    sdir = os.path.join(root, "cityflow", name)
    os.makedirs(sdir)
    os.makedirs(os.path.join(root, "default"))
    shutil.copy(os.path.join(REF_SCEN, "default", "config.yaml"), os.path.join(root, "default"))
    kind = case["roadnet"][0]
    if kind == "grid":
        net = generate_grid(case["roadnet"][1], case["roadnet"][2])
    else:
        with open(os.path.join(REF_SCEN, case["roadnet"][1])) as f:
            net = json.load(f)
    with open(os.path.join(sdir, "roadnet.json"), "w") as f:
        json.dump(net, f)
    rn = RoadNet(net)
    if case["flows"][0] == "poisson":
        entries = poisson_flows(rn, case["flows"][1], 3600, seed=case["flows"][2])
    else:
        entries = load_flow_file(os.path.join(REF_SCEN, case["flows"][1]))
    with open(os.path.join(sdir, "flows.json"), "w") as f:
        json.dump(entries, f)
    import yaml

    scen_cfg = {
        "cityflow": dict({"roadnet_file": "roadnet.json", "flow_file": "flows.json",
                          "roadnet_log_file": "roadnet_log_file.json",
                          "replay_log_file": "replay_log_file.txt", "save_replay": False}, **case.get("cityflow", {})),
        "signal": dict(case.get("signal", {})),
        "misc": dict(case.get("misc", {})),
    }
    if "network" in case:
        scen_cfg["network"] = dict(case["network"])
    with open(os.path.join(sdir, "config.yaml"), "w") as f:
        yaml.safe_dump(scen_cfg, f)
    return net, entries


def generate(name):
    case = CASES[name]
    root = tempfile.mkdtemp(prefix="golden_")
    net, entries = _make_scenario_dir(root, name, case)

    import pytsc.backends.cityflow.config as cf_config
    import pytsc.common.config as common_config
    from pytsc.common.utils import EnvLogger

    EnvLogger.set_log_level(logging.WARNING)
    common_config.CONFIG_DIR = root
    cf_config.CONFIG_DIR = os.path.join(root, "cityflow")
    from pytsc import TrafficSignalNetwork
    from pytsc.controllers import controllers as ref_ctrl
    from pytsc.controllers.controllers import FixedTimeController

    random.seed(0)
    np.random.seed(0)
    wrap = None
    if case.get("wrapper") == "epymarl":
        from pytsc.wrappers.epymarl import EPyMARLTrafficSignalNetwork

        wrap = EPyMARLTrafficSignalNetwork(map_name=name, simulator_backend="cityflow")
        env = wrap.tsc_env
    else:
        env = TrafficSignalNetwork(name, "cityflow")
    A = env.n_agents
    lanes = env.parsed_network.lanes
    ts_list = list(env.traffic_signals.values())
    controllers = [FixedTimeController(ts) for ts in ts_list]
    drivers = {"sotl": ref_ctrl.SOTLController, "greedy": ref_ctrl.GreedyController,
               "max_pressure": ref_ctrl.MaxPressureController}
    if case["policy"] in drivers:
        controllers = [drivers[case["policy"]](ts) for ts in ts_list]
    score_ctrl = [(ref_ctrl.GreedyController(ts), ref_ctrl.MaxPressureController(ts), ref_ctrl.SOTLController(ts))
                  for ts in ts_list]
    P_max = max(ts.controller.n_phases if hasattr(ts.controller, "n_phases") else len(ts.config["phases"]) for ts in ts_list)
    extras = {k: [] for k in ("allow", "score_queue", "score_pressure", "score_flow", "dm_steps", "density_map", "mst",
                              "opt_agent", "opt_lane", "wrap_reward", "wrap_done", "wrap_reset_obs")}
    opt_agent_keys = ("phase", "n_queued", "mean_speed", "mean_delay", "density", "pressure")
    opt_lane_keys = ("n_vehicles", "n_queued", "mean_speed", "occupancy")
    rng = np.random.default_rng(12345)

    def record_scores():
        inp = env.simulator.step_measurements
        al = np.zeros((A, P_max), np.int8)
        sq = np.zeros((A, P_max), np.float64)
        sp = np.zeros((A, P_max), np.float64)
        sf = np.zeros((A, P_max), np.float64)
        for a, ts in enumerate(ts_list):
            n_ph = len(ts.config["phases"])
            al[a, :n_ph] = ts.controller.get_allowable_phase_switches()
            g, m, so = score_ctrl[a]
            for k in range(n_ph):
                sq[a, k] = g._compute_queue_for_phase(inp, k)
                sp[a, k] = m._compute_pressure_for_phase(inp, k)
                sf[a, k] = so._compute_flow_for_phase(inp, k)
        extras["allow"].append(al)
        extras["score_queue"].append(sq)
        extras["score_pressure"].append(sp)
        extras["score_flow"].append(sf)

    def lane_arrays():
        lm = env.simulator.step_measurements["lane"]
        assert list(lm.keys()) == lanes
        return (
            [lm[l]["n_vehicles"] for l in lanes], [lm[l]["n_queued"] for l in lanes],
            [float(lm[l]["mean_speed"]) for l in lanes], [float(lm[l]["occupancy"]) for l in lanes],
        )

    def posmats():
        lm = env.simulator.step_measurements["lane"]
        mx = max(len(lm[l]["position_matrix"]) for l in lanes)
        out = np.full((len(lanes), mx), np.nan)
        for i, l in enumerate(lanes):
            pm = lm[l]["position_matrix"]
            out[i, : len(pm)] = pm
        return out

    def agent_stats():
        return [[ts.n_queued, float(ts.occupancy), float(ts.mean_speed), float(ts.mean_delay),
                 float(ts.outgoing_occupancy), float(ts.pressure), float(ts.time_on_phase),
                 float(ts.sim_step)] for ts in ts_list]

    rec = {k: [] for k in ("actions", "obs", "state", "mask", "local_reward", "global_reward",
                           "done", "terminated", "info", "phase", "lane_n", "lane_q",
                           "lane_mean_speed", "lane_occ", "agent_stats")}
    pm_steps, pm_vals = [], []
    obs0 = np.asarray(env.get_observations(), dtype=np.float64)
    state0 = np.asarray(env.get_state(), dtype=np.float64)
    mask0 = np.asarray(env.get_action_mask(), dtype=np.int64)
    info_keys = ("time_step", "average_travel_time", "n_queued", "mean_speed", "mean_delay",
                 "density", "pressure", "network_flow")
    for s in range(case["steps"]):
        mask = env.get_action_mask()
        if case.get("extras"):
            record_scores()
        if case["policy"] == "fixed_time":
            acts = [int(c.get_action(None)) for c in controllers]
        elif case["policy"] in drivers:
            acts = [int(c.get_action(env.simulator.step_measurements)) for c in controllers]
        else:
            acts = [int(rng.choice(np.nonzero(np.asarray(mask[a]))[0])) for a in range(A)]
        if wrap is not None:
            w_obs, w_rew, done, w_trunc, info = wrap.step(acts)
            assert w_trunc is False and w_obs == env.get_observations()
            greward = w_rew * A  # (common_reward / mean scalarisation: wrappers/epymarl.py:104-111)
            extras["wrap_reward"].append(float(w_rew))
            extras["wrap_done"].append(bool(done))
        else:
            greward, done, info = env.step(acts)
        if case.get("extras"):
            if env.config.misc["return_agent_stats"]:
                extras["opt_agent"].append([[float(info[f"{ts.id}__{k}"]) for k in opt_agent_keys] for ts in ts_list])
            if env.config.misc["return_lane_stats"]:
                extras["opt_lane"].append([[float(info[f"{l}__{k}"]) for k in opt_lane_keys] for l in lanes])
            if s % 6 == 0:
                extras["dm_steps"].append(s)
                extras["density_map"].append(np.asarray(env.metrics.density_map, np.float64))
                extras["mst"].append(np.asarray(env.metrics.mst, np.float64))
        rec["actions"].append(acts)
        rec["obs"].append(env.get_observations())
        rec["state"].append(env.get_state())
        rec["mask"].append(env.get_action_mask())
        rec["local_reward"].append(env.get_rewards())
        rec["global_reward"].append(float(greward))
        rec["done"].append(bool(done))
        rec["terminated"].append(bool(env.simulator.is_terminated))
        rec["info"].append([float(info[k]) for k in info_keys])
        rec["phase"].append([[ts.controller.current_phase_index, ts.controller.time_on_phase,
                              int(ts.controller.program.phase_changed)] for ts in ts_list])
        n, q, ms, oc = lane_arrays()
        rec["lane_n"].append(n)
        rec["lane_q"].append(q)
        rec["lane_mean_speed"].append(ms)
        rec["lane_occ"].append(oc)
        rec["agent_stats"].append(agent_stats())
        if s % 12 == 5:
            pm_steps.append(s)
            pm_vals.append(posmats())
        if done:
            if wrap is not None:
                r_obs, r_state = wrap.reset()
                extras["wrap_reset_obs"].append(np.asarray(r_obs, np.float64).astype(np.int8))
            else:
                env.restart()
    out_dir = os.path.join(HERE, name)
    os.makedirs(out_dir, exist_ok=True)
    inputs = {
        "case": name, "steps": case["steps"],
        "roadnet": {"grid": list(case["roadnet"][1:])} if case["roadnet"][0] == "grid" else {"net": net},
        "flows": compact_flows(entries),
        "signal": case.get("signal", {}), "misc": case.get("misc", {}),
        "network": case.get("network", {}), "backend": case.get("cityflow", {}),
        "info_keys": list(info_keys),
    }
    with gzip.open(os.path.join(out_dir, "inputs.json.gz"), "wt") as f:
        json.dump(inputs, f)
    mxpm = max(p.shape[1] for p in pm_vals)
    pm_arr = np.full((len(pm_vals), len(lanes), mxpm), np.nan)
    for i, p in enumerate(pm_vals):
        pm_arr[i, :, : p.shape[1]] = p
    np.savez_compressed(
        os.path.join(out_dir, "expected.npz"),
        obs0=obs0.astype(np.int8) if np.all(obs0 == np.round(obs0)) else obs0,
        state0=state0, mask0=mask0.astype(np.int8),
        actions=np.asarray(rec["actions"], np.int32),
        obs=np.asarray(rec["obs"], np.float64).astype(np.int8),
        obs_is_int=np.asarray(np.all(np.asarray(rec["obs"], np.float64) == np.round(np.asarray(rec["obs"], np.float64)))),
        state=np.asarray(rec["state"], np.float64).astype(np.int16),
        mask=np.asarray(rec["mask"], np.int8),
        local_reward=np.asarray(rec["local_reward"], np.float64),
        global_reward=np.asarray(rec["global_reward"], np.float64),
        done=np.asarray(rec["done"]), terminated=np.asarray(rec["terminated"]),
        info=np.asarray(rec["info"], np.float64),
        phase=np.asarray(rec["phase"], np.int32),
        lane_n=np.asarray(rec["lane_n"], np.int32), lane_q=np.asarray(rec["lane_q"], np.int32),
        lane_mean_speed=np.asarray(rec["lane_mean_speed"], np.float64),
        lane_occ=np.asarray(rec["lane_occ"], np.float64),
        agent_stats=np.asarray(rec["agent_stats"], np.float64),
        posmat_steps=np.asarray(pm_steps, np.int32), posmat=pm_arr,
        **{"x_" + k: np.asarray(v) for k, v in extras.items() if len(v)},
    )
    pn = env.parsed_network
    parser = {
        "agent_order": list(env.traffic_signals.keys()),
        "traffic_signal_ids": pn.traffic_signal_ids,
        "lanes": pn.lanes,
        "lane_lengths": pn.lane_lengths, "lane_max_speeds": pn.lane_max_speeds,
        "lane_indices": pn.lane_indices, "lane_angles": pn.lane_angles,
        "adjacency_matrix": np.asarray(pn.adjacency_matrix).tolist(),
        "k_hop_neighbors": {t: {str(k): v for k, v in d.items()} for t, d in pn.k_hop_neighbors.items()},
        "neighbors_lanes": pn.neighbors_lanes,
        "signals": {
            t: {k: cfg[k] for k in ("incoming_lanes", "outgoing_lanes", "phases", "n_phases",
                                    "phase_indices", "green_phase_indices", "yellow_phase_indices",
                                    "coordinates", "norm_coordinates", "inc_to_out_lanes")}
            for t, cfg in pn.traffic_signals.items()
        },
        "phases_min_max_times": {t: {str(k): v for k, v in cfg["phases_min_max_times"].items()}
                                 for t, cfg in pn.traffic_signals.items()},
        "phase_to_inc_out_lanes": {t: {str(k): v for k, v in cfg["phase_to_inc_out_lanes"].items()}
                                   for t, cfg in pn.traffic_signals.items()},
        "lane_features": env.observation_space.lane_features,
        "obs_size": env.get_observation_size(), "state_size": env.get_state_size(),
        "action_size": env.get_action_size(), "episode_limit": env.episode_limit,
    }
    with open(os.path.join(out_dir, "parser.json"), "w") as f:
        json.dump(parser, f, default=float)
    shutil.rmtree(root, ignore_errors=True)
    print(name, "ok:", {k: np.asarray(v).shape for k, v in rec.items() if k in ("obs", "info")},
          "n_queued last", rec["info"][-1][2])


if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    for n in names:
        generate(n)
