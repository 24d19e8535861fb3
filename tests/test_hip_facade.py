"""GPU twins of tests/test_facade.py: the façade and the batched API on the real
HIP engine (default engine factory = the C-ABI library)."""
import numpy as np
import pytest

from fixtures import load_case
from test_facade import make_env, run_and_compare

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["syn_2x2_fixed_time", "syn_1x3_switch_rr"])
def test_facade_on_hip_engine(case):
    env, inputs = make_env(case, engine_factory=None)
    from pytsc_amd.vec_env import VecEngine

    assert isinstance(env.simulator.engine, VecEngine)
    _, exp, ref, _, _ = load_case(case)
    run_and_compare(env, inputs, exp, 40)
    lanes = env.simulator.step_measurements["lane"]
    assert list(lanes.keys()) == env.parsed_network.lanes
    env.close()


def test_vec_env_batched_api_and_partial_restart():
    import torch

    from pytsc_amd.vec_env import VecTrafficSignalNetwork

    R = 130  # not a multiple of 64
    env = VecTrafficSignalNetwork("syn_2x2", n_replicas=R, signal={"action_space": "phase_selection",
                                                                   "round_robin": False},
                                  hip={"flow_rate": 800.0, "n_flow_sets": 2, "sim_length": 300,
                                       "episode_limit": 100})
    assert env.get_observations().shape == (R, 4, 324) and env.get_action_mask().shape == (R, 4, 16)
    g = torch.Generator(device="cuda")
    g.manual_seed(0)
    for s in range(60):
        a = env.random_valid_actions(g)
        assert bool((env.get_action_mask().gather(2, a.long()[..., None]) == 1).all())
        rew, done, info = env.step(a)
        assert bool(done.all()) == ((s + 1) % 20 == 0)
    assert bool(env.buf["terminated"].all())  # 60 steps * 5 s = sim_length
    n_before = info["n_queued"].clone()
    assert float(n_before.sum()) > 0
    # restart only the odd replicas
    mask = torch.zeros(R, dtype=torch.uint8, device="cuda")
    mask[1::2] = 1
    env.engine.reset(mask)
    buf = env.engine.observe()
    st = buf["stats"].cpu().numpy()
    assert np.all(st[1::2, 0] == 0.0) and np.all(st[0::2, 0] == 300.0)  # time rewound only where asked
    assert np.all(st[1::2, 2] == 0)
    # replicas with the same flow set stay bit-identical
    assert torch.equal(buf["obs"][1], buf["obs"][3])
    assert env.engine.errors() == 0
    env.close()


def test_vec_smac_wrapper():
    import torch

    from pytsc_amd.wrappers.vec import VecEPyMARLTrafficSignalNetwork

    env = VecEPyMARLTrafficSignalNetwork("syn_2x2", n_replicas=64, signal={"action_space": "phase_selection",
                                                                           "round_robin": False})
    info = env.get_env_info()
    assert info["n_agents"] == 4 and info["n_actions"] == 16 and info["obs_shape"] == 324
    obs, state = env.reset()
    assert obs.shape == (64, 4, 324) and state.shape == (64, 4, 212)
    a = env.tsc_env.random_valid_actions()
    obs, reward, done, truncated, step_info = env.step(a)
    assert reward.shape == (64,) and done.shape == (64,) and truncated is False
    assert torch.allclose(reward * 4, env.tsc_env.get_reward())
    assert set(step_info) >= {"time_step", "n_queued", "episode_limit"}
    env.close()
