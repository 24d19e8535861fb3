"""Batched path and the scenario's observation space (reference common/observations.py:225-382, 72-88): lane_features
observations with per-replica lane dropout on the device; observation noise is refused loudly (it is served by the
single-env facade, with numpy-stream parity)."""
import numpy as np
import pytest


def test_batched_env_refuses_observation_noise():
    from pytsc_amd.vec_env import VecTrafficSignalNetwork

    with pytest.raises(ValueError, match="obs_noise_std"):
        VecTrafficSignalNetwork("syn_2x2", n_replicas=4, signal={"obs_noise_std": 0.3})
    with pytest.raises(ValueError, match="obs_dropout_prob"):
        VecTrafficSignalNetwork("syn_2x2", n_replicas=4, signal={"obs_dropout_prob": 0.3})
    with pytest.raises(ValueError, match="observation_space"):
        VecTrafficSignalNetwork("syn_2x2", n_replicas=4, signal={"observation_space": "nope"})


@pytest.mark.gpu
def test_batched_lane_features_observations_and_dropout(oracle_lib):
    """observation_space: lane_features -> [R, A, 212] rows == the reference's LaneFeatures observation (which the oracle
    produces as the state rows; pinned by the golden case syn_2x2_lane_dropout on the facade), with the three measured
    entries of the dropped lanes zeroed: exactly int(p * n_lanes) lanes per replica, different ones in different replicas."""
    import torch

    from pytsc_amd.config import Config
    from pytsc_amd.scenario import build_tables
    from pytsc_amd.vec_env import VecTrafficSignalNetwork

    R, p = 8, 0.25
    kw = dict(signal={"observation_space": "lane_features", "obs_dropout_prob": p, "action_space": "phase_selection",
                      "round_robin": False}, hip={"flow_rate": 1200.0, "n_flow_sets": 1})
    env = VecTrafficSignalNetwork("syn_2x2", n_replicas=R, **kw)
    orc = oracle_lib.Oracle(build_tables(Config("syn_2x2", **kw), n_replicas=1))
    assert env.get_observation_size() == 212
    rng = np.random.default_rng(0)
    out = orc.observe()
    for s in range(40):
        m = out.mask[0]
        acts = (rng.random(m.shape) * m).argmax(-1).astype(np.int32)
        out = orc.step(acts[None])
        env.step(torch.tensor(np.repeat(acts[None], R, 0), device="cuda"))
    obs = env.get_observations().cpu().numpy()
    full = out.state[0]  # [A, 212]
    nl = env.tables.n_lanes
    n_drop = int(p * nl)
    dropped = env._dropped[:, :nl].cpu().numpy()
    assert (dropped.sum(1) == n_drop).all() and len({tuple(r) for r in dropped.tolist()}) > 1
    slots = env._lane_slot.cpu().numpy()
    some_nonzero_dropped = False
    for r in range(R):
        exp = full.copy().reshape(full.shape[0], -1)
        for a in range(exp.shape[0]):
            for k in range(16):
                l = slots[a, k]
                if l >= 0 and dropped[r, l]:
                    some_nonzero_dropped |= bool(np.any(exp[a, k * 12 + 9: k * 12 + 12] != 0))
                    exp[a, k * 12 + 9: k * 12 + 12] = 0
        assert np.array_equal(obs[r], exp), r
    assert some_nonzero_dropped  # the dropout did hide something
    env.close()
