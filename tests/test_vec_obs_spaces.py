"""Batched path and the scenario's observation space (reference common/observations.py:225-382, 72-88): lane_features
observations with per-replica lane dropout on the device; observation noise drawn on the device where kr_step assembles
the rows (the law of the reference's noise; numpy-stream parity stays with the single-env facade)."""
import math

import numpy as np
import pytest


def test_batched_env_refuses_inconsistent_observation_options():
    from pytsc_amd.vec_env import VecTrafficSignalNetwork

    with pytest.raises(ValueError, match="obs_noise_std"):
        VecTrafficSignalNetwork("syn_2x2", n_replicas=4, signal={"obs_noise_std": 0.3, "observation_space": "lane_features"})
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


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["resident", "list"])
def test_device_side_observation_noise_follows_the_reference_law(family):
    """obs_noise_std on the batched path (reference PositionMatrix._add_gaussian_noise, common/observations.py:72-88, then the
    integer truncation of the padded row).  4096 replicas of ONE demand under one action stream are in the
    same state at every step and differ in their noise draws only, so for every (signal, lane, surviving bin):
      * the row layout -- which entries survive, where every block starts -- is that of the noise-free row (Q2: the
        compaction depends on the values before the noise, one draw per surviving entry);
      * the entry is 1 in a share of the replicas that matches P(val + N(0, std) >= 1) = 1 - Phi((1 - val) / std) for the
        bin's value val from the full position matrices (tsc_lane_detail) -- values from 0.02 to 3 pin mean 0 and the
        standard deviation of the draws;
      * draws of different replicas / entries are independent (no pair of replicas has identical noisy rows over a run);
      * std 0 is today's rows bit for bit."""
    import torch

    from pytsc_amd.vec_env import VecTrafficSignalNetwork

    R, std = 4096, 0.35
    # (round 5: on the list family too -- k_measure keeps the bin values, k_agent draws)
    kw = dict(hip={"flow_rate": 1000.0, "n_flow_sets": 1, "seed": 11, "kernel_family": family},
              signal={"action_space": "phase_selection", "round_robin": False})
    clean = VecTrafficSignalNetwork("syn_2x2", n_replicas=R, **kw)
    kwn = dict(kw, signal=dict(kw["signal"], obs_noise_std=std))
    noisy = VecTrafficSignalNetwork("syn_2x2", n_replicas=R, **kwn)
    zero = VecTrafficSignalNetwork("syn_2x2", n_replicas=8, **dict(kw, signal=dict(kw["signal"], obs_noise_std=0.0)))
    A, vis = clean.n_agents, 10
    rn, ts = clean.tables.rn, clean.parsed_network.traffic_signals
    g = torch.Generator().manual_seed(1)
    z_tot = n_tot = 0
    checked_vals = []
    for s in range(60):
        m = clean.get_action_mask()[0:1]
        acts1 = (torch.rand(m.shape, generator=g).to(m.device) * m).argmax(-1).to(torch.int32)
        acts = acts1.expand(R, A).contiguous()
        clean.step(acts)
        noisy.step(acts)
        zero.step(acts[:8].contiguous())
        assert torch.equal(zero.get_observations(), clean.get_observations()[:8])  # std 0: bit for bit
        oc, on = clean.get_observations(), noisy.get_observations()
        assert torch.equal(oc, oc[0:1].expand_as(oc))  # (every replica in the same state)
        assert torch.equal(noisy.buf["stats"], clean.buf["stats"])  # the noise touches observations only
        diff = oc != on
        # entries differ only where the clean row holds a surviving bin (0 or 1 behind the 9 static entries of a block)
        assert bool(((oc[diff] == 0) | (oc[diff] == 1)).all())
        assert bool(((on == 0) | (on == 1) | (on == -1) | (on == oc)).all())
        if s < 20 or s % 4:
            continue
        _, _, pm = clean.engine.lane_detail(with_posmat=True)
        pm0 = pm[0].cpu().numpy()
        on_h, oc0 = on.cpu().numpy(), oc[0].cpu().numpy()
        for a, ts_id in enumerate(rn.agent_ids):
            w = 0
            for lane in ts[ts_id]["incoming_lanes"]:
                li = rn.lane_index[lane]
                bins = int(clean.tables.lane_bins[li])
                row = pm0[li][: max(bins, vis)][-vis:]
                w += 9
                for val in row[row > 0]:
                    p = 1.0 - 0.5 * (1.0 + math.erf((1.0 - val) / std / math.sqrt(2.0)))
                    ones = float((on_h[:, a, w] == 1).mean())
                    assert oc0[a, w] == (1 if val >= 1 else 0)
                    sd = math.sqrt(max(p * (1 - p), 1e-9) / R)
                    z = (ones - p) / max(sd, 1e-4)
                    assert abs(z) < 5.0, (s, a, lane, val, p, ones)
                    if 0.02 < p < 0.98:
                        z_tot += z * z
                        n_tot += 1
                        checked_vals.append(float(val))
                    w += 1
    assert n_tot > 80 and 0.6 < z_tot / n_tot < 1.5, (n_tot, z_tot / n_tot)  # z-scores ~ N(0, 1): the tail law holds throughout
    assert min(checked_vals) < 0.5 and max(checked_vals) >= 1.0  # (bins of several vehicles, val > 1, end as 1 almost surely)
    rows = noisy.get_observations().reshape(R, -1)
    assert len(torch.unique(rows, dim=0)) > R // 2  # replicas draw independently
    assert noisy.engine.errors() == 0 and noisy.engine.family == family
    for e in (clean, noisy, zero):
        e.close()
