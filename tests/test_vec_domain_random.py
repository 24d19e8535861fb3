"""GPU: the batched forms of the reference's two domain-randomisation devices -- `disrupted=True` (a flow file drawn per
replica and restart: pytsc/backends/cityflow/config.py:106-175, pytsc/__init__.py:26-30,153-155) and the multi-map wrapper
(pytsc/wrappers/epymarl.py:112-336: a map drawn per slot at every simulated-hour restart, outputs padded to the largest
map) -- against the oracle / a single-map environment and the reference wrapper's padding conventions."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_batched_disrupted_config_runs_each_replica_on_its_drawn_file(oracle_lib, tmp_path):
    import torch

    from test_disrupted_config import _write_scenario

    from pytsc_amd.vec_env import VecTrafficSignalNetwork

    section = _write_scenario(tmp_path)
    R = 24
    env = VecTrafficSignalNetwork(
        "syn_2x2", n_replicas=R, disrupted=True, mode="train", scenario_dir=str(tmp_path),
        cityflow=dict(section, seed=4, sim_length=60, episode_limit=30),
        signal={"action_space": "phase_selection", "round_robin": False},
        outputs=("obs", "mask", "stats", "global_reward", "done", "terminated", "valid"))
    cfg = env.config
    owner = np.asarray(cfg.file_domain_class)
    orc = oracle_lib.Oracle(env.tables)
    orc.reset(None, env.current_sets.numpy())
    out = orc.observe()
    assert np.array_equal(env.get_observations().cpu().numpy(), out.obs)
    rng = np.random.default_rng(3)
    seen = set()
    for hour in range(3):
        sets = env.current_sets.numpy()
        info = env.get_env_info()
        assert info["n_domains"] == 3
        assert np.array_equal(info["domain_class"].cpu().numpy(), owner[sets])  # the class of the file it runs
        seen |= set(sets.tolist())
        for s in range(12):
            m = out.mask
            acts = (rng.random(m.shape) * m).argmax(-1).astype(np.int32)
            out = orc.step(acts)
            env.step(torch.tensor(acts, device="cuda"))
            assert np.array_equal(env.get_observations().cpu().numpy(), out.obs), (hour, s)
            assert np.array_equal(env.buf["stats"].cpu().numpy(), out.stats), (hour, s)
        assert bool(env.buf["terminated"].all())
        env.restart()  # every replica's hour is over: each draws anew
        orc.reset(None, env.current_sets.numpy())
        out = orc.observe()
        assert np.array_equal(env.get_observations().cpu().numpy(), out.obs)
    assert len(seen) >= 3  # different replicas ran different files
    assert env.engine.errors() == 0
    env.close()


@pytest.mark.parametrize("list_map", [None, "syn_1x3"])
def test_batched_multi_map_wrapper_pads_like_the_reference_and_redraws_at_the_hour(oracle_lib, list_map):
    """list_map: that map's handle runs on the LIST kernel family (round 5: tsc_park there too -- a large map such as
    manhattan_16_3, which only that family holds, can be one of the maps)."""
    import torch

    from pytsc_amd.vec_env import VecTrafficSignalNetwork
    from pytsc_amd.wrappers.vec import VecDomainRandomizedEPyMARLTrafficSignalNetwork

    R = 16
    kw = dict(hip={"sim_length": 60, "episode_limit": 30, "flow_rate": 900.0, "n_flow_sets": 2},
              signal={"action_space": "phase_selection", "round_robin": False})
    maps = ["syn_2x2", "syn_1x3"]
    mk = {list_map: {"hip": {"kernel_family": "list"}}} if list_map else None
    env = VecDomainRandomizedEPyMARLTrafficSignalNetwork(maps, n_replicas=R, seed=2, map_kwargs=mk, **kw)
    if list_map:
        assert env.envs[maps.index(list_map)].engine.family == "list"
    singles = [VecTrafficSignalNetwork(m, n_replicas=R, **kw) for m in maps]
    nA = [e.n_agents for e in singles]
    assert env.max_n_agents == max(nA) == 4 and min(nA) == 3
    info = env.get_env_info()
    assert info["n_agents"] == 4 and info["adjacency_matrices"].shape == (2, 4, 4)
    assert np.all(info["adjacency_matrices"][1][3] == 0) and np.all(info["adjacency_matrices"][1][:, 3] == 0)
    rng = torch.Generator().manual_seed(7)
    first_map = env.slot_map.clone()
    assert set(first_map.tolist()) == {0, 1}
    for hour in range(3):
        smap = env.slot_map.cpu().numpy()
        for s in range(12):
            avail = env.get_avail_actions()
            u = torch.rand(avail.shape, generator=rng).to(avail.device)
            acts = (u * avail).argmax(-1).to(torch.int32)
            obs, reward, over, _, inf = env.step(acts)
            state, local = env.get_state(), env.get_local_rewards()
            avail2 = env.get_avail_actions()
            for k in range(2):
                sel = torch.tensor(smap == k, device=obs.device)
                if hour == 0:  # the first hour: the very trajectories of a single-map environment under the same actions
                    singles[k].step(acts[:, : nA[k]].contiguous())
                    assert torch.equal(obs[sel][:, : nA[k]], singles[k].get_observations()[sel]), (s, k)
                    assert torch.equal(local[sel][:, : nA[k]], singles[k].get_rewards()[sel])
                    # (mean scalarisation: a float32 division either way, by a tensor here and by a Python scalar there --
                    #  torch multiplies by the reciprocal in the latter form: one unit in the last place apart at most)
                    # (... and the reference's multi-map wrapper divides twice: wrappers/epymarl.py:104-107,326-329)
                    assert torch.allclose(reward[sel], (singles[k].get_reward() / nA[k] / nA[k])[sel], rtol=5e-7, atol=0)
                    assert torch.equal(inf["time_step"][sel], singles[k].buf["stats"][:, 0][sel])
                # the reference's padding conventions (wrappers/epymarl.py:230-300)
                assert bool((obs[sel][:, nA[k]:] == -1).all()) and bool((state[sel][:, nA[k]:] == -1).all())
                assert bool((avail2[sel][:, nA[k]:] == 1).all()) and bool((local[sel][:, nA[k]:] == 0).all())
                # parked replicas are skipped, the playing ones are served
                v = env.envs[k].buf["valid"]
                assert bool((v[sel] == 1).all()) and bool((v[~sel] == 0).all())
            assert bool((inf["time_step"] == 5.0 * (s + 1)).all())
        assert bool(over.all()) and bool(env.is_terminated().all())
        obs0, _ = env.reset()
        assert torch.equal(obs0, obs)  # (the observation taken before the restart)
        assert bool((env.envs[0].buf["stats"][:, 0][env.slot_map == 0] == 0).all())  # restarted: empty network, t = 0
    assert not torch.equal(env.slot_map, first_map)  # slots did draw other maps
    assert all(e.engine.errors() == 0 for e in env.envs)
    env.close()
    for e in singles:
        e.close()
