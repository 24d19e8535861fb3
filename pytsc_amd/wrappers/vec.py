"""Batched SMAC-style adapter: the method set of
pytsc/wrappers/epymarl.py:11-111 with a leading replica axis and torch tensors
that stay in HBM.  Reward scalarisation, `reset` without a simulator reset and the
restart of finished simulated hours follow the reference wrapper."""
import torch

from pytsc_amd.vec_env import VecTrafficSignalNetwork


class VecEPyMARLTrafficSignalNetwork:
    def __init__(self, map_name="syn_4x4", simulator_backend="hip", n_replicas=1024, device=0,
                 common_reward=True, reward_scalarization="mean", **kwargs):
        kwargs.pop("scenario", None)
        self.tsc_env = VecTrafficSignalNetwork(map_name, simulator_backend, n_replicas=n_replicas, device=device,
                                               **kwargs)
        self.n_replicas = n_replicas
        self.episode_limit = self.tsc_env.episode_limit
        self.common_reward = common_reward
        self.reward_scalarization = reward_scalarization
        self.episode_count = torch.zeros(n_replicas, dtype=torch.int64, device=self.tsc_env.engine.device)

    def close(self):
        self.tsc_env.close()

    def get_avail_actions(self):
        return self.tsc_env.get_action_mask()

    def get_env_info(self):
        obs_space = {"lane_obs_dim": 9 + int(self.tsc_env.config.signal["visibility"]),
                     "max_n_controlled_lanes": 16, "max_phases": 20}
        return {
            "agents": list(self.tsc_env.agent_ids),
            "episode_limit": self.episode_limit,
            "n_actions": self.get_total_actions(),
            "adjacency_matrix": self.tsc_env.parsed_network.adjacency_matrix,
            "n_agents": self.tsc_env.n_agents,
            "n_replicas": self.n_replicas,
            "obs_shape": self.get_obs_size(),
            "obs_info": obs_space,
            "state_shape": self.get_state_size(),
        }

    def get_obs(self):
        return self.tsc_env.get_observations()

    def get_obs_size(self):
        return self.tsc_env.get_observation_size()

    def get_state(self):
        return self.tsc_env.get_state()

    def get_state_size(self):
        return self.tsc_env.get_state_size()

    def get_local_rewards(self):
        return self.tsc_env.get_rewards()

    def get_total_actions(self):
        return self.tsc_env.get_action_size()

    def is_terminated(self):
        return self.tsc_env.buf["terminated"]

    def reset(self):
        """As the reference wrapper: counts an episode, returns the current obs/state,
        restarts only the replicas whose simulated hour is over."""
        self.episode_count += 1
        # the reference returns the observation taken BEFORE the restart (wrappers/epymarl.py:97-102);
        # restart() measures again into the same device buffers, hence the copies
        obs, state = self.get_obs().clone(), self.get_state().clone()
        self.tsc_env.restart()
        return obs, state

    def step(self, actions):
        reward, episode_over, env_info = self.tsc_env.step(actions)
        if self.common_reward:
            if self.reward_scalarization == "mean":
                reward = reward / self.tsc_env.n_agents
        else:
            reward = self.get_local_rewards()
        return self.get_obs(), reward, episode_over, False, env_info


class VecDomainRandomizedEPyMARLTrafficSignalNetwork:
    """Batched twin of the reference's domain-randomised wrapper (pytsc/wrappers/epymarl.py:112-336): R environment slots,
    each playing one of `map_names`; a slot draws a new map whenever its simulated hour is over; every per-agent output is
    padded on the device to the largest map with the reference's conventions -- observation / state rows of -1, action
    masks of all ones, local rewards 0, adjacency rows / columns of 0, `n_agents` = the largest map's; surplus actions are
    ignored.

    One handle of R replicas per map: slot s is replica s of the handle of the map it plays and is PARKED (tsc_park: a
    skipped workgroup) in the others, so a step costs the active replicas plus a launch per map.  The maps must agree in
    observation / state / action sizes (the reference pads agents only)."""

    def __init__(self, map_names, simulator_backend="hip", n_replicas=256, device=0, common_reward=True,
                 reward_scalarization="mean", map_kwargs=None, seed=0, **kwargs):
        kwargs.pop("scenario", None)
        kwargs.pop("map_name", None)
        self.map_names = list(map_names)
        self.n_replicas = n_replicas
        self.common_reward, self.reward_scalarization = common_reward, reward_scalarization
        map_kwargs = map_kwargs or {}
        self.envs = []
        for m in self.map_names:
            kw = dict(kwargs)
            for k, v in map_kwargs.get(m, {}).items():
                kw[k] = {**kw.get(k, {}), **v} if isinstance(v, dict) and isinstance(kw.get(k), dict) else v
            self.envs.append(VecTrafficSignalNetwork(m, simulator_backend, n_replicas=n_replicas, device=device, **kw))
        e0 = self.envs[0]
        self.device = e0.engine.device
        for e in self.envs[1:]:
            if (e.get_observation_size(), e.get_state_size(), e.get_action_size(), e.episode_limit) != (
                    e0.get_observation_size(), e0.get_state_size(), e0.get_action_size(), e0.episode_limit):
                raise ValueError("the maps of a domain-randomised environment must agree in observation, state and action sizes")
        self.max_n_agents = max(e.n_agents for e in self.envs)
        self.episode_limit = e0.episode_limit
        self.episode_count = torch.zeros(n_replicas, dtype=torch.int64, device=self.device)
        self._rng = torch.Generator().manual_seed(int(seed))
        self.slot_map = torch.randint(0, len(self.envs), (n_replicas,), generator=self._rng).to(self.device)
        for k, e in enumerate(self.envs):
            e.engine.park(self.slot_map != k)
            e.buf = e.engine.observe()
        self._n_agents_of = torch.tensor([e.n_agents for e in self.envs], device=self.device)

    # -- padded views --------------------------------------------------------------------------------------------------
    def _padded(self, getter, fill, dtype=None):
        out = None
        for k, e in enumerate(self.envs):
            x = getter(e)
            if out is None:
                out = torch.full((self.n_replicas, self.max_n_agents) + tuple(x.shape[2:]), fill, dtype=dtype or x.dtype,
                                 device=self.device)
            sel = self.slot_map == k
            out[:, : e.n_agents][sel] = x[sel].to(out.dtype)
        return out

    def get_obs(self):
        return self._padded(lambda e: e.get_observations(), -1.0)

    def get_state(self):
        return self._padded(lambda e: e.get_state(), -1.0)

    def get_avail_actions(self):
        return self._padded(lambda e: e.get_action_mask(), 1)

    def get_local_rewards(self):
        return self._padded(lambda e: e.get_rewards(), 0.0)

    def get_obs_size(self):
        return self.envs[0].get_observation_size()

    def get_state_size(self):
        return self.envs[0].get_state_size()

    def get_total_actions(self):
        return self.envs[0].get_action_size()

    def current_n_agents(self):
        """[R] agents of the map each slot plays."""
        return self._n_agents_of[self.slot_map]

    def _per_slot(self, getter):
        out = None
        for k, e in enumerate(self.envs):
            x = getter(e)
            sel = (self.slot_map == k).view((-1,) + (1,) * (x.dim() - 1))
            out = x.clone() if out is None else torch.where(sel, x, out)
        return out

    def is_terminated(self):
        return self._per_slot(lambda e: e.buf["terminated"])

    def get_env_info(self):
        """The current map's info with the adjacency matrix padded: here per map (`adjacency_matrices` [K, max, max]) with
        the slots' map indices beside it; `n_agents` = the largest map's."""
        import numpy as np

        n = self.max_n_agents
        adj = np.zeros((len(self.envs), n, n), dtype=np.float32)
        for k, e in enumerate(self.envs):
            a = np.asarray(e.parsed_network.adjacency_matrix, dtype=np.float32)
            adj[k, : a.shape[0], : a.shape[1]] = a
        return {"agents_per_map": [list(e.agent_ids) for e in self.envs], "episode_limit": self.episode_limit,
                "n_actions": self.get_total_actions(), "adjacency_matrices": adj, "slot_map": self.slot_map,
                "n_agents": n, "n_replicas": self.n_replicas, "obs_shape": self.get_obs_size(),
                "state_shape": self.get_state_size(), "map_names": list(self.map_names)}

    # -- episode control -----------------------------------------------------------------------------------------------
    def step(self, actions):
        """actions [R, max_n_agents] int32 (entries beyond a slot's agents are ignored) ->
        (obs [R, max, obs], reward [R] or [R, max], episode_over [R], False, info dict of [R] tensors)."""
        for k, e in enumerate(self.envs):
            e.buf = e.engine.step(actions[:, : e.n_agents].contiguous())
        reward = self._per_slot(lambda e: e.buf["global_reward"])
        if not self.common_reward:
            reward = self.get_local_rewards()
        elif self.reward_scalarization == "mean":
            # (twice: the reference's multi-map wrapper divides the already mean-scalarised reward of the map's own wrapper
            #  again -- wrappers/epymarl.py:104-107 and :326-329)
            n = self.current_n_agents().to(reward.dtype)
            reward = reward / n / n
        over = self._per_slot(lambda e: e.buf["done"])
        stats = self._per_slot(lambda e: e.buf["stats"])
        from pytsc_amd.vec_env import STAT_KEYS

        info = {key: stats[:, i] for i, key in enumerate(STAT_KEYS)}
        info["episode_limit"] = self.episode_limit
        info["map"] = self.slot_map
        # 1: the slot's rows are those of this step; 0: its replica has run out of vehicle records in the map it plays (a
        # handle with parked replicas has no stand-by: include/pytsc_amd.h, tsc_park) -- mask it out
        info["valid"] = self._per_slot(lambda e: e.buf["valid"]) if all("valid" in e.buf for e in self.envs) else None
        return self.get_obs(), reward, over, False, info

    def reset(self):
        """As the reference's reset (wrappers/epymarl.py:308-320): counts an episode, returns the observation taken BEFORE any
        restart; a slot whose simulated hour is over draws a map and starts it from an empty network (its replica in the
        map it leaves is parked)."""
        self.episode_count += 1
        obs, state = self.get_obs(), self.get_state()
        over = self.is_terminated().to(torch.bool)
        if bool(over.any()):  # (one host synchronisation per episode end, as VecEngine.reset of a stand-by handle has anyway)
            draw = torch.randint(0, len(self.envs), (self.n_replicas,), generator=self._rng).to(self.device)
            new_map = torch.where(over, draw, self.slot_map)
            for k, e in enumerate(self.envs):
                leaving = over & (self.slot_map == k) & (new_map != k)
                starting = over & (new_map == k)
                if bool(starting.any()):
                    e.engine.reset(starting.to(torch.uint8))
                if bool(leaving.any()):
                    e.engine.park(leaving)
                e.buf = e.engine.observe()
            self.slot_map = new_map
        return obs, state

    def close(self):
        for e in self.envs:
            e.close()
