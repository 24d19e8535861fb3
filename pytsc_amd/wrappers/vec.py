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
