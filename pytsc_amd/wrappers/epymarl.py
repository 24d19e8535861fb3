"""SMAC-style MultiAgentEnv adapter (reference: pytsc/wrappers/epymarl.py:11-111):
same method names, argument meaning and return tuples."""
from pytsc_amd import TrafficSignalNetwork

try:  # smac is optional, exactly one attribute of it is used
    from smac.env import MultiAgentEnv
except Exception:  # pragma: no cover
    class MultiAgentEnv:
        pass


class EPyMARLTrafficSignalNetwork(MultiAgentEnv):
    step_stats = None

    def __init__(self, map_name="syn_2x2", simulator_backend="hip", **kwargs):
        kwargs.pop("scenario", None)
        self.common_reward = kwargs.pop("common_reward", True)
        self.reward_scalarization = kwargs.pop("reward_scalarization", "mean")
        self.tsc_env = TrafficSignalNetwork(map_name, simulator_backend=simulator_backend, **kwargs)
        self.episode_limit = self.tsc_env.episode_limit

    def apply_actions(self, actions):
        self.tsc_env.action_space.apply(actions)

    def close(self):
        self.tsc_env.close()

    def get_avail_actions(self):
        return self.tsc_env.get_action_mask()

    def get_env_info(self):
        return {
            "agents": list(self.tsc_env.traffic_signals.keys()),
            "episode_limit": self.episode_limit,
            "n_actions": self.get_total_actions(),
            "adjacency_matrix": self.tsc_env.parsed_network.adjacency_matrix,
            "n_agents": self.tsc_env.n_agents,
            "obs_shape": self.get_obs_size(),
            "obs_info": self.tsc_env.observation_space.get_observation_info(),
            "state_shape": self.get_state_size(),
            "lane_features": self.tsc_env.observation_space.get_per_agent_lane_features(),
        }

    def get_obs(self):
        return self.tsc_env.get_observations()

    def get_obs_size(self):
        return self.tsc_env.get_observation_size()

    def get_network_flow(self):
        return self.tsc_env.metrics.network_flow

    def get_pressures(self):
        return self.tsc_env.metrics.pressures

    def get_state(self):
        return self.tsc_env.get_state()

    def get_local_rewards(self):
        return self.tsc_env.get_rewards()

    def get_state_size(self):
        return self.tsc_env.get_state_size()

    def get_stats(self):
        return self.tsc_env.get_env_stats()

    def get_total_actions(self):
        return self.tsc_env.get_action_size()

    def is_terminated(self):
        return self.tsc_env.simulator.is_terminated

    def sim_step(self):
        return self.tsc_env.simulator.sim_step

    def reset(self):
        self.tsc_env.episode_count += 1
        obs, state = self.get_obs(), self.get_state()
        if self.tsc_env.episode_over:
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
