"""SMAC-style MultiAgentEnv adapter (reference: pytsc/wrappers/epymarl.py:11-111):
same method names, argument meaning and return tuples."""
from pytsc_amd import TrafficSignalNetwork

try:  # smac is optional, exactly one attribute of it is used
    from smac.env import MultiAgentEnv
except Exception:  # pragma: no cover
    class MultiAgentEnv:
        pass


def _delegate(path, call=True):
    """A method that forwards to `self.tsc_env.<path>` (called, or read as an attribute)."""
    parts = path.split(".")

    def method(self):
        obj = self.tsc_env
        for p in parts:
            obj = getattr(obj, p)
        return obj() if call else obj

    method.__doc__ = f"tsc_env.{path}{'()' if call else ''}"
    return method


class EPyMARLTrafficSignalNetwork(MultiAgentEnv):
    """The SMAC / EPyMARL method set over one TrafficSignalNetwork.  Most of it is a one-to-one renaming of the env's own
    getters (the table below); what has behaviour of its own -- reward scalarisation, `reset` without a simulator reset --
    is written out."""

    step_stats = None
    # wrapper method -> what it returns of the env (reference wrappers/epymarl.py:35-103)
    _FORWARDED = {
        "close": ("close", True), "get_avail_actions": ("get_action_mask", True), "get_obs": ("get_observations", True),
        "get_obs_size": ("get_observation_size", True), "get_state": ("get_state", True),
        "get_state_size": ("get_state_size", True), "get_local_rewards": ("get_rewards", True),
        "get_stats": ("get_env_stats", True), "get_total_actions": ("get_action_size", True),
        "get_network_flow": ("metrics.network_flow", False), "get_pressures": ("metrics.pressures", False),
        "is_terminated": ("simulator.is_terminated", False), "sim_step": ("simulator.sim_step", False),
    }

    def __init__(self, map_name="syn_2x2", simulator_backend="hip", **kwargs):
        kwargs.pop("scenario", None)
        self.common_reward = kwargs.pop("common_reward", True)
        self.reward_scalarization = kwargs.pop("reward_scalarization", "mean")
        self.tsc_env = TrafficSignalNetwork(map_name, simulator_backend=simulator_backend, **kwargs)
        self.episode_limit = self.tsc_env.episode_limit

    def apply_actions(self, actions):
        self.tsc_env.action_space.apply(actions)

    def get_domain_class(self):
        return self.tsc_env.config.domain_class

    def set_domain_class(self, domain_class):
        self.tsc_env.config.set_domain_class(domain_class)

    def get_env_info(self):
        env, space = self.tsc_env, self.tsc_env.observation_space
        return dict(
            agents=list(env.traffic_signals.keys()), episode_limit=self.episode_limit, n_actions=self.get_total_actions(),
            adjacency_matrix=env.parsed_network.adjacency_matrix, n_agents=env.n_agents, obs_shape=self.get_obs_size(),
            obs_info=space.get_observation_info(), state_shape=self.get_state_size(),
            lane_features=space.get_per_agent_lane_features(),
        )

    def reset(self):
        """SMAC's reset: a new episode of the SAME simulated hour -- the simulator restarts only when its hour is over."""
        env = self.tsc_env
        env.episode_count += 1
        snapshot = (self.get_obs(), self.get_state())
        if env.episode_over:
            env.restart()
        return snapshot

    def step(self, actions):
        env = self.tsc_env
        global_reward, episode_over, info = env.step(actions)
        if not self.common_reward:
            reward = self.get_local_rewards()
        elif self.reward_scalarization == "mean":
            reward = global_reward / env.n_agents
        else:
            reward = global_reward
        return self.get_obs(), reward, episode_over, False, info


for _name, (_path, _call) in EPyMARLTrafficSignalNetwork._FORWARDED.items():
    setattr(EPyMARLTrafficSignalNetwork, _name, _delegate(_path, _call))


def _pad_rows(rows, n_rows, fill_row):
    """rows + copies of fill_row up to n_rows (never truncates)."""
    return list(rows) + [list(fill_row) for _ in range(n_rows - len(rows))]


class DomainRandomizedEPyMARLTrafficSignalNetwork(MultiAgentEnv):
    """Draws one of `map_names` whenever a simulated episode is over and pads every
    per-agent output to the largest map, so that the learner sees a fixed interface
    (reference: pytsc/wrappers/epymarl.py:112-336).  Padding conventions are the
    reference's: observation / state rows of -1, action masks of all ones, local
    rewards 0, adjacency rows / columns of 0, `n_agents` = the largest map's.
    Surplus actions (beyond the current map's agents) are ignored.

    `map_kwargs` (not in the reference) may give per-map keyword arguments — e.g.
    in-memory roadnets — merged over the common `**kwargs`."""

    step_stats = None

    def __init__(self, map_names, simulator_backend="hip", map_kwargs=None, rng=None, **kwargs):
        import random as _random

        kwargs.pop("scenario", None)
        kwargs.pop("map_name", None)
        self.kwargs = kwargs
        self.map_kwargs = map_kwargs or {}
        self.map_names = list(map_names)
        self.simulator_backend = simulator_backend
        self._rng = rng if rng is not None else _random.Random()
        self.current_env = None
        self.current_map = None
        self.max_n_agents = max(self._n_agents_of(m) for m in self.map_names)
        self._draw_map()

    def _kwargs_for(self, map_name):
        kw = dict(self.kwargs)
        for k, v in self.map_kwargs.get(map_name, {}).items():
            kw[k] = {**kw.get(k, {}), **v} if isinstance(v, dict) and isinstance(kw.get(k), dict) else v
        return kw

    def _n_agents_of(self, map_name):
        # the parser alone tells the number of signals: no simulator is started for it
        from pytsc_amd.config import Config
        from pytsc_amd.roadnet import load_roadnet

        kw = {k: v for k, v in self._kwargs_for(map_name).items()
              if k not in ("engine_factory", "common_reward", "reward_scalarization")}
        return load_roadnet(Config(map_name, **kw)).n_agents

    def _draw_map(self):
        if self.current_env is not None:
            self.current_env.close()
        self.current_map = self._rng.choice(self.map_names)
        self.current_env = EPyMARLTrafficSignalNetwork(
            map_name=self.current_map, simulator_backend=self.simulator_backend, **self._kwargs_for(self.current_map))
        self.current_n_agents = self.current_env.tsc_env.n_agents
        self.episode_limit = self.current_env.episode_limit

    # -- padded views ------------------------------------------------------------
    def apply_actions(self, actions):
        self.current_env.apply_actions(actions[: self.current_n_agents])

    def get_avail_actions(self):
        return _pad_rows(self.current_env.get_avail_actions(), self.max_n_agents, [1] * self.get_total_actions())

    def get_obs(self):
        return _pad_rows(self.current_env.get_obs(), self.max_n_agents, [-1] * self.get_obs_size())

    def get_state(self):
        return _pad_rows(self.current_env.get_state(), self.max_n_agents, [-1] * self.get_state_size())

    def get_local_rewards(self):
        rewards = list(self.current_env.get_local_rewards())
        return rewards + [0] * (self.max_n_agents - len(rewards))

    def get_env_info(self):
        info = self.current_env.get_env_info()
        adj = [list(map(float, row)) for row in info["adjacency_matrix"]]
        n = self.max_n_agents
        info["adjacency_matrix"] = [row + [0] * (n - len(row)) for row in adj] + [[0] * n for _ in range(n - len(adj))]
        info["n_agents"] = n
        return info

    def get_obs_size(self):
        return self.current_env.get_obs_size()

    def get_state_size(self):
        return self.current_env.get_state_size()

    def get_total_actions(self):
        return self.current_env.get_total_actions()

    def get_stats(self):
        return self.current_env.get_stats()

    # -- episode control ------------------------------------------------------------
    def reset(self):
        env = self.current_env.tsc_env
        env.episode_count += 1
        obs, state = self.get_obs(), self.get_state()
        if env.episode_over:
            hour_over = env.simulator.is_terminated
            env.restart(reset=False)
            if hour_over:
                self._draw_map()
        return obs, state

    def step(self, actions):
        _, reward, episode_over, _, env_info = self.current_env.step(actions[: self.current_n_agents])
        if self.current_env.common_reward:
            # the reference divides the mean-scalarised reward AGAIN here (wrappers/epymarl.py:326-329, on top of :104-107 inside
            # current_env.step): kept, it is what a learner trained with the reference's wrapper has seen
            if self.current_env.reward_scalarization == "mean":
                reward = reward / self.current_env.tsc_env.n_agents
        else:
            reward = self.get_local_rewards()
        return self.get_obs(), reward, episode_over, False, env_info

    def close(self):
        if self.current_env is not None:
            self.current_env.close()
            self.current_env = None
