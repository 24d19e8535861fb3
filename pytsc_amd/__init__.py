"""pytsc_amd — MI355X-native batched traffic-signal environment.

Keeps the reference's façade and plugin surface for the step() hot path
(pytsc/__init__.py:9-182): `TrafficSignalNetwork(scenario, simulator_backend,
**kwargs)`, `SIMULATOR_MODULES`, `SUPPORTED_SIMULATOR_BACKENDS`; the vehicle
dynamics, measurements, observations, rewards and masks run as HIP kernels
(pytsc_amd/csrc) behind the C-ABI in include/pytsc_amd.h.  There is no CPU
fallback.
"""
import numpy as np

from pytsc_amd.backends.hip import HIP_MODULES
from pytsc_amd.common import ACTION_SPACES, OBSERVATION_SPACES, REWARD_FUNCTIONS
from pytsc_amd.common.utils import validate_input_against_allowed

# "cityflow" is accepted as an alias so reference call sites switch over unchanged
SUPPORTED_SIMULATOR_BACKENDS = ("hip", "cityflow")
SIMULATOR_MODULES = {"hip": HIP_MODULES, "cityflow": HIP_MODULES}


class TrafficSignalNetwork:
    def __init__(self, scenario, simulator_backend="hip", **kwargs):
        self.scenario = scenario
        self.simulator_backend = simulator_backend
        assert (
            self.simulator_backend in SUPPORTED_SIMULATOR_BACKENDS
        ), f"Simulator backend {self.simulator_backend} not supported."
        if kwargs.get("disrupted", False):
            raise NotImplementedError("disrupted (domain-randomised) configs are out of scope")
        engine_factory = kwargs.pop("engine_factory", None)
        modules = SIMULATOR_MODULES[simulator_backend]
        self.config = modules["config"](scenario, **kwargs)
        self._validate_config()
        self.parsed_network = modules["network_parser"](self.config)
        self.simulator = modules["simulator"](self.parsed_network, engine_factory=engine_factory)
        self.simulator.start_simulator()
        self._init_traffic_signals()
        self._init_parsers()
        self._set_n_agents()
        self.hour_count = 0
        self.episode_count = 0

    @property
    def episode_limit(self):
        return int(self.config.simulator["episode_limit"] / self.config.simulator["delta_time"])

    @property
    def episode_over(self):
        if self.simulator.sim_step > 0:
            return self.simulator.sim_step % self.config.simulator["episode_limit"] == 0
        return False

    def _validate_config(self):
        validate_input_against_allowed(self.config.signal["action_space"], ACTION_SPACES.keys())
        validate_input_against_allowed(self.config.signal["observation_space"], OBSERVATION_SPACES.keys())
        if self.config.network["control_scheme"] != "decentralized":
            raise NotImplementedError("centralized control is out of scope")

    def _init_parsers(self):
        self.action_space = ACTION_SPACES[self.config.signal["action_space"]](self.config, self.traffic_signals)
        self.observation_space = OBSERVATION_SPACES[self.config.signal["observation_space"]](
            self.config, self.parsed_network, self.traffic_signals, self.simulator_backend)
        self.metrics = SIMULATOR_MODULES[self.simulator_backend]["metrics_parser"](
            self.parsed_network, self.simulator, self.traffic_signals)
        self.reward_function = REWARD_FUNCTIONS[self.config.signal["reward_function"]](
            self.metrics, self.traffic_signals)

    def _init_traffic_signals(self):
        self.traffic_signals = {}
        for ts_id, signal_config in self.parsed_network.traffic_signals.items():
            ts = SIMULATOR_MODULES[self.simulator_backend]["traffic_signal"](ts_id, signal_config, self.simulator)
            ts.update_stats(self.simulator.step_measurements)
            self.traffic_signals[ts_id] = ts

    def _set_n_agents(self):
        self.n_agents = len(self.traffic_signals)

    def _update_ts_stats(self):
        for ts in self.traffic_signals.values():
            ts.update_stats(self.simulator.step_measurements)

    def get_action_mask(self):
        return self.action_space.get_mask()

    def get_action_size(self):
        return self.action_space.get_size()

    def get_observations(self):
        return self.observation_space.get_observations()

    def get_observation_size(self):
        return self.observation_space.get_size()

    def get_state(self):
        return self.observation_space.get_state()

    def get_state_size(self):
        return self.observation_space.get_state_size()

    def get_reward(self):
        return self.reward_function.get_global_reward()

    def get_rewards(self):
        return self.reward_function.get_local_reward()

    def get_env_info(self):
        stats = self.metrics.get_step_stats()
        stats.update({"episode_count": self.episode_count})
        stats.update({"episode_limit": self.episode_limit})
        return stats

    def get_env_stats(self):
        stats = self.get_env_info()
        for v in self.simulator.step_measurements.values():
            stats.update(v)
        return stats

    def restart(self, reset=True):
        if self.episode_over:
            self.episode_count += 1
        if self.simulator.is_terminated:
            self.hour_count += 1
            self.simulator.close_simulator()
            if reset:
                self.simulator.start_simulator()
                self._init_traffic_signals()
                self._init_parsers()

    def step(self, actions):
        self.action_space.apply(actions)
        self.simulator.simulator_step(n_steps=self.config.simulator["delta_time"])
        self._update_ts_stats()
        return self.get_reward(), self.episode_over, self.get_env_info()

    def close(self):
        self.simulator.shutdown()
