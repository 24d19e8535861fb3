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
    """Façade over one scenario (reference: pytsc/__init__.py:28-182): same constructor,
    getters, `step(actions) -> (reward, done, info)` and `restart()`."""

    def __init__(self, scenario, simulator_backend="hip", **kwargs):
        if simulator_backend not in SUPPORTED_SIMULATOR_BACKENDS:
            raise AssertionError(f"Simulator backend {simulator_backend} not supported.")
        # domain-randomised demand (pytsc/__init__.py:21-30): the registry's `disrupted_config` instead of `config`
        self.disrupted = kwargs.get("disrupted", False)
        self.domain_class = kwargs.get("domain_class", None)
        factory = kwargs.pop("engine_factory", None)
        self.scenario, self.simulator_backend = scenario, simulator_backend
        self._mod = SIMULATOR_MODULES[simulator_backend]
        self.config = self._mod["disrupted_config" if self.disrupted else "config"](scenario, **kwargs)
        self._validate_config()
        self.parsed_network = self._mod["network_parser"](self.config)
        self.simulator = self._mod["simulator"](self.parsed_network, engine_factory=factory)
        self.hour_count = self.episode_count = 0
        self._boot()
        self.n_agents = len(self.traffic_signals)

    # ---- construction ------------------------------------------------------------------
    def _validate_config(self):
        sig = self.config.signal
        validate_input_against_allowed(sig["action_space"], ACTION_SPACES.keys())
        validate_input_against_allowed(sig["observation_space"], OBSERVATION_SPACES.keys())
        if self.config.network["control_scheme"] != "decentralized":
            raise NotImplementedError("centralized control is out of scope")

    def _boot(self):
        """Start the engine, then rebuild everything that hangs off its first measurement."""
        self.simulator.start_simulator()
        self._init_traffic_signals()
        self._init_parsers()

    def _init_traffic_signals(self):
        make = self._mod["traffic_signal"]
        self.traffic_signals = {
            ts_id: make(ts_id, cfg, self.simulator) for ts_id, cfg in self.parsed_network.traffic_signals.items()
        }
        self._update_ts_stats()

    def _init_parsers(self):
        sig, signals = self.config.signal, self.traffic_signals
        self.action_space = ACTION_SPACES[sig["action_space"]](self.config, signals)
        self.observation_space = OBSERVATION_SPACES[sig["observation_space"]](
            self.config, self.parsed_network, signals, self.simulator_backend)
        self.metrics = self._mod["metrics_parser"](self.parsed_network, self.simulator, signals)
        self.reward_function = REWARD_FUNCTIONS[sig["reward_function"]](self.metrics, signals)

    def _update_ts_stats(self):
        meas = self.simulator.step_measurements
        for ts in self.traffic_signals.values():
            ts.update_stats(meas)

    # ---- episode bookkeeping -----------------------------------------------------------
    @property
    def episode_limit(self):
        sim = self.config.simulator
        return int(sim["episode_limit"] / sim["delta_time"])

    @property
    def episode_over(self):
        t = self.simulator.sim_step
        return t > 0 and t % self.config.simulator["episode_limit"] == 0

    # ---- getters -----------------------------------------------------------------------
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
        info = self.metrics.get_step_stats()
        info["episode_count"] = self.episode_count
        info["episode_limit"] = self.episode_limit
        if self.disrupted:  # (pytsc/__init__.py:153-155)
            info["n_domains"] = len(self.config.domain_classes)
            info["domain_class"] = self.config.current_domain_class
        return info

    def get_env_stats(self):
        info = self.get_env_info()
        for group in self.simulator.step_measurements.values():
            info.update(group)
        return info

    # ---- the hot path ------------------------------------------------------------------
    def step(self, actions):
        """One env step = delta_time engine steps + measurement (pytsc/__init__.py:165-179)."""
        self.action_space.apply(actions)
        self.simulator.simulator_step(n_steps=self.config.simulator["delta_time"])
        self._update_ts_stats()
        return self.get_reward(), self.episode_over, self.get_env_info()

    def restart(self, reset=True):
        self.episode_count += bool(self.episode_over)
        if not self.simulator.is_terminated:
            return
        self.hour_count += 1
        self.simulator.close_simulator()
        if reset:
            self.simulator.start_simulator()
            if self.domain_class is not None:  # (pytsc/__init__.py:173-174: after the start, so it binds the NEXT one)
                self.config.set_domain_class(self.domain_class)
            self._init_traffic_signals()
            self._init_parsers()

    def close(self):
        self.simulator.shutdown()
