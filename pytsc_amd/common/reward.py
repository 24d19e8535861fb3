"""Reward functions (reference: pytsc/common/reward.py:43-136); values come from
the device (k_agent / k_stats) in f64."""


class BaseRewardFunction:
    def __init__(self, metrics, traffic_signals):
        self.metrics = metrics
        self.traffic_signals = traffic_signals
        self.parsed_network = metrics.parsed_network
        self.config = metrics.config
        self._sim = metrics.simulator

    def get_global_reward(self):
        return float(self._sim.host("global_reward64"))

    def get_local_reward(self):
        if not self._sim.stepped:
            raise AttributeError("rewards are undefined before the first step")
        return [float(x) for x in self._sim.host("local_reward64")]


class QueueLength(BaseRewardFunction):
    kind = "queue_length"


class MaxPressure(BaseRewardFunction):
    kind = "max_pressure"
