"""Rule-based controllers with the reference's constructor / get_action contract
(pytsc/controllers/controllers.py:26-259).  The reference's own controller
classes also run unchanged against the façade objects (tests/test_facade.py)."""
import numpy as np


class BaseController:
    def __init__(self, traffic_signal, round_robin=True, **kwargs):
        self.traffic_signal = traffic_signal
        self.round_robin = round_robin
        self.visibility = traffic_signal.config["visibility"]
        self.controller = traffic_signal.controller

    def __repr__(self):
        return f"{self.__class__.__name__} ({self.traffic_signal.id})"

    def _phase_lanes(self, phase_index):
        ts = self.traffic_signal
        return ts.config["phase_to_inc_out_lanes"][ts.config["phases"][phase_index]]


class FixedTimeController(BaseController):
    def __init__(self, traffic_signal, green_time=25):
        super().__init__(traffic_signal)
        self.green_time = green_time

    def get_action(self, inp):
        c = self.controller
        if c.current_phase_index in c.green_phase_indices and c.time_on_phase < self.green_time:
            return c.current_phase_index
        return c.next_phase_index


class _ArgmaxController(BaseController):
    def _score(self, inp, phase_index):
        raise NotImplementedError

    def get_action(self, inp):
        c = self.controller
        if c.current_phase_index not in c.green_phase_indices:
            return c.next_phase_index
        mask = c.get_allowable_phase_switches()
        scores = [(self._score(inp, a) if ok else float("-inf"), a) for a, ok in enumerate(mask)]
        best = max(scores, key=lambda x: x[0])[0]
        return np.random.choice([a for s, a in scores if s == best])


class GreedyController(_ArgmaxController):
    def _score(self, inp, phase_index):
        total = 0
        for lane in self._phase_lanes(phase_index).keys():
            total += sum(p == 0.0 for p in inp["lane"][lane]["position_matrix"][-self.visibility:])
        return total


class MaxPressureController(_ArgmaxController):
    def _score(self, inp, phase_index):
        pressure = 0
        for lane, outs in self._phase_lanes(phase_index).items():
            n_in = sum(p >= 0.0 for p in inp["lane"][lane]["position_matrix"][-self.visibility:])
            n_out = 0
            for o in outs:  # the reference keeps only the last outgoing lane's count
                n_out = sum(p >= 0.0 for p in inp["lane"][o]["position_matrix"][: self.visibility])
            pressure += np.abs(n_in - n_out)
        return pressure


class SOTLController(BaseController):
    def __init__(self, traffic_signal, theta=3, mu=4, phi_min=5):
        super().__init__(traffic_signal)
        self.mu, self.theta, self.phi_min = mu, theta, phi_min

    def _flow(self, inp, phase_index):
        return sum(
            sum(p >= 0.0 for p in inp["lane"][lane]["position_matrix"][-self.visibility:])
            for lane in self._phase_lanes(phase_index).keys()
        )

    def get_action(self, inp):
        c = self.controller
        mask = c.get_allowable_phase_switches()
        if not mask[c.current_phase_index]:
            return c.next_phase_index
        red = self._flow(inp, c.next_green_phase_index)
        green = self._flow(inp, c.current_phase_index)
        if c.time_on_phase >= self.phi_min and not 0 < green < self.mu and red >= self.theta:
            return c.next_phase_index
        return c.current_phase_index


class RandomController(BaseController):
    def get_action(self, inp):
        return np.random.choice(np.where(self.controller.get_allowable_phase_switches())[0])


CONTROLLERS = {
    "fixed_time": FixedTimeController,
    "greedy": GreedyController,
    "max_pressure": MaxPressureController,
    "sotl": SOTLController,
    "random": RandomController,
}
