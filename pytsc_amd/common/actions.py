"""Action spaces (reference: pytsc/common/actions.py:87-211).  `apply` stages the
per-signal targets; the engine consumes them in the next simulator_step."""
import random

from pytsc_amd.common.utils import pad_list


class BaseActionSpace:
    def __init__(self, config, traffic_signals):
        self.config = config
        self.traffic_signals = traffic_signals
        self.ts_ids = list(traffic_signals.keys())

    def _controller_targets(self, controller, choices_of):
        """Per signal: the rule-based controller's pick (greedy / max_pressure / sotl / fixed_time, via
        ts.get_controller_action) or, for "random", a draw of Python's `random` weighted by the action mask
        (reference common/actions.py:64-84,190-211)."""
        mask = self.get_mask()
        out = []
        for i, ts in enumerate(self.traffic_signals.values()):
            if controller == "random":
                out.append(random.choices(choices_of(ts), weights=mask[i])[0])
            else:
                out.append(ts.get_controller_action(controller))
        return out

    def get_trad_controller_actions(self, controller):
        """Actions (phase indices) of a traditional controller for every signal."""
        return self._controller_targets(controller, lambda ts: range(ts.controller.n_phases))


class PhaseSelectionActionSpace(BaseActionSpace):
    def apply(self, actions):
        for ts_idx, ts in enumerate(self.traffic_signals.values()):
            ts.action_to_phase(actions[ts_idx])

    def get_size(self):
        return max(ts.controller.n_phases for ts in self.traffic_signals.values())

    def get_mask(self):
        return [pad_list(ts.controller.get_allowable_phase_switches(), self.get_size())
                for ts in self.traffic_signals.values()]


class PhaseSwitchActionSpace(BaseActionSpace):
    def apply(self, actions):
        if not isinstance(actions, list):
            actions = actions.tolist()
        for ts_idx, ts in enumerate(self.traffic_signals.values()):
            cur = ts.controller.program.current_phase_index
            if actions[ts_idx] == 1:
                ts.action_to_phase((cur + 1) % ts.controller.n_phases)
            else:
                ts.action_to_phase(cur)

    def get_size(self):
        return 2

    def get_trad_controller_actions(self, controller):
        """Phase-switch form: 1 where the controller's pick differs from the current phase, else 0 (for "random" the
        pick is itself a 0 / 1 draw weighted by the two-entry mask, compared the same way: reference :190-211)."""
        picks = self._controller_targets(controller, lambda ts: [0, 1])
        return [1 if ts.controller.program.current_phase_index != a else 0
                for ts, a in zip(self.traffic_signals.values(), picks)]

    def get_mask(self):
        masks = []
        for ts in self.traffic_signals.values():
            allow = ts.controller.get_allowable_phase_switches()
            cur = ts.controller.program.current_phase_index
            nxt = (cur + 1) % ts.controller.n_phases
            masks.append([1 if allow[cur] else 0, 1 if allow[nxt] else 0])
        return masks
