"""Action spaces (reference: pytsc/common/actions.py:87-211).  `apply` stages the
per-signal targets; the engine consumes them in the next simulator_step."""
from pytsc_amd.common.utils import pad_list


class BaseActionSpace:
    def __init__(self, config, traffic_signals):
        self.config = config
        self.traffic_signals = traffic_signals
        self.ts_ids = list(traffic_signals.keys())


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

    def get_mask(self):
        masks = []
        for ts in self.traffic_signals.values():
            allow = ts.controller.get_allowable_phase_switches()
            cur = ts.controller.program.current_phase_index
            nxt = (cur + 1) % ts.controller.n_phases
            masks.append([1 if allow[cur] else 0, 1 if allow[nxt] else 0])
        return masks
