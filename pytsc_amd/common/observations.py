"""Observation spaces (reference: pytsc/common/observations.py:46-382).  The
vectors themselves are assembled on the device (k_agent / kr_step); these classes expose
them with the reference's list-of-lists API plus the static lane features.

Observation noise (obs_noise_std > 0) and lane dropout (obs_dropout_prob > 0) draw from numpy's global stream
exactly as the reference does -- one np.random.normal per surviving bin in signal / lane / bin order, one
np.random.choice over the lanes at construction -- so they are applied here on the host, on top of what the
device measured (single-env façade; the batched engine has no noise model)."""
import numpy as np

from pytsc_amd.common.utils import pad_list


class BaseObservationSpace:
    max_lanes_per_direction = 6
    max_lane_speed = 15.0
    max_lane_length = 500
    max_phases = 20
    max_n_controlled_lanes = 16

    def __init__(self, config, parsed_network, traffic_signals, simulator_backend):
        self.config = config
        self.parsed_network = parsed_network
        self.n_agents = len(traffic_signals)
        self.traffic_signals = traffic_signals
        self.simulator_backend = simulator_backend
        self.pad_value = config.misc["pad_value"]
        self.visibility = config.signal["visibility"]
        self._sim = next(iter(traffic_signals.values())).simulator
        self.lane_features = dict(self._sim.tables.lane_features)

    @staticmethod
    def high():
        return 1

    @staticmethod
    def low():
        return -1

    def get_per_agent_lane_features(self):
        out = []
        for ts in self.traffic_signals.values():
            feats = [self.lane_features[l] for l in ts.incoming_lanes]
            for _ in range(self.max_n_controlled_lanes - len(ts.incoming_lanes)):
                feats.append([self.pad_value] * (self.max_lanes_per_direction + 3))
            out.append(feats)
        return out

    def get_state(self):
        return [[_num(x) for x in row] for row in self._sim.host("state").tolist()]

    def get_state_size(self):
        return self.max_n_controlled_lanes * (3 + 9) + self.max_phases


def _num(x):
    return int(x)


class PositionMatrix(BaseObservationSpace):
    def __init__(self, config, parsed_network, traffic_signals, simulator_backend):
        super().__init__(config, parsed_network, traffic_signals, simulator_backend)
        self.obs_noise_std = config.signal.get("obs_noise_std", 0.0)

    def _add_gaussian_noise(self, pos_mat, std):
        """observations.py:72-88: only entries > 0 survive (Q2), each consumes one normal draw."""
        noisy = []
        for val in pos_mat:
            if val > 0:
                noisy.append(np.clip(val + np.random.normal(loc=0.0, scale=std), 0.0, 1.0))
        return noisy

    def get_observations(self):
        if not self.obs_noise_std:
            return [[_num(x) for x in row] for row in self._sim.host("obs").tolist()]
        # the noisy variant needs the float matrices: the lane detail the façade reads back anyway
        lanes = self._sim.step_measurements["lane"]
        observations = []
        for ts in self.traffic_signals.values():
            obs = []
            for lane in ts.incoming_lanes:
                obs.extend(self.lane_features[lane])
                obs.extend(self._add_gaussian_noise(lanes[lane]["position_matrix"][-self.visibility:], self.obs_noise_std))
            obs = pad_list(obs, self.get_size() - self.max_phases, -1)
            obs.extend(pad_list(ts.phase_id, self.max_phases, -1))
            observations.append(obs)
        return observations

    def get_observation_info(self):
        return {"lane_obs_dim": 9 + self.visibility,
                "max_n_controlled_lanes": self.max_n_controlled_lanes,
                "max_phases": self.max_phases}

    def get_size(self):
        return self.max_n_controlled_lanes * (self.visibility + 9) + self.max_phases


class LaneFeatures(BaseObservationSpace):
    """obs == the lane-feature state vector (observations.py:305-329) with the three measured entries of
    dropped lanes zeroed (observations.py:245-253, 321-322)."""

    def __init__(self, config, parsed_network, traffic_signals, simulator_backend):
        super().__init__(config, parsed_network, traffic_signals, simulator_backend)
        self.dropout_prob = config.signal.get("obs_dropout_prob", 0.0)
        self.reset_dropped_lanes()

    def reset_dropped_lanes(self):
        lanes = self.parsed_network.lanes
        self.dropped_lanes = np.random.choice(lanes, int(self.dropout_prob * len(lanes)), replace=False)

    def get_observations(self):
        rows = self.get_state()
        if len(self.dropped_lanes):
            dropped = set(self.dropped_lanes.tolist())
            w = 9 + 3
            for row, ts in zip(rows, self.traffic_signals.values()):
                for i, lane in enumerate(ts.incoming_lanes):
                    if lane in dropped:
                        row[i * w + 9: i * w + 12] = [0, 0, 0]
        return rows

    def get_observation_info(self):
        return {"lane_obs_dim": 9 + 3,
                "max_n_controlled_lanes": self.max_n_controlled_lanes,
                "max_phases": self.max_phases}

    def get_size(self):
        return self.max_n_controlled_lanes * (3 + 9) + self.max_phases
