"""Observation spaces (reference: pytsc/common/observations.py:46-382).  The
vectors themselves are assembled on the device (k_agent); these classes expose
them with the reference's list-of-lists API plus the static lane features."""


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
        if config.signal.get("obs_noise_std", 0.0):
            raise NotImplementedError("obs_noise_std > 0 is not supported by the HIP backend")

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
    def get_observations(self):
        return [[_num(x) for x in row] for row in self._sim.host("obs").tolist()]

    def get_observation_info(self):
        return {"lane_obs_dim": 9 + self.visibility,
                "max_n_controlled_lanes": self.max_n_controlled_lanes,
                "max_phases": self.max_phases}

    def get_size(self):
        return self.max_n_controlled_lanes * (self.visibility + 9) + self.max_phases


class LaneFeatures(BaseObservationSpace):
    """obs == the lane-feature state vector (observations.py:305-329); lane
    dropout is not supported on the device path."""

    def __init__(self, config, parsed_network, traffic_signals, simulator_backend):
        super().__init__(config, parsed_network, traffic_signals, simulator_backend)
        if config.signal.get("obs_dropout_prob", 0.0):
            raise NotImplementedError("obs_dropout_prob > 0 is not supported by the HIP backend")
        self.dropped_lanes = []

    def reset_dropped_lanes(self):
        self.dropped_lanes = []

    def get_observations(self):
        return self.get_state()

    def get_observation_info(self):
        return {"lane_obs_dim": 9 + 3,
                "max_n_controlled_lanes": self.max_n_controlled_lanes,
                "max_phases": self.max_phases}

    def get_size(self):
        return self.max_n_controlled_lanes * (3 + 9) + self.max_phases
