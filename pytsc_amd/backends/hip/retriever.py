"""Lane / sim measurements in the reference's dict format
(pytsc/backends/cityflow/retriever.py:54-112), read back from the device."""
import numpy as np


class Retriever:
    def __init__(self, simulator):
        self.simulator = simulator
        self.parsed_network = simulator.parsed_network
        self.config = simulator.config
        self.visibility = self.config.signal["visibility"]
        self.v_size = self.config.simulator["veh_size_min_gap"]
        self.lane_lengths = self.parsed_network.lane_lengths
        self.lane_max_speeds = self.parsed_network.lane_max_speeds

    def retrieve_lane_measurements(self):
        sim = self.simulator
        counts, sums, pm = sim.engine.lane_detail(with_posmat=True)
        rep = sim.replica
        counts = _np(counts)[rep]
        sums = _np(sums)[rep]
        pm = _np(pm)[rep]
        bins = sim.tables.lane_bins
        out = {}
        for k, lane in enumerate(self.parsed_network.lanes):
            n = int(counts[k, 0])
            mean_speed = 0.0 if n == 0 else float(sums[k] / n)
            lane_length = self.lane_lengths[lane] / self.v_size
            if bins[k] > 0 and n > 0:
                m = max(int(bins[k]), self.visibility)
                posmat = [float(x) for x in pm[k, :m]]
            else:
                posmat = [-1.0] * self.visibility
            out[lane] = {
                "n_vehicles": n,
                "n_queued": int(counts[k, 1]),
                "occupancy": np.float64(n / lane_length),
                "mean_speed": mean_speed,
                "position_matrix": posmat,
            }
        return out

    def retrieve_sim_measurements(self):
        st = self.simulator.host("stats")
        return {
            "n_vehicles": int(round(self.simulator.host_scalar_vehicles())),
            "average_travel_time": float(st[1]),
            "time_step": float(st[0]),
        }

    def retrieve_ts_measurements(self):
        pass


def _np(t):
    return t.cpu().numpy() if hasattr(t, "cpu") else np.asarray(t)
