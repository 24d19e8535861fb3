"""Network-wide metrics (reference: pytsc/backends/cityflow/metrics.py:7-260),
read from the device's per-step stats."""
import numpy as np


class MetricsParser:
    def __init__(self, parsed_network, simulator, traffic_signals):
        self.config = parsed_network.config
        self.simulator = simulator
        self.parsed_network = parsed_network
        self.traffic_signals = traffic_signals

    def _s(self, i):
        return float(self.simulator.host("stats")[i])

    @property
    def flickering_signal(self):
        return np.mean([ts.controller.program.phase_changed for ts in self.traffic_signals.values()])

    time_step = property(lambda s: s._s(0))
    average_travel_time = property(lambda s: s._s(1))
    n_queued = property(lambda s: int(s._s(2)))
    mean_speed = property(lambda s: s._s(3))
    mean_delay = property(lambda s: s._s(4))
    density = property(lambda s: s._s(5))
    pressure = property(lambda s: s._s(6))
    network_flow = property(lambda s: s._s(7))

    @property
    def norm_mean_speed(self):
        return 1 - self.mean_delay

    @property
    def pressures(self):
        return [ts.pressure for ts in self.traffic_signals.values()]

    @property
    def density_map(self):
        nl = self.parsed_network.neighbors_lanes
        ts_ids = list(self.traffic_signals.keys())
        dm = np.zeros((len(ts_ids), len(ts_ids)))
        lanes = self.simulator.step_measurements["lane"]
        for i, ts in enumerate(ts_ids):
            for j, other in enumerate(ts_ids):
                if other in nl.get(ts, {}):
                    ls = nl[ts][other]
                    if ls:
                        m = sum(lanes[l]["occupancy"] for l in ls) / len(ls)
                        dm[i, j] = float(np.clip(m, 0, 1))
        return (dm + dm.T) / 2 + 1e-6 * self.parsed_network.adjacency_matrix

    @property
    def mst(self):
        from scipy.sparse.csgraph import minimum_spanning_tree

        return minimum_spanning_tree(-1 * self.density_map).toarray()

    def get_step_stats(self):
        stats = {
            "time_step": self.time_step,
            "average_travel_time": self.average_travel_time,
            "n_queued": self.n_queued,
            "mean_speed": self.mean_speed,
            "mean_delay": self.mean_delay,
            "density": self.density,
            "pressure": self.pressure,
            "network_flow": self.network_flow,
        }
        if self.config.misc["return_agent_stats"]:
            for ts in self.traffic_signals.values():
                stats.update({
                    f"{ts.id}__phase": ts.controller.current_phase,
                    f"{ts.id}__n_queued": ts.n_queued,
                    f"{ts.id}__mean_speed": ts.mean_speed,
                    f"{ts.id}__mean_delay": ts.mean_delay,
                    f"{ts.id}__density": ts.occupancy,
                    f"{ts.id}__pressure": ts.pressure,
                })
        if self.config.misc["return_lane_stats"]:
            for lane, d in self.simulator.step_measurements["lane"].items():
                for k in ("n_vehicles", "n_queued", "mean_speed", "occupancy"):
                    v = d[k]
                    stats[f"{lane}__{k}"] = v.item() if hasattr(v, "item") else v
        return stats
