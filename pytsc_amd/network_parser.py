"""PyTSC-compatible parsed-network view of a roadnet file.

Mirrors the attribute surface of the reference's CityFlow ``NetworkParser``
(pytsc/backends/cityflow/network_parser.py:11-703) that the rest of the plugin
surface consumes: ``traffic_signals`` (roadnet-file order,
:48-78), ``lanes`` (sorted, :104-118), ``traffic_signal_ids`` (sorted,
:122-134), ``adjacency_matrix`` (:138-163), ``k_hop_neighbors`` (:167-180,
:580-596), ``neighbors_lanes`` (:261-291), ``lane_lengths`` (:325-352),
``lane_max_speeds`` (:356-368), ``lane_indices`` (:372-385), ``lane_angles``
(:389-408), lane maps (:598-629) and the phase tables (:631-703).
"""
import numpy as np

from pytsc_amd.roadnet import load_roadnet


class NetworkParser:
    def __init__(self, config, roadnet=None):
        self.config = config
        self.rn = roadnet if roadnet is not None else load_roadnet(config)
        self.net = self.rn.net
        self._build()

    # -- raw views -----------------------------------------------------------
    @property
    def intersections(self):
        return self.net["intersections"]

    @property
    def roads(self):
        return self.net["roads"]

    def _build(self):
        rn = self.rn
        self.lanes = list(rn.lane_ids)
        self.traffic_signal_ids = sorted(rn.agent_ids)
        self.lane_lengths = {l: float(rn.lane_len_pytsc[k]) for k, l in enumerate(rn.lane_ids)}
        self.lane_max_speeds = {l: float(rn.lane_max_speed[k]) for k, l in enumerate(rn.lane_ids)}
        self.lane_indices = {l: int(rn.lane_idx_in_road[k]) for k, l in enumerate(rn.lane_ids)}
        self.lane_angles = {l: float(rn.lane_angle_deg[k]) for k, l in enumerate(rn.lane_ids)}
        self._build_adjacency()
        self._build_signals()

    # -- graph ---------------------------------------------------------------
    def _build_adjacency(self):
        ids = self.traffic_signal_ids
        idx = {t: i for i, t in enumerate(ids)}
        n = len(ids)
        network_cfg = self.config.network
        adj = np.zeros((n, n))
        if "neighbors" in network_cfg:
            # common/network_parser.py:19-35
            nb = network_cfg["neighbors"]
            for i, a in enumerate(ids):
                for j, b in enumerate(ids):
                    if a in nb and b in nb[a]:
                        adj[i, j] = 1.0
        else:
            for road in self.roads:
                s, e = road["startIntersection"], road["endIntersection"]
                if s in idx and e in idx:
                    adj[idx[s], idx[e]] = 1.0
                    adj[idx[e], idx[s]] = 1.0
        self.adjacency_matrix = adj
        max_hops = self.config.misc["max_hops"]
        self.k_hop_neighbors = {}
        for t in ids:
            self.k_hop_neighbors[t] = {}
            for k in range(1, max_hops + 1):
                p = np.linalg.matrix_power(adj, k)
                self.k_hop_neighbors[t][k] = [ids[j] for j in np.where(p[idx[t]] > 0)[0]]
        if "neighbors_lanes" in network_cfg:
            self.neighbors_lanes = network_cfg["neighbors_lanes"]
        else:
            self.neighbors_lanes = {}
            for t in ids:
                self.neighbors_lanes[t] = {}
                for j in np.where(adj[idx[t]] > 0)[0]:
                    nb_id = ids[j]
                    lanes = []
                    for road in self.roads:
                        if road["startIntersection"] == t and road["endIntersection"] == nb_id:
                            lanes += [f"{road['id']}_{i}" for i in range(len(road["lanes"]))]
                    self.neighbors_lanes[t][nb_id] = lanes

    # -- signals ---------------------------------------------------------------
    def _build_signals(self):
        sig_cfg = self.config.signal
        xs = [it["point"]["x"] for it in self.intersections]
        ys = [it["point"]["y"] for it in self.intersections]
        span = [max(xs) - min(xs), max(ys) - min(ys)]
        self.network_boundary = ((min(xs), min(ys)), (max(xs), max(ys)))
        self.norm_network_boundary = span
        self.traffic_signals = {}
        self.ts_phase_to_inc_out_lanes = {}
        for it in self.intersections:
            if it["virtual"]:
                continue
            ts_id = it["id"]
            inc, out, mapping = [], [], {}
            rl_to_ll = []
            for rl in it["roadLinks"]:
                pairs = []
                for ll in rl["laneLinks"]:
                    a = f"{rl['startRoad']}_{ll['startLaneIndex']}"
                    b = f"{rl['endRoad']}_{ll['endLaneIndex']}"
                    inc.append(a)
                    out.append(b)
                    mapping.setdefault(a, []).append(b)
                    pairs.append((a, b))
                rl_to_ll.append(pairs)
            program = it["trafficLight"]["lightphases"]
            p2l = {}
            for i, ph in enumerate(program):
                p2l[i] = {}
                for rl_idx in ph["availableRoadLinks"]:
                    for a, b in rl_to_ll[rl_idx]:
                        p2l[i].setdefault(a, []).append(b)
            self.ts_phase_to_inc_out_lanes[ts_id] = p2l
            green, yellow, mm = [], [], {}
            for i, ph in enumerate(program):
                if len(ph["availableRoadLinks"]) and ph["time"] > 5:
                    green.append(i)
                    mm[i] = {"min_time": sig_cfg["min_green_time"],
                             "max_time": sig_cfg["max_green_time"]}
                else:
                    yellow.append(i)
                    mm[i] = {"min_time": sig_cfg["yellow_time"],
                             "max_time": sig_cfg["yellow_time"]}
            if len(yellow) == 1:
                phases = [p for g in green for p in (g, yellow[0])]
            else:
                phases = [p for pair in zip(green, yellow) for p in pair]
            green_idx = [phases.index(g) for g in green]
            yellow_idx = [g + 1 for g in green_idx]
            phase_idx = [p for pair in zip(green_idx, yellow_idx) for p in pair]
            if "phase_sequence" in self.config.simulator:
                phases = list(self.config.simulator["phase_sequence"])
                phase_idx = list(range(len(phases)))
                green_idx = phase_idx[0::2]
                yellow_idx = phase_idx[1::2]
            ts = {
                "coordinates": [it["point"]["x"], it["point"]["y"]],
                "norm_coordinates": [it["point"]["x"] / span[0] if span[0] else 0.0,
                                     it["point"]["y"] / span[1] if span[1] else 0.0],
                "incoming_lanes": sorted(set(inc)),
                "outgoing_lanes": sorted(set(out)),
                "inc_to_out_lanes": mapping,
                "phase_to_inc_out_lanes": p2l,
                "phases": phases,
                "n_phases": len(phases),
                "phases_min_max_times": mm,
                "phase_indices": phase_idx,
                "green_phase_indices": green_idx,
                "yellow_phase_indices": yellow_idx,
            }
            ts.update(sig_cfg)
            self.traffic_signals[ts_id] = ts

    def _id_to_intersection(self, intersection_id):
        for it in self.intersections:
            if it["id"] == intersection_id:
                return it
        raise ValueError(f"Intersection {intersection_id} not found")
