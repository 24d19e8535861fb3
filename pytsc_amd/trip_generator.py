"""Flow-file producers for the CityFlow on-disk format.

Host-side input producers, not part of the device path: they write the flow JSON
that `pytsc_amd.flows.FlowCompiler` turns into device tables.  Class names,
constructor arguments, file names and — for a given seed — the generated trips
are those of the reference's generators
(pytsc/backends/cityflow/trip_generator.py:45-1072, common/utils.py:136-155);
`tests/test_trip_generator.py` compares against files written by the reference.

Structure here: one walker over the road graph (`RoadGraph`), one arrival loop
(`_emit`) parameterised by an inter-arrival law, and thin classes that choose
law, entry roads and file name.  Randomness comes from two explicit streams — a
numpy `RandomState` for inter-arrival draws and a `random.Random` for route
choices — seeded like the reference seeds the global ones, so the sequences agree.
"""
import json
import os
import random as _pyrandom

import numpy as np

from pytsc_amd.config import Config
from pytsc_amd.flows import DEFAULT_VEHICLE
from pytsc_amd.roadnet import load_roadnet

TURNS = ("turn_left", "turn_right", "go_straight")


class _Streams:
    """The two random streams; `seed=None` shares the process-global ones."""

    def __init__(self, seed):
        if seed is None:
            self.np, self.py = np.random, _pyrandom
        else:
            self.np, self.py = np.random.RandomState(seed), _pyrandom.Random(seed)


class RoadGraph:
    """Road-level view of a roadnet: boundary roads, turn table, choice weights."""

    def __init__(self, rn, edge_weights=None):
        self.rn = rn
        virtual = {it["id"] for it in rn.intersections if it["virtual"]}
        self.entries, self.exits = [], []
        for road in rn.roads:
            s_out = road["startIntersection"] in virtual
            e_out = road["endIntersection"] in virtual
            if s_out and not e_out:
                self.entries.append(road["id"])
            if e_out and not s_out:
                self.exits.append(road["id"])
        self.turn_table = {}
        for it in rn.intersections:
            for rl in it.get("roadLinks", []):
                self.turn_table.setdefault(rl["startRoad"], {})[rl["type"]] = rl["endRoad"]
        self.weights = self._weights(edge_weights)

    def _weights(self, given):
        rn = self.rn
        if given is not None:
            return {road["id"]: given.get(road["id"], 0.0) for road in rn.roads}
        top = float(max(rn.lane_max_speed))
        out = {}
        for road in rn.roads:
            speeds = [float(lane["maxSpeed"]) for lane in road["lanes"]]
            out[road["id"]] = np.round((sum(speeds) / len(speeds)) / top, 2)
        return out

    def longest_trip(self):
        """Diameter of the directed intersection graph + 2 (a corner-to-corner trip);
        number of signals + 1 when some intersection cannot be reached."""
        adj = {it["id"]: set() for it in self.rn.intersections}
        for road in self.rn.roads:
            adj[road["startIntersection"]].add(road["endIntersection"])
        far = 0
        for src in adj:
            seen = {src: 0}
            wave = [src]
            while wave:
                nxt = []
                for u in wave:
                    for v in adj[u]:
                        if v not in seen:
                            seen[v] = seen[u] + 1
                            nxt.append(v)
                wave = nxt
            if len(seen) < len(adj):
                return len(self.rn.agent_ids) + 1
            far = max(far, max(seen.values()))
        return far + 2

    # -- route walk ------------------------------------------------------------
    def step(self, road, turn_probs, rnd, closed=()):
        options, w = [], []
        table = self.turn_table.get(road)
        if table is None:
            return None
        for k, turn in enumerate(TURNS):
            nxt = table.get(turn)
            if nxt and nxt not in closed:
                options.append(nxt)
                w.append(turn_probs[k] * self.weights.get(nxt, 1.0))
        if not options:
            return None
        total = sum(w)
        if total == 0:
            return None
        return rnd.choices(options, weights=[x / total for x in w], k=1)[0]

    def walk(self, start, turn_probs, rnd, closed=()):
        """Weighted random walk without repeated roads; ends at an exit road, at a dead
        end, or when three redraws in a row land on a road already used."""
        route = [start]
        here = start
        while True:
            nxt = self.step(here, turn_probs, rnd, closed)
            redraws = 0
            while nxt in route:
                nxt = self.step(here, turn_probs, rnd, closed)
                redraws += 1
                if redraws >= len(TURNS):
                    nxt = None
                    break
            if nxt is None:
                return route
            route.append(nxt)
            here = nxt
            if nxt in self.exits:
                return route

    def trip(self, start, turn_probs, rnd, longest, closed=()):
        route = [start]
        while len(route) <= 1 or len(route) > longest:
            route = self.walk(start, turn_probs, rnd, closed)
        return route


def weibull_flow_rates(streams, shape, scale, max_rate, num_segments):
    """Per-segment mean inter-arrival times: a bell over the segments, peak rolled to
    a random segment (common/utils.py:136-155; the Weibull draw only advances the
    stream, as there)."""
    streams.np.weibull(shape, 1000)
    peak = streams.np.randint(0, num_segments)
    x = np.linspace(-2, 2, num_segments)
    bell = np.exp(-(x ** 2))
    return np.roll(bell / max(bell) * max_rate, peak)


class CityFlowTripGenerator:
    """Gaussian inter-arrival times on every entry road, random-walk routes
    (trip_generator.py:45-287)."""

    turns = list(TURNS)
    vehicle_data = dict(DEFAULT_VEHICLE)

    def __init__(self, scenario, start_time, end_time, inter_mu, inter_sigma, seed=0,
                 edge_weights=None, turn_probs=(0.1, 0.3, 0.6), config=None, **kwargs):
        self.scenario = scenario
        self.config = config if config is not None else Config(scenario, **kwargs)
        self.rn = load_roadnet(self.config)
        self.start_time, self.end_time = start_time, end_time
        self.inter_mu, self.inter_sigma = inter_mu, inter_sigma
        self.turn_probabilities = list(turn_probs)
        self.streams = _Streams(seed)
        self.graph = RoadGraph(self.rn, edge_weights)
        self.max_trip_length = self.graph.longest_trip()
        self.edge_weights = self.graph.weights
        self.closed_roads = ()

    # -- the arrival loop --------------------------------------------------------
    def _law(self, road, now):
        """(mean, std) of the next inter-arrival time on `road` at time `now`."""
        return self.inter_mu, self.inter_sigma

    def _emit(self, roads, out, until=None, law=None):
        law = law or self._law
        for road in roads:
            now = self.start_time
            self._emit_road(road, now, self.end_time if until is None else until, law, out)

    def _emit_road(self, road, now, until, law, out):
        while now < until:
            mu, sigma = law(road, now)
            gap = max(0, self.streams.np.normal(mu, sigma))
            t = int(now + gap)
            if t >= until or t >= self.end_time:
                break
            out.append(self._entry(road, t))
            now = t
        return now

    def _entry(self, road, t):
        route = self.graph.trip(road, self.turn_probabilities, self.streams.py, self.max_trip_length,
                                self.closed_roads)
        return {"vehicle": self.vehicle_data, "route": route, "interval": 1.0, "startTime": t, "endTime": t}

    # -- public -------------------------------------------------------------------
    def flows(self):
        """The flow entries, sorted by start time (what generate_flows writes)."""
        out = []
        self._emit(self.graph.entries, out)
        return sorted(out, key=lambda e: e["startTime"])

    def _stem(self):
        return f"{self.scenario}__gaussian_{int((self.end_time - self.start_time) / self.inter_mu)}_flows.json"

    def _filename(self, replicate_no):
        name = self._stem()
        extra = getattr(self.config, "_additional_config", {}) or {}
        if "replicate_no" in extra:
            name = f"{extra['replicate_no']}__{name}"
        if replicate_no is not None:
            name = f"{replicate_no}__{name}"
        return name

    def generate_flows(self, filepath, replicate_no=None):
        path = os.path.join(filepath, self._filename(replicate_no))
        with open(path, "w") as f:
            json.dump(self.flows(), f, indent=4)
        return path


class LinkDisruptedCityFlowTripGenerator(CityFlowTripGenerator):
    """Some interior roads are closed to routing (trip_generator.py:289-388)."""

    def __init__(self, scenario, start_time, end_time, inter_mu, inter_sigma, disruption_ratio=0.1, **kwargs):
        super().__init__(scenario, start_time, end_time, inter_mu, inter_sigma, **kwargs)
        self.disruption_ratio = disruption_ratio
        boundary = set(self.graph.entries + self.graph.exits)
        interior = [road["id"] for road in self.rn.roads if road["id"] not in boundary]
        self.streams.py.shuffle(interior)
        self.disrupted_links = set(interior[: int(disruption_ratio * self.rn.n_agents)])
        self.closed_roads = self.disrupted_links

    def _stem(self):
        rate = int((self.end_time - self.start_time) / self.inter_mu)
        return f"{self.scenario}__dr_{self.disruption_ratio}__gaussian_{rate}_flows.json"


class FlowDisruptedCityFlowTripGenerator(CityFlowTripGenerator):
    """Ten-minute bursts of 4x demand on a share of the entry roads
    (trip_generator.py:391-489)."""

    burst_start_min, burst_start_max, burst_duration, burst_multiplier = 0, 20, 10, 4

    def __init__(self, scenario, start_time, end_time, inter_mu, inter_sigma, disruption_ratio=0.1, **kwargs):
        super().__init__(scenario, start_time, end_time, inter_mu, inter_sigma, **kwargs)
        self.disruption_ratio = disruption_ratio
        picked = self.streams.py.sample(self.graph.entries, int(len(self.graph.entries) * disruption_ratio))
        self.disrupted_links = set(picked)
        self.burst_timings = {}
        for road in self.disrupted_links:
            t0 = self.streams.py.randint(self.burst_start_min * 60, self.burst_start_max * 60)
            self.burst_timings[road] = (t0, t0 + self.burst_duration * 60)

    def _law(self, road, now):
        lo, hi = self.burst_timings.get(road, (float("inf"), float("inf")))
        if road in self.disrupted_links and lo <= now < hi:
            return self.inter_mu / self.burst_multiplier, self.inter_sigma / self.burst_multiplier
        return self.inter_mu, self.inter_sigma

    def _stem(self):
        rate = int((self.end_time - self.start_time) / self.inter_mu)
        return f"{self.scenario}__fd_{self.disruption_ratio}__gaussian_{rate}_flows.json"


class IntervalCityFlowTripGenerator(CityFlowTripGenerator):
    """The mean inter-arrival time changes every `interval_duration` seconds along a
    bell-shaped profile (trip_generator.py:492-554)."""

    def flows(self, interval_duration=360, shape=1.5, scale=300):
        means = weibull_flow_rates(self.streams, shape, scale, self.inter_mu, int(3600 / interval_duration))
        n_intervals = (self.end_time - self.start_time) // interval_duration
        out = []
        for road in self.graph.entries:
            now = self.start_time
            for k in range(n_intervals):
                edge = self.start_time + (k + 1) * interval_duration
                now = self._emit_road(road, now, edge, lambda r, t, m=means[k]: (m, self.inter_sigma), out)
        return sorted(out, key=lambda e: e["startTime"])

    def _stem(self):
        return f"{self.scenario}__interval_{int((self.end_time - self.start_time) / self.inter_mu)}_flows.json"

    def _filename(self, replicate_no):
        name = self._stem()
        extra = getattr(self.config, "_additional_config", {}) or {}
        if "replicate_no" in extra:
            return f"{extra['replicate_no']}__{name}"
        return f"{replicate_no}__{name}"

    def generate_flows(self, filepath, replicate_no, interval_duration=360, shape=1.5, scale=300):
        path = os.path.join(filepath, self._filename(replicate_no))
        with open(path, "w") as f:
            json.dump(self.flows(interval_duration, shape, scale), f, indent=4)
        return path


class VariableDemandTripGenerator(CityFlowTripGenerator):
    """Per-road inter-arrival laws scaled by a ten-minute demand profile
    (trip_generator.py:557-669).  Like the reference it does not seed: pass
    `seed=` to get a reproducible file, or seed numpy / random globally."""

    demand_profile = [0.5, 0.6, 0.75, 1.0, 1.0, 0.5, 0.5, 0.3, 0.3, 1e-6]

    def __init__(self, scenario, start_time, end_time, inter_mus, inter_sigmas, edge_weights, disrupted=False,
                 turn_probs=(1 / 3, 1 / 3, 1 / 3), seed=None, **kwargs):
        super().__init__(scenario, start_time, end_time, None, None, seed=seed, edge_weights=edge_weights,
                         turn_probs=turn_probs, **kwargs)
        self.inter_mus, self.inter_sigmas = inter_mus, inter_sigmas

    def _law(self, road, now):
        share = self.demand_profile[(now % 3600) // 600]
        return self.inter_mus[road] / share, self.inter_sigmas[road] / share

    def flows(self):
        out = []
        self._emit([road for road in self.graph.entries if road in self.inter_mus], out)
        return sorted(out, key=lambda e: e["startTime"])

    def _stem(self):
        rate = (self.end_time - self.start_time) / np.mean(list(self.inter_mus.values()))
        return f"{self.scenario}__gaussian_{int(rate)}_flows.json"


class CityFlowOneWayTripGenerator(CityFlowTripGenerator):
    """Straight-through traffic only, with separate demand on north-south and
    east-west entry roads (trip_generator.py:672-802)."""

    def __init__(self, scenario, start_time, end_time, inter_mu_ns, inter_sigma_ns, inter_mu_ew, inter_sigma_ew,
                 disrupted=False, edge_weights=None, **kwargs):
        kwargs.pop("turn_probs", None)
        super().__init__(scenario, start_time, end_time, inter_mu_ns, inter_sigma_ns, edge_weights=edge_weights,
                         turn_probs=(0.0, 0.0, 1.0), **kwargs)
        self.inter_mu_ns, self.inter_sigma_ns = inter_mu_ns, inter_sigma_ns
        self.inter_mu_ew, self.inter_sigma_ew = inter_mu_ew, inter_sigma_ew

    @staticmethod
    def _heading(road):
        a, b = road["points"][0], road["points"][-1]
        if a["x"] == b["x"] and a["y"] > b["y"]:
            return "ns"
        if a["y"] == b["y"] and a["x"] > b["x"]:
            return "ew"
        return None

    def flows(self):
        entries = set(self.graph.entries)
        out = []
        for heading, mu, sigma in (("ns", self.inter_mu_ns, self.inter_sigma_ns),
                                   ("ew", self.inter_mu_ew, self.inter_sigma_ew)):
            roads = [road["id"] for road in self.rn.roads if road["id"] in entries and self._heading(road) == heading]
            self._emit(roads, out, law=lambda r, t, m=mu, s=sigma: (m, s))
        return sorted(out, key=lambda e: e["startTime"])

    def _stem(self):
        span = self.end_time - self.start_time
        return f"{self.scenario}__oneway_{int(span / self.inter_mu_ns)}_NS_{int(span / self.inter_mu_ew)}_EW_flows.json"


def turn_of(prev_road, cur_road):
    """Turn between two grid roads from their ids `road_<col>_<row>_<dir>`
    (trip_generator.py:25-42)."""
    a, b = prev_road.split("_")[1], cur_road.split("_")[1]
    if a == b:
        return "go_straight"
    return "turn_right" if int(b) > int(a) else "turn_left"


class CityFlowRandomizedTripGenerator(CityFlowTripGenerator):
    """Resamples an existing flow file: per entry road the empirical inter-arrival
    statistics and the empirical route distribution (trip_generator.py:805-1020)."""

    def __init__(self, scenario, start_time, end_time, seed=None, config=None, flow_data=None, **kwargs):
        self.scenario = scenario
        self.config = config if config is not None else Config(scenario, **kwargs)
        self.rn = load_roadnet(self.config)
        self.start_time, self.end_time = start_time, end_time
        self.streams = _Streams(seed)
        self.graph = RoadGraph(self.rn, None)
        self.edge_weights = self.graph.weights
        self.closed_roads = ()
        if flow_data is None:
            from pytsc_amd.flows import load_flow_file

            flow_data = load_flow_file(self.config.flow_files[0])
        ratios, self.flow_info, self.stored_routes, self.route_proportions = self.get_flow_rates(flow_data)
        self.turn_probabilities = [ratios[t] for t in TURNS]
        self.max_trip_length = max(v["max_route_length"] for v in self.flow_info.values())

    @staticmethod
    def get_flow_rates(flow_data):
        per_road = {}
        turn_counts = dict.fromkeys(("go_straight", "turn_right", "turn_left"), 0)
        all_starts = []
        for veh in flow_data:
            route = veh["route"]
            rec = per_road.setdefault(route[0], {"starts": [], "lengths": [], "routes": [], "counts": {}})
            rec["counts"][tuple(route)] = rec["counts"].get(tuple(route), 0) + 1
            if route not in rec["routes"]:
                rec["routes"].append(route)
            rec["starts"].append(veh["startTime"])
            rec["lengths"].append(len(route))
            all_starts.append(veh["startTime"])
            for a, b in zip(route[:-1], route[1:]):
                turn_counts[turn_of(a, b)] += 1
        n_turns = sum(turn_counts.values())
        ratios = {k: v / n_turns for k, v in turn_counts.items()}
        hours = (max(all_starts) - min(all_starts)) / 3600
        info, stored, shares = {}, {}, {}
        for road, rec in per_road.items():
            gaps = np.diff(sorted(rec["starts"]))
            info[road] = {
                "flow_rate": len(rec["starts"]) / hours,
                "arrival_diff_mean": np.mean(gaps),
                "arrival_diff_std": np.std(gaps),
                "mean_route_length": np.mean(rec["lengths"]),
                "min_route_length": np.min(rec["lengths"]),
                "max_route_length": np.max(rec["lengths"]),
            }
            stored[road] = rec["routes"]
            total = sum(rec["counts"].values())
            shares[road] = {route: c / total for route, c in rec["counts"].items()}
        return ratios, info, stored, shares

    def flows(self, flow_type="low"):
        out = []
        for road in self.graph.entries:
            if road not in self.flow_info:
                continue
            stats = self.flow_info[road]
            routes = self.stored_routes[road]
            weights = [self.route_proportions[road][tuple(rt)] for rt in routes]
            now = self.start_time
            while now < self.end_time:
                if flow_type == "low":
                    mean = stats["arrival_diff_mean"] + self.streams.np.uniform(-0.5, 0.0)
                elif flow_type == "medium":
                    mean = stats["arrival_diff_mean"] * 0.75
                elif flow_type == "high":
                    mean = stats["arrival_diff_mean"] * 0.5
                else:
                    raise ValueError("Invalid flow type")
                gap = max(0, self.streams.np.normal(mean, stats["arrival_diff_std"]))
                t = int(now + gap)
                if t >= self.end_time:
                    break
                route = self.streams.py.choices(routes, weights=weights, k=1)[0]
                out.append({"vehicle": self.vehicle_data, "route": route, "interval": 1.0, "startTime": t,
                            "endTime": t})
                now = t
        return sorted(out, key=lambda e: e["startTime"])

    def _stem(self):
        return f"{self.scenario}__gaussian_flows.json"

    def generate_flows(self, filepath, flow_type="low", replicate_no=None):
        out_dir = os.path.join(filepath, flow_type)
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, self._filename(replicate_no))
        with open(path, "w") as f:
            json.dump(self.flows(flow_type), f, indent=4)
        return path
