"""Flow JSON -> per-vehicle spawn schedule + static drivable routes; demand generators.

On-disk format: the reference's CityFlow flow files (SURVEY.md appendix C;
producer pytsc/backends/cityflow/trip_generator.py:252-287): a list of
``{vehicle{...}, route[road ids], interval, startTime, endTime}``.

A vehicle's path through drivables (lane, lane-link, lane, ..., lane) is a pure
function of its road route on these networks, so it is resolved here once
(first lane = a lane of route[0] linked to route[1];
next lane-link = closest end-lane index that still connects to the road after
next) and the device only carries (route id, position in route).
"""
import json

import numpy as np

TMPL_KEYS = (
    "length",
    "width",
    "maxPosAcc",
    "maxNegAcc",
    "usualPosAcc",
    "usualNegAcc",
    "minGap",
    "maxSpeed",
    "headwayTime",
)
DEFAULT_YIELD_DISTANCE = 5.0
DEFAULT_TURN_SPEED = 8.3333

DEFAULT_VEHICLE = {  # reference template: trip_generator.py:47-57
    "length": 5.0,
    "width": 2.0,
    "maxPosAcc": 2.0,
    "maxNegAcc": 4.5,
    "usualPosAcc": 2.0,
    "usualNegAcc": 4.5,
    "minGap": 2.5,
    "maxSpeed": 11.11,
    "headwayTime": 1.5,
}


def load_flow_file(path):
    with open(path, "r") as f:
        return json.load(f)


def template_vector(veh, interval):
    v = [float(veh[k]) for k in TMPL_KEYS]
    v.append(float(veh.get("yieldDistance", DEFAULT_YIELD_DISTANCE)))
    v.append(float(veh.get("turnSpeed", DEFAULT_TURN_SPEED)))
    max_speed, usual_neg = v[7], v[5]
    v.append(max_speed * max_speed / usual_neg / 2 + max_speed * interval * 2)
    return tuple(v)


def _select_link(rn, cur_lane, links):
    """Closest end-lane index to the current lane index (first minimum)."""
    best, best_diff = None, None
    ci = int(rn.lane_idx_in_road[cur_lane])
    for l in links:
        d = abs(int(rn.lane_idx_in_road[rn.links[l]["end"]]) - ci)
        if best_diff is None or d < best_diff:
            best, best_diff = l, d
    return best


def route_to_drivables(rn, roads):
    """Road-id route -> [lane, link, lane, ...] drivable indices, or None if
    the route is not drivable (the engine drops such vehicles)."""
    for r in roads:
        if r not in rn.road_by_id:
            return None
    first_lanes = rn.road_lanes(roads[0])
    if len(roads) == 1:
        cur = first_lanes[0]
    else:
        cands = [l for l in first_lanes if rn.links_to_road(l, roads[1])]
        if not cands:
            return None
        # the engine draws uniformly among candidates; every shipped network has
        # exactly one (left / straight / right lanes), we take the first
        cur = cands[0]
    seq = [cur]
    n = len(roads)
    for i in range(n - 1):
        links = rn.links_to_road(cur, roads[i + 1])
        if i == n - 2:
            sel = _select_link(rn, cur, links)
        else:
            sel = None
            cand = list(links)
            while cand:
                s = _select_link(rn, cur, cand)
                if rn.links_to_road(rn.links[s]["end"], roads[i + 2]):
                    sel = s
                    break
                cand.remove(s)
        if sel is None:
            return None
        cur = rn.links[sel]["end"]
        seq.append(rn.n_lanes + sel)
        seq.append(cur)
    return seq


class FlowSet:
    """One demand file expanded to vehicles, sorted in creation order."""

    def __init__(self, veh_route, veh_start, veh_tmpl, veh_entry=None, veh_k=None):
        self.veh_route = np.asarray(veh_route, dtype=np.int32)
        self.veh_start = np.asarray(veh_start, dtype=np.int32)
        self.veh_tmpl = np.asarray(veh_tmpl, dtype=np.int32)
        # host side only: the flow entry a vehicle came from and its number within it -- CityFlow names vehicles
        # flow_<entry>_<k>, the tables number them in creation order
        self.veh_entry = None if veh_entry is None else np.asarray(veh_entry, dtype=np.int32)
        self.veh_k = None if veh_k is None else np.asarray(veh_k, dtype=np.int32)

    @property
    def n_veh(self):
        return len(self.veh_route)


class FlowCompiler:
    """Accumulates routes/templates across flow sets of one network."""

    def __init__(self, rn, interval=1.0):
        self.rn = rn
        self.interval = float(interval)
        self.routes = []
        self.route_index = {}
        self.tmpls = []
        self.tmpl_index = {}
        self.sets = []
        self._compiled = {}  # id(entries) -> (entries, FlowSet): a demand given several times is expanded once

    def _route_id(self, roads):
        key = tuple(roads)
        if key not in self.route_index:
            seq = route_to_drivables(self.rn, roads)
            if seq is None:
                self.route_index[key] = -1
            else:
                self.route_index[key] = len(self.routes)
                self.routes.append(seq)
        return self.route_index[key]

    def _tmpl_id(self, veh):
        key = template_vector(veh, self.interval)
        if key not in self.tmpl_index:
            self.tmpl_index[key] = len(self.tmpls)
            self.tmpls.append(key)
        return self.tmpl_index[key]

    def add(self, entries, sim_len):
        """Expand flow entries to vehicles created at engine steps < sim_len.

        Per entry: a counter `now` starts at
        `interval`; at each step with startTime <= t (<= endTime unless -1),
        while now >= interval a vehicle is created and now -= interval; then
        now += dt.  Vehicles of one step are ordered by entry index.
        """
        hit = self._compiled.get((id(entries), sim_len))
        if hit is not None and hit[0] is entries:
            self.sets.append(hit[1])
            return hit[1]
        dt = self.interval
        per_step = []
        for fi, e in enumerate(entries):
            rid = self._route_id(e["route"])
            if rid < 0:
                continue  # invalid route: flow is disabled by the engine
            tid = self._tmpl_id(e["vehicle"])
            start = e["startTime"]
            end = e["endTime"]
            ivl = float(e["interval"])
            now = ivl
            # engine time at step s is s*dt; the flow keeps its own clock
            s = 0
            cur = 0.0
            cnt = 0
            # fast path: single-shot entries (all shipped files)
            if end == start and ivl >= dt and float(start) == int(start) and dt == 1.0:
                if start < sim_len:
                    per_step.append((int(start), fi, 0, rid, tid))
                continue
            while s < sim_len:
                if end != -1 and cur > end:
                    break
                if cur >= start:
                    while now >= ivl:
                        per_step.append((s, fi, cnt, rid, tid))
                        cnt += 1
                        now -= ivl
                    now += dt
                cur += dt
                s += 1
        per_step.sort(key=lambda t: (t[0], t[1], t[2]))
        fs = FlowSet(
            [t[3] for t in per_step], [t[0] for t in per_step], [t[4] for t in per_step],
            [t[1] for t in per_step], [t[2] for t in per_step],
        )
        self.sets.append(fs)
        self._compiled[(id(entries), sim_len)] = (entries, fs)
        return fs


# --------------------------------------------------------------------------
# generators (reference format; SURVEY.md §8(d) config 2: Poisson arrivals)
# --------------------------------------------------------------------------
def fringe_roads(rn):
    """(incoming, outgoing) boundary roads (trip_generator.py:150-180)."""
    inc, out = [], []
    for road in rn.roads:
        sv = rn.inter_by_id[road["startIntersection"]]["virtual"]
        ev = rn.inter_by_id[road["endIntersection"]]["virtual"]
        if sv and not ev:
            inc.append(road["id"])
        if ev and not sv:
            out.append(road["id"])
    return inc, out


def _road_turns(rn):
    conn = {}
    for it in rn.intersections:
        for rl in it.get("roadLinks", []):
            conn.setdefault(rl["startRoad"], {})[rl["type"]] = rl["endRoad"]
    return conn


def random_walk_route(rn, conn, out_set, start_road, rng, turn_probs, max_len):
    """Random-walk route in the spirit of trip_generator.py:200-249: weighted
    turn choice, no road twice, stop at a boundary-out road or a dead end."""
    turns = ("turn_left", "turn_right", "go_straight")
    while True:
        route = [start_road]
        cur = start_road
        while True:
            cands, w = [], []
            for i, t in enumerate(turns):
                nxt = conn.get(cur, {}).get(t)
                if nxt is not None and nxt not in route:
                    cands.append(nxt)
                    w.append(turn_probs[i])
            if not cands or sum(w) == 0:
                break
            w = np.asarray(w) / sum(w)
            nxt = cands[int(rng.choice(len(cands), p=w))]
            route.append(nxt)
            cur = nxt
            if nxt in out_set:
                break
        if 1 < len(route) <= max_len:
            return route


def _graph_diameter(rn):
    """Diameter of the directed intersection graph (longest shortest path), BFS."""
    adj = {}
    for road in rn.roads:
        adj.setdefault(road["startIntersection"], set()).add(road["endIntersection"])
        adj.setdefault(road["endIntersection"], set())
    best = 0
    for src in adj:
        dist = {src: 0}
        frontier = [src]
        while frontier:
            nxt = []
            for u in frontier:
                for v in adj[u]:
                    if v not in dist:
                        dist[v] = dist[u] + 1
                        nxt.append(v)
            frontier = nxt
        best = max(best, max(dist.values()))
    return best


def poisson_flows(rn, rate_veh_per_hour=700.0, horizon=3600, seed=0,
                  turn_probs=(0.1, 0.3, 0.6), vehicle=None, max_len=None):
    """Poisson demand: exponential inter-arrival times (mean 3600/rate s) on
    every boundary-in road, integer start second, random-walk routes."""
    rng = np.random.default_rng(seed)
    vehicle = dict(DEFAULT_VEHICLE if vehicle is None else vehicle)
    inc, out = fringe_roads(rn)
    conn = _road_turns(rn)
    out_set = set(out)
    if max_len is None:
        max_len = _graph_diameter(rn) + 2  # as trip_generator.py:95-108 (nx.diameter(G) + 2)
    mean = 3600.0 / rate_veh_per_hour
    flows = []
    for road in inc:
        t = 0.0
        while True:
            t += rng.exponential(mean)
            if t >= horizon:
                break
            route = random_walk_route(rn, conn, out_set, road, rng, turn_probs, max_len)
            st = int(t)
            flows.append(
                {"vehicle": vehicle, "route": route, "interval": 1.0,
                 "startTime": st, "endTime": st}
            )
    flows.sort(key=lambda e: e["startTime"])
    return flows


def mt19937_priorities(seed, n):
    """CityFlow draws a vehicle's `priority` -- the last tie-break of Cross::canPass -- from the engine's
    `std::mt19937 rnd` (seeded with the config `seed`, backends/cityflow/config.py:88-99) in the Vehicle constructor, i.e.
    in creation order (step, flow index, count: the order of the vehicle ids here).  The C++ standard fixes that
    generator bit for bit (10000th output of the default seed 5489: 4123659995); numpy's legacy seeding of an integer
    seed is the same init_genrand and `random_raw` its 32-bit outputs.  Returned as the `int` the engine stores them in.
    A HYPOTHESIS about an engine that is not here: if the engine's router draws first lanes from
    the same generator the streams interleave differently -- which a recorded trace would show (tests/test_cityflow_pin.py)."""
    raw = np.random.RandomState(int(seed) & 0xFFFFFFFF)._bit_generator.random_raw(int(n)) if n else np.zeros(0, np.uint64)
    return raw.astype(np.uint32).view(np.int32).copy()


def enumerate_walk_routes(rn, turn_probs=(0.1, 0.3, 0.6), max_len=None, limit=200000):
    """Every route `random_walk_route` can return from each boundary-in road, with the probability it returns it (the
    walk re-draws a route of one road or of more than max_len roads, so the probabilities of the accepted ones are
    renormalised).  {road: ([route tuples], probabilities)}; None when there are more than `limit` routes (large grids:
    the number grows exponentially with the diameter)."""
    inc, out = fringe_roads(rn)
    if len(inc) > 32:  # (larger than an 8x8 grid: hopeless, and finding that out route by route takes 15 s on the 16x16)
        return None
    conn, out_set = _road_turns(rn), set(out)
    if max_len is None:
        max_len = _graph_diameter(rn) + 2
    turns = ("turn_left", "turn_right", "go_straight")
    table, total = {}, 0
    for start in inc:
        routes, probs = [], []
        stack = [((start,), 1.0)]
        while stack:
            if total + len(routes) > limit or len(stack) > limit:  # (checked as the walk goes: a 16x16 grid has ~1e12 routes per road)
                return None
            route, p = stack.pop()
            cands = [(conn.get(route[-1], {}).get(t), turn_probs[i]) for i, t in enumerate(turns)]
            cands = [(n, w) for n, w in cands if n is not None and n not in route]
            tot = sum(w for _, w in cands)
            if not cands or tot == 0:  # dead end: the walk stops here
                if 1 < len(route) <= max_len:
                    routes.append(route)
                    probs.append(p)
                continue
            for n, w in cands:
                r2 = route + (n,)
                if n in out_set:
                    if len(r2) <= max_len:
                        routes.append(r2)
                        probs.append(p * w / tot)
                elif len(r2) <= max_len:
                    stack.append((r2, p * w / tot))
        total += len(routes)
        if total > limit:
            return None
        pr = np.asarray(probs, dtype=np.float64)
        table[start] = (routes, pr / pr.sum())
    return table


def poisson_flow_sets(fc, rate_veh_per_hour, horizon, set_ids, base_seed=0, turn_probs=(0.1, 0.3, 0.6), vehicle=None):
    """Many Poisson demands at once, straight into FlowCompiler `fc` (one FlowSet per entry of `set_ids`): the process of
    `poisson_flows` -- exponential inter-arrival times per boundary-in road, integer start second, a random-walk route
    per vehicle -- drawn with numpy arrays instead of a Python loop per vehicle: the routes of a road come from the
    enumerated walk distribution (`enumerate_walk_routes`), the arrivals of a set from one generator seeded with
    (base_seed, set id), so that set g is the same demand whichever batch -- or rank -- generates it.  Not the stream of
    `poisson_flows(seed=g)`: the same distribution, other draws.  Returns False (nothing added) when the network's walk
    routes cannot be enumerated (large grids)."""
    rn = fc.rn
    table = enumerate_walk_routes(rn, turn_probs)
    if table is None:
        return False
    vehicle = dict(DEFAULT_VEHICLE if vehicle is None else vehicle)
    tid = fc._tmpl_id(vehicle)
    inc, _ = fringe_roads(rn)
    rids, cums = [], []
    for road in inc:
        routes, pr = table[road]
        rids.append(np.asarray([fc._route_id(list(r)) for r in routes], dtype=np.int32))
        cums.append(np.cumsum(pr))
    mean = 3600.0 / rate_veh_per_hour
    lam = horizon / mean
    K = int(lam + 10.0 * np.sqrt(lam) + 32)
    n_roads = len(inc)
    for g in set_ids:
        rng = np.random.default_rng([int(base_seed), int(g)])
        t = np.cumsum(rng.exponential(mean, size=(n_roads, K)), axis=1)
        u = rng.random((n_roads, K))
        ok = t < horizon
        if ok[:, -1].any():
            raise RuntimeError("poisson_flow_sets: arrival buffer too short")
        route = np.empty((n_roads, K), dtype=np.int32)
        for i in range(n_roads):
            k = np.minimum(np.searchsorted(cums[i], u[i], side="right"), len(rids[i]) - 1)
            route[i] = rids[i][k]
        st = t[ok].astype(np.int64)  # road-major, ascending within a road: the order poisson_flows appends in
        order = np.argsort(st, kind="stable")  # ... and its stable sort by start second
        n = len(order)
        fs = FlowSet(route[ok][order], st[order], np.full(n, tid, np.int32), np.arange(n, dtype=np.int32), np.zeros(n, np.int32))
        fc.sets.append(fs)
    return True


# --------------------------------------------------------------------------
# compact on-disk form of a flow file (fixtures / generated scenarios)
# --------------------------------------------------------------------------
def compact_flows(entries):
    """{vehicles, routes, entries[[veh, route, start, end, interval]]}: the same
    information as the flow JSON with the repeated objects factored out."""
    vehicles, vidx, routes, ridx, rows = [], {}, [], {}, []
    for e in entries:
        vk = json.dumps(e["vehicle"], sort_keys=True)
        if vk not in vidx:
            vidx[vk] = len(vehicles)
            vehicles.append(e["vehicle"])
        rk = tuple(e["route"])
        if rk not in ridx:
            ridx[rk] = len(routes)
            routes.append(list(e["route"]))
        rows.append([vidx[vk], ridx[rk], e["startTime"], e["endTime"], e["interval"]])
    return {"vehicles": vehicles, "routes": routes, "entries": rows}


def expand_flows(compact):
    return [
        {"vehicle": compact["vehicles"][v], "route": compact["routes"][r],
         "interval": ivl, "startTime": st, "endTime": en}
        for v, r, st, en, ivl in compact["entries"]
    ]
