"""Scenarios of the cross-logic known-answer tests (tests/test_cross_kat.py on the CPU oracle, tests/test_hip_cross_kat.py
through the C-ABI): one signalised intersection of the shipped grid family (width 20, three lanes per road, Hermite
lane-links) with a light-phase table that has an ALL-GREEN phase, so that any two movements can meet in the intersection.
What is checked is behaviour the engine's conflict rules (notifyCross, Cross::canPass) imply
whatever the implementation: who keeps its free-flow trajectory, who holds back where, mutual exclusion at every cross,
no deadlock.  Call sites of the reference that see the result: pytsc/backends/cityflow/retriever.py:94-98."""
import numpy as np

from pytsc_amd.config import Config
from pytsc_amd.flows import DEFAULT_VEHICLE
from pytsc_amd.grid import generate_grid
from pytsc_amd.scenario import build_tables

LEN = DEFAULT_VEHICLE["length"]
YIELD = 5.0  # default yieldDistance of a flow that names none
# approach roads of intersection_1_1 by where the vehicle comes FROM, exit roads by heading
FROM = {"W": "road_0_1_0", "S": "road_1_0_1", "E": "road_2_1_2", "N": "road_1_2_3"}
TO = {"E": "road_1_1_0", "N": "road_1_1_1", "W": "road_1_1_2", "S": "road_1_1_3"}


def net_all_green():
    net = generate_grid(1, 1)
    for it in net["intersections"]:
        if not it["virtual"]:
            it["trafficLight"]["lightphases"] = [
                {"time": 5, "availableRoadLinks": []},                # raw 0: all red (the yellow slot)
                {"time": 30, "availableRoadLinks": list(range(12))},  # raw 1: every movement
            ]
    return net


def entry(frm, to, t, vehicle=None):
    return {"vehicle": dict(vehicle or DEFAULT_VEHICLE), "route": [FROM[frm], TO[to]], "interval": 1.0, "startTime": t,
            "endTime": t}


def tables(entries, n_replicas=1, **hip):
    cfg = Config("syn_1x1", signal={"action_space": "phase_selection", "round_robin": False},
                 hip=dict({"roadnet_json": net_all_green(), "flow_entries": [entries], "sim_length": 400}, **hip))
    return build_tables(cfg, n_replicas=n_replicas)


SCENARIOS = {
    # type priority: go_straight (3) beats turn_left (2) -- opposing approaches, both green
    "straight_vs_left": [entry("E", "W", 0), entry("W", "N", 0)],
    # equal types, two straights at right angles, simultaneous: the one nearer to the cross (14 m against 26 m) is there sooner
    "straight_vs_straight": [entry("W", "E", 0), entry("S", "N", 0)],
    # the same with the farther one given a head start of 3 s: now it is there sooner
    "straight_vs_straight_headstart": [entry("W", "E", 0), entry("S", "N", 3)],
    # turn_right (1) against go_straight (3) into the same road
    "right_vs_straight_merge": [entry("S", "E", 0), entry("W", "E", 0)],
    # four left-turners at once, one from every approach: every pair of their paths crosses -- a ring of mutual blockers
    # that only the blocker-chain cycle check resolves
    "four_lefts": [entry("W", "N", 0), entry("S", "W", 0), entry("E", "S", 0), entry("N", "E", 0)],
    # a platoon of three straights against a left-turner: the left-turner waits for a gap, vehicles past the cross do
    # not hold it back (foe past the cross: d2 + length < 0)
    "platoon_vs_left": [entry("E", "W", 0), entry("E", "W", 2), entry("E", "W", 4), entry("W", "N", 0)],
}


class Track:
    """Per-vehicle trace of (drivable, dist, speed) per engine step + the lane-link each vehicle used."""

    def __init__(self, tb):
        self.tb = tb
        self.nl = tb.n_lanes
        self.rows = []  # per step: {vid: (drv, dist, speed)}
        self.link_of = {}

    def add(self, snap):
        d, dist, sp, vid = snap
        row = {int(v): (int(dd), float(x), float(s)) for dd, x, s, v in zip(d, dist, sp, vid)}
        for v, (dd, _, _) in row.items():
            if dd >= self.nl:
                self.link_of[v] = dd - self.nl
        self.rows.append(row)

    def pos_on_link(self, step, v):
        """Front position of v measured along its lane-link (negative: still on the approach lane; beyond the link's
        length: already on the exit lane), or None when it is not in the network."""
        if v not in self.rows[step] or v not in self.link_of:
            return None
        rn, k = self.tb.rn, self.link_of[v]
        dd, x, _ = self.rows[step][v]
        lk = rn.links[k]
        if dd == self.nl + k:
            return x
        if dd == lk["start"]:
            return x - rn.lane_length[lk["start"]]
        if dd == lk["end"]:
            return lk["length"] + x
        return None


def crosses_between(tb, ka, kb):
    """(distance on ka, distance on kb) of the cross of lane-links ka and kb, or None."""
    for (da, _, other, db) in tb.rn.link_crosses[ka]:
        if other == kb:
            return da, db
    return None


def check_mutual_exclusion(track):
    """At no engine step do two vehicles cover the cross of their lane-links at the same time."""
    vids = sorted(track.link_of)
    worst = None
    for step in range(len(track.rows)):
        for i, a in enumerate(vids):
            for b in vids[i + 1:]:
                ka, kb = track.link_of[a], track.link_of[b]
                if ka == kb:
                    continue
                c = crosses_between(track.tb, ka, kb)
                pa, pb = track.pos_on_link(step, a), track.pos_on_link(step, b)
                if c is None or pa is None or pb is None:
                    continue
                on_a = pa - LEN < c[0] < pa
                on_b = pb - LEN < c[1] < pb
                if on_a and on_b:
                    worst = (step, a, b, pa, pb, c)
    return worst


def free_flow_positions(n_steps, v_cap=11.11, acc=2.0):
    x, v, out = 0.0, 0.0, []
    for _ in range(n_steps):
        v_new = min(v_cap, v + acc)
        x += (v + v_new) / 2
        v = v_new
        out.append(x)
    return np.array(out)


def tie_tables(n_replicas=1, **hip):
    """`straight_vs_straight` with the cross of two perpendicular through paths moved to 20 m on BOTH lane-links (it lies at
    26 m / 14 m): two vehicles that start together then reach it in the same number of steps, enter their lane-links at the
    same time and stand at the same distance from it -- the last tie-break of Cross::canPass decides
    (tsc_flows_t::veh_priority).  The lists stay sorted by distance (the engine walks them in order); the geometry is no
    longer that of the intersection, which the conflict logic does not care about."""
    from pytsc_amd.roadnet import load_roadnet

    cfg = Config("syn_1x1", signal={"action_space": "phase_selection", "round_robin": False},
                 hip=dict({"roadnet_json": net_all_green(), "flow_entries": [SCENARIOS["straight_vs_straight"]],
                           "sim_length": 400}, **hip))
    rn = load_roadnet(cfg)
    moved = 0
    for k, cs in enumerate(rn.link_crosses):
        out = []
        for c in cs:
            if abs(c[0] + c[3] - 40.0) < 1e-9 and abs(abs(c[0] - c[3]) - 12.0) < 1e-9:
                c = (20.0,) + tuple(c[1:3]) + (20.0,) + tuple(c[4:])
                moved += 1
            out.append(tuple(c))
        rn.link_crosses[k] = sorted(out, key=lambda t: (t[0], t[1]))
    assert moved == 8
    return build_tables(cfg, n_replicas=n_replicas, roadnet=rn)


def who_braked(track):
    """vehicle ids whose speed ever dropped while they were in the network (a vehicle that is never held back accelerates
    monotonically up to its cap)."""
    out = set()
    for a, b in zip(track.rows, track.rows[1:]):
        for v in a:
            if v in b and b[v][2] < a[v][2] - 1e-9:
                out.add(v)
    return out
