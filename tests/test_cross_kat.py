"""Known-answer tests of the lowest-confidence part of the engine restatement: the conflict logic inside an intersection
(`notifyCross` / `Cross::canPass`) and the cross GEOMETRY that feeds it (appendix B.1) -- the
latter is shared by the oracle and the HIP path (pytsc_amd/roadnet.py::_build_crosses), so HIP == oracle cannot see an
error there.  CPU only (oracle); tests/test_hip_cross_kat.py runs the same scenarios through the C-ABI.
The engine itself (CityFlow) is absent: "equals CityFlow" stays unpinned; these pin the restatement to the published rules."""
import numpy as np
import pytest

from cross_kat_scenarios import (LEN, SCENARIOS, YIELD, Track, check_mutual_exclusion, crosses_between, free_flow_positions,
                                 net_all_green, tables)
from pytsc_amd.roadnet import RoadNet


# ------------------------------------------------------------------ geometry
@pytest.fixture(scope="module")
def rn():
    return RoadNet(net_all_green())


def link_index(rn, start_lane, end_lane):
    for k, l in enumerate(rn.links):
        if rn.lane_ids[l["start"]] == start_lane and rn.lane_ids[l["end"]] == end_lane:
            return k
    raise KeyError((start_lane, end_lane))


def cross(rn, ka, kb):
    for (da, _, other, db) in rn.link_crosses[ka]:
        if other == kb:
            return da, db
    return None


def test_perpendicular_straights_cross_at_closed_form_distances(rn):
    """Width-20 intersection, 4 m lanes, lane 1 (centre line 6 m right of the road axis): the W->E through link runs
    along y = -6 from x = -20, the S->N one along x = +6 from y = -20, the N->S one along x = -6 from y = +20."""
    we = link_index(rn, "road_0_1_0_1", "road_1_1_0_1")
    sn = link_index(rn, "road_1_0_1_1", "road_1_1_1_1")
    ns = link_index(rn, "road_1_2_3_1", "road_1_1_3_1")
    ew = link_index(rn, "road_2_1_2_1", "road_1_1_2_1")
    assert rn.links[we]["length"] == 40.0 and rn.links[sn]["length"] == 40.0
    assert cross(rn, we, sn) == (26.0, 14.0) and cross(rn, sn, we) == (14.0, 26.0)
    assert cross(rn, we, ns) == (14.0, 26.0)
    assert cross(rn, ew, sn) == (14.0, 26.0) and cross(rn, ew, ns) == (26.0, 14.0)
    assert cross(rn, we, ew) is None  # parallel: never


def test_hermite_left_turn_against_a_straight_by_interpolation(rn):
    """A left-turn polyline (cubic Hermite sampled at t = 0, 0.1, ..., 1) against the straight E->W link y = +6: the
    crossing found by walking the polyline and interpolating linearly inside the segment that straddles the line."""
    left = link_index(rn, "road_0_1_0_0", "road_1_1_1_1")
    ew = link_index(rn, "road_2_1_2_1", "road_1_1_2_1")
    pts = np.array(rn.links[left]["points"])
    k = next(i for i in range(len(pts) - 1) if (pts[i, 1] - 6.0) * (pts[i + 1, 1] - 6.0) <= 0 and pts[i, 1] != pts[i + 1, 1])
    f = (6.0 - pts[k, 1]) / (pts[k + 1, 1] - pts[k, 1])
    seg = np.linalg.norm(np.diff(pts, axis=0), axis=1)
    d_left = seg[:k].sum() + f * seg[k]
    x_star = pts[k, 0] + f * (pts[k + 1, 0] - pts[k, 0])
    d_ew = 20.0 - x_star  # the E->W link starts at x = +20
    got = cross(rn, left, ew)
    assert got is not None
    assert got[0] == pytest.approx(d_left, abs=1e-9) and got[1] == pytest.approx(d_ew, abs=1e-9)
    assert cross(rn, ew, left) == (got[1], got[0])


def test_links_sharing_an_end_or_a_start_lane_touch_at_that_end(rn):
    """Two lane-links into the same lane meet at their last point, two out of the same lane at their first: the engine's
    segment test finds these too."""
    we = link_index(rn, "road_0_1_0_1", "road_1_1_0_1")          # W->E straight into lane road_1_1_0_1
    left_in = link_index(rn, "road_1_2_3_0", "road_1_1_0_1")     # from N turning left into the same lane
    c = cross(rn, we, left_in)
    assert c is not None
    assert c[0] == pytest.approx(rn.links[we]["length"], abs=1e-9) and c[1] == pytest.approx(rn.links[left_in]["length"], abs=1e-9)
    a, b = link_index(rn, "road_0_1_0_1", "road_1_1_0_0"), link_index(rn, "road_0_1_0_1", "road_1_1_0_2")  # same start lane
    c = cross(rn, a, b)
    assert c is not None and c[0] == pytest.approx(0.0, abs=1e-9) and c[1] == pytest.approx(0.0, abs=1e-9)


def test_every_pair_against_a_brute_force_segment_solver(rn):
    """All 36 x 35 / 2 pairs: the first crossing along link a by solving the 2 x 2 system of every segment pair with
    numpy, independent of roadnet.first_crossing; crosses are mirrored on both links and sorted by distance."""
    def brute(pa, pb):
        pa, pb = np.array(pa), np.array(pb)
        la = np.concatenate([[0], np.cumsum(np.linalg.norm(np.diff(pa, axis=0), axis=1))])
        lb = np.concatenate([[0], np.cumsum(np.linalg.norm(np.diff(pb, axis=0), axis=1))])
        for i in range(len(pa) - 1):
            for j in range(len(pb) - 1):
                d1, d2 = pa[i + 1] - pa[i], pb[j + 1] - pb[j]
                m = np.array([[d1[0], -d2[0]], [d1[1], -d2[1]]])
                if abs(np.linalg.det(m)) < 1e-9:
                    continue
                t, u = np.linalg.solve(m, pb[j] - pa[i])
                if -1e-9 <= t <= 1 + 1e-9 and -1e-9 <= u <= 1 + 1e-9:
                    return la[i] + t * (la[i + 1] - la[i]), lb[j] + u * (lb[j + 1] - lb[j])
        return None

    n = rn.n_links
    n_cross = 0
    for a in range(n):
        ds = [c[0] for c in rn.link_crosses[a]]
        assert ds == sorted(ds)
        assert len({c[2] for c in rn.link_crosses[a]}) == len(rn.link_crosses[a])  # one cross per pair
        for b in range(a + 1, n):
            exp, got = brute(rn.links[a]["points"], rn.links[b]["points"]), cross(rn, a, b)
            assert (exp is None) == (got is None), (a, b)
            if exp is not None:
                n_cross += 1
                assert got[0] == pytest.approx(exp[0], abs=1e-7) and got[1] == pytest.approx(exp[1], abs=1e-7), (a, b)
                assert cross(rn, b, a) == (got[1], got[0])
    assert n_cross == sum(len(c) for c in rn.link_crosses) // 2 and n_cross > 200


# ------------------------------------------------------------------ behaviour (oracle)
def run(oracle_lib, name, n_steps=200):
    tb = tables(SCENARIOS[name])
    o = oracle_lib.Oracle(tb)
    o.set_tl_phase(0, 0, 1)  # raw phase 1: every movement green
    tr = Track(tb)
    for t in range(n_steps):
        o.next_step(0)
        tr.add(o.snapshot(0))
        if t > 5 and not tr.rows[-1]:
            break
    assert not tr.rows[-1], "deadlock: vehicles left in the network"
    assert o.lib.tsco_engine_finished(o.h, 0) == len(SCENARIOS[name]) and o.errors() == 0
    assert check_mutual_exclusion(tr) is None
    return tb, tr


def speeds(tr, v, first=10):
    return [r[v][2] for r in tr.rows[first:] if v in r]


def unimpeded(tr, v, spawn_step=0):
    """The vehicle ran its whole trip on the free-flow trajectory v_t = min(11.11, 2 t) (its lane-link is straight)."""
    xs = free_flow_positions(len(tr.rows))
    tb = tr.tb
    rn, k = tb.rn, tr.link_of[v]
    for step, row in enumerate(tr.rows):
        p = tr.pos_on_link(step, v)
        if p is None or step < spawn_step:
            continue
        total = p + rn.lane_length[rn.links[k]["start"]]  # metres travelled since the spawn
        if abs(total - xs[step - spawn_step]) > 1e-6:
            return False
    return True


def held_before_cross(tr, waiting, other):
    """While `other` has not cleared the cross of the two lane-links, `waiting` stays yieldDistance short of it."""
    c = crosses_between(tr.tb, tr.link_of[waiting], tr.link_of[other])
    assert c is not None
    for step in range(len(tr.rows)):
        pw, po = tr.pos_on_link(step, waiting), tr.pos_on_link(step, other)
        if pw is None or po is None:
            continue
        if po - LEN < c[1]:  # the other's rear has not passed the cross yet
            assert pw <= c[0] - YIELD + 1e-9, (step, pw, po, c)
    return c


def test_type_priority_straight_beats_left(oracle_lib):
    tb, tr = run(oracle_lib, "straight_vs_left")
    assert tb.rn.links[tr.link_of[0]]["type"] == 3 and tb.rn.links[tr.link_of[1]]["type"] == 2
    assert unimpeded(tr, 0)
    assert min(speeds(tr, 1)) < 3.0  # the left-turner brakes for it ...
    held_before_cross(tr, waiting=1, other=0)  # ... and waits yieldDistance before the cross until the straight is through


def test_equal_types_nearer_vehicle_first(oracle_lib):
    """Two straights at right angles, same step: 14 m against 26 m to the cross -> fewer steps to reach it -> goes first."""
    tb, tr = run(oracle_lib, "straight_vs_straight")
    c = crosses_between(tb, tr.link_of[0], tr.link_of[1])
    assert c == (26.0, 14.0)
    assert unimpeded(tr, 1) and not unimpeded(tr, 0)
    held_before_cross(tr, waiting=0, other=1)


def test_equal_types_head_start_reverses_the_order(oracle_lib):
    tb, tr = run(oracle_lib, "straight_vs_straight_headstart")
    assert unimpeded(tr, 0) and not unimpeded(tr, 1, spawn_step=3)
    held_before_cross(tr, waiting=1, other=0)


def test_movements_that_do_not_cross_do_not_interact(oracle_lib):
    """Right turn into the outer lane, straight into the middle lane of the same road: no cross, nobody yields (the
    right-turner is bounded by the turn speed only)."""
    tb, tr = run(oracle_lib, "right_vs_straight_merge")
    assert crosses_between(tb, tr.link_of[0], tr.link_of[1]) is None
    assert unimpeded(tr, 1)
    assert min(speeds(tr, 0)) == pytest.approx(25.0 / 3.0, abs=1e-3)  # turnSpeed 8.3333


def test_ring_of_four_left_turners_does_not_deadlock(oracle_lib):
    """Every left-turner's path crosses its two neighbours': by the pairwise rule each waits for another, in a ring.  The
    blocker-chain cycle check lets them through; symmetric inputs give symmetric trajectories."""
    tb, tr = run(oracle_lib, "four_lefts")
    assert len(tr.link_of) == 4 and all(tb.rn.links[k]["type"] == 2 for k in tr.link_of.values())
    pairs = [(a, b) for a in range(4) for b in range(a + 1, 4) if crosses_between(tb, tr.link_of[a], tr.link_of[b])]
    assert len(pairs) >= 4
    s0 = speeds(tr, 0, first=0)
    for v in (1, 2, 3):
        assert speeds(tr, v, first=0) == pytest.approx(s0, abs=1e-9)
    assert len(tr.rows) < 70  # nobody stood still for long


def test_left_turner_waits_for_a_gap_in_a_platoon(oracle_lib):
    """Three straights two seconds apart against one left-turner: it lets the platoon's head pass, is not held back by
    a vehicle whose rear is past the cross, and the platoon's head is never slowed."""
    tb, tr = run(oracle_lib, "platoon_vs_left")
    left = next(v for v, k in tr.link_of.items() if tb.rn.links[k]["type"] == 2)
    heads = [v for v, k in tr.link_of.items() if tb.rn.links[k]["type"] == 3]
    assert len(heads) == 3 and unimpeded(tr, min(heads))
    assert min(speeds(tr, left)) < 3.0
    held_before_cross(tr, waiting=left, other=min(heads))
