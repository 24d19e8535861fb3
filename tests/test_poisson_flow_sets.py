"""The array generator behind the bench's default workload -- one Poisson demand per replica, `poisson_flow_sets`
(pytsc_amd/flows.py, reached through `n_flow_sets_batch`, scenario.py) -- against the law it claims and against the
per-vehicle generator `poisson_flows` whose process it restates (exponential inter-arrival,
mean 3600 / 700 s per entry road, integer start second, random-walk routes as trip_generator.py:200-249).  VERDICT r4
"what's weak" 2: the driver-run workload's own inputs had never met a test."""
import numpy as np
import pytest

from pytsc_amd.flows import FlowCompiler, enumerate_walk_routes, fringe_roads, poisson_flow_sets, poisson_flows
from pytsc_amd.grid import generate_grid
from pytsc_amd.roadnet import RoadNet

RATE, HORIZON, N = 700.0, 3600, 32


@pytest.fixture(scope="module")
def rn():
    return RoadNet(generate_grid(4, 4))


@pytest.fixture(scope="module")
def batch(rn):
    fc = FlowCompiler(rn)
    assert poisson_flow_sets(fc, RATE, HORIZON, range(N), base_seed=0)
    return fc


def _roads_of(fc, rn, rid):
    """road ids of compiled route `rid` (the lanes of its drivable sequence)"""
    return tuple(rn.lane_ids[d].rsplit("_", 1)[0] for d in fc.routes[rid] if d < rn.n_lanes)


def test_arrivals_per_entry_road_follow_the_rate(batch, rn):
    inc, _ = fringe_roads(rn)
    lam = RATE * HORIZON / 3600.0
    tot = 0
    for fs in batch.sets:
        first = np.asarray([_roads_of(batch, rn, r)[0] for r in np.unique(fs.veh_route)])
        road_of = dict(zip(np.unique(fs.veh_route).tolist(), first.tolist()))
        counts = {}
        for r in fs.veh_route.tolist():
            counts[road_of[r]] = counts.get(road_of[r], 0) + 1
        assert set(counts) == set(inc)
        for road, c in counts.items():
            assert abs(c - lam) < 4.0 * np.sqrt(lam), (road, c)
        tot += fs.n_veh
    # the pooled count: 32 x 16 roads x 700, within 4 sigma as well
    assert abs(tot - N * len(inc) * lam) < 4.0 * np.sqrt(N * len(inc) * lam)


def test_start_seconds_are_sorted_and_ties_keep_road_order(batch, rn):
    """`poisson_flows` appends road by road and sorts stably by start second: within one second the vehicles come in the
    order of the entry roads.  The array generator must order them the same way (vehicle ids = creation order)."""
    inc, _ = fringe_roads(rn)
    pos = {road: i for i, road in enumerate(inc)}
    for fs in batch.sets[:8]:
        st = fs.veh_start
        assert np.all(np.diff(st) >= 0) and st.min() >= 0 and st.max() < HORIZON
        road_idx = np.asarray([pos[_roads_of(batch, rn, r)[0]] for r in fs.veh_route.tolist()])
        same = np.diff(st) == 0
        assert np.all(np.diff(road_idx)[same] >= 0)
        assert np.array_equal(fs.veh_entry, np.arange(fs.n_veh))  # one flow entry per vehicle, as the reference's files


def test_route_law_is_the_enumerated_walk_distribution(batch, rn):
    table = enumerate_walk_routes(rn)
    for road, (routes, pr) in table.items():
        assert abs(pr.sum() - 1.0) < 1e-12
    obs = {}
    for fs in batch.sets:
        ids, cnt = np.unique(fs.veh_route, return_counts=True)
        for r, c in zip(ids.tolist(), cnt.tolist()):
            key = _roads_of(batch, rn, r)
            obs[key] = obs.get(key, 0) + c
    chi2, dof = 0.0, 0
    for road, (routes, pr) in table.items():
        n_road = sum(c for k, c in obs.items() if k[0] == road)
        exp = n_road * pr
        o = np.asarray([obs.get(tuple(r), 0) for r in routes], dtype=np.float64)
        assert o.sum() == n_road  # every generated route is one the walk can return
        big = exp >= 5.0
        e2 = np.append(exp[big], exp[~big].sum())
        o2 = np.append(o[big], o[~big].sum())
        keep = e2 > 0
        chi2 += float((((o2 - e2) ** 2) / np.where(keep, e2, 1.0))[keep].sum())
        dof += int(keep.sum()) - 1
    assert dof > 100
    assert chi2 < dof + 4.0 * np.sqrt(2.0 * dof), (chi2, dof)


def test_same_law_as_the_per_vehicle_generator(batch, rn):
    """Two-sample chi-square of the route-LENGTH law and of the per-road arrival counts between 32 demands of each
    generator (the per-vehicle one draws its walks step by step, this one from the enumerated distribution)."""
    fc2 = FlowCompiler(rn)
    for i in range(N):
        fc2.add(poisson_flows(rn, RATE, HORIZON, seed=i), HORIZON)

    def length_hist(fc):
        h = np.zeros(64)
        n = 0
        for fs in fc.sets:
            ids, cnt = np.unique(fs.veh_route, return_counts=True)
            for r, c in zip(ids.tolist(), cnt.tolist()):
                h[len(_roads_of(fc, rn, r))] += c
            n += fs.n_veh
        return h, n

    h1, n1 = length_hist(batch)
    h2, n2 = length_hist(fc2)
    assert abs(n1 - n2) < 4.0 * np.sqrt(n1 + n2)  # totals: two Poisson counts with the same mean
    keep = (h1 + h2) >= 10
    k1, k2 = np.sqrt(n2 / n1), np.sqrt(n1 / n2)
    chi2 = float((((k1 * h1 - k2 * h2) ** 2) / (h1 + h2))[keep].sum())
    dof = int(keep.sum()) - 1
    assert dof >= 5
    assert chi2 < dof + 4.0 * np.sqrt(2.0 * dof), (chi2, dof, h1[keep], h2[keep])
    # mean route length to three digits ("route-length law equal to three digits")
    m1 = (h1 * np.arange(64)).sum() / n1
    m2 = (h2 * np.arange(64)).sum() / n2
    assert abs(m1 - m2) < 0.02, (m1, m2)


def test_demand_g_is_the_same_whatever_batch_or_base_generates_it(rn):
    """Rank r of a multi-GPU job generates demands r x R ...: demand g must not depend on the batch it is part of."""

    def demands(ids):
        fc = FlowCompiler(rn)
        poisson_flow_sets(fc, RATE, HORIZON, ids, base_seed=0)
        return {g: (fs.veh_start.copy(), [_roads_of(fc, rn, r) for r in fs.veh_route.tolist()]) for g, fs in zip(ids, fc.sets)}

    a = demands(list(range(0, 6)))
    b = demands([5])
    c = demands(list(range(4, 9)))
    d = demands([4100, 5, 4])
    for g, other in ((5, b), (5, c), (4, c), (5, d), (4, d)):
        assert np.array_equal(a[g][0], other[g][0]) and a[g][1] == other[g][1], g
    assert not np.array_equal(a[4][0][:50], a[5][0][:50])  # ... and different demands do differ
    e = demands([4100])
    assert np.array_equal(d[4100][0], e[4100][0]) and d[4100][1] == e[4100][1]


def test_tables_built_as_the_bench_builds_them_hold_those_demands(rn):
    """bench.py:60-71: `n_flow_sets_batch` = replicas, `flow_set_base` = rank x replicas; set s of the tables is demand
    base + s (the vehicles of a set in creation order, its routes by their drivables)."""
    from pytsc_amd.config import Config
    from pytsc_amd.scenario import build_tables

    def tables(batch_n, base):
        cfg = Config("syn_4x4", signal={"action_space": "phase_selection", "round_robin": False},
                     hip={"flow_rate": RATE, "n_flow_sets": batch_n, "flow_seed": 0, "n_flow_sets_batch": batch_n, "flow_set_base": base})
        return build_tables(cfg, n_replicas=batch_n, roadnet=rn, set_offset=0)

    t0, t1 = tables(6, 0), tables(3, 3)
    for s in range(3):
        f0, f1 = t0.fc.sets[3 + s], t1.fc.sets[s]
        assert np.array_equal(f0.veh_start, f1.veh_start)
        assert [t0.fc.routes[r] for r in f0.veh_route.tolist()] == [t1.fc.routes[r] for r in f1.veh_route.tolist()]
    assert t0.flows.n_sets == 6 and t1.flows.n_sets == 3
