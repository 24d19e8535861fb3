"""Scenario producers: the grid generator reproduces the shipped roadnet family
(compared with the reference's files when the checkout is present), the flow
compiler's routes are drivable, Poisson demand has the requested rate."""
import json
import os

import numpy as np
import pytest

from pytsc_amd.flows import FlowCompiler, compact_flows, expand_flows, fringe_roads, poisson_flows
from pytsc_amd.grid import generate_grid
from pytsc_amd.roadnet import RoadNet

REF = "/root/reference/pytsc/scenarios/cityflow"


def _cmp(a, b, path=""):
    if isinstance(a, dict):
        assert isinstance(b, dict) and list(a.keys()) == list(b.keys()), path
        for k in a:
            _cmp(a[k], b[k], path + "/" + k)
    elif isinstance(a, list):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _cmp(x, y, f"{path}[{i}]")
    elif isinstance(a, (int, float)) and not isinstance(a, bool):
        assert abs(a - b) < 1e-9, (path, a, b)
    else:
        assert a == b, (path, a, b)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
@pytest.mark.parametrize("rows,cols,f", [(1, 1, "syn_1x1/1x1_roadnet.json"), (2, 2, "syn_2x2/2x2_roadnet.json"),
                                         (3, 3, "syn_3x3/3x3_roadnet.json")])
def test_grid_generator_reproduces_shipped_roadnets(rows, cols, f):
    with open(os.path.join(REF, f)) as fh:
        _cmp(json.load(fh), generate_grid(rows, cols))


def test_grid_geometry_invariants():
    rn = RoadNet(generate_grid(4, 4))
    assert rn.n_agents == 16 and rn.n_lanes == 240 and rn.n_links == 16 * 36
    lens = sorted(set(np.round(rn.lane_length, 6)))
    assert lens == [260.0, 280.0]  # trimmed by the intersection width at non-virtual ends
    assert set(np.round(rn.lane_len_pytsc, 6)) == {300.0}
    for a in range(16):
        assert len(rn.agent_links[a]) == 36
    # straight 40 m, left 32-39 m, right 16-23 m
    for l in rn.links:
        lo, hi = {3: (39.9, 40.3), 2: (32.0, 39.0), 1: (16.0, 23.0)}[l["type"]]
        assert lo <= l["length"] <= hi
    # every cross is registered on both links with swapped distances
    for k, cs in enumerate(rn.link_crosses):
        ds = [c[0] for c in cs]
        assert ds == sorted(ds)
        for d_self, _, other, d_other in cs:
            assert any(o == k and abs(a - d_other) < 1e-12 and abs(b - d_self) < 1e-12
                       for a, _, o, b in rn.link_crosses[other])


def test_poisson_flows_and_routes():
    rn = RoadNet(generate_grid(3, 3))
    flows = poisson_flows(rn, 700.0, 3600, seed=0)
    inc, out = fringe_roads(rn)
    assert len(inc) == 12
    rate = len(flows) / len(inc)
    assert 650 < rate < 750
    assert all(e["route"][0] in inc and len(e["route"]) >= 2 for e in flows)
    assert [e["startTime"] for e in flows] == sorted(e["startTime"] for e in flows)
    fc = FlowCompiler(rn)
    fs = fc.add(flows, 3600)
    assert fs.n_veh == len(flows)  # every generated route is drivable
    for seq in fc.routes:
        assert len(seq) % 2 == 1
        for i, d in enumerate(seq):
            assert (d >= rn.n_lanes) == (i % 2 == 1)
            if i % 2 == 1:
                l = rn.links[d - rn.n_lanes]
                assert l["start"] == seq[i - 1] and l["end"] == seq[i + 1]
    assert expand_flows(compact_flows(flows)) == flows


def test_multi_vehicle_flow_entries_expand_like_the_engine():
    rn = RoadNet(generate_grid(1, 1))
    veh = {"length": 5.0, "width": 2.0, "maxPosAcc": 2.0, "maxNegAcc": 4.5, "usualPosAcc": 2.0,
           "usualNegAcc": 4.5, "minGap": 2.5, "maxSpeed": 11.11, "headwayTime": 1.5}
    fc = FlowCompiler(rn)
    fs = fc.add([{"vehicle": veh, "route": ["road_0_1_0", "road_1_1_0"], "interval": 2.0,
                  "startTime": 10, "endTime": 20}], 100)
    assert fs.veh_start.tolist() == [10, 12, 14, 16, 18, 20]
    fs = fc.add([{"vehicle": veh, "route": ["road_0_1_0", "nowhere"], "interval": 1.0,
                  "startTime": 0, "endTime": 0}], 100)
    assert fs.n_veh == 0  # invalid route: the engine disables the flow
