"""Flow-file producers against files written by the
reference's own generators (tests/golden/trips/, made by gen_trips_golden.py):
same class names, same arguments, same seed -> the same trips and file name."""
import gzip
import json
import os
import random

import numpy as np
import pytest

import pytsc_amd.trip_generator as tg
from pytsc_amd.flows import expand_flows

HERE = os.path.dirname(os.path.abspath(__file__))
TRIPS = os.path.join(HERE, "golden", "trips")
CASES = sorted(f[:-8] for f in os.listdir(TRIPS) if f.endswith(".json.gz"))


def _load(name):
    with gzip.open(os.path.join(TRIPS, name + ".json.gz"), "rt") as f:
        return json.load(f)


def _shipped_3x3_flows():
    with gzip.open(os.path.join(HERE, "golden", "syn_3x3_random", "inputs.json.gz"), "rt") as f:
        return expand_flows(json.load(f)["flows"])


@pytest.mark.parametrize("name", CASES)
def test_same_trips_as_the_reference_generator(name, tmp_path):
    rec = _load(name)
    case = rec["case"]
    args = dict(case["args"])
    if "global_seed" in case:  # generators that, like the reference's, use the global streams
        np.random.seed(case["global_seed"])
        random.seed(case["global_seed"])
    if "per_entry" in case:
        args["inter_mus"], args["inter_sigmas"] = rec["inter_mus"], rec["inter_sigmas"]
    if case["cls"] == "CityFlowRandomizedTripGenerator":
        args["flow_data"] = _shipped_3x3_flows()
    gen = getattr(tg, case["cls"])(case["scenario"], **args)
    path = gen.generate_flows(str(tmp_path), **case.get("gen", {}))
    assert os.path.relpath(path, str(tmp_path)) == rec["filename"]
    with open(path) as f:
        got = json.load(f)
    want_routes = rec["routes"]
    assert len(got) == len(rec["trips"])
    for e, (ri, t) in zip(got, rec["trips"]):
        assert e["route"] == want_routes[ri] and e["startTime"] == t and e["endTime"] == t
        assert e["interval"] == 1.0 and e["vehicle"] == rec["vehicle"]


def test_flow_disrupted_bursts_raise_demand():
    """The burst generator draws its burst roads from a set (hash order), so the
    reference's output is not reproducible across processes: check the law."""
    g = tg.FlowDisruptedCityFlowTripGenerator("syn_3x3", 0, 3600, 6.0, 0.8, disruption_ratio=0.5, seed=1)
    flows = g.flows()
    assert len(g.disrupted_links) == 6 and set(g.burst_timings) == g.disrupted_links
    per_road = {}
    for e in flows:
        per_road.setdefault(e["route"][0], []).append(e["startTime"])
    for road, (t0, t1) in g.burst_timings.items():
        inside = sum(t0 <= t < t1 for t in per_road[road])
        assert inside > 2.5 * (t1 - t0) / 6.0  # ~4x the base rate during the burst
    calm = [r for r in g.graph.entries if r not in g.disrupted_links]
    assert all(abs(len(per_road[r]) - 600) < 80 for r in calm)


def test_generated_file_runs_through_the_table_compiler(tmp_path):
    from pytsc_amd.config import Config
    from pytsc_amd.scenario import build_tables

    g = tg.CityFlowTripGenerator("syn_2x2", 0, 300, 5.0, 0.8, seed=0)
    entries = g.flows()
    cfg = Config("syn_2x2", simulator={"flow_entries": [entries]})
    tables = build_tables(cfg, n_replicas=2)
    assert tables.n_lanes > 0
