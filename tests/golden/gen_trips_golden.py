"""Golden flow files of the REFERENCE's trip generators.

Imports pytsc/backends/cityflow/trip_generator.py from /root/reference (never
copied), runs each generator on scenarios the reference ships, and stores what it
wrote in compact form (routes interned, start times) under tests/golden/trips/.
tests/test_trip_generator.py replays the same arguments on pytsc_amd.trip_generator.

Run (build container only):  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_trips_golden.py
"""
import gzip
import json
import os
import random
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.dont_write_bytecode = True
os.environ.setdefault("SUMO_HOME", "/tmp/none")
sys.path[:0] = [os.path.join(HERE, "ref_stubs"), REPO, REF]

CASES = {
    "gaussian_2x2": dict(cls="CityFlowTripGenerator", scenario="syn_2x2",
                         args=dict(start_time=0, end_time=600, inter_mu=6.0, inter_sigma=0.8, seed=3)),
    "gaussian_3x3_turns": dict(cls="CityFlowTripGenerator", scenario="syn_3x3",
                               args=dict(start_time=0, end_time=400, inter_mu=5.0, inter_sigma=1.5, seed=11,
                                         turn_probs=[0.3, 0.3, 0.4])),
    "link_disrupted_3x3": dict(cls="LinkDisruptedCityFlowTripGenerator", scenario="syn_3x3",
                               args=dict(start_time=0, end_time=400, inter_mu=6.0, inter_sigma=0.8,
                                         disruption_ratio=0.3, seed=5)),
    "interval_2x2": dict(cls="IntervalCityFlowTripGenerator", scenario="syn_2x2",
                         args=dict(start_time=0, end_time=720, inter_mu=6.0, inter_sigma=0.8, seed=7),
                         gen=dict(replicate_no=2, interval_duration=360)),
    "oneway_2x2": dict(cls="CityFlowOneWayTripGenerator", scenario="syn_2x2",
                       args=dict(start_time=0, end_time=500, inter_mu_ns=5.0, inter_sigma_ns=0.8,
                                 inter_mu_ew=8.0, inter_sigma_ew=1.0, seed=9)),
    "variable_demand_2x2": dict(cls="VariableDemandTripGenerator", scenario="syn_2x2", global_seed=13,
                                args=dict(start_time=0, end_time=1500, edge_weights=None),
                                per_entry=dict(mu=6.0, sigma=1.0)),
    "randomized_3x3_low": dict(cls="CityFlowRandomizedTripGenerator", scenario="syn_3x3", global_seed=17,
                               args=dict(start_time=0, end_time=500), gen=dict(flow_type="low")),
}


def compact(entries):
    routes, idx, out = [], {}, []
    for e in entries:
        key = tuple(e["route"])
        if key not in idx:
            idx[key] = len(routes)
            routes.append(list(key))
        out.append([idx[key], e["startTime"]])
    return {"routes": routes, "trips": out, "vehicle": entries[0]["vehicle"] if entries else None,
            "interval": 1.0}


def main():
    import pytsc.backends.cityflow.trip_generator as ref

    out_dir = os.path.join(HERE, "trips")
    os.makedirs(out_dir, exist_ok=True)
    for name, case in CASES.items():
        if "global_seed" in case:
            np.random.seed(case["global_seed"])
            random.seed(case["global_seed"])
        args = dict(case["args"])
        gen = getattr(ref, case["cls"])
        if "per_entry" in case:
            probe = ref.CityFlowTripGenerator(case["scenario"], 0, 10, 5.0, 1.0)
            entries, _ = probe._find_fringe_edges()
            args["inter_mus"] = {e: case["per_entry"]["mu"] + 0.5 * k for k, e in enumerate(entries)}
            args["inter_sigmas"] = {e: case["per_entry"]["sigma"] for e in entries}
            np.random.seed(case["global_seed"])
            random.seed(case["global_seed"])
        g = gen(case["scenario"], **args)
        with tempfile.TemporaryDirectory() as tmp:
            g.generate_flows(tmp, **case.get("gen", {}))
            files = []
            for root, _, fs in os.walk(tmp):
                files += [os.path.join(root, f) for f in fs]
            assert len(files) == 1, files
            with open(files[0]) as f:
                entries = json.load(f)
            rel = os.path.relpath(files[0], tmp)
        rec = compact(entries)
        rec["filename"] = rel
        rec["case"] = {k: v for k, v in case.items()}
        if "per_entry" in case:
            rec["inter_mus"], rec["inter_sigmas"] = args["inter_mus"], args["inter_sigmas"]
        with gzip.open(os.path.join(out_dir, name + ".json.gz"), "wt") as f:
            json.dump(rec, f)
        print(name, len(entries), "trips ->", rel)


if __name__ == "__main__":
    main()
