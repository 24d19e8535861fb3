"""Parser fixtures for EVERY road network the reference ships: run the REFERENCE's own CityFlow
`Config` + `NetworkParser` (imported from /root/reference, never copied) on each shipped scenario directory as it
lies there -- its own config.yaml with whatever overrides it carries (`phase_sequence`, `neighbors`, round-robin ...) --
and record the tables the rest of the plugin surface consumes, next to the inputs a build without /root/reference needs
to redo the parse: the roadnet JSON (data) and the scenario's merged config sections.

Run (build container only):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_parser_golden.py [scenario ...]
Writes tests/golden/parsers/<scenario>.json.gz; tests/test_parser_shipped.py replays them (CPU).
A scenario whose parse raises inside the reference is recorded with the exception's type instead of tables.
"""
import gzip
import json
import logging
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
REF_SCEN = os.path.join(REF, "pytsc", "scenarios", "cityflow")
OUT = os.path.join(HERE, "parsers")

sys.dont_write_bytecode = True
os.environ.setdefault("SUMO_HOME", "/tmp/none")
sys.path[:0] = [os.path.join(HERE, "ref_stubs"), REPO, REF]


def parser_dict(pn):
    return {
        "agent_order": list(pn.traffic_signals.keys()),
        "traffic_signal_ids": pn.traffic_signal_ids,
        "lanes": pn.lanes,
        "lane_lengths": pn.lane_lengths, "lane_max_speeds": pn.lane_max_speeds,
        "lane_indices": pn.lane_indices, "lane_angles": pn.lane_angles,
        "adjacency_matrix": np.asarray(pn.adjacency_matrix).tolist(),
        "k_hop_neighbors": {t: {str(k): v for k, v in d.items()} for t, d in pn.k_hop_neighbors.items()},
        "neighbors_lanes": pn.neighbors_lanes,
        "signals": {
            t: {k: cfg[k] for k in ("incoming_lanes", "outgoing_lanes", "phases", "n_phases",
                                    "phase_indices", "green_phase_indices", "yellow_phase_indices",
                                    "coordinates", "norm_coordinates", "inc_to_out_lanes")}
            for t, cfg in pn.traffic_signals.items()
        },
        "phases_min_max_times": {t: {str(k): v for k, v in cfg["phases_min_max_times"].items()}
                                 for t, cfg in pn.traffic_signals.items()},
        "phase_to_inc_out_lanes": {t: {str(k): v for k, v in cfg["phase_to_inc_out_lanes"].items()}
                                   for t, cfg in pn.traffic_signals.items()},
    }


def generate(scenario):
    from pytsc.backends.cityflow.config import Config
    from pytsc.backends.cityflow.network_parser import NetworkParser
    from pytsc.common.utils import EnvLogger

    EnvLogger.set_log_level(logging.WARNING)
    cfg = Config(scenario)
    if not os.path.exists(cfg.cityflow_roadnet_file):
        print(scenario, "skipped: no road network file", cfg.cityflow_roadnet_file)
        return
    with open(cfg.cityflow_roadnet_file) as f:
        net = json.load(f)
    rec = {"scenario": scenario, "roadnet_file": os.path.basename(cfg.cityflow_roadnet_file), "roadnet": net,
           "config": {"network": cfg.network, "signal": cfg.signal, "misc": cfg.misc,
                      "simulator": {k: v for k, v in cfg.simulator.items() if k in ("phase_sequence", "delta_time", "interval")}}}
    try:
        rec["parser"] = parser_dict(NetworkParser(cfg))
    except Exception as e:  # noqa: BLE001 -- whatever the reference raises on its own file is the datum
        rec["parser"] = None
        rec["reference_raises"] = type(e).__name__
    os.makedirs(OUT, exist_ok=True)
    with gzip.open(os.path.join(OUT, scenario + ".json.gz"), "wt", compresslevel=9) as f:
        json.dump(rec, f, separators=(",", ":"), default=float)
    p = rec["parser"]
    print(scenario, "ok:" if p else "reference raises " + rec.get("reference_raises", ""),
          (len(p["agent_order"]), "signals", len(p["lanes"]), "lanes") if p else "")


if __name__ == "__main__":
    names = sys.argv[1:] or sorted(d for d in os.listdir(REF_SCEN) if os.path.exists(os.path.join(REF_SCEN, d, "config.yaml")))
    for n in names:
        generate(n)
