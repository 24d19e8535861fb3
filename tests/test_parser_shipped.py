"""The parser on EVERY road network the reference ships: the tables of
`pytsc_amd.network_parser.NetworkParser` against those the reference's own `NetworkParser` built from the same file with
the scenario's own config (tests/golden/gen_parser_golden.py -> tests/golden/parsers/<scenario>.json.gz: roadnet JSON, merged
config sections, the reference's tables).  Reference: pytsc/backends/cityflow/network_parser.py:24-78,631-703."""
import glob
import gzip
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "parsers", "*.json.gz")))
NAMES = [os.path.basename(f)[: -len(".json.gz")] for f in FILES]


def _load(name):
    with gzip.open(os.path.join(HERE, "golden", "parsers", name + ".json.gz"), "rt") as f:
        return json.load(f)


def _parse(rec):
    from pytsc_amd.config import Config
    from pytsc_amd.network_parser import NetworkParser

    c = rec["config"]
    sim = dict(c["simulator"], roadnet_json=rec["roadnet"])
    cfg = Config("shipped_" + rec["scenario"], network=c["network"], signal=c["signal"], misc=c["misc"], hip=sim)
    return NetworkParser(cfg)


def test_the_fixtures_cover_the_networks_the_survey_names():
    for must in ("hangzhou_4_4", "syn_5x5_oneway", "manhattan_16_3", "new_york_arterial", "1x5_cityflow_grid", "jinan_3_4", "syn_2x2"):
        assert must in NAMES


@pytest.mark.parametrize("name", NAMES)
def test_parser_equals_the_reference_on_every_shipped_network(name):
    rec = _load(name)
    ref = rec["parser"]
    assert ref is not None, "the reference's own parser raised on this file: " + str(rec.get("reference_raises"))
    p = _parse(rec)
    assert list(p.traffic_signals.keys()) == ref["agent_order"]
    assert p.traffic_signal_ids == ref["traffic_signal_ids"]
    assert p.lanes == ref["lanes"]
    for k in ("lane_lengths", "lane_max_speeds", "lane_indices", "lane_angles"):
        mine = getattr(p, k)
        assert set(mine) == set(ref[k]), k
        for lane, v in ref[k].items():
            assert mine[lane] == pytest.approx(v, abs=1e-12), (k, lane)
    assert np.array_equal(np.asarray(p.adjacency_matrix), np.asarray(ref["adjacency_matrix"]))
    for t, d in ref["k_hop_neighbors"].items():
        assert {str(k): v for k, v in p.k_hop_neighbors[t].items()} == d
    assert p.neighbors_lanes == ref["neighbors_lanes"]
    for t, sig in ref["signals"].items():
        for k, v in sig.items():
            mine = p.traffic_signals[t][k]
            if k in ("coordinates", "norm_coordinates"):
                assert list(mine) == pytest.approx(list(v), abs=1e-12), (t, k)
            else:
                assert mine == v, (t, k)
        mm = {str(k): v for k, v in p.traffic_signals[t]["phases_min_max_times"].items()}
        assert mm == ref["phases_min_max_times"][t]
        p2l = {str(k): v for k, v in p.traffic_signals[t]["phase_to_inc_out_lanes"].items()}
        assert p2l == ref["phase_to_inc_out_lanes"][t]
    # every key of the scenario's `signal` section lands in each signal's table (network_parser.py:78)
    for t in ref["agent_order"]:
        for k, v in rec["config"]["signal"].items():
            assert p.traffic_signals[t][k] == v


def test_syn_1x7_parses_like_the_reference_and_the_engine_tables_refuse_it_with_a_reason():
    """The reference ships one internally inconsistent file: 22 road links of syn_1x7 name roads it does not define.  Its
    parser works on names and builds 7 signals / 66 lanes from it; so does ours (above).  Engine tables cannot be built
    from such a file: a ValueError that names the road link, not a bare KeyError (VERDICT r4 "what's missing" 6)."""
    from pytsc_amd.flows import FlowCompiler
    from pytsc_amd.tables import Tables

    rec = _load("syn_1x7")
    p = _parse(rec)
    assert len(p.traffic_signals) == 7 and len(p.lanes) == 66
    assert len(p.rn.inconsistent) > 0
    with pytest.raises(ValueError, match="inconsistent road network .*names lane"):
        Tables(p, FlowCompiler(p.rn), p.config, 1)
