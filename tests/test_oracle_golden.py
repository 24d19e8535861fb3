"""The CPU oracle's restatement of the reference's Python layer (parser tables,
lane measurements, per-agent stats, observations, states, masks, rewards, info,
phase state) against the golden fixtures captured from the reference itself
(tests/golden/gen_golden.py)."""
import numpy as np
import pytest

from fixtures import CASES, load_case

FLOAT_TOL = 1e-9  # fixtures vs oracle: same f64 arithmetic up to summation order


@pytest.mark.parametrize("case", CASES)
def test_parser_tables(case):
    inputs, exp, ref, cfg, tables = load_case(case)
    p = tables.parser
    assert list(p.traffic_signals.keys()) == ref["agent_order"]
    assert p.traffic_signal_ids == ref["traffic_signal_ids"]
    assert p.lanes == ref["lanes"]
    for k in ("lane_lengths", "lane_max_speeds", "lane_indices", "lane_angles"):
        mine = getattr(p, k)
        assert list(mine.keys()) == list(ref[k].keys()) or set(mine) == set(ref[k])
        for lane, v in ref[k].items():
            assert mine[lane] == pytest.approx(v, abs=1e-12), (k, lane)
    assert np.array_equal(np.asarray(p.adjacency_matrix), np.asarray(ref["adjacency_matrix"]))
    for t, d in ref["k_hop_neighbors"].items():
        assert {str(k): v for k, v in p.k_hop_neighbors[t].items()} == d
    assert p.neighbors_lanes == ref["neighbors_lanes"]
    for t, sig in ref["signals"].items():
        for k, v in sig.items():
            assert p.traffic_signals[t][k] == v, (t, k)
        mm = {str(k): v for k, v in p.traffic_signals[t]["phases_min_max_times"].items()}
        assert mm == ref["phases_min_max_times"][t]
        p2l = {str(k): v for k, v in p.traffic_signals[t]["phase_to_inc_out_lanes"].items()}
        assert p2l == ref["phase_to_inc_out_lanes"][t]
    for lane, f in ref["lane_features"].items():
        assert tables.lane_features[lane] == pytest.approx(f, abs=1e-12)
    assert tables.obs_size == ref["obs_size"]
    assert tables.state_size == ref["state_size"]
    assert tables.n_actions == ref["action_size"]


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_reference_layer(case, oracle_lib):
    inputs, exp, ref, cfg, tables = load_case(case)
    o = oracle_lib.Oracle(tables)
    out = o.observe()
    assert np.array_equal(out.obs[0], exp["obs0"].astype(np.float32))
    assert np.array_equal(out.mask[0], exp["mask0"])
    assert bool(exp["obs_is_int"])
    pm_steps = list(exp["posmat_steps"])
    for s in range(inputs["steps"]):
        out = o.step(exp["actions"][s][None])
        assert np.array_equal(out.obs[0], exp["obs"][s].astype(np.float32)), f"obs step {s}"
        assert np.array_equal(out.state[0], exp["state"][s].astype(np.float32)), f"state step {s}"
        assert np.array_equal(out.mask[0], exp["mask"][s]), f"mask step {s}"
        assert np.array_equal(out.phase[0], exp["phase"][s]), f"phase step {s}"
        assert out.done[0] == exp["done"][s] and out.terminated[0] == exp["terminated"][s]
        np.testing.assert_allclose(out.local_reward64[0], exp["local_reward"][s], rtol=FLOAT_TOL, atol=FLOAT_TOL)
        np.testing.assert_allclose(out.global_reward64[0], exp["global_reward"][s], rtol=FLOAT_TOL, atol=FLOAT_TOL)
        np.testing.assert_allclose(out.stats[0], exp["info"][s], rtol=FLOAT_TOL, atol=FLOAT_TOL)
        np.testing.assert_allclose(out.agent_stats[0], exp["agent_stats"][s], rtol=FLOAT_TOL, atol=FLOAT_TOL)
        counts, sum_speed, pm = o.lane_detail(with_posmat=s in pm_steps)
        assert np.array_equal(counts[0, :, 0], exp["lane_n"][s])
        assert np.array_equal(counts[0, :, 1], exp["lane_q"][s])
        n = np.maximum(counts[0, :, 0], 1)
        ms = np.where(counts[0, :, 0] > 0, sum_speed[0] / n, 0.0)
        np.testing.assert_allclose(ms, exp["lane_mean_speed"][s], rtol=FLOAT_TOL, atol=FLOAT_TOL)
        if s in pm_steps:
            e = exp["posmat"][pm_steps.index(s)]
            for l in range(tables.n_lanes):
                row = e[l][~np.isnan(e[l])]
                np.testing.assert_allclose(pm[0, l, : len(row)], row, rtol=0, atol=1e-12)
    assert o.errors() == 0
