"""GPU: the cross-logic scenarios of tests/test_cross_kat.py through the C-ABI, both kernel families -- every env step's
vehicle snapshot (drivable, position, speed, id) and outputs against the oracle, and the behavioural invariants
(mutual exclusion at every cross, no deadlock) on what the HIP path itself produced."""
import numpy as np
import pytest

from cross_kat_scenarios import SCENARIOS, Track, check_mutual_exclusion, tables, tie_tables, who_braked

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("family", ["list", "resident"])
@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_cross_scenarios_match_oracle(name, family, oracle_lib):
    import torch

    from pytsc_amd.vec_env import VecEngine

    R = 2
    eng = VecEngine(tables(SCENARIOS[name], n_replicas=R), family=family)
    orc = oracle_lib.Oracle(tables(SCENARIOS[name], n_replicas=1))
    acts = np.zeros((1, 1), np.int32)  # phase index 0 = raw phase 1: every movement green
    tr = Track(eng.tables)
    for s in range(40):
        out = orc.step(acts)
        buf = eng.step(torch.zeros((R, 1), dtype=torch.int32, device="cuda"))
        a = orc.snapshot(0)
        for r in range(R):
            b = eng.snapshot(r)
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), (s, r)
            for k in ("obs", "stats", "agent_stats", "local_reward64"):
                assert np.array_equal(buf[k][r].cpu().numpy(), getattr(out, k)[0]), (s, k)
        tr.add(eng.snapshot(R - 1))
        if s > 2 and not tr.rows[-1]:
            break
    assert not tr.rows[-1], "vehicles left in the network"
    assert check_mutual_exclusion(tr) is None
    c = eng.replica_counts().cpu().numpy()
    assert (c[:, 2] == len(SCENARIOS[name])).all() and eng.errors() == 0 and orc.errors() == 0
    eng.close()


@pytest.mark.parametrize("family", ["list", "resident"])
@pytest.mark.parametrize("seed", [0, 1])
def test_priority_rule_on_a_tie_matches_oracle(seed, family, oracle_lib):
    """tsc_flows_t::veh_priority through the C-ABI: the tie scenario of tests/test_priority_rule.py under the mt19937 rule
    (seed 0: the higher id goes first; seed 1: the lower), every vehicle of every step against the oracle."""
    import torch

    from pytsc_amd.flows import mt19937_priorities
    from pytsc_amd.vec_env import VecEngine

    kw = dict(priority_rule="mt19937", seed=seed)
    eng = VecEngine(tie_tables(n_replicas=2, **kw), family=family)
    orc = oracle_lib.Oracle(tie_tables(**kw))
    tr = Track(eng.tables)
    for s in range(40):
        orc.step(np.zeros((1, 1), np.int32))
        eng.step(torch.zeros((2, 1), dtype=torch.int32, device="cuda"))
        for r in range(2):
            assert all(np.array_equal(x, y) for x, y in zip(orc.snapshot(0), eng.snapshot(r))), (s, r)
        tr.add(eng.snapshot(1))
    p = mt19937_priorities(seed, 2)
    assert who_braked(tr) == {1 if p[0] > p[1] else 0}
    assert check_mutual_exclusion(tr) is None and eng.errors() == 0
    eng.close()
