"""GPU parity: the HIP path through the C-ABI against the CPU oracle on the same
seeded inputs (bit-exact: integers and f64 alike) and against the golden
fixtures captured from the reference's Python layer."""
import numpy as np
import pytest

from fixtures import CASES, load_case

pytestmark = pytest.mark.gpu


def _cmp_outputs(buf, out, s, R):
    g = {k: v.cpu().numpy() for k, v in buf.items()}
    for r in range(R):
        assert np.array_equal(g["obs"][r], out.obs[0]), f"obs step {s} rep {r}"
        assert np.array_equal(g["state"][r], out.state[0]), f"state step {s} rep {r}"
        assert np.array_equal(g["mask"][r], out.mask[0]), f"mask step {s} rep {r}"
        assert np.array_equal(g["phase"][r], out.phase[0]), f"phase step {s}"
        assert g["done"][r] == out.done[0] and g["terminated"][r] == out.terminated[0]
        # f64 outputs: bit-exact against the oracle
        assert np.array_equal(g["local_reward64"][r], out.local_reward64[0]), f"local reward step {s}"
        assert g["global_reward64"][r] == out.global_reward64[0], f"global reward step {s}"
        assert np.array_equal(g["stats"][r], out.stats[0]), (s, g["stats"][r], out.stats[0])
        assert np.array_equal(g["agent_stats"][r], out.agent_stats[0]), f"agent stats step {s}"
        assert np.array_equal(g["local_reward"][r], out.local_reward[0])


@pytest.mark.parametrize("family", ["list", "resident"])
@pytest.mark.parametrize("case", CASES)
def test_hip_matches_oracle_and_golden(case, family, oracle_lib):
    from pytsc_amd.vec_env import VecEngine

    R = 3
    inputs, exp, ref, cfg, tables1 = load_case(case, n_replicas=1)
    _, _, _, _, tablesR = load_case(case, n_replicas=R)
    o = oracle_lib.Oracle(tables1)
    eng = VecEngine(tablesR, family=family)
    assert eng.family == family
    _cmp_outputs(eng.observe(), o.observe(), -1, R)
    import torch

    for s in range(inputs["steps"]):
        acts = exp["actions"][s]
        out = o.step(acts[None])
        buf = eng.step(torch.tensor(np.repeat(acts[None], R, 0), dtype=torch.int32, device="cuda"))
        _cmp_outputs(buf, out, s, R)
        # golden (reference Python layer) — integers exact, floats 1e-5 (north star tolerance)
        g_obs = buf["obs"][0].cpu().numpy()
        assert np.array_equal(g_obs, exp["obs"][s].astype(np.float32))
        assert np.array_equal(buf["mask"][0].cpu().numpy(), exp["mask"][s])
        assert np.array_equal(buf["phase"][0].cpu().numpy(), exp["phase"][s])
        np.testing.assert_allclose(buf["local_reward64"][0].cpu().numpy(), exp["local_reward"][s], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(buf["stats"][0].cpu().numpy(), exp["info"][s], rtol=1e-5, atol=1e-5)
        if s % 24 == 7:
            d1, x1, v1, i1 = o.snapshot(0)
            for r in (0, R - 1):
                d2, x2, v2, i2 = eng.snapshot(r)
                assert np.array_equal(d1, d2) and np.array_equal(i1, i2), f"vehicle lists step {s}"
                assert np.array_equal(x1, x2) and np.array_equal(v1, v2), f"positions/speeds step {s}"
            c1, s1, p1 = o.lane_detail(True)
            c2, s2, p2 = eng.lane_detail(True)
            assert np.array_equal(c1[0], c2[0].cpu().numpy())
            assert np.array_equal(s1[0], s2[0].cpu().numpy())
            assert np.array_equal(p1[0], p2[0].cpu().numpy())
    assert eng.errors() == 0 and o.errors() == 0
    eng.close()
