"""Diagnostic (GPU box): HIP against the oracle on randomly drawn configurations -- network, demand, visibility, action
space, kernel family -- a few replicas each, every output compared at every step.  Not part of the test suite (the
suite's cases are fixed); run after kernel changes:  python tools/fuzz_parity.py [n_cases] [seed]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch

from oracle import pyoracle
from pytsc_amd.config import Config
from pytsc_amd.scenario import build_tables
from pytsc_amd.vec_env import VecEngine

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
names = ("obs", "mask", "stats", "local_reward64", "global_reward64", "agent_stats", "phase", "state", "done", "terminated", "phase_scores")
for case in range(n_cases):
    scen = str(rng.choice(["syn_1x1", "syn_2x2", "syn_3x3", "syn_4x4"]))
    space = str(rng.choice(["phase_selection", "phase_switch"]))
    rr = bool(rng.integers(0, 2))
    vis = int(rng.choice([6, 8, 10, 12, 14]))
    rate = float(rng.choice([300.0, 700.0, 1200.0, 1800.0]))
    family = str(rng.choice(["resident", "list"]))
    R = int(rng.choice([3, 6, 9]))
    steps = int(rng.choice([70, 70, 160]))  # (160 at the higher rates: the 3x3 / 4x4 cross from the small class to the full-size one)

    def tables():
        cfg = Config(scen, signal={"action_space": space, "round_robin": rr, "visibility": vis},
                     hip=dict(flow_rate=rate, n_flow_sets=int(min(R, 3)), sim_length=600, flow_seed=int(case)))
        return build_tables(cfg, n_replicas=R)

    eng, orc = VecEngine(tables(), family=family), pyoracle.Oracle(tables())
    buf, out = eng.observe(), orc.observe()
    for s in range(steps):
        acts = eng.random_valid_actions(buf["mask"], seed=1000 + case)
        buf = eng.step(acts)
        out = orc.step(acts.cpu().numpy())
        for k in names:
            got = buf[k].cpu().numpy()
            assert np.array_equal(got, getattr(out, k)), (case, scen, space, rr, vis, rate, family, k, s)
    for r in range(R):
        assert all(np.array_equal(x, y) for x, y in zip(orc.snapshot(r), eng.snapshot(r))), (case, r)
    assert eng.errors() == 0 and orc.errors() == 0
    print(f"case {case}: {scen} {space} rr={rr} vis={vis} rate={rate:g} {family} ({eng.kernel_variant if family == 'resident' else '-'}) R={R}: "
          f"{eng.mean_vehicles():.0f} vehicles per replica, exact", flush=True)
    eng.close()
    orc.close()
print("all exact")
