"""Diagnostic (GPU box): HIP against the oracle on randomly drawn configurations -- network, demand, visibility, action
space, kernel family -- a few replicas each, every output compared at every step.  Not part of the test suite (the
suite's cases are fixed); run after kernel changes:  python tools/fuzz_parity.py [n_cases] [seed] [handover]
`handover` (round 5): `auto` handles with a small record capacity under heavy demand instead -- promotions to the spill
pool, a partial restart, the move to the list family (state written across, or the replay when a replica had stalled):
every row marked valid is the oracle's at every step, every vehicle afterwards."""
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
HANDOVER = len(sys.argv) > 3 and sys.argv[3] == "handover"
names = ("obs", "mask", "stats", "local_reward64", "global_reward64", "agent_stats", "phase", "state", "done", "terminated", "phase_scores")
for case in range(n_cases if HANDOVER else 0):
    scen = str(rng.choice(["syn_2x2", "syn_3x3", "syn_4x4"]))
    space = str(rng.choice(["phase_selection", "phase_switch"]))
    vis = int(rng.choice([8, 10, 14]))
    rate = float(rng.choice([1500.0, 1800.0, 2400.0]))
    R = int(rng.choice([4, 6, 9]))
    cap = int(rng.choice([512, 768, 1024, 1472]))
    margin = int(rng.choice([32, 128, 256]))
    limit = int(rng.choice([-1, 1, 2]))
    restart_at = int(rng.integers(20, 90))
    n_sets = int(min(R, 3))

    def tables(**hip):
        cfg = Config(scen, signal={"action_space": space, "round_robin": False, "visibility": vis},
                     hip=dict(flow_rate=rate, n_flow_sets=n_sets, sim_length=1500, episode_limit=1500, flow_seed=100 + int(case), **hip))
        return build_tables(cfg, n_replicas=R)

    try:
        eng = VecEngine(tables(resident_capacity=cap, resident_margin=margin, resident_spill_limit=limit), family="auto")
    except Exception as e:  # (a capacity the network's measurement scratch does not fit)
        print(f"case {case}: {scen} capacity {cap}: {str(e)[:80]} -- skipped", flush=True)
        continue
    if eng.family != "resident":
        print(f"case {case}: {scen} capacity {cap}: not on the resident family -- skipped", flush=True)
        eng.close()
        continue
    orc = pyoracle.Oracle(tables())
    buf, out = eng.observe(), orc.observe()
    moved_at, n_invalid, s = None, 0, 0
    while s < 400 and (moved_at is None or s < moved_at + 30):
        acts = eng.random_valid_actions(buf["mask"], seed=2000 + case)
        buf = eng.step(acts)
        out = orc.step(acts.cpu().numpy())
        ok = buf["valid"].cpu().numpy().astype(bool)
        n_invalid += int((~ok).sum())
        for k in names:
            got = buf[k].cpu().numpy()
            assert np.array_equal(got[ok], getattr(out, k)[ok]), (case, scen, space, vis, rate, cap, margin, limit, k, s)
        if moved_at is None and eng.family == "list":
            moved_at = s
        if s == restart_at and moved_at is None:
            mask = (np.arange(R) % 2 == 0).astype(np.uint8)
            orc.reset(mask)
            eng.reset(torch.tensor(mask))
            out, buf = orc.observe(), eng.observe()
        s += 1
    if moved_at is not None:
        assert bool(buf["valid"].all())
        for r in range(R):
            assert all(np.array_equal(x, y) for x, y in zip(orc.snapshot(r), eng.snapshot(r))), (case, r)
    c = eng.path_counts()
    assert eng.errors() == 0 and orc.errors() == 0
    print(f"case {case}: {scen} {space} vis={vis} rate={rate:g} capacity {cap} margin {margin} pool limit {limit} R={R}: "
          f"moved to the list family at step {moved_at}, {c['stalled']} stall(s), {n_invalid} rows marked invalid, "
          f"{eng.mean_vehicles():.0f} vehicles per replica, exact", flush=True)
    eng.close()
    orc.close()
for case in range(0 if HANDOVER else n_cases):
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
