"""GPU: the two classes of the resident family (include/pytsc_amd.h, tsc_class_counts) -- replicas that need
few vehicle records are stepped by 512-thread workgroups, two per CU, out of the same blobs as the full-size class; a
replica changes class by one byte.  Every output of every step against the ORACLE across the hand-overs in both
directions, the path that abandons a step inside the small class and has the full-size launch redo it (forced: ordinary
traffic next to never reaches it), the launch the host skips while hardly any replica is small, and equality with a
handle that has one class only."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NAMES = ("obs", "state", "mask", "stats", "local_reward64", "global_reward64", "agent_stats", "phase", "phase_scores", "done", "terminated")


def _tables(R, rate, **hip):
    from pytsc_amd.config import Config
    from pytsc_amd.scenario import build_tables

    cfg = Config("syn_4x4", signal={"action_space": "phase_selection", "round_robin": False},
                 hip=dict(flow_rate=rate, n_flow_sets=4, **hip))
    return build_tables(cfg, n_replicas=R)


def _run(oracle_lib, R, rate, steps, env=None, restart_at=None, **hip):
    import torch

    from pytsc_amd.vec_env import VecEngine

    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        eng = VecEngine(_tables(R, rate, **hip), family="resident")
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    orc = oracle_lib.Oracle(_tables(R, rate, **hip))
    orc.reset(sets=np.arange(R, dtype=np.int32) % 4)
    buf, out = eng.observe(), orc.observe()
    hist = []
    for s in range(steps):
        acts = eng.random_valid_actions(buf["mask"], seed=5)
        buf = eng.step(acts)
        out = orc.step(acts.cpu().numpy())
        for k in NAMES:
            assert np.array_equal(buf[k].cpu().numpy(), getattr(out, k)), (k, s)
        assert bool(buf["valid"].all()), s
        if s % 10 == 9:
            hist.append(eng.class_counts())
        if restart_at is not None and s == restart_at:  # half of the replicas start again from an empty network
            mask = (np.arange(R) % 2 == 0).astype(np.uint8)
            orc.reset(mask)
            eng.reset(torch.tensor(mask))
            out, buf = orc.observe(), eng.observe()
            assert np.array_equal(buf["obs"].cpu().numpy(), out.obs)
            hist.append(eng.class_counts())
    for r in (0, R // 2, R - 1):
        assert all(np.array_equal(x, y) for x, y in zip(orc.snapshot(r), eng.snapshot(r))), r
    c = eng.path_counts()
    assert eng.errors() == 0 and orc.errors() == 0 and c["stalled"] == 0
    eng.close()
    orc.close()
    return hist, c


def test_replicas_cross_from_the_small_class_to_the_full_size_one_and_back(oracle_lib):
    """1 100 veh/h: most replicas outgrow the small class within ~200 env steps; half of them are then restarted and are
    small again while the others stay full-size; all of it bit for bit the oracle's."""
    hist, c = _run(oracle_lib, 64, 1100.0, 230, restart_at=200, sim_length=1500)
    assert hist[0] == (64, 0)  # an empty network: everybody small
    assert max(b for _, b in hist) >= 32  # ... then most of them full-size (need > 1 856 - 32 records)
    assert hist[-1][0] >= 32  # the restarted half is small again
    assert c["class_change"] >= 32 and c["class_redo"] <= 8  # (a margin of 32 records: next to no step is redone)


def test_a_step_abandoned_inside_the_small_class_is_redone_by_the_full_size_launch(oracle_lib):
    """PYTSC_AMD_CLS_MARGIN=0: a replica leaves the small class only by running out of records WITHIN a step -- nothing has
    been stored by then, the full-size launch of the same tsc_step steps it from the same state.  A small capacity
    (PYTSC_AMD_SMALL_P) brings that forward."""
    hist, c = _run(oracle_lib, 48, 1100.0, 140, env={"PYTSC_AMD_CLS_MARGIN": "0", "PYTSC_AMD_SMALL_P": "1024",
                                                         "PYTSC_AMD_SMALL_HEAVY_PCT": "0"}, sim_length=1200)  # (... and the small launch is never left out for small replicas near their capacity: every hand-over is a redo)
    assert c["class_redo"] >= 40, c  # (nearly) every replica went over through the redo path
    assert any(b == 48 for _, b in hist)


def test_one_class_and_two_classes_agree_and_the_small_launch_is_skipped_when_nobody_is_small(oracle_lib):
    """`resident_threads` = 768: the full-size class only (round 4's handle).  The same run on both handles gives the same
    outputs; on the two-class handle the host stops launching the small class once (a stale count says) fewer than an eighth
    of the replicas are small, and the full-size launch steps the stragglers too."""
    import torch

    from pytsc_amd.vec_env import VecEngine

    R = 64
    two = VecEngine(_tables(R, 1300.0, sim_length=1500), family="resident")
    one = VecEngine(_tables(R, 1300.0, sim_length=1500, resident_threads=768), family="resident")
    assert one.class_counts() == (0, R) and two.class_counts() == (R, 0)
    b2, b1 = two.observe(), one.observe()
    for s in range(260):
        acts = two.random_valid_actions(b2["mask"], seed=9)
        b2, b1 = two.step(acts), one.step(acts)
        for k in ("obs", "stats", "mask", "phase_scores", "local_reward64"):
            assert torch.equal(b2[k], b1[k]), (k, s)
    n_big = two.class_counts()[1]
    assert n_big >= R * 3 // 4 and two.path_counts()["class_change"] >= n_big, n_big
    assert two.errors() == 0 and one.errors() == 0
    two.close()
    one.close()


def test_a_two_class_handle_that_outgrows_the_family_writes_its_state_across(oracle_lib, capfd):
    """(round 5, the demand cliff of VERDICT r4 item 6) An `auto` handle with two classes and a spill pool of one slot: at
    1 500 veh/h the first replica to come near its capacity is promoted, the second one takes the whole handle to the list
    family -- no replica has stalled, so the blobs' state (full-size replicas, the half that was restarted and is in the
    small class again, the pool's replica) is written across by kr_to_list instead of replaying 200-odd logged steps.
    Every output of every step before, at and after the hand-over is the oracle's, every vehicle too."""
    import torch

    from pytsc_amd.vec_env import VecEngine

    R, steps = 8, 330
    hip = dict(sim_length=2000, episode_limit=2000)
    eng = VecEngine(_tables(R, 1500.0, resident_capacity=2304, resident_margin=128, resident_spill_limit=1, **hip), family="auto")
    assert eng.family == "resident" and eng.class_counts() == (R, 0)
    orc = oracle_lib.Oracle(_tables(R, 1500.0, **hip))
    orc.reset(sets=np.arange(R, dtype=np.int32) % 4)
    buf, out = eng.observe(), orc.observe()
    moved_at, classes_before = None, None
    for s in range(steps):
        acts = eng.random_valid_actions(buf["mask"], seed=9)
        if moved_at is None:
            classes_before = eng.class_counts()
        buf = eng.step(acts)
        out = orc.step(acts.cpu().numpy())
        for k in NAMES:
            assert np.array_equal(buf[k].cpu().numpy(), getattr(out, k)), (k, s)
        assert bool(buf["valid"].all()), s
        if moved_at is None and eng.family == "list":
            moved_at = s
        if s == 100:  # half of the replicas start again: they are small when the others ask for room
            mask = (np.arange(R) % 2 == 0).astype(np.uint8)
            orc.reset(mask)
            eng.reset(torch.tensor(mask))
            out, buf = orc.observe(), eng.observe()
        if moved_at is not None and s >= moved_at + 40:
            break
    assert moved_at is not None and moved_at > 105, moved_at
    assert classes_before[0] >= 1 and classes_before[1] >= 1, (classes_before, moved_at)  # both classes were populated at the hand-over
    assert "state written across" in capfd.readouterr().err
    for r in range(R):
        assert all(np.array_equal(x, y) for x, y in zip(orc.snapshot(r), eng.snapshot(r))), r
    assert eng.errors() == 0 and orc.errors() == 0 and eng.path_counts()["stalled"] == 0
    eng.close()
    orc.close()
