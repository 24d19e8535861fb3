"""Diagnostic (GPU box): the hour's transition between the two classes of the resident family -- env steps 0-360 of the default
4x4 @ 4096 workload in 10-step windows: ms per step, class split, steps redone, class changes, mean vehicles; PROBE_MODE=mix<share>
then restarts that share of the replicas (a steady mix of full-size and light small replicas) and times 80 more steps.
A/B through the environment knobs of INTEGRATION.md (PYTSC_AMD_CLS_MARGIN, PYTSC_AMD_SMALL_MIN_PCT, PYTSC_AMD_SMALL_HEAVY_PCT, PYTSC_AMD_NO_SMALL)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
tables = bench.workload_tables(4, 4, 4096, 0, 700.0, 4096)
import torch
from pytsc_amd.vec_env import VecEngine
eng = VecEngine(tables, device=0, want=tuple(bench.DEFAULT_OUTPUTS.split(",")), family="auto")
R, A = 4096, tables.n_agents
acts = torch.empty((R, A), device="cuda", dtype=torch.int32)
buf = eng.observe()
mode = os.environ.get("PROBE_MODE", "")
rows = []
prev = eng.path_counts()
for w in range(36):
    cs, cb = eng.class_counts()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(10):
        buf = eng.step(eng.random_valid_actions(buf["mask"], seed=1234, out=acts))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 100.0
    c = eng.path_counts()
    rows.append((w * 10, round(ms, 4), cs, cb, c["class_redo"] - prev["class_redo"], c["class_change"] - prev["class_change"], round(eng.mean_vehicles(), 0), c["serial_scan"] - prev["serial_scan"]))
    prev = c
    print(rows[-1], flush=True)
print("total ms", sum(r[1] for r in rows) * 10)
if mode.startswith("mix"):
    # a steady mix: a share of the replicas starts again (light, small class), the others stay full-size
    share = float(mode[3:]) / 100.0
    import numpy as np
    mask = (np.arange(R) < int(R * share)).astype(np.uint8)
    eng.reset(torch.tensor(mask))
    buf = eng.observe()
    for w in range(8):
        cs, cb = eng.class_counts()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(10):
            buf = eng.step(eng.random_valid_actions(buf["mask"], seed=1234, out=acts))
        torch.cuda.synchronize()
        print("mix", share, w * 10, round((time.perf_counter() - t0) * 100.0, 4), cs, cb, round(eng.mean_vehicles(), 0), flush=True)
