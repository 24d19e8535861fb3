"""Diagnostic (GPU box, -DTSC_DIAG library built by `python tools/phase_prof.py --build`): two fallback paths of the
resident kernel that ordinary runs reach rarely or never, forced on copies of one engine and compared bit for bit:
* 0x400  the measure phase numbers the records from their entry numbers (lanes are FIFO) and walks the lists only when
         a vehicle ever left from behind another one;
* 0x800  hand-offs relink themselves unless two of them touch the same list end in one sub-step: then (here: always)
         the per-drivable two-phase path runs (unlink per source, ordered append per target)."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch

from pytsc_amd import _lib

_lib.LIB_PATH = os.path.join(REPO, "pytsc_amd", "lib", "libpytsc_amd_diag.so")
import bench
from pytsc_amd.vec_env import VecEngine

R = 64
engs = [VecEngine(bench.workload_tables(3, 3, R, 0, 1200.0, 4), family="resident") for _ in range(3)]
engs[1].lib.tsc_debug_set_ablate.argtypes = [C.c_void_p, C.c_int]
engs[1].lib.tsc_debug_set_ablate(engs[1].h, 0x400)
engs[2].lib.tsc_debug_set_ablate(engs[2].h, 0x800)
bufs = [e.observe() for e in engs]
g = torch.Generator(device="cuda")
g.manual_seed(0)
for s in range(120):
    u = torch.rand(bufs[0]["mask"].shape, device="cuda", generator=g).clamp_min(1e-6)
    acts = (u * bufs[0]["mask"]).argmax(-1).to(torch.int32)
    bufs = [e.step(acts) for e in engs]
    for k in ("obs", "state", "mask", "stats", "local_reward64", "agent_stats"):
        assert torch.equal(bufs[0][k], bufs[1][k]), (s, k, "walk")
        assert torch.equal(bufs[0][k], bufs[2][k]), (s, k, "two-phase hand-offs")
for r in (0, 17, R - 1):
    a, b, c = engs[0].snapshot(r), engs[1].snapshot(r), engs[2].snapshot(r)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)), r
    assert all(np.array_equal(x, y) for x, y in zip(a, c)), r
print("forced fallbacks == default paths over 120 steps,", engs[0].mean_vehicles(), "vehicles/replica, errors",
      [e.errors() for e in engs])
