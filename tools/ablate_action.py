"""Diagnostic: price the parts of k_action on a frozen, congested state."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import workload_tables
from pytsc_amd.vec_env import VecEngine

R = int(os.environ.get('TSC_R', '4096'))
tables = workload_tables(4, 4, R, 0, 700.0, 8)
eng = VecEngine(tables, want=("mask",))
buf = eng.observe()
g = torch.Generator(device="cuda"); g.manual_seed(0)
u = torch.empty((R, 16, 16), device="cuda")
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 150):
    u.uniform_(1e-6, 1.0, generator=g)
    buf = eng.step((u * buf["mask"]).argmax(-1).to(torch.int32))
print("vehicles/replica", eng.mean_vehicles())
eng.lib.tsc_debug_time_action.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
ms = C.c_double()
def t(which, ab=0, iters=5):
    eng.lib.tsc_debug_time_action(eng.h, ab, which, iters, C.byref(ms))
    return round(ms.value * 1e3, 1)

print("notify", t(8), "enum", t(1), "commit", t(32))
for ab in (0, 256, 512, 768):
    print("enum lanes ablate", ab, t(1, ab))
t(1 | 64, 0, 1)  # lists and batch table valid for k_exec
print("enum+plan", t(1 | 64))
B = 1024 + 16
for ab in (0, 1024, B, B + 16384, B + 32768, B + 65536, B + 131072, B + 16384 + 32768 + 65536 + 131072):
    print("exec lanes ablate", ab, "us", t(4, ab))

import numpy as np
t(1 | 64, 0, 1)
buf_ = (C.c_int32 * 100000)()
eng.lib.tsc_debug_read_lists.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
n = eng.lib.tsc_debug_read_lists(eng.h, buf_, 100000)
a = np.frombuffer(buf_, dtype=np.int32)[:n]
nb = a[:2]; cnt = a[2:]
nl = tables.n_lanes
lane = cnt[: nl * 5].reshape(nl, 5)
link = cnt[nl * 5:]
print("batches lanes, links", nb, "n_lanes", nl, "n_links", len(link))
print("items per q", lane.sum(0), "links", link.sum())
print("batches per q", ((lane + 63) // 64).sum(0), "links", ((link + 63) // 64).sum())
print("nonempty lists per q", (lane > 0).sum(0))
