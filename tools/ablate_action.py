"""Diagnostic: price the parts of k_action on a frozen, congested state."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import workload_tables
from pytsc_amd.vec_env import VecEngine

tables = workload_tables(4, 4, 4096, 0, 700.0, 8)
eng = VecEngine(tables, want=("mask",))
buf = eng.observe()
g = torch.Generator(device="cuda"); g.manual_seed(0)
u = torch.empty((4096, 16, 16), device="cuda")
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 150):
    u.uniform_(1e-6, 1.0, generator=g)
    buf = eng.step((u * buf["mask"]).argmax(-1).to(torch.int32))
print("vehicles/replica", eng.mean_vehicles())
eng.lib.tsc_debug_time_action.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
ms = C.c_double()
def t(which, ab=0, iters=5):
    eng.lib.tsc_debug_time_action(eng.h, ab, which, iters, C.byref(ms))
    return round(ms.value * 1e3, 1)

print("notify", t(8), "enum", t(1), "enum(8rb)", t(1, 8 << 16), "commit", t(32))
t(1, 0, 1)  # lists valid for k_heavy
for ab in (0, 128, 64, 16, 1, 8 << 16, 16 << 16, 32 << 16):
    print("heavy ablate", ab, "us", t(4, ab))
for ab in (0, 128, 64, 16):
    print("links ablate", ab, "us", t(2, ab))
