"""Diagnostic (GPU box): how many vehicle records does a resident replica of the bench workload need over a whole
simulated hour under the random policy?  Prints the per-replica maximum of live vehicles every 60 env steps, the record
capacity of the handle and the device error counter (a spawn that finds no free record raises it)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch

import bench
from pytsc_amd.vec_env import VecEngine

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rate = float(sys.argv[2]) if len(sys.argv) > 2 else 700.0
tables = bench.workload_tables(4, 4, R, 0, rate, 8)
eng = VecEngine(tables, want=("mask", "terminated"), family=sys.argv[3] if len(sys.argv) > 3 else "auto")
acts = torch.empty((R, tables.n_agents), device="cuda", dtype=torch.int32)
buf = eng.observe()
counts = torch.zeros((R, 3), dtype=torch.int32, device="cuda")
import ctypes as C

for s in range(719):
    buf = eng.step(eng.random_valid_actions(buf["mask"], seed=7, out=acts))
    if s % 60 == 59 or s == 718:
        eng.lib.tsc_replica_counts(eng.h, C.c_void_p(counts.data_ptr()), None)
        torch.cuda.synchronize()
        live = counts[:, 1]
        q = torch.quantile(live.float(), torch.tensor([0.5, 0.9, 0.99], device=live.device)).tolist()
        print(f"step {s + 1}: live vehicles per replica mean {live.float().mean():.0f} p50 {q[0]:.0f} p90 {q[1]:.0f} p99 {q[2]:.0f} max {int(live.max())}, errors {eng.errors()}, family {eng.lib.tsc_kernel_family(eng.h)}", flush=True)
