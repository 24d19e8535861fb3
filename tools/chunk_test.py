"""Does running the env step chunk by chunk of replicas (state of a chunk stays in the
Infinity Cache across its 5 sub-steps) beat one pass over all replicas?  K handles of R/K
replicas each, stepped one after the other (same stream), against one handle of R."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import workload_tables
from pytsc_amd.vec_env import VecEngine

R = 4096
def run(K, streams=1, warm=120, steps=120):
    engs = [VecEngine(workload_tables(4, 4, R // K, k * (R // K), 700.0, 8), want=("obs", "mask", "local_reward", "global_reward", "done", "terminated")) for k in range(K)]
    bufs = [e.observe() for e in engs]
    ss = [torch.cuda.Stream() for _ in range(streams)]
    def step_all():
        for k, e in enumerate(engs):
            if streams > 1:
                with torch.cuda.stream(ss[k % streams]):
                    bufs[k] = e.step(e.random_valid_actions(bufs[k]["mask"], seed=k))
            else:
                bufs[k] = e.step(e.random_valid_actions(bufs[k]["mask"], seed=k))
    for _ in range(warm):
        step_all()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        step_all()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    v = sum(e.mean_vehicles() for e in engs) / K
    for e in engs:
        e.close()
    print(f"K={K} streams={streams}: {R * 16 * steps / dt / 1e6:.2f} M agent-env-steps/s, {dt / steps * 1e3:.3f} ms/step, {v:.0f} veh/replica", flush=True)

if __name__ == "__main__":
    for K, st in ((1, 1), (2, 1), (4, 1), (8, 1), (4, 2), (8, 2), (8, 4)):
        run(K, st)
