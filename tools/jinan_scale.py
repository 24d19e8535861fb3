"""BASELINE config 4 size check: Jinan 3x4 (irregular lanes / phases, real flows) at 8192 replicas."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from fixtures import load_case
from pytsc_amd.vec_env import VecEngine

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
FAMILY = sys.argv[2] if len(sys.argv) > 2 else "auto"
_, _, _, _, tables = load_case("jinan_3_4_random", n_replicas=R)
eng = VecEngine(tables, want=("obs", "mask", "global_reward", "done", "terminated", "stats"), family=FAMILY)
buf = eng.observe()
A = buf["mask"].shape[1]
g = torch.Generator(device="cuda"); g.manual_seed(0)
u = torch.empty(buf["mask"].shape, device="cuda")
def step():
    u.uniform_(1e-6, 1.0, generator=g)
    return eng.step((u * buf["mask"]).argmax(-1).to(torch.int32))
for _ in range(60):
    buf = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 120
for _ in range(n):
    buf = step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"jinan R={R} A={A} family={eng.family}: {R * A * n / dt / 1e6:.2f} M agent-env-steps/s, {dt / n * 1e3:.2f} ms/step, "
      f"{eng.mean_vehicles():.0f} vehicles/replica, errors {eng.errors()}")
