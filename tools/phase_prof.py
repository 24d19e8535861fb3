"""Diagnostic (GPU box): where kr_step's time goes, phase by phase.

Runs the bench workload on a -DTSC_DIAG build of the library (built beforehand with
`python tools/phase_prof.py --build`, in-tree as pytsc_amd/lib/libpytsc_amd_diag.so; the product
library has no timers) and prints the cycles thread 0 of a workgroup spends in each phase of an
env step (mean over workgroups and steps).  Usage: python tools/phase_prof.py [--replicas 4096]
"""
import argparse
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
DIAG = os.path.join(REPO, "pytsc_amd", "lib", "libpytsc_amd_diag.so")
PHASES = ["load", "links", "spawn0", "classify", "exec", "unlink", "append", "measure", "store", "agent",
          "stats"]


def build():
    from pytsc_amd import build as b

    os.environ["PYTSC_AMD_BUILD_ANY_VGPRS"] = "1"  # (the stamps cost registers: a diagnostic build's small class may run one per CU)
    b.build(force=True, verbose=True, extra=["-DTSC_DIAG"], out=DIAG)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--replicas", type=int, default=4096)
    ap.add_argument("--preroll", type=int, default=150)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rows", type=int, default=4)
    ap.add_argument("--cols", type=int, default=4)
    ap.add_argument("--threads", type=int, default=0, help="resident family: threads per workgroup")
    ap.add_argument("--ablate", type=lambda x: int(x, 0), default=0, help="ablation bits set after the pre-roll")
    ap.add_argument("--lib", default=DIAG, help="the -DTSC_DIAG build to profile")
    args = ap.parse_args()
    if args.build:
        build()
        return
    import torch

    from pytsc_amd import _lib

    _lib.LIB_PATH = os.path.abspath(args.lib)
    import bench
    from pytsc_amd.vec_env import VecEngine

    R = args.replicas
    tables = bench.workload_tables(args.rows, args.cols, R, 0, 700.0, 8)
    tables.cfg.resident_threads = args.threads
    eng = VecEngine(tables, device=0, want=tuple(bench.DEFAULT_OUTPUTS.split(",")), family="resident")
    A = tables.n_agents
    acts = torch.empty((R, A), device="cuda:0", dtype=torch.int32)
    buf = eng.observe()
    for _ in range(args.preroll):
        buf = eng.step(eng.random_valid_actions(buf["mask"], seed=1234, out=acts))
    out = (C.c_ulonglong * 64)()
    eng.lib.tsc_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int32]
    eng.lib.tsc_debug_phase_cycles(eng.h, out, 64)  # drop the pre-roll
    if args.ablate:
        eng.lib.tsc_debug_set_ablate(eng.h, args.ablate)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        buf = eng.step(eng.random_valid_actions(buf["mask"], seed=1234, out=acts))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    n = eng.lib.tsc_debug_phase_cycles(eng.h, out, 64)
    assert n == 64, n
    tot = sum(out[i] for i in range(len(PHASES)))
    print(f"R={R} V/replica={eng.mean_vehicles():.0f} {ms:.3f} ms/env step (diag build, ablate {args.ablate:#x}), errors {eng.errors()}")
    for i, name in enumerate(PHASES):
        per = out[i] / (R * args.steps)
        print(f"{name:10s} {per:12.0f} cycles/workgroup/env-step  {100.0 * out[i] / tot:5.1f} %")
    print(f"{'total':10s} {tot / (R * args.steps):12.0f}")
    n = R * args.steps
    print(f"exec, per wave per env step: heavy chunks {out[11] / n:.0f} cycles (all waves), follower chunks {out[12] / n:.0f} "
          f"(16 waves; the phase lasts until the slowest wave is done)")
    print("agent sub-phases (wave 0): sums %.0f, stats+mask %.0f, obs row build %.0f, obs row store %.0f" % tuple(out[i] / n for i in (16, 17, 18, 19)))
    print("stats sums (one wave): %.0f" % (out[28] / n))
    print("exec stages per env step: A + followers %.0f, B (tasks) %.0f, C (finish) %.0f" % tuple(out[i] / n for i in (20, 21, 22)))
    print("stage A, first heavy chunk, per env step: half A1 (leader/gap + car-following) %.0f, half A2 (signal + tasks) %.0f" % (out[23] / n, out[24] / n))
    for name, slot, parts in (("stage A1", 32, ("gathers", "leader + gap", "car-following")), ("stage A2", 36, ("signal gating", "tasks")),
                              ("stage B task", 40, ("gathers", "cross evaluation")), ("stage C", 44, ("join", "finish + commit")),
                              ("  cross evaluation", 52, ("cross record", "foe", "canPass"))):
        cnt = max(int(out[slot]), 1)
        print(f"{name}: {cnt / n / 5:.1f} wavefront runs per sub-step; cycles per run: " +
              ", ".join(f"{q} {out[slot + 1 + i] / cnt:.0f}" for i, q in enumerate(parts)))
    print(f"items per sub-step: heavy {out[13] / n / 5:.1f}, followers {out[14] / n / 5:.1f}, cross candidates {out[15] / n / 5:.1f}")


if __name__ == "__main__":
    main()
