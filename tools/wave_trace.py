"""Diagnostic (GPU box, -DTSC_DIAG library built by `python tools/phase_prof.py --build`): the timeline of ONE workgroup of
kr_step -- when does each of its twelve wavefronts arrive at and leave every barrier of an env step?  Prints, per phase
(the stretch between two barriers), the cycles each wavefront worked before it reached the barrier: the phase lasts as long
as its slowest wavefront, the others wait.  usage: python tools/wave_trace.py [--lib path] [--preroll 150] [--replicas 4096]"""
import argparse
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
NAMES = {1: "load", 2: "links (scan)", 3: "links + actions", 10: "spawn of sub-step 0", 11: "classify + notify", 12: "stage A",
         13: "stage B + followers", 14: "commits + stage C", 15: "two-phase unlink", 16: "hand-offs + spawn", 20: "measure (scan)",
         21: "measure", 22: "store + lane terms", 23: "signals + sums", 30: "  (zeroing)", 31: "  (notify: lane-links)",
         32: "  (signal sums)", 33: "  (stats + mask)", 34: "  (obs row build)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(REPO, "pytsc_amd", "lib", "libpytsc_amd_diag.so"))
    ap.add_argument("--preroll", type=int, default=150)
    ap.add_argument("--replicas", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=8, help="env steps traced (workgroup 0 = replica 0 each time)")
    ap.add_argument("--rate", type=float, default=700.0, help="veh/h per entry road (150: a near-empty network -- the floor of a step)")
    ap.add_argument("--sets", type=int, default=8)
    ap.add_argument("--waves", type=int, default=8, help="wavefronts of the traced workgroup: 8 = the small class (512 threads: what replica 0 "
                    "runs in while it needs few records), 12 = the full-size class (PYTSC_AMD_NO_SMALL=1 in the environment)")
    args = ap.parse_args()
    import numpy as np
    import torch

    from pytsc_amd import _lib

    _lib.LIB_PATH = os.path.abspath(args.lib)
    import bench
    from pytsc_amd.vec_env import VecEngine

    R = args.replicas
    tables = bench.workload_tables(4, 4, R, 0, args.rate, args.sets)
    eng = VecEngine(tables, device=0, want=tuple(bench.DEFAULT_OUTPUTS.split(",")), family="resident")
    acts = torch.empty((R, tables.n_agents), device="cuda:0", dtype=torch.int32)
    buf = eng.observe()
    for _ in range(args.preroll):
        buf = eng.step(eng.random_valid_actions(buf["mask"], seed=1234, out=acts))
    NW, EV = 12, 128  # (the buffer holds twelve wavefronts' events; the first `--waves` are looked at)
    out = (C.c_ulonglong * (2 * EV * NW))()
    eng.lib.tsc_debug_wave_trace.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int32]
    eng.lib.tsc_debug_wave_trace(eng.h, out, len(out))  # drop the pre-roll
    work = {}   # (order, id) -> list of [NW] work cycles
    waitv = {}
    for s in range(args.samples):
        buf = eng.step(eng.random_valid_actions(buf["mask"], seed=1234, out=acts))
        n = eng.lib.tsc_debug_wave_trace(eng.h, out, len(out))
        assert n == len(out)
        a = np.array(out[:], dtype=np.uint64).reshape(NW, EV, 2)[: args.waves]
        ids = (a[0, :, 0] >> np.uint64(56)).astype(int)
        arrive = (a[:, :, 0] & np.uint64((1 << 56) - 1)).astype(np.int64)
        leave = (a[:, :, 1] & np.uint64((1 << 56) - 1)).astype(np.int64)
        n_ev = int((ids > 0).sum())
        prev = None
        for e in range(n_ev):
            start = leave[:, e - 1] if e else np.full(args.waves, arrive[:, 0].min())
            w = arrive[:, e] - start           # cycles each wavefront worked in this phase
            wt = leave[:, e].max() - arrive[:, e]  # ... and waited at the barrier
            work.setdefault((e, ids[e]), []).append(w)
            waitv.setdefault((e, ids[e]), []).append(wt)
    print(f"kr_step, workgroup 0 (replica 0), mean of {args.samples} env steps after {args.preroll}; cycles per wavefront between barriers")
    print(f"{'#':>3} {'phase':24s} {'longest':>8} {'mean':>7} {'shortest':>8}   work per wavefront 0..{args.waves - 1}")
    tot_long = tot_mean = 0
    for (e, i) in sorted(work):
        w = np.mean(work[(e, i)], axis=0)
        tot_long += w.max()
        tot_mean += w.mean()
        print(f"{e:3d} {NAMES.get(i, str(i)):24s} {w.max():8.0f} {w.mean():7.0f} {w.min():8.0f}   " + " ".join(f"{x:5.0f}" for x in w))
    print(f"sum of the longest wavefront per phase {tot_long:.0f}; sum of the mean wavefront {tot_mean:.0f}: "
          f"{100 * (1 - tot_mean / tot_long):.0f} % of the wave-cycles between barriers are waiting for the slowest wavefront")


if __name__ == "__main__":
    main()
