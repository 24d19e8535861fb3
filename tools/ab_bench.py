"""Diagnostic (GPU box): same-box A/B of builds of the library.

  python tools/ab_bench.py [--steps 40] [--warmup 5] [--rounds 2] [--bench-args "--rate 150"] libA.so libB.so ...

Runs bench.py's workload (no CPU baseline) once per library and round, each in a fresh process with
pytsc_amd._lib.LIB_PATH pointed at that build, and prints ms per env step and agent-env-steps/s side by side.
Variants are built beforehand into ab/ (git-ignored, travels with gpurun):
  python -c "from pytsc_amd import build; build.build(force=True, out='ab/x.so', extra=['-DSOMETHING'])"
"""
import argparse
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(lib, steps, warmup, extra):
    sys.path.insert(0, REPO)
    from pytsc_amd import _lib

    _lib.LIB_PATH = os.path.abspath(lib)
    import bench

    return bench.main(["--no-cpu-baseline", "--no-extras", "--steps", str(steps), "--warmup", str(warmup)] + extra)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--one", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--bench-args", default="", help='further bench.py arguments as ONE string, e.g. --bench-args "--rate 150 --family resident" '
                    "(round 5: unknown options used to be passed through, and the positional `libs` swallowed their values)")
    ap.add_argument("libs", nargs="*")
    args = ap.parse_args()
    extra = args.bench_args.split()
    if args.one:
        sys.exit(one(args.one, args.steps, args.warmup, extra))
    res = {lib: [] for lib in args.libs}
    for _ in range(args.rounds):
        for lib in args.libs:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", lib, "--steps", str(args.steps), "--warmup",
                                str(args.warmup), "--bench-args", args.bench_args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            line = [x for x in p.stdout.splitlines() if x.startswith("{")]
            if p.returncode != 0 or not line:
                print(f"{lib}: FAILED rc={p.returncode}\n{p.stderr[-2000:]}")
                continue
            d = json.loads(line[-1])
            res[lib].append((d["ms_per_step"], d["value"], d["config"].get("device_errors")))
    for lib in args.libs:
        print(f"{lib:28s} " + "  ".join(f"{ms:.4f} ms {v / 1e6:.2f} M err {e}" for ms, v, e in res[lib]))


if __name__ == "__main__":
    main()
