"""Build the HIP library in-tree: pytsc_amd/lib/libpytsc_amd.so (gfx950 only)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "tsc_kernels.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "tsc_api.inc"), os.path.join(HERE, "csrc", "tsc_device.h"),
        os.path.join(HERE, "csrc", "tsc_resident.inc"),
        os.path.join(HERE, "..", "include", "pytsc_amd.h")]
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libpytsc_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared",
         "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False, extra=()):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [HIPCC] + FLAGS + list(extra) + ["-o", LIB, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="-f" in sys.argv, verbose=True,
          extra=["-Rpass-analysis=kernel-resource-usage"] if "-r" in sys.argv else ())
