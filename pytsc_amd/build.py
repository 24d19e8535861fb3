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
         "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
         # the kernels are chains of dependent LDS / f64 operations at an occupancy the LDS fixes: schedule for
         # instruction-level parallelism, not for register pressure (+0.8 % on kr_step, same-box A/B; same results)
         "-mllvm", "-amdgpu-sched-strategy=max-ilp"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def check_no_vector_spills(report, kernel="kr_step"):
    """The resident kernel must not spill vector registers: a spill store placed under a partial EXEC mask
    loses the inactive lanes (seen with the 1024-thread build of round 2: a loop counter reloaded as garbage,
    the launch never ended).  `report` is hipcc's -Rpass-analysis=kernel-resource-usage output."""
    bad, name = [], None
    for line in report.splitlines():
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split()[0]
        elif name and kernel in name and ("VGPRs Spill:" in line or "ScratchSize [bytes/lane]:" in line):
            if int(line.split(":")[-1].split()[0]) != 0:
                bad.append((name, line.split("remark:")[-1].strip()))
    if bad:
        raise RuntimeError("vector register spills in the resident kernel: %r" % bad)
    # the small class (512 threads) is worth building only when TWO of its workgroups fit a CU: 16 wavefronts = 4 per SIMD
    # = at most 128 vector registers each (1.5-1.7x; with more the hardware silently runs one workgroup per CU)
    fat, name = [], None
    for line in report.splitlines():
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split()[0]
        elif name and kernel in name and "ILi" in name and "Li512E" in name and " VGPRs:" in line and "Spill" not in line:
            n = int(line.split("VGPRs:")[1].split()[0])
            if n > 128:
                fat.append((name, n))
    if fat and not os.environ.get("PYTSC_AMD_BUILD_ANY_VGPRS"):
        raise RuntimeError("512-thread instantiations of the resident kernel need <= 128 vector registers: %r" % fat)


def build(force=False, verbose=False, extra=(), out=None, report=False):
    out = out or LIB
    if out == LIB and not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [HIPCC] + FLAGS + ["-Rpass-analysis=kernel-resource-usage"] + list(extra) + ["-o", out, SRC]
    if verbose:
        print(" ".join(cmd))
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if p.returncode != 0 or report:
        print(p.stdout)
    else:
        for line in p.stdout.splitlines():
            if ": warning:" in line:
                print(line)
    if p.returncode != 0:
        raise subprocess.CalledProcessError(p.returncode, cmd)
    try:
        check_no_vector_spills(p.stdout)
    except RuntimeError:
        os.remove(out)
        raise
    return out


if __name__ == "__main__":
    build(force="-f" in sys.argv, verbose=True, report="-r" in sys.argv)
