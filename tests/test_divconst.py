"""div_c of the HIP kernels (pytsc_amd/csrc/tsc_kernels.hip): a / b by five multiply-adds with the correctly rounded reciprocal
of a compile-time divisor, used by the vehicle-specialised instantiations of kr_step for the divisions by maxNegAcc,
usualNegAcc, maxSpeed and turnSpeed.  Markstein's theorem makes the result THE correctly rounded quotient; this checks the
same C sequence against the IEEE division on the host, bit for bit: random numerators over the ranges the engine divides
(squared speeds, distances), the neighbours of exact multiples and of rounding midpoints' pre-images, tiny and huge values
-- for every divisor the shipped vehicles have.  The GPU parity suite covers the kernels' use of it."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO

DIVISORS = [4.5, 11.11, 11.111, 8.3333, 2.5, 5.0, 7.5, 13.89, 3.0, 1.5]


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("divconst") / "libdivconst.so")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", out,
                           os.path.join(REPO, "tests", "native", "divconst.c"), "-lm"])
    lib = ctypes.CDLL(out)
    lib.divconst_mismatches.restype = ctypes.c_long
    lib.divconst_mismatches.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
    return lib


def _inputs(b, rng):
    xs = [rng.uniform(0.0, 200.0, 2_000_000),            # 0.5 v^2, v^2, distances
          rng.uniform(-300.0, 300.0, 500_000),             # (distance - dut) may be negative
          10.0 ** rng.uniform(-30, 30, 500_000),           # far from the engine's ranges: still exact
          -(10.0 ** rng.uniform(-30, 30, 100_000))]
    # quotients that sit next to a representable number or to a midpoint between two: q * b for q near k * 2^-e
    q = rng.uniform(0.0, 64.0, 300_000)
    q = np.round(q * 2.0 ** 20) / 2.0 ** 20
    near = q * b
    for k in range(-3, 4):
        xs.append(near * (1.0 + k * 2.0 ** -52))
    half = (q + 2.0 ** -53 * np.maximum(1.0, 2.0 ** np.floor(np.log2(np.maximum(q, 1e-9))))) * b
    for k in range(-3, 4):
        xs.append(half * (1.0 + k * 2.0 ** -52))
    xs.append(np.array([0.0, 1e-290, 1e290, b, 2 * b, b / 3, np.nextafter(b, 0), np.nextafter(b, 100)]))
    return np.ascontiguousarray(np.concatenate(xs))


@pytest.mark.parametrize("b", DIVISORS)
def test_div_c_is_the_ieee_division(lib, b):
    a = _inputs(b, np.random.default_rng(int(b * 1000)))
    first = ctypes.c_double(0)
    bad = lib.divconst_mismatches(a.ctypes.data, a.size, b, ctypes.byref(first))
    assert bad == 0, (b, bad, first.value)


def test_div_c_with_per_lane_divisors(lib):
    """Round 4: k_net_sums divides a lane's vehicle count by its slot count (pytsc length / 7.5) and its mean speed by its
    speed limit through div_c with reciprocals the host computes per lane (tsc_api.inc: lane_rslots, lane_rmax; a network
    with a divisor whose significand is all ones keeps the divisions).  Checked here for the shipped lane lengths and speed
    limits and for random divisors of their kind: counts are small integers, means are averages of speeds."""
    rng = np.random.default_rng(2026)
    lens = [300.0, 292.5, 280.0, 260.0, 150.0, 400.0, 800.0, 742.31, 87.3, 1000.0, 33.33]
    divisors = [L / 7.5 for L in lens] + [11.11, 11.111, 13.89, 16.67, 8.33, 5.56, 22.22, 10000.0]
    divisors += list(rng.uniform(0.5, 300.0, 16))
    counts = np.arange(0, 600, dtype=np.float64)
    for b in divisors:
        sums = np.round(rng.uniform(0.0, 600.0, 200_000) * 1e6) / 1e6
        n = rng.integers(1, 64, sums.size)
        a = np.ascontiguousarray(np.concatenate([counts, rng.uniform(0.0, 40.0, 200_000), sums / n, b * counts[:64], b * (counts[:64] + 0.5)]))
        first = ctypes.c_double(0)
        bad = lib.divconst_mismatches(a.ctypes.data, a.size, float(b), ctypes.byref(first))
        assert bad == 0, (b, bad, first.value)
