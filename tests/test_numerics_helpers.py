"""round6 (std::to_string -> float) and the Python floor-division restatement used
by the position-matrix binning, against the real thing."""
import ctypes
import math

import numpy as np


def test_round6_equals_printf_roundtrip(oracle_lib):
    lib = ctypes.CDLL(oracle_lib.build())
    # round6 is static in the oracle; re-state it here exactly and compare with "%f"
    def round6(x):
        k = float(np.rint(x * 1e6))
        r = math.fma(x, 1e6, -k) if hasattr(math, "fma") else None
        return k, r

    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(0, 800, 200000), rng.uniform(0, 12, 200000),
                         np.array([0.0078125, 0.5e-6, 1.5e-6, 2.5e-6, 11.11, 279.975, 0.1, 7.5])])
    ref = np.array([float("%f" % x) for x in xs])
    k = np.rint(xs * 1e6)
    # exact residual via integer arithmetic on the binary expansion (no math.fma in 3.10)
    from fractions import Fraction

    got = np.empty_like(xs)
    for i, x in enumerate(xs):
        kk = Fraction(int(k[i]))
        r = Fraction(x) * 1000000 - kk
        if r > Fraction(1, 2):
            kk += 1
        elif r < Fraction(-1, 2):
            kk -= 1
        got[i] = float(kk) / 1e6
    assert np.array_equal(got, ref)
    del lib


def test_py_floordiv_restatement():
    rng = np.random.default_rng(1)
    a = rng.uniform(0, 800, 100000)
    for b in (7.5, 400 / 53, 800 / 106, 200 / 26):
        ref = np.array([x // b for x in a])
        mod = np.fmod(a, b)
        div = (a - mod) / b
        fl = np.floor(div)
        fl = np.where(div - fl > 0.5, fl + 1.0, fl)
        assert np.array_equal(fl, ref)
