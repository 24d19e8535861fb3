"""The two shortcuts of kr_step's measure phase (pytsc_amd/csrc/tsc_resident.inc, "The bin: ..." and "normalised speed")
against the expressions of the reference they stand for (pytsc/backends/cityflow/retriever.py:20-52 through
oracle/tsc_oracle.c): wherever the shortcut applies it must give the reference's number -- checked here on the host, on
random and on adversarial values (bin edges +- the rounding of round(x, 6)), with exact rational arithmetic standing in
for the kernel's fma.  The GPU parity tests then cover the kernel's use of them."""
import math
import random
from fractions import Fraction

import numpy as np


def _fast_bin(dist, length, bins):
    """kr_step's guess; None where the kernel takes the divisions instead."""
    bs = length / bins
    inv = 1.0 / bs
    kq = float(math.floor(dist * inv))
    rem = float(Fraction(dist) - Fraction(kq) * Fraction(bs))  # fma(-kq, bs, dist): one rounding of the exact value
    if kq >= 0.0 and rem > 1e-5 and rem < bs - 1e-5 and dist < length - 1e-5:
        return int(kq)
    return None


def _ref_bin(dist, length, bins):
    bs = length / bins
    pos = round(dist, 6)
    pos = min(max(pos, 0.0), length)
    idx = int(pos // bs)
    return min(idx, bins - 1)


def test_bin_guess_is_the_reference_bin_wherever_it_applies():
    rng = random.Random(5)
    lanes = [(300.0, 43), (292.5, 41), (150.0, 21), (742.31, 106), (87.3, 12), (7.0, 1), (1000.0, 142), (33.33, 4)]
    n_fast = n_slow = 0
    for length, bins in lanes:
        bs = length / bins
        cases = [rng.uniform(0.0, length) for _ in range(20000)]
        for k in range(bins + 1):
            for eps in (0.0, 1e-9, 1e-7, 4.9e-7, 5e-7, 5.1e-7, 1e-6, 9e-6, 9.99e-6, 1.0e-5, 1.01e-5, 1e-4, 1e-3):
                for sgn in (-1.0, 1.0):
                    x = k * bs + sgn * eps
                    if 0.0 <= x <= length:
                        cases.append(x)
                        cases.append(float(np.nextafter(x, 0.0)))
                        cases.append(float(np.nextafter(x, 1e9)))
        for x in cases:
            g = _fast_bin(x, length, bins)
            if g is None:
                n_slow += 1
                continue
            n_fast += 1
            assert min(g, bins - 1) == _ref_bin(x, length, bins), (x, length, bins)
    assert n_fast > 100000 and n_slow > 100  # both sides were exercised


def _k_of(speed):
    """round6 of the kernels without its final division: round(speed, 6) == k / 1e6."""
    k = float(np.rint(speed * 1e6))
    rr = float(Fraction(speed) * 1000000 - Fraction(k))
    if rr > 0.5:
        k += 1.0
    elif rr < -0.5:
        k -= 1.0
    return k


def _k_ge1(mx):
    """tsc_api.inc: smallest k >= 1 with (k / 1e6) / mx >= 1.0."""
    k = max(1, math.floor(mx * 1e6) - 4)
    while not (k / 1e6) / mx >= 1.0:
        k += 1
    return k


def test_speed_predicates_without_divisions():
    rng = random.Random(9)
    for mx in (11.11, 11.111, 13.89, 16.67, 8.333333, 5.0, 30.0, 1e-3, 11.110000000000001):
        K = _k_ge1(mx)
        assert (K / 1e6) / mx >= 1.0 and (K == 1 or not ((K - 1) / 1e6) / mx >= 1.0)
        speeds = [rng.uniform(0.0, 1.2 * mx) for _ in range(4000)] + [0.0, 4e-7, 5e-7, 5.0000001e-7, 6e-7, 1e-6, mx]
        for j in range(-30, 31):
            speeds.append(max(0.0, mx + j * 1e-7))
            speeds.append(max(0.0, (K + j) / 1e6))
            speeds.append(max(0.0, (K + j) / 1e6 + 4.9e-7))
            speeds.append(max(0.0, (K + j) / 1e6 - 4.9e-7))
        for s in speeds:
            k = _k_of(s)
            assert k / 1e6 == round(s, 6), s
            norm = 0.0 if s == 0.0 else round(s, 6) / mx
            assert (k == 0.0) == (norm == 0.0), (s, mx)
            assert (k >= K) == (norm >= 1.0), (s, mx)


def _no_collision_speed(vL, dL, vF, dF, gap, interval, target_gap):
    """pytsc_amd/csrc/tsc_kernels.hip::no_collision_speed, operation by operation (numpy f64, no fused operations)."""
    a = 0.5 / dF
    ia = 0.5 / a
    lb = 0.5 * vL * vL / dL
    c = vF * interval / 2 + target_gap - lb - gap
    b = 0.5 * interval
    with np.errstate(invalid="ignore"):
        v1 = ia * (np.sqrt(b * b - 4 * a * c) - b)
    v2 = 2 * vL - dL * interval + 2 * (gap - target_gap) / interval
    out = np.where(v1 < v2, v1, v2)
    return np.where(b * b < 4 * a * c, -100.0, out)


def test_car_following_bound_at_maximal_deceleration_is_redundant_when_the_decelerations_are_equal():
    """car_follow_speed skips the first of its two no_collision_speed evaluations when maxNegAcc == usualNegAcc for
    leader and follower: min(first, second) must then be the second, bit for bit."""
    rng = np.random.default_rng(3)
    n = 2_000_000
    for dec, min_gap, interval in ((4.5, 2.5, 1.0), (4.5, 0.0, 1.0), (3.0, 1.0, 0.5), (7.3, 2.5, 1.0), (4.5, 1e-9, 1.0)):
        vL = rng.uniform(0.0, 20.0, n)
        vF = rng.uniform(0.0, 20.0, n)
        gap = np.concatenate([rng.uniform(-10.0, 200.0, n // 2), rng.uniform(-1e-3, 5.0, n - n // 2)])
        for arr in (vL, vF):
            arr[rng.integers(0, n, n // 20)] = 0.0
        gap[rng.integers(0, n, n // 50)] = min_gap
        first = _no_collision_speed(vL, dec, vF, dec, gap, interval, 0.0)
        second = _no_collision_speed(vL, dec, vF, dec, gap, interval, min_gap)
        both = np.where(first < second, first, second)  # min2(first, second)
        assert np.array_equal(both, second)
