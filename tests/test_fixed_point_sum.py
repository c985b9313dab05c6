"""Exact floating SUMs (pinot_amd/csrc/pg_fixed_point.h): the digit extraction the kernels run and the limb combination the
host runs, compiled into a CPU harness and compared with the exact rational sum rounded once (fractions.Fraction).  north_star:
floating SUM within 1 ulp of the reference; the reference's own sequential double sum is what drifts — see
tests/test_gpu_sum_exactness.py for the GPU leg."""
import math
import os
import struct
import subprocess
from fractions import Fraction

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "harness", "fxsum_check.cpp")
BIN = os.path.join(ROOT, "tools", "harness", "fxsum_check")


@pytest.fixture(scope="module")
def harness():
    hdr = os.path.join(ROOT, "pinot_amd", "csrc", "pg_fixed_point.h")
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", BIN, SRC])
    return BIN


def fx_exp_of(max_abs):
    if not max_abs > 0:
        return 0
    _, e = math.frexp(max_abs)
    return 16 * math.floor((e + 15) / 16)


def round_fraction(fr: Fraction) -> float:
    """nearest-even double of an exact rational (Python's int / int true division is correctly rounded)"""
    return fr.numerator / fr.denominator


def run(harness, mode, q, limbs, values):
    if mode == "d":
        body = "\n".join(f"{struct.unpack('<Q', struct.pack('<d', float(v)))[0]:x}" for v in values)
    else:
        body = "\n".join(str(int(v)) for v in values)
    out = subprocess.run([harness], input=f"{mode} {q} {limbs} {len(values)}\n{body}\n", capture_output=True, text=True, check=True)
    return struct.unpack("<d", struct.pack("<Q", int(out.stdout.strip(), 16)))[0]


def ulp(x):
    return math.ulp(x) if x != 0 else 5e-324


@pytest.mark.parametrize("seed", range(6))
def test_double_sums_are_correctly_rounded(harness, seed):
    rng = np.random.default_rng(seed)
    n = 20000
    kinds = [rng.normal(0, 1e3, n), rng.uniform(0, 1, n) * 1e-3, rng.lognormal(0, 6, n) * rng.choice([-1, 1], n),
             rng.integers(-2**40, 2**40, n).astype(np.float64) / 7.0, np.full(n, 0.1), rng.uniform(-1, 1, n) * 2.0**300]
    v = kinds[seed % len(kinds)]
    q = fx_exp_of(float(np.abs(v).max())) - 32 * 4 + 1
    got = run(harness, "d", q, 4, v)
    exact = sum(Fraction(float(x)) for x in v)
    want = round_fraction(exact)
    # every value here lies within 2^-59 of the largest magnitude or the truncation stays far below an ulp: correctly rounded
    assert abs(Fraction(got) - exact) <= Fraction(ulp(want)), (got, want)
    if seed in (0, 3, 4):
        assert got == want
    # the reference's order (sequential double accumulation) is the one that drifts
    seq = 0.0
    for x in v:
        seq += float(x)
    assert abs(Fraction(got) - exact) <= abs(Fraction(seq) - exact)


def test_float_sources_three_limbs(harness):
    rng = np.random.default_rng(9)
    v = rng.uniform(-1e6, 1e6, 50000).astype(np.float32).astype(np.float64)
    q = fx_exp_of(float(np.abs(v).max())) - 32 * 3 + 1
    got = run(harness, "d", q, 3, v)
    assert got == round_fraction(sum(Fraction(float(x)) for x in v))


def test_tiny_values_are_truncated_not_lost_wholesale(harness):
    """values far below the column's largest magnitude lose only what lies under 2^q"""
    v = np.array([1e30] + [1.0] * 1000 + [-1e30])
    q = fx_exp_of(1e30) - 32 * 4 + 1
    got = run(harness, "d", q, 4, v)
    assert got == 1000.0          # 2^q = 2^(112 - 127) < 1: the ones are exact
    v = np.array([2.0**100] + [2.0**-40] * 4096 + [-(2.0**100)])
    got = run(harness, "d", fx_exp_of(2.0**100) - 127, 4, v)
    assert got == 0.0             # below 2^q = 2^-15: truncated (an absolute error of 2^-28 against a largest magnitude of 2^100)


def test_long_two_digit_sums(harness):
    rng = np.random.default_rng(3)
    v = rng.integers(-2**62, 2**62, 30000, dtype=np.int64)
    got = run(harness, "l", 0, 2, v)
    exact = sum(int(x) for x in v)
    assert got == round_fraction(Fraction(exact))
    v = np.full(100000, 2**62, dtype=np.int64)      # 2^62 * 1e5 wraps int64 many times over
    assert run(harness, "l", 0, 2, v) == float(2**62 * 100000)
    v = np.array([-2**63, -2**63, 2**63 - 1, -1, 0, 1], dtype=np.int64)
    assert run(harness, "l", 0, 2, v) == float(sum(int(x) for x in v))


def test_denormals_and_signed_zero(harness):
    v = np.array([5e-324, 5e-324, -5e-324, 0.0, -0.0, 2.2250738585072014e-308])
    q = fx_exp_of(2.2250738585072014e-308) - 127
    got = run(harness, "d", q, 4, v)
    assert got == round_fraction(sum(Fraction(float(x)) for x in v))
