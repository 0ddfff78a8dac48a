"""The proof obligation behind the parallel second pass of the correspondence kernels (serial_kernels.hip, "second pass without a
chain"), checked on the CPU with exact rational arithmetic.

Claim: float terms t_j > 0, q = min_j (exponent(t_j) - 23).  If U * (1 + 2^-30) < 2^(q + 53) for the computed sum U, then the
reference's member-by-member double chain (DmsaOptimizer.h:259-264), ANY other summation order and the exact sum are the same
double.  The kernel evaluates the condition with the integer logic restated below (high words of the terms as doubles)."""
import math
from fractions import Fraction

import numpy as np
import pytest


def kernel_condition(terms32: np.ndarray, U: float) -> bool:
    """The test of parallel_second_pass: key = smallest high word of the terms as doubles, 0 for a term <= 0 or NaN."""
    t = terms32.astype(np.float32)
    if not np.all(t > 0):  # NaN compares false
        return False
    hi = (t.astype(np.float64).view(np.uint64) >> np.uint64(32)).astype(np.uint32)
    q = int(hi.min() >> 20) - 1023 - 23
    pe = min(max(q + 53 + 1023, 0), 2046)
    limit = np.array([pe << 52], np.uint64).view(np.float64)[0]
    return bool(U * (1.0 + 2.0 ** -30) < limit)


def chain(terms32):
    acc = 0.0
    for t in terms32:
        acc = acc + float(t)
    return acc


def tree(terms32):
    v = [float(t) for t in terms32]
    while len(v) > 1:
        v = [v[i] + v[i + 1] if i + 1 < len(v) else v[i] for i in range(0, len(v), 2)]
    return v[0]


def strided(terms32, k=7):
    parts = [0.0] * k
    for i, t in enumerate(terms32):
        parts[i % k] += float(t)
    return sum(parts[1:], parts[0])


@pytest.mark.parametrize("seed", range(8))
def test_condition_implies_order_independence(seed):
    rng = np.random.default_rng(seed)
    checked = 0
    for n in (10, 129, 1000, 14213):
        for spread in (4, 12, 20, 28, 36):  # binary orders of magnitude between the smallest and the largest term
            e = rng.uniform(-spread, 0, n)
            terms = (rng.uniform(1.0, 2.0, n) * 2.0 ** e * 2.0 ** rng.integers(-20, 20)).astype(np.float32)
            U = tree(terms)
            if kernel_condition(terms, U):
                exact = sum(Fraction(float(t)) for t in terms)
                assert Fraction(chain(terms)) == exact == Fraction(U) == Fraction(strided(terms))
                checked += 1
    assert checked >= 10  # the narrow spreads all satisfy the condition


def test_condition_is_not_vacuous():
    """Wide spreads fail the condition, and then the orders really differ -- the chain fallback is needed, not decoration."""
    rng = np.random.default_rng(99)
    differing = 0
    for _ in range(50):
        n = 2000
        terms = (rng.uniform(1.0, 2.0, n) * 2.0 ** rng.uniform(-45, 0, n)).astype(np.float32)
        U = tree(terms)
        assert not kernel_condition(terms, U)
        differing += chain(terms) != U
    assert differing > 25


def test_condition_rejects_zero_negative_and_nan_terms():
    base = np.full(100, 0.25, np.float32)
    assert kernel_condition(base, tree(base))
    for bad in (0.0, -0.25, np.nan, -0.0):
        t = base.copy()
        t[37] = bad
        assert not kernel_condition(t, tree(np.nan_to_num(t)))


def test_bench_like_margins():
    """Rebalanced sums of ~69 with smallest terms of ~2^-18 (what the bench window's largest Gaussians look like): 4 bits to spare."""
    rng = np.random.default_rng(5)
    n = 14213
    terms = np.maximum(rng.gamma(1.5, 69.0 / (1.5 * n), n), 2.0 ** -18).astype(np.float32)
    U = chain(terms)
    assert kernel_condition(terms, U) and U == tree(terms) == math.fsum(float(t) for t in terms)
