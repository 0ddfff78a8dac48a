"""The exactness argument behind the block-wise fall-back of the latency tier (csrc/serial_kernels.hip, DESIGN.md 4 K4 iii), checked on the CPU
with Python floats (IEEE doubles, round to nearest even) against the reference's chain `acc += double(float term)` (DmsaOptimizer.h:259-264):

    S a multiple of 2^a (a = its lowest set bit), every term of a block a positive float, i.e. a multiple of 2^q_b (q_b = exponent of the
    smallest term - 23), g = min(a, q_b), S + s_b < 2^(g + 53)   =>   the chain leaves the block with exactly S + s_b, and s_b, summed in ANY
    order, is exact.

The model makes the same decisions as the kernel (the bound with its 2^-30 slack on a block sum that was added up in a shuffled order) and chains
the blocks that fail; the result must have the chain's bits."""
import math
import random
from fractions import Fraction

import numpy as np


def chain(terms, acc=0.0):
    for t in terms:
        acc = acc + float(t)
    return acc


def lowest_set_bit_exponent(x):
    m, e = math.frexp(x)  # x = m 2^e, 0.5 <= m < 1
    f, k = Fraction(m), 0
    while f.denominator != 1:  # m 2^k is an odd integer for the smallest such k: x = odd * 2^(e - k)
        f *= 2
        k += 1
    return e - k


def block_ok(S, sb, smallest):
    if not (smallest > 0.0):
        return False
    g = math.frexp(float(smallest))[1] - 1 - 23  # exponent of the smallest float term - 23
    if S != 0.0:
        if S < 2.0 ** -1022 or math.isinf(S) or math.isnan(S):
            return False
        g = min(g, lowest_set_bit_exponent(S))
    limit = 2.0 ** min(max(g + 53, -1023), 1023)
    return (S + sb) * (1.0 + 2.0 ** -30) < limit


def block_walk(terms, nblk, rng):
    n = len(terms)
    bsz = (n + nblk - 1) // nblk
    S, chained = 0.0, 0
    for w in range(nblk):
        blk = terms[w * bsz:(w + 1) * bsz]
        if len(blk) == 0:
            continue
        shuffled = list(blk)
        rng.shuffle(shuffled)
        sb = chain(shuffled)  # "any order": what the lanes of the parallel pass add up
        if block_ok(S, sb, float(min(blk))):
            assert Fraction(sb) == sum(Fraction(float(t)) for t in blk)  # the bound also makes the block's own sum exact
            S = S + sb
        else:
            S = chain(blk, S)
            chained += 1
    return S, chained


def test_block_walk_has_the_bits_of_the_chain():
    rng = random.Random(7)
    nprng = np.random.default_rng(7)
    total_chained = total_blocks = 0
    for case in range(300):
        n = rng.randrange(50, 4000)
        # terms like the kernel's: float32, positive, spread over ~2^-18 .. 2^-2, scaled so that the sum lands anywhere between 2^-3 and 2^7
        t = np.exp2(nprng.uniform(-18.0, -2.0, n)).astype(np.float32) * np.float32(2.0 ** rng.randrange(-4, 6))
        for _ in range(rng.randrange(0, 5)):  # members (almost) on the mean: tiny or zero terms anywhere in the list
            t[rng.randrange(n)] = np.float32(2.0 ** rng.uniform(-70.0, -25.0)) if rng.random() < 0.8 else np.float32(0.0)
        terms = [np.float32(v) for v in t]
        nblk = rng.choice([8, 10])
        got, chained = block_walk(terms, nblk, rng)
        want = chain(terms)
        assert got == want and math.copysign(1.0, got) == math.copysign(1.0, want), (case, got, want)
        total_chained += chained
        total_blocks += nblk
    # the cases really exercise both branches
    assert 0 < total_chained < total_blocks


def test_lowest_set_bit_exponent():
    assert lowest_set_bit_exponent(1.0) == 0 and lowest_set_bit_exponent(0.75) == -2 and lowest_set_bit_exponent(40.0) == 3
    assert lowest_set_bit_exponent(1.0 + 2.0 ** -52) == -52
