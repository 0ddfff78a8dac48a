"""The voxelisation's hand-written device primitives on caller data (test hooks of the C ABI): the onesweep radix sort (32-bit keys, and
64-bit keys as two halves), the prefix scans and the single-pass leaf segmentation -- no vendor library is linked.  Bit-exact against numpy: a stable sort is unique, and so is the run-length structure of a sorted
array.  Sizes straddle the tile sizes (8192 pairs per sort tile, 8192 positions per segmentation tile)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 63, 64, 65, 4095, 4096, 8191, 8192, 8193, 16385, 100_003, 1_510_720]


def _keys(rng, n, bits, kind):
    hi = (1 << bits) - 1 if bits < 32 else 0xFFFFFFFF
    if kind == "random":
        return rng.integers(0, hi + 1, n, dtype=np.uint64).astype(np.uint32)
    if kind == "equal":
        return np.full(n, hi // 3, np.uint32)
    if kind == "sorted":
        return np.sort(rng.integers(0, hi + 1, n, dtype=np.uint64).astype(np.uint32))
    if kind == "reverse":
        return np.sort(rng.integers(0, hi + 1, n, dtype=np.uint64).astype(np.uint32))[::-1].copy()
    if kind == "coherent":  # like leaf codes of a scan: long runs of equal upper digits, few distinct values
        base = rng.integers(0, max(1, hi >> 6) + 1, max(1, n // 500 + 1), dtype=np.uint64)
        k = (np.repeat(base, 500)[:n] << 6) | rng.integers(0, 64, n, dtype=np.uint64)
        return (k & hi).astype(np.uint32)
    raise ValueError(kind)


@pytest.mark.parametrize("n", SIZES)
def test_radix_sort_is_the_stable_sort(hip, n):
    rng = np.random.default_rng(n)
    opt = hip.DmsaOptimizer()
    cases = [(8, "random"), (17, "coherent"), (24, "random"), (32, "random")] if n > 100_000 else \
            [(b, k) for b in (1, 7, 8, 9, 16, 17, 24, 25, 32) for k in ("random", "equal", "sorted", "reverse", "coherent")]
    for bits, kind in cases:
        keys = _keys(rng, n, bits, kind)
        vals = rng.permutation(n).astype(np.uint32)
        ks, vs = opt.sortPairs(keys, vals, bits)
        order = np.argsort(keys, kind="stable")
        assert np.array_equal(ks, keys[order]), (n, bits, kind)
        assert np.array_equal(vs, vals[order]), (n, bits, kind)
    # bits above end_bit are ignored: sorting on the low byte only keeps the input order inside a bucket
    keys = _keys(rng, n, 32, "random")
    vals = np.arange(n, dtype=np.uint32)
    ks, vs = opt.sortPairs(keys, vals, 8)
    order = np.argsort(keys & 0xFF, kind="stable")
    assert np.array_equal(ks, keys[order]) and np.array_equal(vs, order.astype(np.uint32))
    # ... also when end_bit is not a multiple of 8: the last pass must mask its digit (random bits above end_bit)
    for bits in (1, 5, 11, 17, 21, 27, 31):
        keys = _keys(rng, n, 32, "random")
        ks, vs = opt.sortPairs(keys, vals, bits)
        order = np.argsort(keys & np.uint32((1 << bits) - 1), kind="stable")
        assert np.array_equal(ks, keys[order]), (n, bits)
        assert np.array_equal(vs, order.astype(np.uint32)), (n, bits)
    opt.close()


@pytest.mark.parametrize("n", SIZES)
def test_leaf_segments_match_run_lengths(hip, n):
    rng = np.random.default_rng(7 * n + 1)
    opt = hip.DmsaOptimizer()
    bits = 20
    invalid = np.uint32(1 << bits)
    for leaves_per_point, tail in ((0.1, 0), (1.0, 0), (0.001, 0), (0.1, min(n - 1, 37)), (0.1, n)):
        codes = np.sort(rng.integers(0, max(1, int(n * leaves_per_point)) + 1, n, dtype=np.uint64).astype(np.uint32))
        if tail:
            codes[n - tail:] = invalid  # non-finite points sort to the end and belong to no leaf
        of_pos, start = opt.leafSegments(codes, bits)
        valid = codes != invalid
        nv = int(valid.sum())
        if nv == 0:
            assert start.size == 1 or start.size == 0
            continue
        c = codes[:nv]
        head = np.ones(nv, bool)
        head[1:] = c[1:] != c[:-1]
        ref_of_pos = np.cumsum(head).astype(np.int32)
        ref_start = np.concatenate([np.flatnonzero(head), [nv]]).astype(np.int32)
        assert np.array_equal(of_pos[:nv], ref_of_pos)
        assert np.array_equal(start, ref_start)
    opt.close()


@pytest.mark.parametrize("n", [1, 65, 8193, 100_003, 1_510_720])
def test_radix_sort_of_64_bit_keys_is_the_stable_sort(hip, n):
    """Leaf codes of trees deeper than ten levels: sorted as two stable 32-bit sorts that carry positions (csrc/radix_sort.hip)."""
    rng = np.random.default_rng(3 * n + 5)
    opt = hip.DmsaOptimizer()
    for bits in (20, 32, 33, 40, 47, 63, 64):
        hi = (1 << bits) - 1
        keys = rng.integers(0, hi, n, dtype=np.uint64, endpoint=True)
        if n > 100:  # long runs of equal upper words, like leaf codes
            keys = (keys & np.uint64(0xFFFF_FFFF_FFFF_FFFF ^ 0x3FFF_FF00)) | (keys & np.uint64(0xFF))
        keys |= rng.integers(0, 1 << (64 - bits), n, dtype=np.uint64) << np.uint64(bits) if bits < 64 else np.uint64(0)  # bits above end_bit: ignored
        vals = rng.permutation(n).astype(np.uint32)
        ks, vs = opt.sortPairs64(keys, vals, bits)
        order = np.argsort(keys & np.uint64(hi), kind="stable")
        assert np.array_equal(ks, keys[order]), (n, bits)
        assert np.array_equal(vs, vals[order]), (n, bits)
    opt.close()


@pytest.mark.parametrize("n", [1, 2, 4095, 4096, 4097, 100_003, 3_021_440])
def test_prefix_scans(hip, n):
    rng = np.random.default_rng(n)
    opt = hip.DmsaOptimizer()
    for kind in ("flags", "counts"):
        x = (rng.random(n) < 0.3).astype(np.int32) if kind == "flags" else rng.integers(0, 700, n).astype(np.int32)
        incl = np.cumsum(x, dtype=np.int64).astype(np.int32)
        assert np.array_equal(opt.scan(x, True), incl), (n, kind)
        assert np.array_equal(opt.scan(x, False), incl - x), (n, kind)
    opt.close()
