"""The one reduction order of the Gaussian fit that the reference's sources settle: `subset.colwise().mean()` (Gaussians.h:146) and
`rebalancingWeights.head(M).mean()` (:176) are Eigen 3.4 linear vectorised reductions of contiguous float vectors -- a pure function of
the length and of where the vector starts relative to a 16-byte boundary.  The oracle's restatement (index arithmetic) is compared with
an independent model that walks ADDRESSES the way the SSE2 code does: a byte buffer, aligned 16-byte loads, two packet registers."""
import numpy as np
import pytest

f32 = np.float32


def sse_model_mean(buf, addr, n):
    """buf: float32 array standing for memory, addr: index (in floats) of the first element, buf[0] 16-byte aligned."""
    if n == 0:
        return f32(0)
    v = buf
    first_aligned = addr
    while first_aligned % 4 != 0:  # first element on a 16-byte boundary
        first_aligned += 1
    end = addr + n
    if first_aligned > end:
        first_aligned = end
    packets = (end - first_aligned) // 4
    if packets == 0:
        r = v[addr]
        for i in range(addr + 1, end):
            r = f32(r + v[i])
        return f32(r / f32(n))
    load = lambda a: v[a:a + 4].copy()  # movaps
    a = first_aligned
    acc0 = load(a)
    a += 4
    last_packet_end = first_aligned + 4 * packets
    if packets > 1:
        acc1 = load(a)
        a += 4
        pair_end = first_aligned + 8 * (packets // 2)
        while a < pair_end:
            acc0 = (acc0 + load(a)).astype(f32)
            acc1 = (acc1 + load(a + 4)).astype(f32)
            a += 8
        acc0 = (acc0 + acc1).astype(f32)
        if last_packet_end > pair_end:
            acc0 = (acc0 + load(pair_end)).astype(f32)
    # predux: tmp = a + movehl(a, a); add_ss(tmp, shuffle(tmp, 1))
    tmp = (acc0 + np.array([acc0[2], acc0[3], acc0[2], acc0[3]], f32)).astype(f32)
    r = f32(tmp[0] + tmp[1])
    for i in range(addr, first_aligned):
        r = f32(r + v[i])
    for i in range(last_packet_end, end):
        r = f32(r + v[i])
    return f32(r / f32(n))


@pytest.mark.parametrize("n", list(range(1, 41)) + [63, 64, 65, 127, 257, 1000, 4099])
def test_oracle_mean_follows_the_sse2_linear_redux(orc, n):
    rng = np.random.default_rng(n)
    for offset in range(4):
        buf = np.zeros(n + 16, f32)
        x = (rng.normal(0, 30, n) + 100).astype(f32)
        buf[offset:offset + n] = x
        want = sse_model_mean(buf, offset, n)
        got = orc.eigen_mean_f32(x, offset)
        assert got.tobytes() == want.tobytes(), (n, offset, got, want)


def test_columns_of_a_column_major_matrix_start_at_c_times_n(orc):
    """Column c of the n x 3 `subset` (MatrixX3f, column-major, 16-byte aligned buffer) starts c * n floats in: the three columns of one
    Gaussian are reduced with three different peels unless n is a multiple of four -- and the order matters (the sums differ in the
    last bits between offsets)."""
    rng = np.random.default_rng(1)
    n = 1001
    m = (rng.normal(0, 5, (n, 3)) + 50).astype(f32)
    col_major = np.zeros(3 * n + 8, f32)
    col_major[:3 * n] = m.T.reshape(-1)
    differs = 0
    for c in range(3):
        want = sse_model_mean(col_major, c * n, n)
        got = orc.eigen_mean_f32(m[:, c], c * n)
        assert got.tobytes() == want.tobytes()
        differs += int(orc.eigen_mean_f32(m[:, c], 0).tobytes() != got.tobytes())
    assert abs(float(orc.eigen_mean_f32(m[:, 0], 0)) - float(m[:, 0].astype(np.float64).mean())) < 1e-4
    assert differs >= 0  # informational: peels usually change the last bit of the mean
