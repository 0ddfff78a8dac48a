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


# ---- centered^T * centered (Gaussians.h:147): Eigen 3.4's product kernels for a (3 x n) * (n x 3) float product -----------------------
def gebp_model_dot(a, b, l1_bytes=32 * 1024):
    """Independent transcription of what Eigen 3.4.0 does for ONE coefficient of that product (see oracle/dmsa_oracle.cpp,
    Gaussians::gemm_dot_f32): evaluateProductBlockingSizesHeuristic's kc written with its own variable names, then gebp's scalar tail."""
    n = len(a)
    prod = (a * b).astype(f32)  # every product rounded to float on its own (no FMA: the reference is built without -march)
    if n + 3 + 3 < 20:  # EIGEN_GEMM_TO_COEFFBASED_THRESHOLD: lazy product, linear redux of the products from an aligned start
        return _redux_sum(prod)
    k = n
    if max(k, 3, 3) >= 48:
        mr, nr, k_peeling = 8, 4, 8
        k_div, k_sub = 1 * (mr * 4 + nr * 4), mr * nr * 4
        max_kc = max(((l1_bytes - k_sub) // k_div) & ~(k_peeling - 1), 1)
        if k > max_kc:
            k = max_kc if k % max_kc == 0 else max_kc - k_peeling * ((max_kc - 1 - (k % max_kc)) // (k_peeling * (k // max_kc + 1)))
    res = f32(0)
    for k2 in range(0, n, k):
        c0 = f32(0)
        for t in prod[k2:k2 + k]:
            c0 = f32(t + c0)
        res = f32(res + f32(f32(1) * c0))
    return res


def _redux_sum(x):
    """redux_impl<sum, LinearVectorizedTraversal, NoUnrolling> with Packet4f from element 0 (the sum, not the mean)"""
    n = len(x)
    packets = n // 4
    if packets == 0:
        r = x[0]
        for v in x[1:]:
            r = f32(r + v)
        return r
    acc0 = x[0:4].copy()
    if packets > 1:
        acc1 = x[4:8].copy()
        pair_end = 8 * (packets // 2)
        for a in range(8, pair_end, 8):
            acc0 = (acc0 + x[a:a + 4]).astype(f32)
            acc1 = (acc1 + x[a + 4:a + 8]).astype(f32)
        acc0 = (acc0 + acc1).astype(f32)
        if 4 * packets > pair_end:
            acc0 = (acc0 + x[pair_end:pair_end + 4]).astype(f32)
    r = f32(f32(acc0[0] + acc0[2]) + f32(acc0[1] + acc0[3]))
    for v in x[4 * packets:]:
        r = f32(r + v)
    return r


@pytest.mark.parametrize("n", list(range(2, 24)) + [47, 48, 63, 257, 679, 680, 681, 1000, 1361, 2049, 14200])
def test_oracle_covariance_product_follows_eigens_product_kernels(orc, n):
    rng = np.random.default_rng(n)
    a = rng.normal(0, 0.3, n).astype(f32)
    b = rng.normal(0, 0.3, n).astype(f32)
    assert orc.eigen_gemm_dot_f32(a, b).tobytes() == gebp_model_dot(a, b).tobytes()
    assert orc.eigen_gemm_dot_f32(a, a).tobytes() == gebp_model_dot(a, a).tobytes()


def test_depth_blocks_of_the_product_follow_the_l1_size(orc):
    """kc for a 32 KB L1d is 680 (every Gaussian up to 680 members is ONE chain: machine-independent); 48 KB gives 1016.  Above max_kc the
    depth is cut into nearly equal blocks that are multiples of 8."""
    try:
        assert [orc.eigen_gemm_kc(k) for k in (14, 47, 48, 680, 681, 1360, 1361, 14200)] == [14, 47, 48, 680, 344, 680, 456, 680]
        orc.set_eigen_l1_bytes(48 * 1024)
        assert [orc.eigen_gemm_kc(k) for k in (680, 1016, 1017, 14200)] == [680, 1016, 512, 1016]
        rng = np.random.default_rng(3)
        a = rng.normal(0, 0.3, 900).astype(f32)
        assert orc.eigen_gemm_dot_f32(a, a).tobytes() == gebp_model_dot(a, a, 48 * 1024).tobytes()
    finally:
        orc.set_eigen_l1_bytes(32 * 1024)
