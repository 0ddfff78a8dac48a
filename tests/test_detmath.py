"""include/dmsa_detmath.h: the sin / cos / acos / atan2 of the pose-table path as fixed IEEE operation sequences.

CPU: pinned against this machine's libm (numpy) to <= 1 ulp (atan2 outside the first quadrant: <= 2 ulp) on dense samples of the
ranges the pose tables use, plus exact special values.  GPU: the same header evaluated on the device returns the same BITS.
"""
import numpy as np
import pytest

from oracle import oracle_py as orc


def _ulps(a, b):
    ia = np.ascontiguousarray(a, np.float64).view(np.int64).copy()
    ib = np.ascontiguousarray(b, np.float64).view(np.int64).copy()
    ia[ia < 0] = np.iinfo(np.int64).min - ia[ia < 0]
    ib[ib < 0] = np.iinfo(np.int64).min - ib[ib < 0]
    return np.abs(ia - ib)


def _samples(seed=3, n=400_000):
    rng = np.random.default_rng(seed)
    x_trig = np.concatenate([rng.uniform(-10, 10, n), rng.uniform(-1, 1, n), rng.uniform(0, 1.6, n), rng.uniform(-1e-3, 1e-3, n),
                             rng.uniform(-1, 1, n // 4) * 2.0 ** -rng.integers(0, 60, n // 4), np.array([0.0, -0.0, np.pi / 4, np.pi / 2, np.pi, 2 * np.pi])])
    x_acos = np.concatenate([rng.uniform(-1, 1, n), rng.uniform(0.99, 1.0, n), 1.0 - rng.uniform(0, 1, n) * 2.0 ** -rng.integers(0, 52, n),
                             np.array([1.0, -1.0, 0.0, 0.5, -0.5])])
    y = rng.uniform(0, 1, n) * 2.0 ** -rng.integers(0, 40, n)
    x = rng.uniform(0, 1, n) * 2.0 ** -rng.integers(0, 40, n)
    x[::5] = 0.0
    return x_trig, x_acos, y, x


def test_detmath_is_within_one_ulp_of_libm():
    x_trig, x_acos, y, x = _samples()
    assert _ulps(orc.detmath_eval(0, x_trig), np.sin(x_trig)).max() <= 1
    assert _ulps(orc.detmath_eval(1, x_trig), np.cos(x_trig)).max() <= 1
    assert _ulps(orc.detmath_eval(2, x_acos), np.arccos(x_acos)).max() <= 1
    # the pose tables call atan2(|vec|, |w|): first quadrant
    assert _ulps(orc.detmath_eval(3, x, y), np.arctan2(y, x)).max() <= 1
    ys, xs = np.concatenate([y, -y]), np.concatenate([-x, x])
    assert _ulps(orc.detmath_eval(3, xs, ys), np.arctan2(ys, xs)).max() <= 2


def test_detmath_special_values():
    assert orc.detmath_eval(0, [0.0])[0] == 0.0 and orc.detmath_eval(1, [0.0])[0] == 1.0
    assert orc.detmath_eval(2, [1.0])[0] == 0.0 and orc.detmath_eval(2, [-1.0])[0] == np.pi
    assert np.isnan(orc.detmath_eval(2, [1.5])[0])
    assert orc.detmath_eval(3, [0.0], [1.0])[0] == np.pi / 2 and orc.detmath_eval(3, [1.0], [0.0])[0] == 0.0
    assert orc.detmath_eval(3, [-1.0], [0.0])[0] == np.pi and orc.detmath_eval(3, [1e-300], [1.0])[0] == np.pi / 2


@pytest.mark.gpu
def test_detmath_device_bits_equal_host_bits():
    from dmsa_lidar_slam_amd.api import DmsaOptimizer

    opt = DmsaOptimizer(device=0)
    x_trig, x_acos, y, x = _samples(seed=5)
    for fn, a, b in ((0, x_trig, None), (1, x_trig, None), (2, x_acos, None), (3, np.concatenate([x, -x]), np.concatenate([y, -y]))):
        dev = opt.detmathEval(fn, a, b)
        host = orc.detmath_eval(fn, a, b)
        assert np.array_equal(dev.view(np.int64), host.view(np.int64)), f"fn {fn}: {np.count_nonzero(dev.view(np.int64) != host.view(np.int64))} differ"
