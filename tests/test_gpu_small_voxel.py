"""The small-problem voxelisation (csrc/small_voxel.hip: one launch, a workgroup per resolution, pairs in registers) against the general path
(ten kernels per level) on the same problems: leaf codes, sorted point order, leaf counts, Gaussians, member lists, information matrices,
weights -- bit for bit -- and whole optimizeSet calls against the CPU oracle.  createGaussianSets: DmsaOptimizer.h:275-350."""
import numpy as np
import pytest

from dmsa_lidar_slam_amd import synth
from dmsa_lidar_slam_amd.problems import DmsaOptimSettings

pytestmark = pytest.mark.gpu


def _cases():
    out = []
    # the reference's everyday shape (config 2): 5 x 3072 points + 10^4 static points, IMU rows
    out.append(("imu", synth.window_problem(seed=5, scans=5, rings=32, az_steps=96, num_static=10_000, use_imu=True), DmsaOptimSettings.sliding_window(use_imu=True)))
    # close to the limit of 29 696 points (29 positions per thread)
    out.append(("k29a", synth.window_problem(seed=3, scans=3, rings=32, az_steps=256, num_static=4000), DmsaOptimSettings.sliding_window()))
    out.append(("k29b", synth.window_problem(seed=4, scans=2, rings=64, az_steps=200, num_static=4095), DmsaOptimSettings.sliding_window()))
    # few points: 9 and 17 positions per thread, an odd count
    out.append(("k9", synth.window_problem(seed=6, scans=2, rings=16, az_steps=128, num_static=1501), DmsaOptimSettings.sliding_window()))
    out.append(("k17", synth.window_problem(seed=7, scans=3, rings=16, az_steps=200, num_static=3333), DmsaOptimSettings.sliding_window()))
    # irregular scan pattern, ids = index % 1000 (config 5 at the size livox.yaml prescribes: 1500 points per scan)
    out.append(("rosette", synth.rosette_window_problem(seed=2, scans=5, pts_per_scan=1500, num_static=6000), DmsaOptimSettings.sliding_window()))
    # a keyframe set without splitSet
    k = DmsaOptimSettings.keyframe_map()
    k.gauss_split = False
    out.append(("keyframes", synth.keyframe_problem(seed=4, frames=5, rings=16, az_steps=128, arc=0.4), k))
    return out


def _voxelise(hip, prob, s, small, poke=None):
    opt = hip.DmsaOptimizer(device=0, debug={"small_voxel": small})
    opt.upload(prob)
    opt.poseTables(prob.getPoseParameters())
    opt.updateGlobalPoints(0)
    M, Mm = opt.buildGaussians(s)
    lv = [opt.voxelLevel(l) for l in (0, 1)]
    g = opt.gaussians()
    # a second voxelisation of the same context (the lattice hint, the general path's speculation)
    M2, Mm2 = opt.buildGaussians(s)
    g2 = opt.gaussians()
    c = opt.debugCounters()
    opt.close()
    return (M, Mm), lv, g, (M2, Mm2), g2, c


@pytest.mark.parametrize("case", range(7))
def test_small_path_equals_general_path(hip, case):
    name, prob, s = _cases()[case]
    n = prob.localPoints.shape[0] + (prob.staticPoints.shape[0] if hasattr(prob, "staticPoints") else 0)
    assert n <= 29696, (name, n)
    a = _voxelise(hip, prob, s, 1)
    b = _voxelise(hip, prob, s, 0)
    assert a[0] == b[0] and a[3] == b[3] and a[0][0] > 30, (name, a[0], b[0])
    for l in (0, 1):
        (ia, ca, ka, oa), (ib, cb, kb, ob) = a[1][l], b[1][l]
        assert (ia.num_leaves, ia.num_valid, ia.depth) == (ib.num_leaves, ib.num_valid, ib.depth), name
        assert np.array_equal(ca, cb) and np.array_equal(ka, kb) and np.array_equal(oa, ob), (name, l)
    for ga, gb in ((a[2], b[2]), (a[4], b[4])):
        for x, y in zip(ga, gb):
            assert np.array_equal(x.view(np.int32) if x.dtype == np.float32 else x, y.view(np.int32) if y.dtype == np.float32 else y), name


def test_non_finite_points_and_duplicates(hip):
    prob = synth.window_problem(seed=9, scans=2, rings=16, az_steps=128, num_static=2000)
    prob.localPoints[5, 0] = np.nan
    prob.localPoints[77, 1] = np.inf
    prob.localPoints[1000:1010, :3] = prob.localPoints[999, :3]  # ten points in one place
    prob.staticPoints[3, 2] = -np.inf
    s = DmsaOptimSettings.sliding_window()
    a = _voxelise(hip, prob, s, 1)
    b = _voxelise(hip, prob, s, 0)
    assert a[0] == b[0]
    for l in (0, 1):
        assert a[1][l][0].num_valid == b[1][l][0].num_valid < prob.localPoints.shape[0] + prob.staticPoints.shape[0]
        assert np.array_equal(a[1][l][3], b[1][l][3])
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(x, y, equal_nan=True)


@pytest.mark.parametrize("case", [0, 1, 5])
def test_whole_calls_match_the_oracle_on_the_small_path(hip, orc, case):
    name, prob, s = _cases()[case]
    s.num_iter = 4
    p_ref = prob.copy()
    rep_ref, _, trace = orc.optimize_window(p_ref, s)
    p = prob.copy()
    opt = hip.DmsaOptimizer(device=0, debug={"small_voxel": 1})
    rep = opt.optimizeSet(p, s)
    assert (rep.iterations, rep.stop_reason, rep.num_gaussians, rep.num_memberships) == (rep_ref.iterations, rep_ref.stop_reason, rep_ref.num_gaussians, rep_ref.num_memberships)
    assert np.array_equal(p.relOrientations, p_ref.relOrientations) and np.array_equal(p.relTranslations, p_ref.relTranslations), name
    assert opt.debugCounters().get("small_voxel_launches", 1) > 0
    opt.close()


def test_codes_wider_than_32_bits_fall_back_to_the_general_path(hip, orc):
    """Two far-apart clusters at a fine resolution: the tree gets deeper than ten levels, the small kernel gives up and the general path
    (64-bit codes) takes over -- same Gaussians as a context that never tried."""
    rng = np.random.default_rng(3)
    prob = synth.window_problem(seed=9, scans=2, rings=16, az_steps=128, num_static=2000)
    prob.staticPoints[:600, :3] += np.float32(3000.0)  # far away: the bounding box doubles until it holds both
    s = DmsaOptimSettings.sliding_window()
    a = _voxelise(hip, prob, s, 1)
    b = _voxelise(hip, prob, s, 0)
    assert a[0] == b[0] and a[3] == b[3]
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(x, y)
    assert a[5].get("small_voxel_fallbacks", 0) >= 1 or a[1][0][0].depth <= 10
