"""GPU parity of SURVEY.md 8(f) rows f1/f2 through include/dmsa_static_points.h: DmsaSlam::addStaticPoints selection +
isVisible, getOverlap, randomGridDownsampling (seeded).  Everything here is index/flag work: the bar is bit-exact."""
import numpy as np
import pytest

from dmsa_lidar_slam_amd import synth
from dmsa_lidar_slam_amd.static_points import StaticPointSelector, StaticSelectProblem

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(scope="module")
def gpu():
    s = StaticPointSelector(device=0)
    yield s
    s.close()


def _pts(rng, n, lo, hi):
    return np.concatenate([rng.uniform(lo, hi, (n, 3)), np.ones((n, 1))], axis=1).astype(f32)


def _same_selection(a, b):
    assert np.array_equal(a.staticPoints, b.staticPoints) and np.array_equal(a.staticIds, b.staticIds)
    assert np.array_equal(a.overlapPerKeyframe, b.overlapPerKeyframe)
    assert (a.keyframeId, a.minRelatedKeyId, a.maxOverlap) == (b.keyframeId, b.minRelatedKeyId, b.maxOverlap)


def test_radius_exists_matches_oracle_and_brute_force(gpu, orc):
    rng = np.random.default_rng(3)
    cloud, query = _pts(rng, 20000, -4, 4), _pts(rng, 9000, -5, 5)
    cloud[7, 0], cloud[8, 1], query[3, 2] = np.nan, np.inf, np.nan
    query[4] = [1e30, 0, 0, 1]
    query[5] = [-1e30, 3e38, 0, 1]
    for r in (0.02, 0.3, 1.7):
        got = gpu.radiusExists(cloud, query, r)
        assert np.array_equal(got, orc.radius_exists(cloud, query, r))
    got = gpu.radiusExists(cloud[:3000], query[:2000], 0.3)
    assert np.array_equal(got, orc.radius_exists(cloud[:3000], query[:2000], 0.3, brute=True))
    assert not gpu.radiusExists(np.zeros((0, 4), f32), query, 0.3).any()
    assert gpu.radiusExists(cloud, np.zeros((0, 4), f32), 0.3).shape == (0,)
    # every cloud point is within any radius of itself; all-NaN cloud matches nothing
    assert gpu.radiusExists(cloud[100:200], cloud[100:200], 1e-3).all()
    assert not gpu.radiusExists(np.full((10, 4), np.nan, f32), query[:50], 0.3).any()


def test_radius_boundary_inclusive(gpu):
    r = f32(0.3)
    r2 = r * r
    x_in = f32(np.sqrt(np.float64(r2)))
    while f32(x_in * x_in) > r2:
        x_in = np.nextafter(x_in, f32(0))
    x_out = x_in
    while f32(x_out * x_out) <= r2:
        x_out = np.nextafter(x_out, f32(1))
    q = np.array([[x_in, 0, 0, 1], [x_out, 0, 0, 1], [0, -x_in, 0, 1], [0, 0, x_out, 1]], f32)
    assert gpu.radiusExists(np.array([[0, 0, 0, 1]], f32), q, r).tolist() == [True, False, True, False]


def test_select_static_points_small(gpu, orc):
    p = synth.static_select_problem(seed=2, scans=2, rings=32, az_steps=256, frames=3, key_rings=16, key_az=128)
    p.keyNormals[5:400] *= -1.0
    ref = orc.select_static_points(p)
    assert 0 < ref.staticPoints.shape[0] < p.keyPoints.shape[0]
    _same_selection(gpu.selectStaticPoints(p), ref)


def test_select_static_points_edge_cases(gpu, orc):
    p = synth.static_select_problem(seed=4, scans=1, rings=16, az_steps=128, frames=2, key_rings=8, key_az=64)
    n0 = p.frameOffsets[1]
    dup = StaticSelectProblem(windowPoints=p.windowPoints, keyframeIds=np.array([9, 4], np.int32), frameOffsets=np.array([0, n0, 2 * n0]),
                              keyPoints=np.concatenate([p.keyPoints[:n0]] * 2), keyNormals=np.concatenate([p.keyNormals[:n0]] * 2),
                              keyRingIds=np.concatenate([p.keyRingIds[:n0]] * 2), currPos=p.currPos, minGridSize=p.minGridSize)
    got = gpu.selectStaticPoints(dup)
    _same_selection(got, orc.select_static_points(dup))
    assert got.keyframeId == 9 and got.minRelatedKeyId == 4  # first of equal overlaps; smallest contributing id
    far = StaticSelectProblem(windowPoints=p.windowPoints + f32(1000.0), keyframeIds=p.keyframeIds, frameOffsets=p.frameOffsets, keyPoints=p.keyPoints,
                              keyNormals=p.keyNormals, keyRingIds=p.keyRingIds, currPos=p.currPos, minGridSize=p.minGridSize)
    got = gpu.selectStaticPoints(far)
    assert got.staticPoints.shape[0] == 0 and (got.keyframeId, got.minRelatedKeyId, got.maxOverlap) == (0, -1, 0)
    empty_frame = StaticSelectProblem(windowPoints=p.windowPoints, keyframeIds=np.array([3, 8, 1], np.int32), frameOffsets=np.array([0, n0, n0, p.keyPoints.shape[0]]),
                                      keyPoints=p.keyPoints, keyNormals=p.keyNormals, keyRingIds=p.keyRingIds, currPos=p.currPos, minGridSize=p.minGridSize)
    _same_selection(gpu.selectStaticPoints(empty_frame), orc.select_static_points(empty_frame))
    no_window = StaticSelectProblem(windowPoints=np.zeros((0, 4), f32), keyframeIds=p.keyframeIds, frameOffsets=p.frameOffsets, keyPoints=p.keyPoints,
                                    keyNormals=p.keyNormals, keyRingIds=p.keyRingIds, currPos=p.currPos, minGridSize=p.minGridSize)
    _same_selection(gpu.selectStaticPoints(no_window), orc.select_static_points(no_window))
    nan_prob = synth.static_select_problem(seed=4, scans=1, rings=16, az_steps=128, frames=2, key_rings=8, key_az=64)
    nan_prob.windowPoints[::7, 1] = np.nan
    nan_prob.keyPoints[::11, 0] = np.nan
    _same_selection(gpu.selectStaticPoints(nan_prob), orc.select_static_points(nan_prob))


def test_get_overlap(gpu, orc):
    rng = np.random.default_rng(5)
    a, b = _pts(rng, 15000, -3, 3), _pts(rng, 22000, -3.5, 3.5)
    assert gpu.getOverlap(a, b, 0.25) == orc.get_overlap(a, b, 0.25)
    assert gpu.getOverlap(np.zeros((0, 4), f32), b, 0.25) == (0.0, 0) and gpu.getOverlap(a, np.zeros((0, 4), f32), 0.25) == (0.0, 0)


@pytest.mark.parametrize("seed", [1, 77, 0])
def test_random_grid_downsampling(gpu, orc, seed):
    rng = np.random.default_rng(11)
    pts = _pts(rng, 30000, -6, 6)
    pts[17, 0] = np.nan
    pts[0, 2] = np.inf  # first point non-finite: the lattice anchors at the next one
    for grid in (f32(0.4), f32(0.15), f32(0.075)):
        got = gpu.randomGridDownsampling(pts, grid, seed)
        assert np.array_equal(got, orc.random_grid_downsampling(pts, grid, seed))
    assert gpu.randomGridDownsampling(np.zeros((0, 4), f32), 0.4, seed).shape == (0,)
    assert gpu.randomGridDownsampling(np.full((5, 4), np.nan, f32), 0.4, seed).shape == (0,)
    one = gpu.randomGridDownsampling(pts[5:6], 0.4, seed)
    assert one.tolist() == [0]


def test_add_static_points_full_size(gpu, orc):
    """The whole step at the bench size: 10 x 131 072-point window cloud, 3 keyframes (slam_settings.yaml), thinning at
    minGridSize / 2 with srand(7), overlap ratio of the 1.3 M window points against the thinned map."""
    p = synth.static_select_problem(seed=1)
    sel, active, active_ids, overlap = gpu.addStaticPoints(p, seed=7)
    ref = orc.select_static_points(p)
    _same_selection(sel, ref)
    pick = orc.random_grid_downsampling(ref.staticPoints, f32(p.minGridSize) / f32(2.0), 7)
    assert np.array_equal(active, ref.staticPoints[pick]) and np.array_equal(active_ids, ref.staticIds[pick])
    assert overlap == orc.get_overlap(ref.staticPoints[pick], p.windowPoints, p.minGridSize)[0]
    assert 0.5 < overlap < 1.0 and active.shape[0] > 10_000
    # size-independent properties: idempotent; one DISTINCT point per occupied leaf, in leaf order whatever the seed (thinning again
    # can only merge leaves: the lattice re-anchors at the new first point); every selected point has a window point within minGridSize
    again, *_ = gpu.addStaticPoints(p, seed=7)
    _same_selection(again, sel)
    half = f32(p.minGridSize) / f32(2.0)
    pick_a, pick_b = gpu.randomGridDownsampling(sel.staticPoints, half, 7), gpu.randomGridDownsampling(sel.staticPoints, half, 8)
    assert pick_a.shape == pick_b.shape and np.unique(pick_a).shape == pick_a.shape and not np.array_equal(pick_a, pick_b)
    assert gpu.randomGridDownsampling(active, half, 3).shape[0] <= active.shape[0]
    assert gpu.radiusExists(p.windowPoints, sel.staticPoints, p.minGridSize).all()


def _room_scan(rng, n, half=(25.0, 18.0, 3.0)):
    d = rng.normal(size=(n, 3))
    d[:, 2] *= 0.3
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t = np.min(np.asarray(half)[None, :] / np.maximum(np.abs(d), 1e-9), axis=1)
    p = d * t[:, None] + rng.normal(scale=0.01, size=(n, 3))
    return np.concatenate([p, rng.uniform(0, 1, (n, 1))], axis=1).astype(f32)


def _same_scan(a, b):
    assert a[2] == b[2], "gridSize of the kept filter pass"
    assert np.array_equal(a[1], b[1]), "indices into the raw scan"
    assert np.array_equal(a[0], b[0]), "filtered points, bit-exact"


@pytest.mark.parametrize("n,max_pts,min_dist_ds,min_dist", [(131072, 3000, 30.0, 0.0), (60000, 1000, 10.0, 4.0), (1500, 3000, 30.0, 0.0), (20000, 100000, 5.0, 1.0)])
def test_preprocess_scan(gpu, orc, n, max_pts, min_dist_ds, min_dist):
    """DmsaSlam::preProcess (DmsaSlam.h:569-634): adaptive grid filter, range threshold, gates, lidar->IMU transform, w = 1."""
    from scipy.spatial.transform import Rotation

    rng = np.random.default_rng(n + max_pts)
    raw = _room_scan(rng, n)
    raw[5, 1] = np.nan
    T = np.eye(4, dtype=f32)
    T[:3, :3] = Rotation.from_euler("xyz", [0.02, -0.01, 1.3]).as_matrix().astype(f32)
    T[:3, 3] = [0.05, -0.12, 0.3]
    for seed in (9, 10):
        got = gpu.preProcess(raw, seed, max_pts, min_dist_ds, min_dist, T)
        _same_scan(got, orc.preprocess_scan(raw, seed, max_pts, min_dist_ds, min_dist, T))
        assert np.all(got[0][:, 3] == 1.0)
    _same_scan(gpu.preProcess(raw, 9), orc.preprocess_scan(raw, 9))  # Config.h defaults, identity transform


def test_preprocess_scan_edge_cases(gpu, orc):
    rng = np.random.default_rng(2)
    assert gpu.preProcess(np.zeros((0, 4), f32), 1)[0].shape == (0, 4)
    assert gpu.preProcess(np.full((7, 4), np.nan, f32), 1)[0].shape == (0, 4)
    one = np.array([[3.0, 4.0, 0.0, 0.5]], f32)
    _same_scan(gpu.preProcess(one, 1), orc.preprocess_scan(one, 1))  # thresRange = max(5, 30): kept
    _same_scan(gpu.preProcess(one, 1, 3000, 4.0, 0.0), orc.preprocess_scan(one, 1, 3000, 4.0, 0.0))  # range == thresRange: dropped (strict <)
    assert gpu.preProcess(one, 1, 3000, 4.0, 0.0)[0].shape[0] == 0
    _same_scan(gpu.preProcess(one, 1, 3000, 30.0, 5.0), orc.preprocess_scan(one, 1, 3000, 30.0, 5.0))  # range == min_dist: dropped (strict >)
    # the scratch is shared with the other static-point functions: interleave them
    raw = _room_scan(rng, 40000)
    a = gpu.preProcess(raw, 3)
    gpu.radiusExists(raw[:5000], raw[5000:9000], 0.3)
    gpu.randomGridDownsampling(raw, 0.2, 3)
    _same_scan(gpu.preProcess(raw, 3), a)
    _same_scan(a, orc.preprocess_scan(raw, 3))


def test_resident_window_cloud(orc):
    """NULL window pointers = the global points of the problem uploaded to the shared context: same results as with host arrays, and
    the keyframe cloud is built from the resident points and ring ids."""
    from dmsa_lidar_slam_amd import posemath
    from dmsa_lidar_slam_amd.api import DmsaOptimizer
    from dmsa_lidar_slam_amd.keyframe_cloud import KeyframeCloudBuilder

    w = synth.window_problem(seed=3, scans=4, rings=32, az_steps=256, num_static=0)
    opt = DmsaOptimizer(device=0)
    opt.upload(w)
    opt.poseTables(opt.getPoseParameters(), download=False)
    window_global = opt.updateGlobalPoints(0)
    n = window_global.shape[0]
    p = synth.static_select_problem(seed=2, scans=1, rings=16, az_steps=64, frames=3, key_rings=32, key_az=160)
    host = StaticSelectProblem(windowPoints=window_global, keyframeIds=p.keyframeIds, frameOffsets=p.frameOffsets, keyPoints=p.keyPoints, keyNormals=p.keyNormals,
                               keyRingIds=p.keyRingIds, currPos=p.currPos, minGridSize=p.minGridSize)
    res = StaticSelectProblem(windowPoints=None, numWindowResident=n, keyframeIds=p.keyframeIds, frameOffsets=p.frameOffsets, keyPoints=p.keyPoints,
                              keyNormals=p.keyNormals, keyRingIds=p.keyRingIds, currPos=p.currPos, minGridSize=p.minGridSize)
    shared = StaticPointSelector(optimizer=opt)
    a = shared.addStaticPoints(res, seed=7)
    own = StaticPointSelector(device=0)
    b = own.addStaticPoints(host, seed=7)
    _same_selection(a[0], b[0])
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3] and a[0].staticPoints.shape[0] > 100
    _same_selection(a[0], orc.select_static_points(host))
    go, gt = posemath.relative2global(w.relOrientations, w.relTranslations)
    kb = KeyframeCloudBuilder(optimizer=opt)
    r = kb.addNewKeyframeCloud(None, None, w.minGridSize, 5, gt[0], go[0], numResident=n)
    h = orc.make_keyframe_cloud(window_global, w.ringIds, w.minGridSize, 5, gt[0], go[0])
    assert np.array_equal(r[0], h[0]) and np.array_equal(r[2], h[2]) and np.array_equal(r[3], h[3]) and r[0].shape[0] > 1000
    # the resident problem is untouched: optimizeSet still runs on it
    from dmsa_lidar_slam_amd.problems import DmsaOptimSettings

    rep = opt.optimizeResident(DmsaOptimSettings.sliding_window(num_iter=1))
    assert rep.iterations == 1
    with pytest.raises(Exception):
        own.addStaticPoints(res, seed=7)  # nothing uploaded to that context
    kb.close(), shared.close(), own.close(), opt.close()
