"""GPU parity of keyframe creation through include/dmsa_keyframe_cloud.h.  Neighbour lists, thinning picks and
ring ids are index work: bit-exact.  Normals: every operation is a correctly rounded float operation on both sides (the float
atan2 / cos / sin of eigen33 are evaluated in double and rounded once), so they are bit-exact too."""
import numpy as np
import pytest

from dmsa_lidar_slam_amd import synth
from dmsa_lidar_slam_amd.keyframe_cloud import KeyframeCloudBuilder

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(scope="module")
def gpu():
    b = KeyframeCloudBuilder(device=0)
    yield b
    b.close()


def _check_normals(got, ref):
    assert np.array_equal(got, ref, equal_nan=True), int(np.sum(np.any((got != ref) & ~(np.isnan(got) & np.isnan(ref)), axis=1)))
    return 1.0


@pytest.mark.parametrize("n,cell", [(20000, 0.3), (20000, 0.05), (20000, 3.0), (300, 0.3), (7, 0.3)])
def test_update_normals(gpu, orc, n, cell):
    """Scene surfaces at keyframe density; the search grid's cell size (far too small, right, far too large) must not matter."""
    rng = np.random.default_rng(n)
    scene = synth.Scene.room_with_stairs()
    pts = scene.sample_surfaces(0.15, rng, n, 0.01).astype(f32) - f32([4.0, 3.0, 1.5])
    cloud = np.concatenate([pts, np.ones((pts.shape[0], 1), f32)], axis=1)
    if n > 100:
        cloud[17, 1] = np.nan
        cloud[31] = cloud[30]
    nrm, nn = gpu.updateNormals(cloud, cell, neighbours=True)
    ref, ref_nn = orc.update_normals(cloud, neighbours=True)
    assert np.array_equal(nn, ref_nn)  # exact k-NN, ascending (float distance, index)
    _check_normals(nrm, ref)
    if n > 100:
        assert np.isnan(nrm[17]).all() and np.all(nn[17] == -1) and 17 not in nn


def test_update_normals_small_and_empty(gpu, orc):
    assert gpu.updateNormals(np.zeros((0, 4), f32), 0.3).shape == (0, 4)
    two = np.array([[0, 0, 0, 1], [1, 0, 0, 1]], f32)
    assert np.isnan(gpu.updateNormals(two, 0.3)).all()
    far = np.array([[0, 0, 0, 1], [50, 0, 0, 1], [0, 70, 0, 1], [0, 0, 90, 1], [-60, 5, 5, 1]], f32)  # neighbours dozens of rings away
    nrm, nn = gpu.updateNormals(far, 0.3, neighbours=True)
    ref, ref_nn = orc.update_normals(far, neighbours=True)
    assert np.array_equal(nn, ref_nn) and np.all(nn[:, 5] == -1)
    _check_normals(nrm, ref)


def test_add_new_keyframe_cloud(gpu, orc):
    """addNewKeyframeToMap (:497-531) on one window's global points (5 x 8192 points): thinning picks, ring ids and local coordinates
    bit-exact; normals as above."""
    p = synth.window_problem(seed=3, scans=5, rings=32, az_steps=256, num_static=0)
    from dmsa_lidar_slam_amd import posemath

    go, gt = posemath.relative2global(p.relOrientations, p.relTranslations)
    # global points of the window through the oracle's own transform (any consistent cloud would do)
    from scipy.spatial.transform import Rotation as Rot

    glob = p.localPoints.copy()
    glob[:, :3] = (Rot.from_rotvec(go[0]).apply(p.localPoints[:, :3].astype(np.float64)) + gt[0]).astype(f32)
    xyz, nrm, ring, src = gpu.addNewKeyframeCloud(glob, p.ringIds, p.minGridSize, 7, gt[0], go[0])
    rxyz, rnrm, rring, rsrc = orc.make_keyframe_cloud(glob, p.ringIds, p.minGridSize, 7, gt[0], go[0])
    assert np.array_equal(src, rsrc) and np.array_equal(ring, rring) and np.array_equal(xyz, rxyz)
    frac = _check_normals(nrm, rnrm)
    assert xyz.shape[0] > 5000 and frac > 0.97
    assert np.allclose(xyz[:, :3], p.localPoints[src, :3], atol=2e-5)  # back in the sensor frame of control pose 0
