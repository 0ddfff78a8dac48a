"""N > 1 path on CPU: two gloo ranks shard a keyframe map into neighbourhoods, optimise them (with the CPU oracle standing
in for the per-rank GPU optimiser) and all-gather the relative poses; the result must equal the single-process composition."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_optimize(sub, settings):
    from oracle import oracle_py as orc

    rep, _, _ = orc.optimize_keyframes(sub, settings)
    return rep


def _make_map():
    from dmsa_lidar_slam_amd import synth

    return synth.keyframe_problem(seed=9, frames=7, rings=12, az_steps=96, arc=0.45)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dmsa_lidar_slam_amd.problems import DmsaOptimSettings
    from dmsa_lidar_slam_amd.sharding import optimize_neighbourhoods

    m = _make_map()
    s = DmsaOptimSettings.keyframe_map(num_iter=2)
    optimize_neighbourhoods(m, s, _oracle_optimize, rank=rank, world=world, dist=dist)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ro=m.relOrientations, rt=m.relTranslations)
    dist.barrier()
    dist.destroy_process_group()


def _strong_map():
    from dmsa_lidar_slam_amd import synth

    return synth.keyframe_problem(seed=12, frames=25, rings=8, az_steps=64, arc=1.2)


def _strong_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dmsa_lidar_slam_amd.problems import DmsaOptimSettings
    from dmsa_lidar_slam_amd.sharding import gather_owned_neighbourhood_poses, neighbourhood_ranges, owned_neighbourhoods

    m = _strong_map()
    s = DmsaOptimSettings.keyframe_map(num_iter=1)
    s.min_num_gaussians = 1
    ranges = neighbourhood_ranges(m.numFrames, 8)
    subs = {}
    for nb in owned_neighbourhoods(8, rank, world):
        subs[nb] = m.getSubmap(*ranges[nb])
        _oracle_optimize(subs[nb], s)
    gather_owned_neighbourhood_poses(m, subs, ranges, rank, world, dist)
    np.savez(os.path.join(out_dir, f"w{world}_rank{rank}.npz"), ro=m.relOrientations, rt=m.relTranslations)
    dist.barrier()
    dist.destroy_process_group()


def test_fixed_cut_gives_the_same_map_at_every_world_size(tmp_path):
    """The strong-scaling layout of `bench.py --workload keyframes`: the map is cut into 8 neighbourhoods whatever the world size, rank r
    runs r, r + N, ... and ONE all-gather carries ceil(8 / N) slots per rank.  N = 1 (no collective), 2, 4 and 8 gloo ranks end with
    bit-identical relative poses on every rank -- the sequential composition of the eight updatePosesFromSubmap calls."""
    sys.path.insert(0, ROOT)
    from dmsa_lidar_slam_amd.problems import DmsaOptimSettings
    from dmsa_lidar_slam_amd.sharding import gather_owned_neighbourhood_poses, neighbourhood_ranges, owned_neighbourhoods

    assert owned_neighbourhoods(8, 1, 4) == [1, 5] and owned_neighbourhoods(8, 0, 1) == list(range(8)) and owned_neighbourhoods(8, 7, 8) == [7]
    m = _strong_map()
    s = DmsaOptimSettings.keyframe_map(num_iter=1)
    s.min_num_gaussians = 1
    ranges = neighbourhood_ranges(m.numFrames, 8)
    seq = m.copy()
    subs = {}
    for nb, (f, t) in enumerate(ranges):
        subs[nb] = m.getSubmap(f, t)
        _oracle_optimize(subs[nb], s)
        seq.updatePosesFromSubmap(f, t, subs[nb])
    assert np.abs(seq.relTranslations - m.relTranslations).max() > 1e-6  # the pass changed something
    one = m.copy()
    gather_owned_neighbourhood_poses(one, subs, ranges, 0, 1)  # N = 1: the same function, no collective
    assert np.array_equal(one.relOrientations, seq.relOrientations) and np.array_equal(one.relTranslations, seq.relTranslations)
    for world in (2, 4, 8):
        port = 29500 + ((os.getpid() + 11 * world) % 2000)
        mp.spawn(_strong_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
        for r in range(world):
            o = np.load(tmp_path / f"w{world}_rank{r}.npz")
            assert np.array_equal(o["ro"], seq.relOrientations) and np.array_equal(o["rt"], seq.relTranslations), (world, r)


def test_keyframe_map_dump_round_trip(tmp_path):
    """bench.py generates the 249-frame map once (rank 0) and hands it to the other ranks as a flat dump (dump.py: 'DMSAKF01')."""
    sys.path.insert(0, ROOT)
    from dmsa_lidar_slam_amd import dump

    m = _strong_map()
    path = str(tmp_path / "map.bin")
    dump.write_keyframe_map(path, m)
    r = dump.read_keyframe_map(path)
    for k in ("relOrientations", "relTranslations", "frameOffsets", "localPoints", "localNormals", "ringIds", "measuredGravity", "gravityPlausible", "gravity", "Cov_grav_inv"):
        assert np.array_equal(getattr(m, k), getattr(r, k)), k
    assert (m.minGridSize, m.useGravityErrorTerms, m.balancingFactorGrav) == (r.minGridSize, r.useGravityErrorTerms, r.balancingFactorGrav)


def test_neighbourhood_ranges():
    from dmsa_lidar_slam_amd.sharding import neighbourhood_ranges

    r = neighbourhood_ranges(256, 8)
    assert r[0][0] == 0 and r[-1][1] == 255
    assert all(r[i][1] == r[i + 1][0] for i in range(7))           # one shared boundary frame
    assert all(30 <= t - f <= 33 for f, t in r)
    assert neighbourhood_ranges(5, 1) == [(0, 4)]
    with pytest.raises(ValueError):
        neighbourhood_ranges(2, 2)


def test_two_rank_gloo_sharded_pass_matches_sequential(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(a["ro"], b["ro"]) and np.array_equal(a["rt"], b["rt"])  # every rank ends with the same poses
    # single-process composition of the same neighbourhoods
    sys.path.insert(0, ROOT)
    from dmsa_lidar_slam_amd.problems import DmsaOptimSettings
    from dmsa_lidar_slam_amd.sharding import neighbourhood_ranges

    m = _make_map()
    s = DmsaOptimSettings.keyframe_map(num_iter=2)
    ref = m.copy()
    for f, t in neighbourhood_ranges(m.numFrames, world):
        sub = m.getSubmap(f, t)
        _oracle_optimize(sub, s)
        ref.updatePosesFromSubmap(f, t, sub)
    assert np.array_equal(a["ro"], ref.relOrientations) and np.array_equal(a["rt"], ref.relTranslations)
    assert np.abs(a["rt"] - m.relTranslations).max() > 1e-5  # the pass changed something


def test_keyframe_map_seam_of_the_c_abi():
    """include/dmsa_keyframe_map.h (host-only entry points of libdmsa_hip.so, what a C++ host links): getSubmap's poses against an
    independent scipy statement, the odometry measurements it stores, and updatePosesFromSubmap as the inverse cut."""
    sys.path.insert(0, ROOT)
    from scipy.spatial.transform import Rotation as Rot

    from dmsa_lidar_slam_amd import posemath

    m = _make_map()
    m.useOdometryErrorTerms = True
    sub = m.getSubmap(2, 5)
    go, gt = posemath.relative2global(m.relOrientations, m.relTranslations)   # scipy rotations, independent of the library's math
    ro, rt = posemath.global2relative(go[2:6], gt[2:6])
    assert np.abs(sub.relOrientations - ro).max() < 1e-12 and np.abs(sub.relTranslations - rt).max() < 1e-12
    assert np.array_equal(sub.odomRelTransl, sub.relTranslations)
    assert np.abs(sub.odomRelOrientMat - Rot.from_rotvec(sub.relOrientations).as_matrix()).max() < 1e-12
    assert sub.localPoints.shape[0] == m.frameOffsets[6] - m.frameOffsets[2]
    # writing an unchanged submap back leaves the map where it was (to rounding), a changed one lands in columns 3..5 only
    back = m.copy()
    back.updatePosesFromSubmap(2, 5, sub)
    assert np.abs(back.relTranslations - m.relTranslations).max() < 1e-12 and np.abs(back.relOrientations - m.relOrientations).max() < 1e-12
    sub.relTranslations[1:] += 0.01
    back.updatePosesFromSubmap(2, 5, sub)
    changed = np.abs(back.relTranslations - m.relTranslations).max(axis=1) > 1e-6
    assert list(np.nonzero(changed)[0]) == [3, 4, 5]
    with pytest.raises(ValueError):
        m.getSubmap(3, 99)
