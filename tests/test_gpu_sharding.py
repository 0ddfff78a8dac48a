"""The sharded keyframe pass with the HIP library under REAL multi-process execution: two ranks (gloo collectives, both on GPU 0)
each optimise their neighbourhood with DmsaOptimizer (default path) and exchange relative poses with one all-gather.  The result
must equal the sequential composition of the same neighbourhoods on one process, and the oracle's run of that composition."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _make_map():
    from dmsa_lidar_slam_amd import synth

    return synth.keyframe_problem(seed=17, frames=9, rings=24, az_steps=160, arc=0.5)


def _hip_optimize(sub, settings):
    from dmsa_lidar_slam_amd.api import DmsaOptimizer

    return DmsaOptimizer(device=0).optimizeSet(sub, settings)  # raises without the HIP library / a device: no CPU fallback


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dmsa_lidar_slam_amd.problems import DmsaOptimSettings
    from dmsa_lidar_slam_amd.sharding import optimize_neighbourhoods

    m = _make_map()
    rep = optimize_neighbourhoods(m, DmsaOptimSettings.keyframe_map(num_iter=2), _hip_optimize, rank=rank, world=world, dist=dist)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ro=m.relOrientations, rt=m.relTranslations, iters=rep.iterations)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_sequential_composition_and_oracle(tmp_path, orc):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(a["ro"], b["ro"]) and np.array_equal(a["rt"], b["rt"])  # every rank ends with the same map
    sys.path.insert(0, ROOT)
    from dmsa_lidar_slam_amd.problems import DmsaOptimSettings
    from dmsa_lidar_slam_amd.sharding import neighbourhood_ranges

    m = _make_map()
    s = DmsaOptimSettings.keyframe_map(num_iter=2)
    seq, ora = m.copy(), m.copy()
    for f, t in neighbourhood_ranges(m.numFrames, world):
        sub = m.getSubmap(f, t)
        _hip_optimize(sub, s)
        seq.updatePosesFromSubmap(f, t, sub)
        sub_o = m.getSubmap(f, t)
        orc.optimize_keyframes(sub_o, s)
        ora.updatePosesFromSubmap(f, t, sub_o)
    assert np.array_equal(a["ro"], seq.relOrientations) and np.array_equal(a["rt"], seq.relTranslations)
    go_a, gt_a = orc.relative2global(a["ro"], a["rt"])
    go_o, gt_o = orc.relative2global(ora.relOrientations, ora.relTranslations)
    assert np.abs(gt_a - gt_o).max() < 1e-4 and np.abs(go_a - go_o).max() < 1e-4
    assert np.abs(a["rt"] - m.relTranslations).max() > 1e-5  # the pass changed something


def test_python_free_rccl_driver_matches_the_python_pass(tmp_path, hip):
    """examples/keyframe_shard_rccl (C++ host, C ABI, ncclAllGather from librccl) on a dumped map: one process per GPU for
    min(2, visible GPUs) ranks -- on a multi-GPU box this is RCCL with N > 1 over xGMI -- must write the poses the Python glue
    produces for the same neighbourhoods (MapManagement.h:254-288)."""
    import subprocess

    import torch

    from dmsa_lidar_slam_amd import dump
    from dmsa_lidar_slam_amd.problems import DmsaOptimSettings
    from dmsa_lidar_slam_amd.sharding import neighbourhood_ranges

    exe = os.path.join(ROOT, "examples", "keyframe_shard_rccl")
    if not os.path.exists(exe):
        pytest.fail(f"{exe} missing: run __graft_entry__.build() (make -C dmsa_lidar_slam_amd/csrc driver)")
    ranks = min(2, torch.cuda.device_count())
    m = _make_map()
    src, out = str(tmp_path / "map.bin"), str(tmp_path / "poses.bin")
    dump.write_keyframe_map(src, m)
    r = subprocess.run([exe, src, out, str(ranks), "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"ranks={ranks}" in r.stdout + r.stderr, r.stdout + r.stderr  # the driver reports the RCCL world size it ran with
    ro, rt = dump.read_poses(out)
    ref = m.copy()
    s = DmsaOptimSettings.keyframe_map(num_iter=2)
    for f, t in neighbourhood_ranges(m.numFrames, ranks):
        sub = m.getSubmap(f, t)
        opt = hip.DmsaOptimizer(device=0, fixed_iters=True)
        opt.optimizeSet(sub, s)
        opt.close()
        ref.updatePosesFromSubmap(f, t, sub)
    assert np.array_equal(ro, ref.relOrientations) and np.array_equal(rt, ref.relTranslations)


# ---- BASELINE.json config 4 at its full shape: the 249-frame map of `bench.py --workload keyframes --gpus 8` ---------------------
def _full_shape_worker(rank, world, port, map_path, out_dir):
    import pickle

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dmsa_lidar_slam_amd.problems import DmsaOptimSettings
    from dmsa_lidar_slam_amd.sharding import optimize_neighbourhoods

    with open(map_path, "rb") as f:
        m = pickle.load(f)
    rep = optimize_neighbourhoods(m, DmsaOptimSettings.keyframe_map(num_iter=2), _hip_optimize, rank=rank, world=world, dist=dist)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ro=m.relOrientations, rt=m.relTranslations, iters=rep.iterations)
    dist.barrier()
    dist.destroy_process_group()


def test_config4_full_shape_eight_neighbourhoods(tmp_path, orc):
    """The map `bench.py --workload keyframes --gpus 8` shards -- (32 - 1) * 8 + 1 = 249 keyframes, 2.5 M points -- cut into its eight
    neighbourhoods (dmsa_neighbourhood_ranges), each run through getSubmap -> optimizeSet -> updatePosesFromSubmap (the C seam of
    include/dmsa_keyframe_map.h) sequentially on GPU 0; two of the neighbourhoods against the oracle (1e-4 m / 1e-4 rad), and the
    composition bit for bit against an 8-rank run (gloo collectives, the ranks share GPU 0) of the same pass."""
    import pickle

    sys.path.insert(0, ROOT)
    from dmsa_lidar_slam_amd import synth
    from dmsa_lidar_slam_amd.problems import DmsaOptimSettings
    from dmsa_lidar_slam_amd.sharding import neighbourhood_ranges

    world, frames = 8, 32
    total = (frames - 1) * world + 1
    m = synth.keyframe_problem(seed=1, frames=total, arc=2 * np.pi * total / 256.0)
    assert m.numFrames == 249 and m.localPoints.shape[0] > 2_400_000
    ranges = neighbourhood_ranges(total, world)
    assert ranges[0] == (0, 31) and ranges[-1] == (217, 248) and all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
    s = DmsaOptimSettings.keyframe_map(num_iter=2)
    seq = m.copy()
    for i, (f, t) in enumerate(ranges):
        sub = m.getSubmap(f, t)
        assert sub.numParams == 186
        rep = _hip_optimize(sub, s)
        assert rep.iterations == 2
        if i in (0, 5):
            sub_o = m.getSubmap(f, t)
            rep_o, _, _ = orc.optimize_keyframes(sub_o, s)
            assert (rep.iterations, rep.stop_reason, rep.num_gaussians, rep.num_memberships) == (rep_o.iterations, rep_o.stop_reason, rep_o.num_gaussians, rep_o.num_memberships)
            go_a, gt_a = orc.relative2global(sub.relOrientations, sub.relTranslations)
            go_o, gt_o = orc.relative2global(sub_o.relOrientations, sub_o.relTranslations)
            assert np.abs(gt_a - gt_o).max() < 1e-4 and np.abs(go_a - go_o).max() < 1e-4
        seq.updatePosesFromSubmap(f, t, sub)
    assert np.abs(seq.relTranslations - m.relTranslations).max() > 1e-5
    map_path = str(tmp_path / "map.pkl")
    with open(map_path, "wb") as fh:
        pickle.dump(m, fh)
    port = 29500 + ((os.getpid() + 7) % 2000)
    mp.spawn(_full_shape_worker, args=(world, port, map_path, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for o in outs[1:]:
        assert np.array_equal(o["ro"], outs[0]["ro"]) and np.array_equal(o["rt"], outs[0]["rt"])
    assert np.array_equal(outs[0]["ro"], seq.relOrientations) and np.array_equal(outs[0]["rt"], seq.relTranslations)


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher must start two ranks itself (torch.distributed.run, 127.0.0.1) and say who ran
    where -- not quietly measure one GPU.  gloo collectives so that the two ranks can share the one GPU of this box; the line must
    carry n_gpus = 2, both (rank, device) pairs, the collective's world size, and the sharded keyframe pass at the same world size -- as a
    STRONG-scaling workload: a fixed map and a fixed cut (here 17 frames in 4 neighbourhoods), two neighbourhoods per rank, with the
    per-rank telemetry that makes load imbalance visible."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["DMSA_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-iters", "0", "--scans", "2", "--rings", "32",
           "--az", "256", "--static", "2000", "--map-frames", "17", "--neighbourhoods", "4", "--keyframe-steps", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0
    assert sorted(p[0] for p in out["config"]["rank_device"]) == [0, 1]
    assert out["config"]["collective"] == {"backend": "gloo", "world_size": 2}
    kp = out["keyframe_pass"]
    assert kp["n_gpus"] == 2 and kp["frames_total"] == 17 and kp["neighbourhoods"] == 4 and kp["scaling"] == "strong" and kp["steps"] == 2
    assert [(r["rank"], r["neighbourhoods"]) for r in kp["per_rank"]["ranks"]] == [(0, 2), (1, 2)]
    assert all(r["ms"] > 0 and r["points"] > 0 and r["gaussians"] > 0 for r in kp["per_rank"]["ranks"]) and kp["per_rank"]["max_over_mean_ms"] >= 1.0
    # a launcher that disagrees with --gpus is an error, not a silent fallback
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=300, env=env2)
    assert r2.returncode != 0 and "WORLD_SIZE=1" in (r2.stdout + r2.stderr)
