"""Keyframe-neighbourhood sharding: one neighbourhood per GPU, one all-gather of relative poses.

The reference optimises ONE neighbourhood (fromId..newest) per new keyframe.  The seam that makes sharding possible is
already in it: keyframe poses are parameterised as RELATIVE poses (ConsecutivePoses.h:45-67) and
MapManagement::getSubmap / updatePosesFromSubmap (MapManagement.h:254-288) cut out an independent problem whose first
frame is fixed and write its relative poses back.  Here the ring buffer is cut into `world` contiguous neighbourhoods
that share one boundary frame (to_i == from_{i+1}); rank i owns the relative-pose columns from_i+1..to_i, runs a full
optimizeSet on its own GPU with only its frames resident, and the only exchange is ONE all-gather of
(to_i - from_i) x 6 doubles per rank (<= 1.5 KB: latency-bound over xGMI, no collective inside the iterations).
"""
from __future__ import annotations

import numpy as np

from .problems import DmsaOptimSettings, MapManagement


def neighbourhood_ranges(num_frames: int, world: int):
    """Contiguous [from, to] frame ranges (inclusive) sharing one boundary frame; every rank gets >= 2 frames."""
    import ctypes as C

    from . import _capi as capi

    if world < 1 or num_frames < world + 1:
        raise ValueError("need at least world + 1 keyframes")
    f, t = (C.c_int32 * world)(), (C.c_int32 * world)()
    if capi.load_library().dmsa_neighbourhood_ranges(int(num_frames), int(world), f, t) != 0:  # include/dmsa_keyframe_map.h
        raise ValueError("dmsa_neighbourhood_ranges")
    return [(int(f[i]), int(t[i])) for i in range(world)]


def gather_neighbourhood_poses(full_map: MapManagement, sub: MapManagement, ranges, rank: int, world: int, dist=None, device=None):
    """The one exchange step of the sharded pass: all-gather every rank's optimised relative poses ((to - from) x 6
    doubles) and apply updatePosesFromSubmap (MapManagement.h:278-288) for every neighbourhood on every rank."""
    f0, f1 = ranges[rank]
    width = max(t - f for f, t in ranges)
    # updatePosesFromSubmap on the owner (its global2relative round trip needs the submap's first pose), then the owner's columns travel
    full_map.updatePosesFromSubmap(f0, f1, sub)
    mine = np.zeros((width, 6))
    mine[: f1 - f0, :3] = full_map.relOrientations[f0 + 1:f1 + 1]
    mine[: f1 - f0, 3:] = full_map.relTranslations[f0 + 1:f1 + 1]
    if world > 1:
        import torch

        t = torch.from_numpy(mine)
        if device is not None:
            t = t.to(device)
        gathered = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        parts = [g.cpu().numpy() for g in gathered]
    else:
        parts = [mine]
    for (f, t_), part in zip(ranges, parts):  # disjoint columns: the order of application does not matter
        k = t_ - f
        full_map.relOrientations[f + 1:t_ + 1] = part[:k, :3]
        full_map.relTranslations[f + 1:t_ + 1] = part[:k, 3:]


def owned_neighbourhoods(num_neighbourhoods: int, rank: int, world: int):
    """Strong-scaling ownership: the map is cut into `num_neighbourhoods` whatever the world size; rank r runs r, r + world, ... one after
    the other (world == num_neighbourhoods is the one-per-GPU layout above)."""
    return list(range(rank, num_neighbourhoods, world))


def gather_owned_neighbourhood_poses(full_map: MapManagement, subs: dict, ranges, rank: int, world: int, dist=None, device=None):
    """gather_neighbourhood_poses for a FIXED cut (len(ranges) neighbourhoods, any world size): `subs` maps the indices this rank owns
    (owned_neighbourhoods) to their optimised submaps.  Still ONE all-gather: every rank sends ceil(len(ranges) / world) slots of
    (width x 6) doubles, a slot per owned neighbourhood; afterwards every rank holds the same relative poses, and they are the same for
    every world size (updatePosesFromSubmap reads nothing but the submap and writes disjoint columns)."""
    num = len(ranges)
    width = max(t - f for f, t in ranges)
    slots = (num + world - 1) // world
    mine = np.zeros((slots, width, 6))
    for j, nb in enumerate(owned_neighbourhoods(num, rank, world)):
        f0, f1 = ranges[nb]
        full_map.updatePosesFromSubmap(f0, f1, subs[nb])
        mine[j, : f1 - f0, :3] = full_map.relOrientations[f0 + 1:f1 + 1]
        mine[j, : f1 - f0, 3:] = full_map.relTranslations[f0 + 1:f1 + 1]
    if world > 1:
        import torch

        t = torch.from_numpy(mine)
        if device is not None:
            t = t.to(device)
        gathered = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        parts = [g.cpu().numpy() for g in gathered]
    else:
        parts = [mine]
    for r, part in enumerate(parts):
        for j, nb in enumerate(owned_neighbourhoods(num, r, world)):
            f, t_ = ranges[nb]
            k = t_ - f
            full_map.relOrientations[f + 1:t_ + 1] = part[j, :k, :3]
            full_map.relTranslations[f + 1:t_ + 1] = part[j, :k, 3:]


def optimize_neighbourhoods(full_map: MapManagement, settings: DmsaOptimSettings, optimize_fn, rank: int = 0, world: int = 1,
                            dist=None, device=None):
    """Shard `full_map` over `world` ranks and optimise this rank's neighbourhood with `optimize_fn(submap, settings)`
    (which updates the submap's relative poses in place).  With world > 1, `dist` is an initialised torch.distributed
    module (backend nccl == RCCL on the GPU box, gloo in the CPU tests).  Every rank returns with the SAME updated
    relative poses in full_map."""
    ranges = neighbourhood_ranges(full_map.numFrames, world)
    sub = full_map.getSubmap(*ranges[rank])
    report = optimize_fn(sub, settings)
    gather_neighbourhood_poses(full_map, sub, ranges, rank, world, dist, device)
    return report
