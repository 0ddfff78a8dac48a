"""Seeded synthetic inputs for the DMSA hot path.

There is no dataset access on the build or GPU boxes, so every BASELINE.json config is driven
by ray-casting an analytic scene of axis-aligned boxes:
  * `window_problem`   — 128-ring x 1024-azimuth spinning-LiDAR scans (131 072 pts/scan) with per-point
                         stamps along a smooth trajectory, a static map sampled from the same scene,
                         C control poses perturbed from the truth (configs 1/2/3)
  * `keyframe_problem` — F keyframes (undistorted scans with normals) on a closed ring path (config 4)
  * `rosette_window_problem` — non-repetitive (Livox-like) pattern, ids = index % 1000, no IMU (config 5)
Everything is numpy (PCG64 with fixed seeds); nothing here is on the timed path.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
from scipy.spatial.transform import Rotation as Rot

from .posemath import global2relative
from .problems import ContinuousTrajectory, MapManagement


# ------------------------------------------------------------------------------------------------
# scene = one enclosing box seen from inside + obstacle boxes seen from outside
# ------------------------------------------------------------------------------------------------
@dataclass
class Scene:
    room: np.ndarray       # (2,3) min/max
    boxes: np.ndarray      # (nb,2,3)

    @staticmethod
    def room_with_stairs() -> "Scene":
        """Closed 20 x 15 x 6 m room + 4 interior walls + a staircase of 10 boxes."""
        room = np.array([[0.0, 0.0, 0.0], [20.0, 15.0, 6.0]])
        b = [
            [[8.0, 0.0, 0.0], [8.2, 9.0, 6.0]],
            [[12.0, 6.0, 0.0], [12.2, 15.0, 6.0]],
            [[2.0, 10.0, 0.0], [8.0, 10.2, 4.0]],
            [[14.0, 4.0, 0.0], [20.0, 4.2, 3.0]],
        ]
        for k in range(10):  # staircase climbing in +x
            b.append([[14.0 + 0.3 * k, 8.0, 0.0], [14.3 + 0.3 * k, 10.0, 0.2 * (k + 1)]])
        return Scene(room, np.array(b))

    @staticmethod
    def pillar_hall(half: float = 35.0, height: float = 8.0) -> "Scene":
        """70 x 70 x 8 m hall with a ring of pillars and wall stubs (keyframe ring path of radius 20 m)."""
        room = np.array([[-half, -half, 0.0], [half, half, height]])
        b = []
        for k in range(24):
            a = 2 * np.pi * k / 24
            for r in (12.0, 28.0):
                c = np.array([r * np.cos(a + (0.13 if r > 20 else 0.0)), r * np.sin(a + (0.13 if r > 20 else 0.0))])
                b.append([[c[0] - 0.6, c[1] - 0.6, 0.0], [c[0] + 0.6, c[1] + 0.6, height]])
        for k in range(8):
            a = 2 * np.pi * k / 8 + 0.2
            c = np.array([20.0 * np.cos(a), 20.0 * np.sin(a)])
            b.append([[c[0] - 2.5, c[1] - 0.15, 0.0], [c[0] + 2.5, c[1] + 0.15, 2.5]])
        return Scene(room, np.array(b))

    def raycast(self, o: np.ndarray, d: np.ndarray):
        """Nearest hit along rays o + r d (d unit).  Returns ranges (n,) and outward hit normals (n,3)."""
        n = o.shape[0]
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / d
        # enclosing room: exit distance
        t1 = (self.room[0] - o) * inv
        t2 = (self.room[1] - o) * inv
        tfar = np.maximum(t1, t2)
        ax = np.argmin(tfar, axis=1)
        best = tfar[np.arange(n), ax]
        nrm = np.zeros((n, 3))
        nrm[np.arange(n), ax] = -np.sign(d[np.arange(n), ax])
        for bx in self.boxes:
            t1 = (bx[0] - o) * inv
            t2 = (bx[1] - o) * inv
            tn = np.minimum(t1, t2)
            tf = np.maximum(t1, t2)
            axn = np.argmax(tn, axis=1)
            tnear = tn[np.arange(n), axn]
            tfar_b = tf.min(axis=1)
            hit = (tnear > 1e-6) & (tnear <= tfar_b) & (tnear < best)
            idx = np.nonzero(hit)[0]
            best[idx] = tnear[idx]
            nrm[idx] = 0.0
            nrm[idx, axn[idx]] = -np.sign(d[idx, axn[idx]])
        return best, nrm

    def sample_surfaces(self, grid: float, rng: np.random.Generator, count: int, sigma: float = 0.01) -> np.ndarray:
        """`count` points on the scene surfaces, drawn from a `grid`-spaced lattice on every face."""
        pts = []

        def face_grid(lo, hi, axis, value):
            u, v = [a for a in range(3) if a != axis]
            gu = np.arange(lo[u] + grid / 2, hi[u], grid)
            gv = np.arange(lo[v] + grid / 2, hi[v], grid)
            if len(gu) == 0 or len(gv) == 0:
                return
            uu, vv = np.meshgrid(gu, gv, indexing="ij")
            p = np.zeros((uu.size, 3))
            p[:, u], p[:, v], p[:, axis] = uu.ravel(), vv.ravel(), value
            pts.append(p)

        for axis in range(3):
            face_grid(self.room[0], self.room[1], axis, self.room[0][axis])
            face_grid(self.room[0], self.room[1], axis, self.room[1][axis])
        for bx in self.boxes:
            for axis in range(3):
                face_grid(bx[0], bx[1], axis, bx[0][axis])
                face_grid(bx[0], bx[1], axis, bx[1][axis])
        allp = np.concatenate(pts, axis=0)
        sel = rng.choice(allp.shape[0], size=min(count, allp.shape[0]), replace=False)
        sel.sort()
        out = allp[sel] + rng.normal(0.0, sigma, size=(len(sel), 3))
        return out


# ------------------------------------------------------------------------------------------------
# sensor model + trajectory
# ------------------------------------------------------------------------------------------------
def spinning_lidar_dirs(rings: int, az_steps: int, fov_deg: float = 22.5):
    """Unit ray directions in the sensor frame, azimuth-major (all rings fire per azimuth step)."""
    elev = np.deg2rad(np.linspace(-fov_deg, fov_deg, rings))
    az = 2 * np.pi * np.arange(az_steps) / az_steps
    ce, se = np.cos(elev), np.sin(elev)
    d = np.stack([np.outer(np.cos(az), ce), np.outer(np.sin(az), ce), np.outer(np.ones_like(az), se)], axis=-1)
    ring = np.tile(np.arange(rings, dtype=np.int32), az_steps)
    frac = np.repeat(np.arange(az_steps) / az_steps, rings)
    return d.reshape(-1, 3), ring, frac


@dataclass
class SmoothTrajectory:
    """1 m/s forward speed, 0.3 rad/s yaw rate, small roll/pitch/height sinusoids."""

    p0: np.ndarray
    yaw0: float = 0.2
    speed: float = 1.0
    yaw_rate: float = 0.3

    def pose(self, t: np.ndarray):
        t = np.asarray(t, dtype=np.float64)
        yaw = self.yaw0 + self.yaw_rate * t
        x = self.p0[0] + self.speed / self.yaw_rate * (np.sin(yaw) - np.sin(self.yaw0))
        y = self.p0[1] - self.speed / self.yaw_rate * (np.cos(yaw) - np.cos(self.yaw0))
        z = self.p0[2] + 0.05 * np.sin(2 * np.pi * 0.5 * t)
        roll = 0.02 * np.sin(2 * np.pi * 0.7 * t)
        pitch = 0.02 * np.cos(2 * np.pi * 0.4 * t)
        R = Rot.from_euler("ZYX", np.stack([yaw, pitch, roll], axis=-1))
        return R, np.stack([x, y, z], axis=-1)


def _scan(scene: Scene, traj: SmoothTrajectory, t_start: float, period: float, rings: int, az_steps: int,
          rng: np.random.Generator, sigma: float):
    d_local, ring, frac = spinning_lidar_dirs(rings, az_steps)
    t_rel = t_start + frac * period
    R, p = traj.pose(t_rel)
    d_world = R.apply(d_local)
    rng_m, _ = scene.raycast(p, d_world)
    r = rng_m + rng.normal(0.0, sigma, size=rng_m.shape)
    keep = r >= 0.1
    return (d_local[keep] * r[keep, None]).astype(np.float32), ring[keep], t_rel[keep]


def window_problem(seed: int = 1, scans: int = 10, rings: int = 128, az_steps: int = 1024, num_static: int = 200_000,
                   num_control_poses: int = 6, dt_res: float = 1e-3, grid_size: float = 0.15, sigma: float = 0.01,
                   perturb_t: float = 0.02, perturb_r_deg: float = 0.5, scan_period: float = 0.1,
                   use_imu: bool = False, epoch: float = 1.6e9, scene: Scene | None = None) -> ContinuousTrajectory:
    """Sliding-window problem: `scans` x (rings*az_steps) points + `num_static` map points."""
    rng = np.random.default_rng(seed)
    scene = scene or Scene.room_with_stairs()
    traj = SmoothTrajectory(p0=np.array([4.0, 3.0, 1.5]))
    pts, ids, stamps = [], [], []
    for s in range(scans):
        p, r, t = _scan(scene, traj, s * scan_period, scan_period, rings, az_steps, rng, sigma)
        pts.append(p), ids.append(r), stamps.append(t)
    pts = np.concatenate(pts)
    ids = np.concatenate(ids)
    stamps_rel = np.concatenate(stamps)
    stamps_abs = epoch + stamps_rel  # UNIX-epoch sized doubles, as PointStampId::stamp carries them
    # initTraj (ContinuousTrajectory.h:301-346)
    t0 = stamps_abs.min()
    horizon = stamps_abs.max() - t0 + dt_res
    n_total = int(round(horizon / dt_res)) + 1
    traj_time = np.linspace(0.0, horizon, n_total)
    ctrl_stamps = np.linspace(0.0, horizon, num_control_poses)
    # registerPcBuffer (:240-260): lower_bound(trajTime, stamp - t0), clamped
    tform_idx = np.minimum(np.searchsorted(traj_time, stamps_abs - t0, side="left"), n_total - 1).astype(np.int32)
    # control poses: truth perturbed (pose 0 stays exact — it is the fixed anchor)
    Rk, pk = traj.pose((t0 - epoch) + ctrl_stamps)
    go = Rk.as_rotvec()
    gt = pk.copy()
    noise_r = Rot.from_rotvec(rng.normal(0.0, np.deg2rad(perturb_r_deg), size=(num_control_poses, 3)))
    go_p = (Rk * noise_r).as_rotvec()
    gt_p = gt + rng.normal(0.0, perturb_t, size=gt.shape)
    go_p[0], gt_p[0] = go[0], gt[0]
    ro, rt = global2relative(go_p, gt_p)
    static = scene.sample_surfaces(grid_size / 2, rng, num_static, sigma) if num_static > 0 else np.zeros((0, 3))
    static_ids = rng.integers(0, rings, size=static.shape[0]).astype(np.int32)
    kw = {}
    if use_imu:
        kw = _imu_factors(traj, t0 - epoch, ctrl_stamps, dt_res, rng)
    prob = ContinuousTrajectory(
        relOrientations=ro, relTranslations=rt, stamps=ctrl_stamps, trajTime=traj_time, localPoints=pts,
        tformIdPerPoint=tform_idx, ringIds=ids, staticPoints=static.astype(np.float32), staticRingIds=static_ids,
        minGridSize=grid_size, useImuErrorTerms=use_imu, dt_res=dt_res, **kw)
    prob.truth_global = (go, gt)  # for convergence checks
    # what the ring buffer of scans holds (PointStampId::stamp per point, one cloud per scan) and the origin of the time grid
    prob.pointStamps, prob.t0 = stamps_abs, float(t0)
    prob.scanOffsets = np.concatenate([[0], np.cumsum([len(t) for t in stamps])]).astype(np.int64)
    return prob


def rosette_scan_sequence(seed: int = 1, scans: int = 12, pts_per_scan: int = 12_000, scan_period: float = 0.1, sigma: float = 0.01,
                          grid_size: float = 0.15, epoch: float = 1.6e9):
    """Config 5's scan stream: non-repetitive scan pattern (Livox Mid-360-like), no rings -- the node assigns id = k % 1000 while
    decoding (dmsa_slam_ros.cpp:449-469).  Same tuple layout as scan_sequence; the third entry is the point index inside its scan."""
    rng = np.random.default_rng(seed)
    scene = Scene.room_with_stairs()
    traj = SmoothTrajectory(p0=np.array([4.0, 3.0, 1.5]))
    clouds = []
    for s in range(scans):
        frac = np.arange(pts_per_scan) / pts_per_scan
        tt = s * scan_period + frac * scan_period
        a = 2 * np.pi * (17.0 * frac + 0.37 * s)  # Mid-360-like: full azimuth, elevation -7 .. 52 degrees, never the same line twice
        el = np.deg2rad(-7.0 + 59.0 * np.abs(np.cos(2 * np.pi * 3.4 * frac + 0.11 * s)))
        d = np.stack([np.cos(el) * np.cos(a), np.cos(el) * np.sin(a), np.sin(el)], axis=-1)
        R, p = traj.pose(tt)
        r, _ = scene.raycast(p, R.apply(d))
        r = r + rng.normal(0, sigma, r.shape)
        keep = r >= 0.1
        clouds.append(((d[keep] * r[keep, None]).astype(np.float32), epoch + tt[keep], np.arange(int(keep.sum()), dtype=np.int32), np.float32(grid_size)))
    return clouds, traj


def scan_sequence(seed: int = 1, scans: int = 12, rings: int = 32, az_steps: int = 256, scan_period: float = 0.1, sigma: float = 0.01,
                  grid_size: float = 0.15, epoch: float = 1.6e9, scene: Scene | None = None):
    """A stream of scans as the node buffers them (PointCloudBuffer): list of (xyz_local (n,3) f32, absolute stamps (n,) f64,
    ring ids (n,) i32, gridSize) plus the generating trajectory — input of window_setup.prepareTrajectoryForOptimization."""
    rng = np.random.default_rng(seed)
    scene = scene or Scene.room_with_stairs()
    traj = SmoothTrajectory(p0=np.array([4.0, 3.0, 1.5]))
    clouds = []
    for s in range(scans):
        p, r, t = _scan(scene, traj, s * scan_period, scan_period, rings, az_steps, rng, sigma)
        clouds.append((p, epoch + t, r.astype(np.int32), np.float32(grid_size)))
    return clouds, traj


def imu_stream(traj: SmoothTrajectory, t_begin: float, t_end: float, rate: float = 400.0, epoch: float = 1.6e9, rng=None, sigma_acc: float = 0.0,
               sigma_gyr: float = 0.0):
    """Body-frame specific force and angular velocity along `traj` (what an IMU rigidly mounted at the sensor origin reports,
    gravity (0, 0, -9.805) as in ContinuousTrajectory.h:345): stamps (absolute), acc (n,3), ang_vel (n,3)."""
    g = np.array([0.0, 0.0, -9.805])
    t = np.arange(t_begin, t_end, 1.0 / rate)
    h = 1e-4
    R, _ = traj.pose(t)
    _, pp = traj.pose(t + h)
    _, p0 = traj.pose(t)
    _, pm = traj.pose(t - h)
    a_world = (pp - 2.0 * p0 + pm) / (h * h)
    acc = R.inv().apply(a_world - g)
    Rp, _ = traj.pose(t + h)
    Rm, _ = traj.pose(t - h)
    ang = (Rm.inv() * Rp).as_rotvec() / (2.0 * h)  # body-frame rate
    if rng is not None:
        acc = acc + rng.normal(0.0, sigma_acc, acc.shape)
        ang = ang + rng.normal(0.0, sigma_gyr, ang.shape)
    return epoch + t, acc, ang


def _imu_factors(traj: SmoothTrajectory, t_off: float, ctrl_stamps: np.ndarray, dt_res: float, rng):
    """Preintegrated factors consistent with the truth (stand-in for ImuPreintegration, which is outside the hot path)."""
    c = len(ctrl_stamps)
    g = np.array([0.0, 0.0, -9.805])
    Rk, pk = traj.pose(t_off + ctrl_stamps)
    _, p_plus = traj.pose(t_off + ctrl_stamps + dt_res)
    _, p_minus = traj.pose(t_off + ctrl_stamps - dt_res)
    rot = np.tile(np.eye(3), (c, 1, 1))
    pos = np.zeros((c, 3))
    vel = np.zeros((c, 3))
    cov = np.tile(np.eye(9), (c, 1, 1))
    Rm = Rk.as_matrix()
    for k in range(1, c):
        dt = ctrl_stamps[k] - ctrl_stamps[k - 1]
        v0 = (p_plus[k - 1] - pk[k - 1]) / dt_res
        v1 = (pk[k] - p_minus[k]) / dt_res
        rot[k] = Rm[k - 1].T @ Rm[k] @ Rot.from_rotvec(rng.normal(0, 1e-4, 3)).as_matrix()
        vel[k] = Rm[k - 1].T @ (v1 - v0 - g * dt) + rng.normal(0, 1e-3, 3)
        pos[k] = Rm[k - 1].T @ (pk[k] - pk[k - 1] - v0 * dt - 0.5 * g * dt * dt) + rng.normal(0, 1e-4, 3)
        cov[k] = np.diag([1e6] * 3 + [1e4] * 3 + [1e6] * 3)
    idx = np.round(ctrl_stamps / dt_res).astype(np.int32)
    return dict(paramIndices=idx, preintImuRots=rot, preintRelPositions=pos, preintRelVelocity=vel, CovPVRot_inv=cov,
                balancingImu=0.001, gravity=g)


def rosette_window_problem(seed: int = 1, scans: int = 5, pts_per_scan: int = 24_000, num_static: int = 20_000,
                           grid_size: float = 0.15, dt_res: float = 1e-3, sigma: float = 0.01) -> ContinuousTrajectory:
    """Config 5: non-repetitive rosette pattern (Livox-like), ids = k % 1000 (dmsa_slam_ros.cpp livox branch), no IMU."""
    rng = np.random.default_rng(seed)
    scene = Scene.room_with_stairs()
    traj = SmoothTrajectory(p0=np.array([4.0, 3.0, 1.5]))
    period = 0.1
    pts, ids, stamps = [], [], []
    for s in range(scans):
        frac = np.arange(pts_per_scan) / pts_per_scan
        tt = s * period + frac * period
        a = 2 * np.pi * (17.0 * frac + 0.37 * s)
        rad = np.deg2rad(35.0) * np.abs(np.cos(2 * np.pi * 3.4 * frac + 0.11 * s))
        d = np.stack([np.cos(rad), np.sin(rad) * np.cos(a), np.sin(rad) * np.sin(a)], axis=-1)
        R, p = traj.pose(tt)
        r, _ = scene.raycast(p, R.apply(d))
        r = r + rng.normal(0, sigma, r.shape)
        keep = r >= 0.1
        pts.append((d[keep] * r[keep, None]).astype(np.float32))
        ids.append((np.arange(pts_per_scan)[keep] % 1000).astype(np.int32))
        stamps.append(tt[keep])
    pts, ids, st = np.concatenate(pts), np.concatenate(ids), np.concatenate(stamps)
    t0 = st.min()
    horizon = st.max() - t0 + dt_res
    n_total = int(round(horizon / dt_res)) + 1
    traj_time = np.linspace(0.0, horizon, n_total)
    cs = np.linspace(0.0, horizon, 6)
    tidx = np.minimum(np.searchsorted(traj_time, st - t0, side="left"), n_total - 1).astype(np.int32)
    Rk, pk = traj.pose(t0 + cs)
    go, gt = Rk.as_rotvec(), pk.copy()
    go_p = (Rk * Rot.from_rotvec(rng.normal(0, np.deg2rad(0.5), (6, 3)))).as_rotvec()
    gt_p = gt + rng.normal(0, 0.02, gt.shape)
    go_p[0], gt_p[0] = go[0], gt[0]
    ro, rt = global2relative(go_p, gt_p)
    static = scene.sample_surfaces(grid_size / 2, rng, num_static, sigma)
    prob = ContinuousTrajectory(relOrientations=ro, relTranslations=rt, stamps=cs, trajTime=traj_time, localPoints=pts,
                                tformIdPerPoint=tidx, ringIds=ids, staticPoints=static.astype(np.float32),
                                staticRingIds=rng.integers(0, 1000, static.shape[0]).astype(np.int32),
                                minGridSize=grid_size, dt_res=dt_res)
    prob.truth_global = (go, gt)
    return prob


def keyframe_problem(seed: int = 1, frames: int = 256, rings: int = 32, az_steps: int = 320, radius: float = 20.0,
                     grid_size: float = 0.25, sigma: float = 0.01, perturb_t: float = 0.01, perturb_r_deg: float = 0.1,
                     use_gravity: bool = True, arc: float = 2 * np.pi) -> MapManagement:
    """Config 4: `frames` keyframes (~rings*az_steps points each, with normals) on a ring path of `radius` m."""
    rng = np.random.default_rng(seed)
    scene = Scene.pillar_hall()
    d_local, ring, _ = spinning_lidar_dirs(rings, az_steps)
    ang = arc * np.arange(frames) / frames
    pos = np.stack([radius * np.cos(ang), radius * np.sin(ang), 1.5 + 0.1 * np.sin(3 * ang)], axis=-1)
    R = Rot.from_euler("ZYX", np.stack([ang + np.pi / 2, 0.02 * np.sin(5 * ang), 0.02 * np.cos(4 * ang)], axis=-1))
    pts, nrms, ids, off = [], [], [], [0]
    for k in range(frames):
        dw = R[k].apply(d_local)
        r, n_w = scene.raycast(np.broadcast_to(pos[k], dw.shape).copy(), dw)
        r = r + rng.normal(0, sigma, r.shape)
        keep = r >= 0.1
        n_l = R[k].inv().apply(n_w[keep])
        p_l = d_local[keep] * r[keep, None]
        flip = np.sum(n_l * p_l, axis=1) > 0  # point normals towards the sensor (viewpoint 0,0,0; DmsaSlam.h:557-568)
        n_l[flip] *= -1.0
        pts.append(p_l.astype(np.float32)), nrms.append(n_l.astype(np.float32)), ids.append(ring[keep])
        off.append(off[-1] + int(keep.sum()))
    go, gt = R.as_rotvec(), pos.copy()
    ro, rt = global2relative(go, gt)
    ro_p = ro + rng.normal(0, np.deg2rad(perturb_r_deg), ro.shape)
    rt_p = rt + rng.normal(0, perturb_t, rt.shape)
    ro_p[0], rt_p[0] = ro[0], rt[0]
    g = np.array([0.0, 0.0, -9.805])
    meas = R.inv().apply(np.broadcast_to(g, (frames, 3))) + rng.normal(0, 0.02, (frames, 3))
    prob = MapManagement(relOrientations=ro_p, relTranslations=rt_p, frameOffsets=np.array(off, np.int64),
                         localPoints=np.concatenate(pts), localNormals=np.concatenate(nrms), ringIds=np.concatenate(ids),
                         minGridSize=grid_size, useGravityErrorTerms=use_gravity, measuredGravity=meas,
                         gravityPlausible=np.ones(frames, np.int32), gravity=g)
    prob.truth_relative = (ro, rt)
    return prob


def static_select_problem(seed: int = 1, scans: int = 10, rings: int = 128, az_steps: int = 1024, frames: int = 3, key_rings: int = 32,
                          key_az: int = 320, grid_size: float = 0.15, sigma: float = 0.01):
    """Inputs of DmsaSlam::addStaticPoints (DmsaSlam.h:264-358): the window cloud in the world frame (scans along the
    trajectory) and `frames` keyframe clouds with normals and ring ids recorded earlier in the same room, already transformed
    to the world frame (getGlobalKeyframeCloud).  slam_settings.yaml uses the 3 closest keyframes."""
    from .static_points import StaticSelectProblem

    rng = np.random.default_rng(seed)
    scene = Scene.room_with_stairs()
    traj = SmoothTrajectory(p0=np.array([4.0, 3.0, 1.5]))
    win = []
    for s in range(scans):
        d_local, _, frac = spinning_lidar_dirs(rings, az_steps)
        t_rel = s * 0.1 + frac * 0.1
        R, p = traj.pose(t_rel)
        dw = R.apply(d_local)
        r, _ = scene.raycast(p, dw)
        r = r + rng.normal(0.0, sigma, r.shape)
        keep = r >= 0.1
        win.append((p[keep] + dw[keep] * r[keep, None]).astype(np.float32))
    window = np.concatenate(win)
    d_local, ring, _ = spinning_lidar_dirs(key_rings, key_az)
    pts, nrm, ids, off, kf_ids = [], [], [], [0], []
    for k in range(frames):
        Rk, pk = traj.pose(np.array([-0.8 * (frames - k)]))  # keyframes recorded before the window, oldest first
        dw = Rk.apply(d_local)
        r, n_w = scene.raycast(np.broadcast_to(pk[0], dw.shape).copy(), dw)
        r = r + rng.normal(0.0, sigma, r.shape)
        keep = r >= 0.1
        pw = pk[0] + dw[keep] * r[keep, None]
        nw = n_w[keep].copy()
        flip = np.sum(nw * (pw - pk[0]), axis=1) > 0  # normals point towards the sensor that saw them
        nw[flip] *= -1.0
        pts.append(pw.astype(np.float32)), nrm.append(nw.astype(np.float32)), ids.append(ring[keep])
        off.append(off[-1] + int(keep.sum()))
        kf_ids.append(5 + k)
    order = np.arange(frames)[::-1]  # closestKeyIds: nearest (= newest) keyframe first
    o2 = [0]
    for k in order:
        o2.append(o2[-1] + (off[k + 1] - off[k]))
    _, cur = traj.pose(np.array([0.0]))
    return StaticSelectProblem(windowPoints=window, keyframeIds=np.array([kf_ids[k] for k in order], np.int32), frameOffsets=np.array(o2, np.int64),
                               keyPoints=np.concatenate([pts[k] for k in order]), keyNormals=np.concatenate([nrm[k] for k in order]),
                               keyRingIds=np.concatenate([ids[k] for k in order]), currPos=cur[0].astype(np.float32), minGridSize=grid_size)
