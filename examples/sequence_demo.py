"""A miniature of DmsaSlam::processPointCloud (DmsaSlam.h:115-202) on synthetic scans, composed only of this library's calls:

    PointCloud2 bytes --decode--> scan --preProcess--> ring buffer --prepareTrajectoryForOptimization--> window problem
      --addStaticPoints (keyframe map)--> optimizeSet --> keyframe decision --addNewKeyframeCloud--> map ... --> TUM lines

The orchestration itself (ring buffers, when to add a keyframe) is deliberately tiny and lives here, not in the library:
SURVEY.md 8 puts it outside the hot path.  Run:  python examples/sequence_demo.py [--scans 14]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dmsa_lidar_slam_amd import posemath, synth  # noqa: E402
from dmsa_lidar_slam_amd import window_setup as ws  # noqa: E402
from dmsa_lidar_slam_amd import wire_formats as wf  # noqa: E402
from dmsa_lidar_slam_amd.api import DmsaOptimizer  # noqa: E402
from dmsa_lidar_slam_amd.keyframe_cloud import KeyframeCloudBuilder  # noqa: E402
from dmsa_lidar_slam_amd.problems import DmsaOptimSettings, MapManagement  # noqa: E402
from dmsa_lidar_slam_amd.static_points import StaticPointSelector, StaticSelectProblem  # noqa: E402

f32 = np.float32
OUSTER = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("pad0", "<f4"), ("intensity", "<f4"), ("t", "<u4"), ("reflectivity", "<u2"), ("ring", "u1"),
                   ("pad1", "u1"), ("ambient", "<u2"), ("pad2", "<u2"), ("range", "<u4")])
OUSTER_FIELDS = ["x", "y", "z", "intensity", "t", "reflectivity", "ring", "ambient", "range"]


HESAI = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<f4"), ("timestamp", "<f8"), ("ring", "<u2")])  # PandarXT: absolute f64 stamps
HESAI_FIELDS = ["x", "y", "z", "intensity", "timestamp", "ring"]


def scan_to_hesai_pointcloud2(xyz, stamps, rings):
    """What the Hesai driver publishes (dmsa_slam_ros.cpp:411-419: stamp = double @fields[4], ring = uint16 @fields[5])."""
    rec = np.zeros(xyz.shape[0], HESAI)
    rec["x"], rec["y"], rec["z"], rec["timestamp"], rec["ring"] = xyz[:, 0], xyz[:, 1], xyz[:, 2], stamps, rings
    offs = np.array([HESAI.fields[n][1] for n in HESAI_FIELDS], np.uint32)
    return wf.PointCloud2Msg(height=1, width=xyz.shape[0], point_step=HESAI.itemsize, field_offsets=offs, data=np.frombuffer(rec.tobytes(), np.uint8).copy(),
                             stamp=float(stamps.min()))


LIVOX = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("reflectivity", "<f4"), ("tag", "u1"), ("line", "u1"), ("timestamp", "<f8")])  # XYZRTLT, packed
LIVOX_FIELDS = ["x", "y", "z", "reflectivity", "tag", "line", "timestamp"]


def scan_to_livox_pointcloud2(xyz, stamps):
    """What livox_ros_driver2 publishes (XYZRTLT, per-point timestamp in NANOSECONDS as a double: the driver bug the node's
    livoxXYZRTLT_ns branch works around, dmsa_slam_ros.cpp:459-469).  No ring field: the node assigns id = k % 1000."""
    rec = np.zeros(xyz.shape[0], LIVOX)
    rec["x"], rec["y"], rec["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    rec["timestamp"] = np.round(stamps * 1e9)
    offs = np.array([LIVOX.fields[n][1] for n in LIVOX_FIELDS], np.uint32)
    return wf.PointCloud2Msg(height=1, width=xyz.shape[0], point_step=LIVOX.itemsize, field_offsets=offs, data=np.frombuffer(rec.tobytes(), np.uint8).copy(),
                             stamp=float(stamps.min()))


def scan_to_pointcloud2(xyz, stamps, rings):
    """What an Ouster driver would publish for this scan (header stamp = first point, t = nanoseconds since then)."""
    rec = np.zeros(xyz.shape[0], OUSTER)
    rec["x"], rec["y"], rec["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    head = float(stamps.min())
    rec["t"] = np.round((stamps - head) * 1e9).astype(np.uint32)
    rec["ring"] = rings
    offs = np.array([OUSTER.fields[n][1] for n in OUSTER_FIELDS], np.uint32)
    return wf.PointCloud2Msg(height=1, width=xyz.shape[0], point_step=OUSTER.itemsize, field_offsets=offs, data=np.frombuffer(rec.tobytes(), np.uint8).copy(),
                             stamp=head)


class GpuBackend:
    """The library calls MiniSlam strings together.  (tests/test_gpu_sequence.py runs the same MiniSlam on a backend made of the CPU
    oracle's functions and compares the two trajectories.)"""

    def __init__(self, parity: bool = False, sensor: str = "ouster"):
        self.sensor = sensor
        self.optimizer = DmsaOptimizer(device=0)
        self.kf_optimizer = DmsaOptimizer(device=0)  # keyframeMapOptimizer (DmsaSlam.h:53)
        self.decoder = wf.PointCloud2Decoder(sensor)
        self.scan_filter = StaticPointSelector(0)                      # preProcess works on raw scans: its own context
        self.static = StaticPointSelector(optimizer=self.optimizer)     # these two share the optimizer's context: the window cloud
        self.kf_builder = KeyframeCloudBuilder(optimizer=self.optimizer)  # stays resident in HBM between the steps
        self.setup = ws.WindowSetup(0)

    def close(self):
        for o in (self.decoder, self.scan_filter, self.static, self.setup, self.kf_builder, self.optimizer, self.kf_optimizer):
            o.close()

    def decode(self, msg):
        return self.decoder.decode(msg)

    def preProcess(self, xyz, seed, max_pts):
        return self.scan_filter.preProcess(xyz, seed, max_pts)

    def newImuBuffer(self):
        return ws.ImuBuffer(10000)

    def prepare(self, buffer, old_traj, initialized, C, dt_res, imu=None, cov_gyr=None, cov_acc=None):
        return self.setup.prepareTrajectoryForOptimization(buffer, old_traj, initialized, C, dt_res, imu, cov_gyr, cov_acc)

    def gravityEstimate(self, traj):
        return self.setup.getSubmapGravityEstimate(traj)

    def addStaticPoints(self, prob, key_xyz, key_nrm, key_ring, offsets, curr_pos, seed):
        self.optimizer.upload(prob)
        self.optimizer.poseTables(self.optimizer.getPoseParameters(), download=False)
        self.optimizer.updateGlobalPoints(0, download=False)  # trajIn.globalPoints: resident, never downloaded
        sp = StaticSelectProblem(windowPoints=None, numWindowResident=prob.localPoints.shape[0], keyframeIds=np.arange(len(offsets) - 1, dtype=np.int32),
                                 frameOffsets=offsets, keyPoints=key_xyz, keyNormals=key_nrm, keyRingIds=key_ring, currPos=curr_pos, minGridSize=prob.minGridSize)
        _, active, active_ids, overlap = self.static.addStaticPoints(sp, seed)
        return active, active_ids, overlap

    def optimizeSet(self, prob, settings):
        return self.optimizer.optimizeSet(prob, settings)

    def optimizeKeyframes(self, submap, settings):
        return self.kf_optimizer.optimizeSet(submap, settings)

    def keyframeCloud(self, prob, pos0, orient0, seed):
        # the optimised window is still resident (final updateGlobalPoints of optimizeSet, DmsaOptimizer.h:149)
        xyz, nrm, ring, _ = self.kf_builder.addNewKeyframeCloud(None, None, prob.minGridSize, seed, pos0, orient0, numResident=prob.localPoints.shape[0])
        return xyz, nrm, ring

    def tumLine(self, stamp, pos, orient):
        return wf.addPoseToFile(stamp, pos, orient)


class MiniSlam:
    def __init__(self, backend=None, n_clouds=5, num_control_poses=6, dt_res=1e-3, max_points_per_scan=3000, min_overlap_new_keyframe=0.7, dist_new_keyframe=1.0,
                 seed=7, num_iter=5, num_iter_keyframe_optim=0, use_imu=False, gravity_outlier_thresh=1.0):
        self.be = backend or GpuBackend()
        self.imu = self.be.newImuBuffer() if use_imu else None  # processImuMeasurements (DmsaSlam.h:100-113) feeds it
        self.cov_gyr, self.cov_acc, self.gravity_thresh = np.diag([1e-4] * 3), np.diag([1e-2] * 3), gravity_outlier_thresh
        self.kf_iters = num_iter_keyframe_optim  # > 0: keyframeOptimization after every new keyframe (DmsaSlam.h:181-184, :212-238)
        self.n_clouds, self.C, self.dt_res, self.max_pts, self.seed, self.num_iter = n_clouds, num_control_poses, dt_res, max_points_per_scan, seed, num_iter
        self.min_overlap, self.dist_kf = min_overlap_new_keyframe, dist_new_keyframe
        self.buffer, self.old_traj, self.initialized = [], None, False
        self.map = None       # MapManagement: the keyframe map
        self.lines, self.log = [], []

    def close(self):
        self.be.close()

    @property
    def keyframes(self):
        return 0 if self.map is None else self.map.numFrames

    def _keyframe_poses(self):
        return posemath.relative2global(self.map.relOrientations, self.map.relTranslations)

    def _map_global(self):
        """getGlobalKeyframeCloud (MapManagement.h:290-299) for every keyframe: float transform of points and normals."""
        from scipy.spatial.transform import Rotation as Rot

        go, gt = self._keyframe_poses()
        xyz, nrm = self.map.localPoints.copy(), self.map.localNormals.copy()
        for k in range(self.map.numFrames):
            a, b = int(self.map.frameOffsets[k]), int(self.map.frameOffsets[k + 1])
            R = Rot.from_rotvec(go[k]).as_matrix().astype(f32)
            xyz[a:b, :3] = (self.map.localPoints[a:b, :3] @ R.T + gt[k].astype(f32)).astype(f32)
            nrm[a:b, :3] = (np.nan_to_num(self.map.localNormals[a:b, :3]) @ R.T).astype(f32)
        return xyz, nrm

    def process(self, msg, first_pose=None):
        # callbackPointCloud + preProcess
        xyz, stamps, ids = self.be.decode(msg)
        fxyz, src, grid = self.be.preProcess(xyz, self.seed, self.max_pts)
        self.buffer.append((fxyz[:, :3].copy(), stamps[src], ids[src], f32(grid)))
        if len(self.buffer) > self.n_clouds:
            self.buffer.pop(0)
        if len(self.buffer) < self.n_clouds:
            return
        # prepareTrajectoryForOptimization
        traj, prob, self.initialized = self.be.prepare(self.buffer, self.old_traj, self.initialized, self.C, self.dt_res, self.imu, self.cov_gyr, self.cov_acc)
        if self.old_traj is None and first_pose is not None:  # the demo starts in motion: seed the first window with its true poses
            prob.relOrientations[...], prob.relTranslations[...] = first_pose(traj)
        settings = DmsaOptimSettings.sliding_window(use_imu=self.imu is not None, num_iter=self.num_iter)
        overlap = 0.0
        if self.keyframes:  # addStaticPoints against the (here: all) keyframes
            kx, kn = self._map_global()
            go, gt = posemath.relative2global(prob.relOrientations, prob.relTranslations)
            active, active_ids, overlap = self.be.addStaticPoints(prob, kx, kn, self.map.ringIds, self.map.frameOffsets, gt[0].astype(f32), self.seed)
            prob.staticPoints, prob.staticRingIds = active, active_ids
        rep = self.be.optimizeSet(prob, settings)
        traj.relOrientations[...], traj.relTranslations[...] = prob.relOrientations, prob.relTranslations
        self.old_traj = traj
        go, gt = posemath.relative2global(prob.relOrientations, prob.relTranslations)
        self.lines.append(self.be.tumLine(traj.t0, gt[0], go[0]))
        # keyframe decision (DmsaSlam.h:170-186)
        need = not self.keyframes or overlap < self.min_overlap or np.linalg.norm(gt[0] - self._keyframe_poses()[1][-1]) > self.dist_kf
        kf_rep = None
        if need:
            kxyz, knrm, kring = self.be.keyframeCloud(prob, gt[0], go[0], self.seed)
            grav, plausible = None, True
            if self.imu is not None:  # measuredGravity + plausibility gate (DmsaSlam.h:533-539)
                traj.globOrientations[...], traj.globTranslations[...] = go, gt
                grav = self.be.gravityEstimate(traj)
                plausible = bool(abs(np.linalg.norm(grav) - 9.805) < self.gravity_thresh)
            self.map = MapManagement.addKeyframe(self.map, gt[0], go[0], kxyz, knrm, kring, prob.minGridSize, measuredGravity=grav, gravityPlausible=plausible,
                                                 useOdometryErrorTerms=True, useGravityErrorTerms=self.imu is not None)
            if self.kf_iters > 0 and self.map.numFrames >= 3:  # keyframeOptimization(fromId = 0, KeyframeMap)
                last = self.map.numFrames - 1
                sub = self.map.getSubmap(0, last)
                kf_rep = self.be.optimizeKeyframes(sub, DmsaOptimSettings.keyframe_map(num_iter=self.kf_iters))
                self.map.updatePosesFromSubmap(0, last, sub)
                kgo, kgt = self._keyframe_poses()
                traj.relOrientations[0], traj.relTranslations[0] = kgo[-1], kgt[-1]  # "update curr trajectory" (:233-237)
        self.log.append({"t0": traj.t0, "pos": gt[0].copy(), "orient": go[0].copy(), "iterations": rep.iterations, "gaussians": rep.num_gaussians,
                         "static": int(prob.staticPoints.shape[0]), "overlap": float(overlap), "keyframes": self.keyframes,
                         "rel": (prob.relOrientations.copy(), prob.relTranslations.copy()),
                         "keyframe_opt": None if kf_rep is None else (kf_rep.iterations, kf_rep.num_gaussians),
                         "map_rel": (self.map.relOrientations.copy(), self.map.relTranslations.copy())})


def run(scans=14, rings=64, az_steps=512, seed=1, backend=None, livox=False, hesai=False, record=None, replay=None, **slam_args):
    """record: write the message stream (PointCloud2 + Imu, bag order) to a flat dump (include/dmsa_raw_sequence.h) while running;
    replay: run from such a dump instead of generating the messages -- what a recorded sequence converted by scripts/rosbag_to_raw.py
    goes through (src/dmsa_slam_ros.cpp:240-307)."""
    from dmsa_lidar_slam_amd import raw_sequence as rs

    if livox:  # BASELINE.json config 5: Livox-like rosette scans, livoxXYZRTLT_ns messages, no IMU
        clouds, truth = synth.rosette_scan_sequence(seed=seed, scans=scans)
        backend = backend or GpuBackend(sensor="livoxXYZRTLT_ns")
    else:
        clouds, truth = synth.scan_sequence(seed=seed, scans=scans, rings=rings, az_steps=az_steps)
        if hesai:  # BASELINE.json config 2: PandarXT-32 messages (+ IMU through use_imu=True)
            backend = backend or GpuBackend(sensor="hesai")
    slam = MiniSlam(backend, **slam_args)
    writer = rs.RawWriter(record) if record else None

    def first_pose(traj):
        R, p = truth.pose(traj.t0 - 1.6e9 + traj.stamps)
        return posemath.global2relative(R.as_rotvec(), p)

    if replay:  # the bag loop (:270-281): messages in file order, Imu -> processImuMeasurements, PointCloud2 -> processPointCloud
        for kind, m in rs.RawReader(replay):
            if kind == "imu":
                if slam.imu is not None:
                    slam.imu.addMeasurement(m[2], m[1], m[0])
            else:
                slam.process(m, first_pose)
    else:
        if slam.imu is not None:  # the whole IMU stream up front (the node interleaves the two callbacks): 50 samples at rest for the gyro bias, then the drive
            st, acc, ang = synth.imu_stream(truth, -0.3, scans * 0.1 + 0.3, rate=400.0, rng=np.random.default_rng(seed + 50), sigma_acc=0.02, sigma_gyr=0.002)
            for t in st[0] - (50 - np.arange(50)) * 0.0025:
                slam.imu.addMeasurement([0.0, 0.0, 9.805], np.zeros(3), t)
                if writer:
                    writer.writeImu(t, np.zeros(3), [0.0, 0.0, 9.805])
            for t, a, w in zip(st, acc, ang):
                slam.imu.addMeasurement(a, w, t)
                if writer:
                    writer.writeImu(t, w, a)
        for xyz, stamps, ring, _ in clouds:
            msg = scan_to_livox_pointcloud2(xyz, stamps) if livox else scan_to_hesai_pointcloud2(xyz, stamps, ring) if hesai else scan_to_pointcloud2(xyz, stamps, ring)
            if writer:
                writer.writePointCloud2(msg)
            slam.process(msg, first_pose)
    if writer:
        writer.close()
    errs = []
    for e in slam.log:
        _, p = truth.pose(np.array([e["t0"] - 1.6e9]))
        errs.append(float(np.linalg.norm(e["pos"] - p[0])))
    out = {"windows": len(slam.log), "keyframes": slam.keyframes, "max_position_error_m": max(errs) if errs else None, "log": slam.log, "tum": slam.lines}
    slam.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=14)
    ap.add_argument("--keyframe-dist", type=float, default=0.25, help="dist_new_keyframe [m]")
    ap.add_argument("--keyframe-iters", type=int, default=3, help="num_iter_keyframe_optim (0 = no keyframe optimisation)")
    ap.add_argument("--imu", action="store_true", help="IMU rows in the window, gravity rows in the keyframe pass")
    ap.add_argument("--hesai", action="store_true", help="32-ring scans as Hesai PandarXT messages (absolute double stamps, uint16 ring)")
    ap.add_argument("--livox", action="store_true", help="rosette scans as livoxXYZRTLT_ns messages (ids = k % 1000)")
    ap.add_argument("--record", help="also write the message stream to this flat dump (include/dmsa_raw_sequence.h)")
    ap.add_argument("--replay", help="take the messages from this flat dump instead of generating them (same sensor flags as when it was recorded)")
    a = ap.parse_args()
    r = run(a.scans, record=a.record, replay=a.replay, dist_new_keyframe=a.keyframe_dist, num_iter_keyframe_optim=a.keyframe_iters, use_imu=a.imu, livox=a.livox, hesai=a.hesai, **(dict(rings=32, az_steps=512) if a.hesai else {}),
            **(dict(max_points_per_scan=1000) if a.livox else {}))
    for e in r["log"]:
        print(f"t0={e['t0']:.3f} pos=({e['pos'][0]:.3f} {e['pos'][1]:.3f} {e['pos'][2]:.3f}) iters={e['iterations']} M={e['gaussians']} static={e['static']} "
              f"overlap={e['overlap']:.2f} keyframes={e['keyframes']} keyframe_opt={e['keyframe_opt']}")
    print("".join(r["tum"]), end="")
    print(f"windows={r['windows']} keyframes={r['keyframes']} max |position error| = {r['max_position_error_m']:.3f} m")
