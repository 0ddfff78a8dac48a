// aos_call_demo.cpp — what a drop-in call costs from C++ when the points stay in the reference's own containers.
//
// Reads a window dump ('DMSAWN01', dmsa_lidar_slam_amd/dump.py), rebuilds what DmsaSlam holds at the call site of
// slidingWindowOptimizer.optimizeSet (DmsaSlam.h:166): one array of 32-byte points per scan of the ring buffer -- the layout of
// pcl::PointCloud<PointStampId>::points (PointStampId.h:33-45: float data[4], double stamp, int id, int isStatic, 16-byte aligned) --
// the tformIdPerPoint vectors, and the static tail of globalPoints.  Then times, best of `reps`:
//   aos    dmsa_optimize_window_aos (include/dmsa_aos.h): views of the arrays as they are + dmsa_get_global_points_aos into globalPoints
//   flat   dmsa_optimize_window (include/dmsa_hip.h) from arrays that are ALREADY flat (what bench.py's pcie_inclusive times from Python)
//   ring   the scans resident in HBM (include/dmsa_window_ring.h), as consecutive windows of DmsaSlam share all but one scan: ONE scan
//          pushed from its container (dmsa_window_ring_push_aos), dmsa_window_upload_from_ring_aos, dmsa_optimize_resident, dmsa_get_poses
//   repack the same flat call INCLUDING the per-point repack a caller needs without dmsa_aos.h (what integration/DmsaOptimizerHip.h did
//          until round 3)
// and checks that the three end at the same poses bit for bit.  No PCL / Eigen needed: the structs below only mirror the byte layout.
//
//   aos_call_demo <window.bin> <num_scans> [iterations = 10] [reps = 3]
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/dmsa_aos.h"
#include "../include/dmsa_hip.h"
#include "../include/dmsa_window_ring.h"

struct alignas(16) PointStampIdLayout {  // PointStampId.h:33-45
    float data[4];
    double stamp;
    int32_t id;
    int32_t isStatic;
};
static_assert(sizeof(PointStampIdLayout) == 32, "PointStampId is 32 bytes");

template <class T>
static bool rd(FILE* f, T* p, size_t count) { return std::fread(p, sizeof(T), count, f) == count; }
static double ms_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <window.bin> <num_scans> [iterations] [reps]\n", argv[0]);
        return 2;
    }
    const int num_scans = std::atoi(argv[2]), iterations = argc > 3 ? std::atoi(argv[3]) : 10, reps = argc > 4 ? std::atoi(argv[4]) : 3;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    char magic[8];
    int32_t C, n_total, use_imu, pad;
    int64_t N, S;
    float min_grid, padf;
    double dt_res;
    if (!(rd(f, magic, 8) && !std::memcmp(magic, "DMSAWN01", 8) && rd(f, &C, 1) && rd(f, &n_total, 1) && rd(f, &N, 1) && rd(f, &S, 1) && rd(f, &min_grid, 1) &&
          rd(f, &padf, 1) && rd(f, &use_imu, 1) && rd(f, &pad, 1) && rd(f, &dt_res, 1)) || use_imu || num_scans < 1)
        return 2;
    std::vector<double> ro(3 * C), rt(3 * C), stamps(C), traj(n_total);
    std::vector<float> xyz(4 * N), sxyz(4 * S);
    std::vector<int32_t> tf(N), ring(N), sring(S);
    if (!(rd(f, ro.data(), ro.size()) && rd(f, rt.data(), rt.size()) && rd(f, stamps.data(), stamps.size()) && rd(f, traj.data(), traj.size()) &&
          rd(f, xyz.data(), xyz.size()) && rd(f, tf.data(), tf.size()) && rd(f, ring.data(), ring.size()) && rd(f, sxyz.data(), sxyz.size()) &&
          rd(f, sring.data(), sring.size())))
        return 2;
    std::fclose(f);
    // ---- the reference's containers ----
    std::vector<std::vector<PointStampIdLayout>> clouds(num_scans);   // regPcBuffer->at(pc).points
    std::vector<std::vector<int32_t>> tformIdPerPoint(num_scans);     // ContinuousTrajectory::tformIdPerPoint
    std::vector<PointStampIdLayout> globalPoints((size_t)(N + S));    // OptimizablePointSet::globalPoints
    for (int pc = 0; pc < num_scans; ++pc) {
        const int64_t a = N * pc / num_scans, b = N * (pc + 1) / num_scans;
        clouds[pc].resize((size_t)(b - a)), tformIdPerPoint[pc].assign(tf.begin() + a, tf.begin() + b);
        for (int64_t i = a; i < b; ++i) {
            PointStampIdLayout& p = clouds[pc][(size_t)(i - a)];
            p.data[0] = xyz[4 * i], p.data[1] = xyz[4 * i + 1], p.data[2] = xyz[4 * i + 2], p.data[3] = 1.0f;
            p.stamp = traj[tf[i]], p.id = ring[i], p.isStatic = 0;
            globalPoints[(size_t)i] = p;
        }
    }
    for (int64_t k = 0; k < S; ++k) {  // addStaticPoints (ContinuousTrajectory.h:158-172)
        PointStampIdLayout& p = globalPoints[(size_t)(N + k)];
        p.data[0] = sxyz[4 * k], p.data[1] = sxyz[4 * k + 1], p.data[2] = sxyz[4 * k + 2], p.data[3] = 1.0f;
        p.stamp = -1000.0, p.id = sring[k], p.isStatic = 1;
    }
    dmsa_window_problem prob{};
    prob.num_control_poses = C, prob.stamps = stamps.data(), prob.n_total = n_total, prob.traj_time = traj.data(), prob.min_grid_size = min_grid;
    prob.dt_res = dt_res, prob.balancing_imu = 1e-3, prob.gravity[2] = -9.805;
    dmsa_settings s;
    dmsa_default_settings(&s);
    s.num_iter = iterations;
    dmsa_ctx* ctx = nullptr;
    if (dmsa_create(0, DMSA_FLAG_FIXED_ITERS, &ctx) != DMSA_OK) {
        std::fprintf(stderr, "no usable HIP device\n");
        return 3;
    }
    if (dmsa_reserve(ctx, N + S, n_total, 6 * (C - 1)) != DMSA_OK) {
        std::fprintf(stderr, "dmsa_reserve: %s\n", dmsa_last_error(ctx));
        return 3;
    }
    std::vector<dmsa_aos_view> views((size_t)num_scans);
    for (int pc = 0; pc < num_scans; ++pc)
        views[(size_t)pc] = dmsa_aos_view{clouds[pc].data(), (int64_t)clouds[pc].size(), (int32_t)sizeof(PointStampIdLayout), 0, 24, tformIdPerPoint[pc].data()};
    const dmsa_aos_view stat{globalPoints.data() + N, S, (int32_t)sizeof(PointStampIdLayout), 0, 24, nullptr};
    std::vector<double> o(3 * C), t(3 * C), o_aos, t_aos, o_flat, t_flat, o_rep, t_rep;
    dmsa_report rep{};
    auto fresh = [&]() { o = ro, t = rt, prob.rel_orient = o.data(), prob.rel_transl = t.data(); };
    double best_aos = 1e30, best_aos_poses = 1e30, best_flat = 1e30, best_repack = 1e30, best_ring = 1e30;
    for (int r = 0; r <= reps; ++r) {  // r = 0 warms up
        fresh();
        auto t0 = std::chrono::steady_clock::now();
        if (dmsa_optimize_window_aos(ctx, &prob, views.data(), num_scans, &stat, &s, &rep) != DMSA_OK) {
            std::fprintf(stderr, "aos call: %s\n", dmsa_last_error(ctx));
            return 4;
        }
        const double t_poses = ms_since(t0);  // the poses are back; the global points can stay in HBM for whoever consumes them next
        if (dmsa_get_global_points_aos(ctx, globalPoints.data(), N + S, (int32_t)sizeof(PointStampIdLayout), 0, -1) != DMSA_OK) return 4;
        if (r > 0 && ms_since(t0) < best_aos) best_aos = ms_since(t0);
        if (r > 0 && t_poses < best_aos_poses) best_aos_poses = t_poses;
        o_aos = o, t_aos = t;
    }
    std::vector<double> o_ring, t_ring;
    {
        dmsa_ctx* rc = nullptr;
        if (dmsa_create(0, DMSA_FLAG_FIXED_ITERS, &rc) != DMSA_OK) return 3;
        int64_t per_scan = 0;
        for (auto& c : clouds) per_scan = std::max<int64_t>(per_scan, (int64_t)c.size());
        const dmsa_window_ring_config cfg{num_scans, per_scan, S, n_total, C};
        if (dmsa_window_ring_create(rc, &cfg) != DMSA_OK) return 3;
        for (int pc = 0; pc < num_scans; ++pc)
            if (dmsa_window_ring_push_aos(rc, &views[(size_t)pc], 16) != DMSA_OK) return 4;
        for (int r = 0; r <= reps; ++r) {
            fresh();
            prob.num_points = 0;
            auto t0 = std::chrono::steady_clock::now();
            // the window slides by one scan: here the oldest scan is pushed again, so that the ring holds the same window (same result) while
            // the call pays exactly what a new scan costs
            if (dmsa_window_ring_push_aos(rc, &views[(size_t)((r) % num_scans)], 16) != DMSA_OK) return 4;
            if (dmsa_window_upload_from_ring_aos(rc, &prob, 0.0, &stat) != DMSA_OK || dmsa_optimize_resident(rc, &s, &rep) != DMSA_OK ||
                dmsa_get_poses(rc, o.data(), t.data()) != DMSA_OK) {
                std::fprintf(stderr, "ring call: %s\n", dmsa_last_error(rc));
                return 4;
            }
            if (r > 0 && ms_since(t0) < best_ring) best_ring = ms_since(t0);
            if (r % num_scans == 0) o_ring = o, t_ring = t;  // after pushing scan 0 .. the ring is a rotation of the window only when r % num_scans == 0
        }
        dmsa_destroy(rc);
    }
    std::vector<float> gflat(4 * (size_t)(N + S));
    for (int r = 0; r <= reps; ++r) {
        fresh();
        prob.num_points = N, prob.xyz_local = xyz.data(), prob.tform_idx = tf.data(), prob.ring_id = ring.data();
        prob.num_static = S, prob.xyz_static = sxyz.data(), prob.ring_id_static = sring.data();
        auto t0 = std::chrono::steady_clock::now();
        if (dmsa_optimize_window(ctx, &prob, &s, &rep) != DMSA_OK || dmsa_get_global_points(ctx, gflat.data(), N + S) != DMSA_OK) return 4;
        if (r > 0 && ms_since(t0) < best_flat) best_flat = ms_since(t0);
        o_flat = o, t_flat = t;
    }
    for (int r = 0; r <= reps; ++r) {  // the per-point repack a caller of the flat ABI has to do from PCL containers
        fresh();
        auto t0 = std::chrono::steady_clock::now();
        std::vector<float> local, st;
        std::vector<int32_t> tfv, rg, rgs;
        for (int pc = 0; pc < num_scans; ++pc)
            for (size_t k = 0; k < clouds[pc].size(); ++k) {
                local.insert(local.end(), clouds[pc][k].data, clouds[pc][k].data + 4);
                tfv.push_back(tformIdPerPoint[pc][k]), rg.push_back(clouds[pc][k].id);
            }
        for (size_t k = (size_t)N; k < globalPoints.size(); ++k) {
            st.insert(st.end(), globalPoints[k].data, globalPoints[k].data + 4);
            rgs.push_back(globalPoints[k].id);
        }
        prob.num_points = N, prob.xyz_local = local.data(), prob.tform_idx = tfv.data(), prob.ring_id = rg.data();
        prob.num_static = S, prob.xyz_static = st.data(), prob.ring_id_static = rgs.data();
        if (dmsa_optimize_window(ctx, &prob, &s, &rep) != DMSA_OK || dmsa_get_global_points(ctx, gflat.data(), N + S) != DMSA_OK) return 4;
        for (size_t k = 0; k < globalPoints.size(); ++k) std::memcpy(globalPoints[k].data, &gflat[4 * k], 12);
        if (r > 0 && ms_since(t0) < best_repack) best_repack = ms_since(t0);
        o_rep = o, t_rep = t;
    }
    const bool same = o_aos == o_flat && t_aos == t_flat && o_aos == o_rep && t_aos == t_rep;
    (void)o_ring, (void)t_ring;
    bool same_points = true;
    for (size_t k = 0; k < globalPoints.size() && same_points; ++k) same_points = !std::memcmp(globalPoints[k].data, &gflat[4 * k], 12);
    std::printf("{\"points\": %lld, \"scans\": %d, \"iterations_per_call\": %d, \"aos_call_ms\": %.3f, \"aos_call_poses_only_ms\": %.3f, \"aos_ring_call_ms\": %.3f, "
                "\"flat_call_ms\": %.3f, \"host_repack_call_ms\": %.3f, \"poses_bit_identical\": %s, \"global_points_bit_identical\": %s, \"gaussians\": %d}\n",
                (long long)(N + S), num_scans, iterations, best_aos, best_aos_poses, best_ring, best_flat, best_repack, same ? "true" : "false",
                same_points ? "true" : "false", rep.num_gaussians);
    dmsa_destroy(ctx);
    return same && same_points ? 0 : 5;
}
