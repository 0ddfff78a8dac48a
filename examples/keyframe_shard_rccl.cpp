// keyframe_shard_rccl.cpp — the sharded keyframe pass (BASELINE.json config 4) without Python: one process per GPU, the C ABI of
// libdmsa_hip.so for the optimiser and the map seam, ONE ncclAllGather (RCCL over xGMI) of relative poses.
//
//   keyframe_shard_rccl <map.bin> <poses_out.bin> [ranks] [iterations]
//
// The parent forks `ranks` children (default: one per visible GPU) BEFORE any HIP call; rank r binds GPU r, cuts its neighbourhood
// with dmsa_neighbourhood_ranges / dmsa_submap_poses (MapManagement::getSubmap, MapManagement.h:254-276), runs
// dmsa_optimize_keyframes (keyframeMapOptimizer.optimizeSet, DmsaSlam.h:228) on it, applies dmsa_update_poses_from_submap
// (MapManagement.h:278-288) to its own columns and all-gathers them; every rank ends with the same map, rank 0 writes it.
// There is no collective inside the iterations: the neighbourhoods share one boundary frame and own disjoint pose columns.
// Input / output formats: dmsa_lidar_slam_amd/dump.py.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/dmsa_hip.h"
#include "../include/dmsa_keyframe_map.h"

namespace {

struct KeyframeMap {
    int32_t F = 0, use_gravity = 0;
    int64_t n = 0;
    float min_grid_size = 0.f;
    double gravity[3], cov_grav_inv[9], balancing_grav = 1.0;
    std::vector<double> rel_o, rel_t, meas_grav;
    std::vector<int64_t> off;
    std::vector<float> xyz, nrm;
    std::vector<int32_t> ring, plausible;
};

template <class T>
bool rd(FILE* f, T* p, size_t count) { return std::fread(p, sizeof(T), count, f) == count; }

bool load(const char* path, KeyframeMap& m) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    char magic[8];
    float pad;
    bool ok = rd(f, magic, 8) && std::memcmp(magic, "DMSAKF01", 8) == 0 && rd(f, &m.F, 1) && rd(f, &m.use_gravity, 1) && rd(f, &m.n, 1) &&
              rd(f, &m.min_grid_size, 1) && rd(f, &pad, 1) && rd(f, m.gravity, 3) && rd(f, m.cov_grav_inv, 9) && rd(f, &m.balancing_grav, 1);
    if (ok && m.F >= 2 && m.n >= 0) {
        m.rel_o.resize(3 * (size_t)m.F), m.rel_t.resize(3 * (size_t)m.F), m.off.resize((size_t)m.F + 1), m.xyz.resize(4 * (size_t)m.n), m.nrm.resize(4 * (size_t)m.n);
        m.ring.resize((size_t)m.n), m.meas_grav.resize(3 * (size_t)m.F), m.plausible.resize((size_t)m.F);
        ok = rd(f, m.rel_o.data(), m.rel_o.size()) && rd(f, m.rel_t.data(), m.rel_t.size()) && rd(f, m.off.data(), m.off.size()) &&
             rd(f, m.xyz.data(), m.xyz.size()) && rd(f, m.nrm.data(), m.nrm.size()) && rd(f, m.ring.data(), m.ring.size()) &&
             rd(f, m.meas_grav.data(), m.meas_grav.size()) && rd(f, m.plausible.data(), m.plausible.size());
    } else {
        ok = false;
    }
    std::fclose(f);
    return ok;
}

#define CHECK(expr, what)                                                          \
    do {                                                                           \
        if (!(expr)) {                                                             \
            std::fprintf(stderr, "[rank %d] %s failed (%s:%d)\n", rank, what, __FILE__, __LINE__); \
            return 1;                                                              \
        }                                                                          \
    } while (0)

struct Rendezvous {  // shared anonymous mapping: rank 0 creates the RCCL id AFTER the fork (no HIP / RCCL state is ever inherited)
    volatile int ready;
    ncclUniqueId id;
};

int run_rank(int rank, int world, Rendezvous* rv, const KeyframeMap& full, int iterations, const char* out_path) {
    if (rank == 0) {
        CHECK(ncclGetUniqueId(&rv->id) == ncclSuccess, "ncclGetUniqueId");
        __sync_synchronize();
        rv->ready = 1;
    } else {
        for (long spins = 0; !rv->ready; ++spins) {
            usleep(200);
            CHECK(spins < 150000, "waiting for rank 0's RCCL id (30 s)");
        }
        __sync_synchronize();
    }
    const ncclUniqueId id = rv->id;
    int ndev = 0;
    CHECK(hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0, "hipGetDeviceCount");
    const int dev = rank % ndev;
    CHECK(hipSetDevice(dev) == hipSuccess, "hipSetDevice");
    ncclComm_t comm;
    CHECK(ncclCommInitRank(&comm, world, id, rank) == ncclSuccess, "ncclCommInitRank");
    int comm_ranks = 0;
    CHECK(ncclCommCount(comm, &comm_ranks) == ncclSuccess && comm_ranks == world, "ncclCommCount");
    if (rank == 0) std::printf("[rccl] ranks=%d visible_gpus=%d\n", comm_ranks, ndev);  // the world size the communicator really has
    hipStream_t stream;
    CHECK(hipStreamCreate(&stream) == hipSuccess, "hipStreamCreate");

    std::vector<int32_t> from((size_t)world), to((size_t)world);
    CHECK(dmsa_neighbourhood_ranges(full.F, world, from.data(), to.data()) == DMSA_OK, "dmsa_neighbourhood_ranges");
    const int f0 = from[(size_t)rank], f1 = to[(size_t)rank], nf = f1 - f0 + 1;
    std::vector<double> sub_o(3 * (size_t)nf), sub_t(3 * (size_t)nf);
    CHECK(dmsa_submap_poses(full.F, full.rel_o.data(), full.rel_t.data(), f0, f1, sub_o.data(), sub_t.data(), nullptr, nullptr) == DMSA_OK, "dmsa_submap_poses");
    const int64_t a = full.off[(size_t)f0];
    std::vector<int64_t> off((size_t)nf + 1);
    for (int k = 0; k <= nf; ++k) off[(size_t)k] = full.off[(size_t)(f0 + k)] - a;

    dmsa_keyframe_problem p;
    std::memset(&p, 0, sizeof(p));
    p.num_frames = nf, p.rel_orient = sub_o.data(), p.rel_transl = sub_t.data(), p.frame_offset = off.data();
    p.xyz_local = full.xyz.data() + 4 * a, p.normal_local = full.nrm.data() + 4 * a, p.ring_id = full.ring.data() + a;
    p.min_grid_size = full.min_grid_size, p.use_gravity = full.use_gravity, p.use_odometry = 0;
    std::memcpy(p.gravity, full.gravity, sizeof(p.gravity)), std::memcpy(p.cov_grav_inv, full.cov_grav_inv, sizeof(p.cov_grav_inv));
    p.balancing_grav = full.balancing_grav, p.balancing_odom = 1000.0;
    p.measured_gravity = full.meas_grav.data() + 3 * (size_t)f0, p.gravity_plausible = full.plausible.data() + f0;

    dmsa_ctx* ctx = nullptr;
    CHECK(dmsa_create(dev, DMSA_FLAG_FIXED_ITERS, &ctx) == DMSA_OK, "dmsa_create");
    dmsa_settings s;  // optimSettingsMap (DmsaSlam.h:91-99)
    dmsa_default_settings(&s);
    s.num_iter = iterations, s.epsilon = 1e-4, s.step_length_optim = 0.2, s.max_step = 0.01, s.gauss_split = 1, s.min_num_points_per_set = 10;
    dmsa_report rep;
    CHECK(dmsa_keyframes_upload(ctx, &p) == DMSA_OK, "dmsa_keyframes_upload");  // points resident before the timed region
    // barrier = a first, tiny all-reduce (also sets up the xGMI peer connections)
    double* d_buf = nullptr;
    int width = 0;
    for (int i = 0; i < world; ++i) width = to[(size_t)i] - from[(size_t)i] > width ? to[(size_t)i] - from[(size_t)i] : width;
    const size_t slot = (size_t)width * 6;
    CHECK(hipMalloc(reinterpret_cast<void**>(&d_buf), (size_t)(world + 1) * slot * 8) == hipSuccess, "hipMalloc");
    CHECK(hipMemsetAsync(d_buf, 0, (size_t)(world + 1) * slot * 8, stream) == hipSuccess, "hipMemset");
    CHECK(ncclAllReduce(d_buf, d_buf, 1, ncclDouble, ncclSum, comm, stream) == ncclSuccess && hipStreamSynchronize(stream) == hipSuccess, "warm-up all-reduce");

    const auto t0 = std::chrono::steady_clock::now();
    CHECK(dmsa_optimize_resident(ctx, &s, &rep) == DMSA_OK, "dmsa_optimize_resident");
    CHECK(dmsa_get_poses(ctx, sub_o.data(), sub_t.data()) == DMSA_OK, "dmsa_get_poses");
    std::vector<double> map_o(full.rel_o), map_t(full.rel_t);
    CHECK(dmsa_update_poses_from_submap(full.F, map_o.data(), map_t.data(), f0, f1, sub_o.data(), sub_t.data()) == DMSA_OK, "dmsa_update_poses_from_submap");
    std::vector<double> mine(slot, 0.0), all((size_t)world * slot);
    for (int k = 1; k < nf; ++k)
        for (int c = 0; c < 3; ++c) mine[(size_t)(k - 1) * 6 + c] = map_o[3 * (size_t)(f0 + k) + c], mine[(size_t)(k - 1) * 6 + 3 + c] = map_t[3 * (size_t)(f0 + k) + c];
    CHECK(hipMemcpyAsync(d_buf, mine.data(), slot * 8, hipMemcpyHostToDevice, stream) == hipSuccess, "H2D");
    CHECK(ncclAllGather(d_buf, d_buf + slot, slot, ncclDouble, comm, stream) == ncclSuccess, "ncclAllGather");  // the ONE exchange step
    CHECK(hipMemcpyAsync(all.data(), d_buf + slot, (size_t)world * slot * 8, hipMemcpyDeviceToHost, stream) == hipSuccess, "D2H");
    CHECK(hipStreamSynchronize(stream) == hipSuccess, "sync");
    for (int i = 0; i < world; ++i)  // disjoint columns: every rank ends with the same map
        for (int k = 1; k <= to[(size_t)i] - from[(size_t)i]; ++k)
            for (int c = 0; c < 3; ++c) {
                map_o[3 * (size_t)(from[(size_t)i] + k) + c] = all[(size_t)i * slot + (size_t)(k - 1) * 6 + c];
                map_t[3 * (size_t)(from[(size_t)i] + k) + c] = all[(size_t)i * slot + (size_t)(k - 1) * 6 + 3 + c];
            }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("[rank %d/%d gpu %d] frames %d..%d  points %lld  P %d  iterations %d  %.3f ms  (%.1f it/s on this rank)  M %d\n", rank, world, dev, f0, f1,
                (long long)off[(size_t)nf], 6 * (nf - 1), rep.iterations, 1e3 * sec, rep.iterations / sec, rep.num_gaussians);
    if (rank == 0 && out_path) {
        FILE* f = std::fopen(out_path, "wb");
        CHECK(f != nullptr, "open output");
        const int32_t hdr[2] = {full.F, 0};
        std::fwrite("DMSAPO01", 1, 8, f), std::fwrite(hdr, 4, 2, f);
        std::fwrite(map_o.data(), 8, map_o.size(), f), std::fwrite(map_t.data(), 8, map_t.size(), f);
        std::fclose(f);
    }
    dmsa_destroy(ctx);
    (void)hipFree(d_buf);
    ncclCommDestroy(comm);
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <map.bin> <poses_out.bin> [ranks] [iterations]\n", argv[0]);
        return 2;
    }
    KeyframeMap full;
    if (!load(argv[1], full)) {
        std::fprintf(stderr, "cannot read %s\n", argv[1]);
        return 2;
    }
    int world = argc > 3 ? std::atoi(argv[3]) : 0;
    const int iterations = argc > 4 ? std::atoi(argv[4]) : 10;
    auto* rv = static_cast<Rendezvous*>(mmap(nullptr, sizeof(Rendezvous), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0));
    if (rv == MAP_FAILED) {
        std::fprintf(stderr, "mmap failed\n");
        return 1;
    }
    rv->ready = 0;
    if (world <= 0) {
        const char* vis = std::getenv("DMSA_RANKS");
        world = vis ? std::atoi(vis) : 1;
    }
    if (full.F < world + 1) {
        std::fprintf(stderr, "%d keyframes cannot be cut into %d neighbourhoods\n", full.F, world);
        return 2;
    }
    std::vector<pid_t> kids;
    for (int r = 0; r < world; ++r) {
        const pid_t pid = fork();
        if (pid == 0) return run_rank(r, world, rv, full, iterations, argv[2]);
        kids.push_back(pid);
    }
    int bad = 0;
    for (pid_t k : kids) {
        int st = 0;
        waitpid(k, &st, 0);
        bad += !(WIFEXITED(st) && WEXITSTATUS(st) == 0);
    }
    return bad ? 1 : 0;
}
