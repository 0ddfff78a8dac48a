// window_ring.cpp — include/dmsa_window_ring.h: the scans of the sliding window resident in HBM (RingBuffer.h:31-88,
// PointCloudBuffer.h:24-49), one scan uploaded per window, the window problem assembled on the device (ContinuousTrajectory.h:228-261).
#include "dmsa_ctx.h"

#include "../../include/dmsa_window_ring.h"

extern "C" {

int dmsa_window_ring_create(dmsa_ctx* ctx, const dmsa_window_ring_config* cfg) {
    if (!ctx || !cfg || cfg->num_scans < 1 || cfg->num_scans > 1024 || cfg->max_points_per_scan < 1 || cfg->max_static_points < 0 || cfg->max_n_total < 2 ||
        cfg->max_control_poses < 3 || cfg->max_control_poses > 64)
        return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    WindowRing& r = ctx->ring;
    r.num_scans = cfg->num_scans, r.cap = cfg->max_points_per_scan, r.head = 0, r.filled = 0;
    r.count.assign((size_t)cfg->num_scans, 0);
    const size_t slots = (size_t)cfg->num_scans * (size_t)cfg->max_points_per_scan;
    HIPCHK(r.xyz.ensure(slots * 16));
    HIPCHK(r.stamp.ensure(slots * 8));
    HIPCHK(r.id.ensure(slots * 4));
    // one pinned staging area for a scan (28 bytes per point) or the static points of a window (20 bytes per point)
    const size_t stage = std::max((size_t)cfg->max_points_per_scan * 28, (size_t)cfg->max_static_points * 20) + 256;
    if (stage > ctx->h_stage_cap) {
        if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
        ctx->h_stage = nullptr, ctx->h_stage_cap = 0;
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_stage), stage, hipHostMallocDefault));
        ctx->h_stage_cap = stage;
    }
    // every buffer whose size depends on the point count, for the full window: nothing is allocated when the windows start to slide
    struct Restore {  // the sizes of an already uploaded problem come back on every exit path, a failed allocation included
        dmsa_ctx* c;
        int64_t n;
        int rows;
        ~Restore() { c->n = n, c->rows = rows; }
    } restore{ctx, ctx->n, ctx->rows};
    ctx->n = (int64_t)slots + cfg->max_static_points, ctx->rows = cfg->max_n_total + 1;
    HIPCHK(ctx->d_local.ensure((size_t)ctx->n * 16 + 16));
    HIPCHK(ctx->d_ring.ensure((size_t)ctx->n * 4 + 16));
    const int rc = alloc_point_buffers(ctx);
    const int P = 6 * (cfg->max_control_poses - 1);
    if (rc == DMSA_OK) {
        HIPCHK(ctx->d_tables.ensure((size_t)(P + 1) * ctx->rows * 48));
        HIPCHK(ctx->d_tablesT.ensure((size_t)(P + 1) * ctx->rows * 48));
        HIPCHK(ctx->d_table0.ensure((size_t)ctx->rows * 48));
        HIPCHK(ctx->d_stamps.ensure((size_t)cfg->max_control_poses * 8));
        HIPCHK(ctx->d_fhw.ensure((size_t)cfg->max_control_poses * 8));
        HIPCHK(ctx->d_trajtime.ensure((size_t)cfg->max_n_total * 8));
        // residual batches: at most one Gaussian per two points and level
        HIPCHK(ctx->d_E.ensure((size_t)(P + 1) * (size_t)(ctx->n / 8 + 4096) * 8));
    }
    return rc;
}

int dmsa_window_ring_push(dmsa_ctx* ctx, const float* xyz_local, const double* stamps, const int32_t* ring_id, int64_t n) {
    if (!ctx || ctx->ring.num_scans == 0 || n < 0 || n > ctx->ring.cap || (n > 0 && (!xyz_local || !stamps || !ring_id))) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    WindowRing& r = ctx->ring;
    const int slot = r.head;
    // RingBuffer::addElem (RingBuffer.h:67-88): the oldest element is overwritten once the buffer is full
    const size_t off = (size_t)slot * (size_t)r.cap;
    if (n > 0) {
        HIPCHK(hipStreamSynchronize(ctx->stream));  // the staging area may still feed the previous copy
        char* st = ctx->h_stage;
        std::memcpy(st, xyz_local, (size_t)n * 16);
        std::memcpy(st + (size_t)n * 16, stamps, (size_t)n * 8);
        std::memcpy(st + (size_t)n * 24, ring_id, (size_t)n * 4);
        HIPCHK(hipMemcpyAsync(r.xyz.as<char>() + off * 16, st, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(r.stamp.as<char>() + off * 8, st + (size_t)n * 16, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(r.id.as<char>() + off * 4, st + (size_t)n * 24, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    r.count[(size_t)slot] = n;
    r.head = (r.head + 1) % r.num_scans;
    r.filled = std::min(r.filled + 1, r.num_scans);
    return DMSA_OK;
}

int dmsa_window_ring_points(dmsa_ctx* ctx, int32_t* scans_out, int64_t* points_out) {
    if (!ctx || ctx->ring.num_scans == 0) return DMSA_ERR_INVALID;
    int64_t total = 0;
    for (int k = 0; k < ctx->ring.filled; ++k) total += ctx->ring.count[(size_t)((ctx->ring.head - ctx->ring.filled + k + 2 * ctx->ring.num_scans) % ctx->ring.num_scans)];
    if (scans_out) *scans_out = ctx->ring.filled;
    if (points_out) *points_out = total;
    return DMSA_OK;
}

int dmsa_window_upload_from_ring(dmsa_ctx* ctx, const dmsa_window_problem* p, double t0) {
    if (!ctx || !p || ctx->ring.num_scans == 0 || p->num_static < 0) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    WindowRing& r = ctx->ring;
    int64_t N = 0;
    CHK(dmsa_window_ring_points(ctx, nullptr, &N));
    if ((p->num_points != 0 && p->num_points != N) || (p->num_static > 0 && (!p->xyz_static || !p->ring_id_static)) || !(p->min_grid_size > 0.0f)) {
        ctx->err = "invalid window problem for the resident ring (point count differs from the ring's, null static arrays or min_grid_size <= 0)";
        return DMSA_ERR_INVALID;
    }
    if (!ctx->win.init(*p)) {
        ctx->err = "invalid window problem (fewer than 3 control poses, coincident stamps, null pose arrays or IMU parameter indices outside the time grid)";
        return DMSA_ERR_INVALID;
    }
    if (ctx->win.ctrl.n > 64) {
        ctx->err = "more than 64 control poses";
        return DMSA_ERR_INVALID;
    }
    ctx->model = MODEL_WINDOW;
    const int64_t S = p->num_static;
    ctx->N = N, ctx->S = S, ctx->n = N + S;
    ctx->rows = p->n_total + 1;
    const size_t n = (size_t)ctx->n;
    HIPCHK(ctx->d_local.ensure(n * 16 + 16));
    HIPCHK(ctx->d_ring.ensure(n * 4 + 16));
    const int C = ctx->win.ctrl.n;
    HIPCHK(ctx->d_stamps.ensure((size_t)C * 8));
    HIPCHK(ctx->d_fhw.ensure((size_t)C * 8));
    HIPCHK(ctx->d_trajtime.ensure((size_t)p->n_total * 8));
    // time grid, control stamps and weights, static points: one pinned staging area, asynchronous copies
    const size_t small = ((size_t)2 * C + (size_t)p->n_total) * 8;
    const size_t need = small + (size_t)S * 20 + 64;
    if (need > ctx->h_stage_cap) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
        ctx->h_stage = nullptr, ctx->h_stage_cap = 0;
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_stage), need + need / 8, hipHostMallocDefault));
        ctx->h_stage_cap = need + need / 8;
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));  // the staging area may still feed an earlier copy
    double* sd = reinterpret_cast<double*>(ctx->h_stage);
    std::memcpy(sd, ctx->win.stamps.data(), (size_t)C * 8);
    std::memcpy(sd + C, ctx->win.fh.w.data(), (size_t)C * 8);
    std::memcpy(sd + 2 * C, ctx->win.traj_time.data(), (size_t)p->n_total * 8);
    HIPCHK(hipMemcpyAsync(ctx->d_stamps.p, sd, (size_t)C * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_fhw.p, sd + C, (size_t)C * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_trajtime.p, sd + 2 * C, (size_t)p->n_total * 8, hipMemcpyHostToDevice, ctx->stream));
    if (S > 0) {
        float* loc = reinterpret_cast<float*>(ctx->h_stage + small);
        int32_t* ring = reinterpret_cast<int32_t*>(ctx->h_stage + small + (size_t)S * 16);
        const int32_t id_row = p->n_total;  // the identity row appended to every pose table
        auto pack = [&](int64_t k0, int64_t k1) {
            for (int64_t k = k0; k < k1; ++k) {
                loc[4 * k] = p->xyz_static[4 * k], loc[4 * k + 1] = p->xyz_static[4 * k + 1], loc[4 * k + 2] = p->xyz_static[4 * k + 2];
                std::memcpy(&loc[4 * k + 3], &id_row, 4);
                ring[k] = p->ring_id_static[k];
            }
        };
        if (S < 131072)
            pack(0, S);
        else
            workers(ctx).run_all([&](int t, int nt) { pack(S * t / nt, S * (t + 1) / nt); });
        HIPCHK(hipMemcpyAsync(ctx->d_local.as<float4>() + N, loc, (size_t)S * 16, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->d_ring.as<int32_t>() + N, ring, (size_t)S * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    // registerPcBuffer (:240-260) on the resident scans, oldest first (RingBuffer::at is chronological)
    int64_t at = 0;
    for (int k = 0; k < r.filled; ++k) {
        const int slot = (r.head - r.filled + k + 2 * r.num_scans) % r.num_scans;
        const int64_t cnt = r.count[(size_t)slot];
        const size_t off = (size_t)slot * (size_t)r.cap;
        launch_ring_assemble(r.xyz.as<float4>() + off, r.stamp.as<double>() + off, r.id.as<int32_t>() + off, cnt, t0, ctx->d_trajtime.as<double>(), p->n_total,
                             ctx->d_local.as<float4>() + at, ctx->d_ring.as<int32_t>() + at, ctx->stream);
        at += cnt;
    }
    HIPCHK(hipGetLastError());
    ctx->win.ctrl.relative_to_global();
    ctx->min_grid_size = p->min_grid_size;
    CHK(upload_loop_model(ctx));
    return upload_common(ctx);
}

}  // extern "C"
