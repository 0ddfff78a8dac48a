// aos_upload.cpp — include/dmsa_aos.h: the reference's point containers (pcl::PointCloud<PointStampId>, 32-byte points,
// PointStampId.h:33-45; pcl::PointCloud<pcl::PointNormal>, 48-byte points, KeyframeData.h:20) handed over as they lie in memory.
// Host side: the worker threads move what the device needs into pinned staging -- of a window point the 16 bytes x, y, z, id plus its
// tform index, so that a scan crosses PCIe as 20 bytes per point like the flat arrays (measured: the whole 32-byte points cost 0.7 ms more
// per call than the gather saves) -- and the pack kernels (static_kernels.hip) write the layouts of dmsa_ctx.h.
#include "dmsa_ctx.h"

#include "../../include/dmsa_window_ring.h"

namespace {

bool view_ok(const dmsa_aos_view& v, int min_aux_bytes) {
    if (v.count < 0 || (v.count > 0 && !v.base)) return false;
    if (v.stride < 16 || (v.stride & 3) != 0 || v.xyz_offset < 0 || (v.xyz_offset & 3) != 0 || v.xyz_offset + 12 > v.stride) return false;
    return v.aux_offset >= 0 && (v.aux_offset & 3) == 0 && v.aux_offset + min_aux_bytes <= v.stride;
}

// bytes [0, total) of a cloud copied by the context's worker threads
void copy_parallel(dmsa_ctx* ctx, char* dst, const char* src, size_t total) {
    if (total < (size_t)4 << 20) {
        std::memcpy(dst, src, total);
        return;
    }
    workers(ctx).run_all([&](int t, int nt) {
        const size_t a = total * (size_t)t / (size_t)nt, b = total * (size_t)(t + 1) / (size_t)nt;
        std::memcpy(dst + a, src + a, b - a);
    });
}

}  // namespace

extern "C" {

int dmsa_window_upload_aos(dmsa_ctx* ctx, const dmsa_window_problem* p, const dmsa_aos_view* clouds, int32_t num_clouds, const dmsa_aos_view* static_points) {
    if (!ctx || !p || num_clouds < 0 || (num_clouds > 0 && !clouds)) return DMSA_ERR_INVALID;
    int64_t N = 0;
    size_t raw_bytes = 0;
    for (int c = 0; c < num_clouds; ++c) {
        if (!view_ok(clouds[c], 4) || (clouds[c].count > 0 && !clouds[c].index)) {
            ctx->err = "invalid window cloud view (stride / offsets not multiples of four, fields outside the point, or no tform indices)";
            return DMSA_ERR_INVALID;
        }
        N += clouds[c].count, raw_bytes += (size_t)clouds[c].count * 16;
    }
    const int64_t S = static_points ? static_points->count : 0;
    if (S < 0 || (S > 0 && !view_ok(*static_points, 4))) {
        ctx->err = "invalid static point view";
        return DMSA_ERR_INVALID;
    }
    if (S > 0) raw_bytes += (size_t)S * 16;
    CHK(window_upload_begin(ctx, p, N, S));
    // staging: [x y z id of every window point | of the static points | tform indices]: ONE pass of the worker threads over all clouds, two
    // DMAs, one pack kernel for the window points and one for the static points
    const size_t idx_off = (raw_bytes + 15) & ~(size_t)15;
    CHK(ensure_stage(ctx, idx_off + (size_t)N * 4 + 64));
    HIPCHK(ctx->d_aos_raw.ensure(idx_off + 64));
    HIPCHK(ctx->d_aos_idx.ensure((size_t)N * 4 + 64));
    int32_t* d_bad = ctx->d_aos_idx.as<int32_t>() + N;  // one flag word behind the indices
    HIPCHK(hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    char* st = ctx->h_stage;
    std::vector<int64_t> first((size_t)num_clouds + 2, 0);  // prefix of the point counts: clouds, then the static points
    for (int c = 0; c < num_clouds; ++c) first[(size_t)c + 1] = first[(size_t)c] + clouds[c].count;
    first[(size_t)num_clouds + 1] = N + S;
    auto gather = [&](int64_t a, int64_t b) {  // points [a, b) of the concatenation
        int c = (int)(std::upper_bound(first.begin(), first.end(), a) - first.begin()) - 1;
        for (int64_t i = a; i < b;) {
            while (c <= num_clouds && first[(size_t)c + 1] <= i) ++c;
            const dmsa_aos_view& v = c == num_clouds ? *static_points : clouds[c];
            const int64_t end = std::min(b, first[(size_t)c + 1]);
            const char* src = static_cast<const char*>(v.base);
            for (; i < end; ++i) {
                const int64_t k = i - first[(size_t)c];
                const char* pt = src + (size_t)k * v.stride;
                std::memcpy(st + (size_t)i * 16, pt + v.xyz_offset, 12);
                std::memcpy(st + (size_t)i * 16 + 12, pt + v.aux_offset, 4);
                if (c < num_clouds) std::memcpy(st + idx_off + (size_t)i * 4, v.index + k, 4);
            }
        }
    };
    const int64_t total = N + S;
    // The worker threads gather a slice while the DMA engine moves the slice before it.  (At the bench size the 30 MB over the bus bound the
    // upload either way: 1.8 ms with and without the overlap.)
    const int slices = total < 262144 ? 1 : 8;
    for (int sl = 0; sl < slices; ++sl) {
        const int64_t a = total * sl / slices, b = total * (sl + 1) / slices;
        if (b - a < 131072)
            gather(a, b);
        else
            workers(ctx).run_all([&](int t, int nt) { gather(a + (b - a) * t / nt, a + (b - a) * (t + 1) / nt); });
        HIPCHK(hipMemcpyAsync(ctx->d_aos_raw.as<char>() + (size_t)a * 16, st + (size_t)a * 16, (size_t)(b - a) * 16, hipMemcpyHostToDevice, ctx->stream));
        const int64_t ib = std::min(b, N);
        if (a < ib)
            HIPCHK(hipMemcpyAsync(ctx->d_aos_idx.as<char>() + (size_t)a * 4, st + idx_off + (size_t)a * 4, (size_t)(ib - a) * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    launch_pack_aos_window(ctx->d_aos_raw.as<uint8_t>(), N, 16, 0, 12, ctx->d_aos_idx.as<int32_t>(), p->n_total, p->n_total, ctx->d_local.as<float4>(),
                           ctx->d_ring.as<int32_t>(), d_bad, ctx->stream);
    launch_pack_aos_window(ctx->d_aos_raw.as<uint8_t>() + (size_t)N * 16, S, 16, 0, 12, nullptr, p->n_total, p->n_total, ctx->d_local.as<float4>() + N,
                           ctx->d_ring.as<int32_t>() + N, d_bad, ctx->stream);
    HIPCHK(hipGetLastError());
    int32_t bad = 0;
    HIPCHK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));  // the staging area is reused by the next upload
    if (bad) {
        ctx->err = "tform_idx out of range";
        return DMSA_ERR_INVALID;
    }
    return window_upload_finish(ctx, p);
}

int dmsa_keyframes_upload_aos(dmsa_ctx* ctx, const dmsa_keyframe_problem* p, const dmsa_aos_view* frames, int32_t num_frames) {
    if (!ctx || !p || !frames || num_frames != p->num_frames || num_frames < 2) return DMSA_ERR_INVALID;
    int64_t n = 0;
    size_t raw_bytes = 0;
    for (int k = 0; k < num_frames; ++k) {
        if (!view_ok(frames[k], 16) || (frames[k].count > 0 && !frames[k].index)) {
            ctx->err = "invalid keyframe cloud view (stride / offsets not multiples of four, fields outside the point, or no ring ids)";
            return DMSA_ERR_INVALID;
        }
        n += frames[k].count, raw_bytes += (size_t)frames[k].count * frames[k].stride;
    }
    CHK(keyframes_upload_begin(ctx, p, n));
    const size_t idx_off = (raw_bytes + 15) & ~(size_t)15;
    CHK(ensure_stage(ctx, idx_off + (size_t)n * 4 + 64));
    HIPCHK(ctx->d_aos_raw.ensure(idx_off + 64));
    char* st = ctx->h_stage;
    size_t at = 0;
    int64_t first = 0;
    for (int k = 0; k < num_frames; ++k) {
        const dmsa_aos_view& v = frames[k];
        if (v.count == 0) continue;
        const size_t bytes = (size_t)v.count * v.stride;
        copy_parallel(ctx, st + at, static_cast<const char*>(v.base), bytes);
        HIPCHK(hipMemcpyAsync(ctx->d_aos_raw.as<char>() + at, st + at, bytes, hipMemcpyHostToDevice, ctx->stream));
        std::memcpy(st + idx_off + (size_t)first * 4, v.index, (size_t)v.count * 4);  // ring ids go up as they are
        HIPCHK(hipMemcpyAsync(ctx->d_ring.as<int32_t>() + first, st + idx_off + (size_t)first * 4, (size_t)v.count * 4, hipMemcpyHostToDevice, ctx->stream));
        launch_pack_aos_keyframe(ctx->d_aos_raw.as<uint8_t>() + at, v.count, v.stride, v.xyz_offset, v.aux_offset, k, ctx->d_local.as<float4>() + first,
                                 ctx->d_nlocal.as<float4>() + first, ctx->stream);
        at += bytes, first += v.count;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return keyframes_upload_finish(ctx, p);
}

int dmsa_optimize_window_aos(dmsa_ctx* ctx, dmsa_window_problem* p, const dmsa_aos_view* clouds, int32_t num_clouds, const dmsa_aos_view* static_points,
                             const dmsa_settings* s, dmsa_report* rep) {
    if (!ctx || !p || !s) return DMSA_ERR_INVALID;
    CHK(dmsa_window_upload_aos(ctx, p, clouds, num_clouds, static_points));
    CHK(optimize(ctx, *s, rep));
    write_back_poses(ctx->win.ctrl, p->rel_orient, p->rel_transl);
    return DMSA_OK;
}

int dmsa_optimize_keyframes_aos(dmsa_ctx* ctx, dmsa_keyframe_problem* p, const dmsa_aos_view* frames, int32_t num_frames, const dmsa_settings* s, dmsa_report* rep) {
    if (!ctx || !p || !s) return DMSA_ERR_INVALID;
    CHK(dmsa_keyframes_upload_aos(ctx, p, frames, num_frames));
    CHK(optimize(ctx, *s, rep));
    write_back_poses(ctx->key.frames, p->rel_orient, p->rel_transl);
    return DMSA_OK;
}

int dmsa_get_global_points_aos(dmsa_ctx* ctx, void* base, int64_t count, int32_t stride, int32_t xyz_offset, int32_t normal_offset) {
    if (!ctx || ctx->model == MODEL_NONE || !base || count < ctx->n || stride < 16 || (stride & 3) || xyz_offset < 0 || xyz_offset + 12 > stride ||
        (normal_offset >= 0 && (normal_offset + 12 > stride || ctx->model != MODEL_KEYFRAMES)))
        return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    const size_t n = (size_t)ctx->n, per = normal_offset >= 0 ? 32 : 16;
    CHK(ensure_stage(ctx, n * per + 64));
    float* xyz = reinterpret_cast<float*>(ctx->h_stage);
    float* nrm = xyz + 4 * n;
    // Pieces of <= 2^18 points come down one after the other; the worker threads scatter a piece into the caller's container while the
    // DMA engine fetches the next ones.
    struct Piece {
        size_t first, count;   // points
        const float* src;      // staging
        int field;             // byte offset of the three floats inside a point of the container
        hipEvent_t done;
    };
    std::vector<Piece> pieces;
    auto fetch = [&](float* dst, const void* dev, size_t first, size_t count, int field) -> int {
        constexpr size_t kPiece = (size_t)1 << 18;
        for (size_t a = 0; a < count; a += kPiece) {
            const size_t c = std::min(kPiece, count - a);
            HIPCHK(hipMemcpyAsync(dst + 4 * (first + a), static_cast<const char*>(dev) + a * 16, c * 16, hipMemcpyDeviceToHost, ctx->stream));
            Piece pc{first + a, c, dst, field, nullptr};
            HIPCHK(hipEventCreateWithFlags(&pc.done, hipEventDisableTiming));
            pieces.push_back(pc);
            HIPCHK(hipEventRecord(pc.done, ctx->stream));
        }
        return DMSA_OK;
    };
    int rc = fetch(xyz, ctx->d_global.p, 0, (size_t)ctx->N, xyz_offset);
    if (rc == DMSA_OK && ctx->model == MODEL_WINDOW && ctx->S > 0)  // static points are not moved by updateGlobalPoints; they sit (de-centralised again) in the local array
        rc = fetch(xyz, ctx->d_local.as<float4>() + ctx->N, (size_t)ctx->N, (size_t)ctx->S, xyz_offset);
    if (rc == DMSA_OK && normal_offset >= 0) rc = fetch(nrm, ctx->d_nglobal.p, 0, n, normal_offset);
    char* out = static_cast<char*>(base);
    for (Piece& pc : pieces) {
        if (rc == DMSA_OK && hipEventSynchronize(pc.done) != hipSuccess) ctx->err = "hipEventSynchronize failed", rc = DMSA_ERR_HIP;
        if (rc == DMSA_OK) {
            auto scatter = [&](size_t a, size_t b) {
                for (size_t i = a; i < b; ++i) std::memcpy(out + i * (size_t)stride + pc.field, pc.src + 4 * i, 12);
            };
            if (pc.count < 65536)
                scatter(pc.first, pc.first + pc.count);
            else
                workers(ctx).run_all([&](int t, int nt) {
                    scatter(pc.first + pc.count * (size_t)t / (size_t)nt, pc.first + pc.count * (size_t)(t + 1) / (size_t)nt);
                });
        }
        (void)hipEventDestroy(pc.done);
    }
    if (rc != DMSA_OK) (void)hipStreamSynchronize(ctx->stream);  // nothing may still write into the staging area
    return rc;
}

int dmsa_window_ring_push_aos(dmsa_ctx* ctx, const dmsa_aos_view* scan, int32_t stamp_offset) {
    if (!ctx || !scan || ctx->ring.num_scans == 0 || !view_ok(*scan, 4) || scan->count > ctx->ring.cap || stamp_offset < 0 || (stamp_offset & 3) != 0 ||
        stamp_offset + 8 > scan->stride)
        return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    WindowRing& r = ctx->ring;
    const int slot = r.head;  // RingBuffer::addElem (RingBuffer.h:67-88): the oldest element is overwritten once the buffer is full
    const size_t off = (size_t)slot * (size_t)r.cap, bytes = (size_t)scan->count * scan->stride;
    if (scan->count > 0) {
        CHK(ensure_stage(ctx, bytes + 64));
        HIPCHK(ctx->d_aos_raw.ensure(bytes + 64));
        HIPCHK(hipStreamSynchronize(ctx->stream));  // the staging area may still feed the previous copy
        copy_parallel(ctx, ctx->h_stage, static_cast<const char*>(scan->base), bytes);
        HIPCHK(hipMemcpyAsync(ctx->d_aos_raw.p, ctx->h_stage, bytes, hipMemcpyHostToDevice, ctx->stream));
        launch_unpack_ring_scan(ctx->d_aos_raw.as<uint8_t>(), scan->count, scan->stride, scan->xyz_offset, stamp_offset, scan->aux_offset, r.xyz.as<float4>() + off,
                                r.stamp.as<double>() + off, r.id.as<int32_t>() + off, ctx->stream);
        HIPCHK(hipGetLastError());
    }
    r.count[(size_t)slot] = scan->count;
    r.head = (r.head + 1) % r.num_scans;
    r.filled = std::min(r.filled + 1, r.num_scans);
    return DMSA_OK;
}

int dmsa_window_upload_from_ring_aos(dmsa_ctx* ctx, const dmsa_window_problem* p, double t0, const dmsa_aos_view* static_points) {
    if (!ctx || !p) return DMSA_ERR_INVALID;
    const int64_t S = static_points ? static_points->count : 0;
    if (S < 0 || (S > 0 && !view_ok(*static_points, 4))) return DMSA_ERR_INVALID;
    // the static tail of globalPoints as flat arrays in pinned staging (x y z 1 | id): what dmsa_window_upload_from_ring reads
    dmsa_window_problem q = *p;
    q.num_static = S, q.xyz_static = nullptr, q.ring_id_static = nullptr;
    q.num_points = 0, q.xyz_local = nullptr, q.tform_idx = nullptr, q.ring_id = nullptr;  // the window points come from the ring: p's point arrays are ignored
    std::vector<float> xyz;
    std::vector<int32_t> ids;
    if (S > 0) {
        xyz.resize((size_t)S * 4), ids.resize((size_t)S);
        const char* src = static_cast<const char*>(static_points->base);
        auto run = [&](int64_t a, int64_t b) {
            for (int64_t i = a; i < b; ++i) {
                const char* pt = src + (size_t)i * static_points->stride;
                std::memcpy(&xyz[4 * (size_t)i], pt + static_points->xyz_offset, 12);
                xyz[4 * (size_t)i + 3] = 1.0f;
                std::memcpy(&ids[(size_t)i], pt + static_points->aux_offset, 4);
            }
        };
        if (S < 131072)
            run(0, S);
        else
            workers(ctx).run_all([&](int t, int nt) { run(S * t / nt, S * (t + 1) / nt); });
        q.xyz_static = xyz.data(), q.ring_id_static = ids.data();
    }
    return dmsa_window_upload_from_ring(ctx, &q, t0);
}

int dmsa_reserve(dmsa_ctx* ctx, int64_t max_points, int32_t max_table_rows, int32_t max_params) {
    if (!ctx || max_points < 1 || max_table_rows < 1 || max_params < 6) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    struct Restore {  // a resident problem keeps its own sizes
        dmsa_ctx* c;
        int64_t n;
        int rows;
        ~Restore() { c->n = n, c->rows = rows; }
    } restore{ctx, ctx->n, ctx->rows};
    ctx->n = max_points, ctx->rows = max_table_rows + 1;
    const size_t n = (size_t)max_points;
    const int P = max_params;
    HIPCHK(ctx->d_local.ensure(n * 16 + 16));
    HIPCHK(ctx->d_nlocal.ensure(n * 16 + 16));
    HIPCHK(ctx->d_nglobal.ensure(n * 16 + 16));
    HIPCHK(ctx->d_ring.ensure(n * 4 + 16));
    HIPCHK(ctx->d_aos_raw.ensure(n * 48 + 64));
    HIPCHK(ctx->d_aos_idx.ensure(n * 4 + 64));
    HIPCHK(ctx->d_static_keep.ensure(n * 16));
    CHK(ensure_stage(ctx, n * 52 + 128));
    CHK(alloc_point_buffers(ctx));
    HIPCHK(ctx->d_tables.ensure((size_t)(P + 1) * ctx->rows * 48));
    HIPCHK(ctx->d_tablesT.ensure((size_t)(P + 1) * ctx->rows * 48));
    HIPCHK(ctx->d_table0.ensure((size_t)ctx->rows * 48));
    HIPCHK(ctx->d_trajtime.ensure((size_t)max_table_rows * 8));
    HIPCHK(ctx->d_E.ensure((size_t)(P + 1) * (size_t)(n / 8 + 4096) * 8));  // residual batches: far fewer Gaussians than points in practice (grown on demand otherwise)
    return DMSA_OK;
}

}  // extern "C"
