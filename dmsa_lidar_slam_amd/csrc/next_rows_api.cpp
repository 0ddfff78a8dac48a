// next_rows_api.cpp — the C ABI of the rows around the hot path: include/dmsa_static_points.h, dmsa_window_setup.h (device
// part), dmsa_wire_formats.h (device part), dmsa_keyframe_cloud.h.  Kernels in static_kernels.hip.
#include "dmsa_ctx.h"

// ---- include/dmsa_static_points.h ---------------------------------------------------------------------------------------
namespace {

StaticState* sp_state(dmsa_ctx* ctx) {
    if (!ctx->sp) ctx->sp = new (std::nothrow) StaticState();
    return ctx->sp;
}

// Uniform cell grid over `n` host points (cells of 1.001 * radius): bounds -> [sync] -> codes -> radix sort -> sorted copies + hash of
// the occupied cells.  Leaves the grid in sp->grid / table / pts_sorted / code_s.
int sp_build_grid_device(dmsa_ctx* ctx, int64_t n, float radius);
// n points into a device buffer: from the host, or (host pointer NULL) the first n global points of the uploaded problem as the
// last dmsa_transform_points / optimizeSet left them -- the window cloud never leaves HBM between the hot path and the steps around it
int sp_stage_points(dmsa_ctx* ctx, void* dst, const float* host_xyz, int64_t n) {
    if (host_xyz) {
        HIPCHK(hipMemcpyAsync(dst, host_xyz, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
        return DMSA_OK;
    }
    if (ctx->model == MODEL_NONE || n > ctx->n || !ctx->d_global.p) return DMSA_ERR_INVALID;
    HIPCHK(hipMemcpyAsync(dst, ctx->d_global.p, (size_t)n * 16, hipMemcpyDeviceToDevice, ctx->stream));
    return DMSA_OK;
}
int sp_build_grid(dmsa_ctx* ctx, const float* cloud_xyz, int64_t n, float radius) {
    StaticState* sp = sp_state(ctx);
    if (!sp) return DMSA_ERR_NOMEM;
    if (!(radius > 0.0f) || n < 0 || n > (int64_t)0x7FFFFFF0) return DMSA_ERR_INVALID;
    sp->n_cloud = n;
    if (n == 0) return DMSA_OK;
    HIPCHK(sp->cloud.ensure((size_t)n * 16));
    CHK(sp_stage_points(ctx, sp->cloud.p, cloud_xyz, n));
    return sp_build_grid_device(ctx, n, radius);
}
// the same on a cloud that already sits in sp->cloud
int sp_build_grid_device(dmsa_ctx* ctx, int64_t n, float radius) {
    StaticState* sp = ctx->sp;
    if (!(radius > 0.0f) || n < 0 || n > (int64_t)0x7FFFFFF0) return DMSA_ERR_INVALID;
    sp->n_cloud = n;
    if (n == 0) return DMSA_OK;
    HIPCHK(sp->small.ensure(256));
    CloudBounds* d_b = sp->small.as<CloudBounds>();
    launch_cloud_bounds_init(d_b, ctx->stream);
    launch_cloud_bounds(sp->cloud.as<float4>(), n, d_b, ctx->stream);
    CloudBounds hb{};
    HIPCHK(hipMemcpyAsync(&hb, d_b, sizeof(hb), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(sync_spin(ctx->stream));
    sp->num_finite = hb.num_finite;
    CellGrid g{};
    g.inv = 1.0 / (1.001 * (double)radius);
    g.nx = g.ny = g.nz = 1;
    if (hb.num_finite > 0) {
        int64_t* dims[3] = {&g.nx, &g.ny, &g.nz};
        for (int a = 0; a < 3; ++a) {
            g.lo[a] = (double)ordered_to_float(hb.lo[a]);
            const double ext = ((double)ordered_to_float(hb.hi[a]) - g.lo[a]) * g.inv;
            if (!(ext < 2097150.0)) return DMSA_ERR_DEPTH;  // more than 2^21 cells along an axis
            *dims[a] = (int64_t)std::floor(ext) + 1;
        }
    }
    sp->grid = g;
    const double cells = (double)g.nx * (double)g.ny * (double)g.nz;
    unsigned bits = 1;
    while (bits < 63 && std::ldexp(1.0, (int)bits) < cells) ++bits;
    sp->key32 = bits < 32;  // the invalid marker ~0 needs one more value than the largest code
    const unsigned end_bit = sp->key32 ? 32u : 64u;
    HIPCHK(sp->code.ensure((size_t)n * 8));
    HIPCHK(sp->idx.ensure((size_t)n * 4));
    HIPCHK(sp->code_s.ensure((size_t)n * 8));
    HIPCHK(sp->idx_s.ensure((size_t)n * 4));
    HIPCHK(sp->pts_sorted.ensure((size_t)n * 16));
    HIPCHK(sp->sort_tmp.ensure(sort_pairs_temp_bytes((size_t)n)));
    size_t cap = 1024;
    while (cap < 2 * (size_t)n) cap <<= 1;
    sp->table_mask = (uint32_t)(cap - 1);
    HIPCHK(sp->table.ensure(cap * sizeof(CellHashEntry)));
    HIPCHK(hipMemsetAsync(sp->table.p, 0xFF, cap * sizeof(CellHashEntry), ctx->stream));
    if (sp->key32) {
        launch_cell_codes32(sp->cloud.as<float4>(), n, g, sp->code.as<uint32_t>(), sp->idx.as<uint32_t>(), ctx->stream);
        // sort on the bits that can differ; the all-ones marker of non-finite points has every bit set, so it still sorts last
        HIPCHK(sort_pairs_u32_u32(sp->sort_tmp.p, sp->sort_tmp.cap, sp->code.as<uint32_t>(), sp->code_s.as<uint32_t>(), sp->idx.as<uint32_t>(),
                                  sp->idx_s.as<uint32_t>(), (size_t)n, std::min(end_bit, bits + 1), ctx->stream));
    } else {
        launch_cell_codes(sp->cloud.as<float4>(), n, g, sp->code.as<uint64_t>(), sp->idx.as<uint32_t>(), ctx->stream);
        HIPCHK(sort_pairs_u64_u32(sp->sort_tmp.p, sp->sort_tmp.cap, sp->code.as<uint64_t>(), sp->code_s.as<uint64_t>(), sp->idx.as<uint32_t>(),
                                  sp->idx_s.as<uint32_t>(), (size_t)n, std::min(end_bit, bits + 1), ctx->stream));
    }
    launch_cell_table(sp->cloud.as<float4>(), sp->idx_s.as<uint32_t>(), sp->code_s.p, sp->key32, n, sp->pts_sorted.as<float4>(), sp->table.as<CellHashEntry>(),
                      sp->table_mask, ctx->stream);
    HIPCHK(hipGetLastError());
    return DMSA_OK;
}

// flags (device, sp->flags) of `nq` host queries against the grid built last
int sp_query(dmsa_ctx* ctx, const float* query_xyz, int64_t nq, float r2) {
    StaticState* sp = ctx->sp;
    if (nq <= 0) return DMSA_OK;
    HIPCHK(sp->query.ensure((size_t)nq * 16));
    HIPCHK(sp->flags.ensure((size_t)nq));
    CHK(sp_stage_points(ctx, sp->query.p, query_xyz, nq));
    if (sp->n_cloud == 0) {
        HIPCHK(hipMemsetAsync(sp->flags.p, 0, (size_t)nq, ctx->stream));
        return DMSA_OK;
    }
    launch_radius_exists(sp->query.as<float4>(), nq, sp->grid, sp->pts_sorted.as<float4>(), sp->code_s.p, sp->key32, sp->n_cloud, sp->table.as<CellHashEntry>(),
                         sp->table_mask, r2, sp->flags.as<uint8_t>(), ctx->stream);
    HIPCHK(hipGetLastError());
    return DMSA_OK;
}

// randomGridDownsampling (helpers.h:67-182) in three stages on the device-resident cloud sp->cloud.
int sp_grid_upload(dmsa_ctx* ctx, const float* xyz, int64_t n) {
    StaticState* sp = ctx->sp;
    HIPCHK(sp->cloud.ensure((size_t)n * 16));
    HIPCHK(sp->code.ensure((size_t)n * 8));
    HIPCHK(sp->idx.ensure((size_t)n * 4));
    HIPCHK(sp->code_s.ensure((size_t)n * 8));
    HIPCHK(sp->idx_s.ensure((size_t)n * 4));
    HIPCHK(sp->head.ensure((size_t)n * 4));
    HIPCHK(sp->incl.ensure((size_t)n * 4));
    HIPCHK(sp->leaf_start.ensure(((size_t)n + 1) * 4));
    HIPCHK(sp->sort_tmp.ensure(sort_pairs_temp_bytes((size_t)n)));
    HIPCHK(sp->scan_tmp.ensure(scan_temp_bytes((size_t)n)));
    HIPCHK(sp->counts.ensure(sizeof(GaussCounts)));
    HIPCHK(sp->lattice.ensure(2 * sizeof(LatticeTable)));
    HIPCHK(sp->aabb.ensure((size_t)((n + kAabbBlock - 1) / kAabbBlock) * 8 * sizeof(float)));
    CHK(sp_stage_points(ctx, sp->cloud.p, xyz, n));
    launch_block_aabb(sp->cloud.as<float4>(), n, sp->aabb.as<float>(), nullptr, 0, ctx->stream);  // independent of the resolution
    return DMSA_OK;
}
// the same PCL-exact lattice / key / leaf machinery as createGaussianSets (DmsaOptimizer.h:282-298): leaves sp->leaf_start (leaf
// boundaries in depth-first order) and sp->idx_s (point indices, ascending inside a leaf); *leaves = octree.getLeafCount()
int sp_grid_leaves(dmsa_ctx* ctx, int64_t n, float grid_size, int64_t* leaves) {
    StaticState* sp = ctx->sp;
    *leaves = 0;
    const double res = (double)grid_size;  // OctreePointCloud(gridSize): float -> double resolution
    const int nb = (int)((n + kAabbBlock - 1) / kAabbBlock);
    HIPCHK(hipMemsetAsync(sp->counts.p, 0, sizeof(GaussCounts), ctx->stream));
    launch_lattice(sp->cloud.as<float4>(), n, sp->aabb.as<float>(), nb, res, res, false, sp->lattice.as<LatticeTable>(), nullptr, nullptr, ctx->stream);
    LatticeTable lat[2];
    HIPCHK(hipMemcpyAsync(lat, sp->lattice.p, 2 * sizeof(LatticeTable), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(sync_spin(ctx->stream));
    if (lat[0].status != 0) return lat[0].status;
    if (!lat[0].defined) return DMSA_OK;  // no finite point: empty octree
    const unsigned end_bit = (unsigned)(3 * lat[0].final_depth + 1);
    const bool k32 = end_bit <= 32;
    LatticeTable* tab = sp->lattice.as<LatticeTable>();
    GaussCounts* counts = sp->counts.as<GaussCounts>();
    launch_voxel_keys(sp->cloud.as<float4>(), n, tab, res, sp->code.p, k32, sp->idx.as<uint32_t>(), 0ull, nullptr, ctx->stream);
    if (k32)
        HIPCHK(sort_pairs_u32_u32(sp->sort_tmp.p, sp->sort_tmp.cap, sp->code.as<uint32_t>(), sp->code_s.as<uint32_t>(), sp->idx.as<uint32_t>(),
                                  sp->idx_s.as<uint32_t>(), (size_t)n, end_bit, ctx->stream));
    else
        HIPCHK(sort_pairs_u64_u32(sp->sort_tmp.p, sp->sort_tmp.cap, sp->code.as<uint64_t>(), sp->code_s.as<uint64_t>(), sp->idx.as<uint32_t>(),
                                  sp->idx_s.as<uint32_t>(), (size_t)n, end_bit, ctx->stream));
    launch_head_flags(sp->code_s.p, k32, n, tab, sp->head.as<int32_t>(), ctx->stream);
    HIPCHK(inclusive_scan_i32(sp->scan_tmp.p, sp->scan_tmp.cap, sp->head.as<int32_t>(), sp->incl.as<int32_t>(), (size_t)n, ctx->stream));
    launch_leaf_starts(sp->head.as<int32_t>(), sp->incl.as<int32_t>(), sp->code_s.p, k32, tab, n, sp->leaf_start.as<int32_t>(), &counts->level[0], ctx->stream);
    GaussCounts hc{};
    HIPCHK(hipMemcpyAsync(&hc, counts, sizeof(hc), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(sync_spin(ctx->stream));
    *leaves = hc.level[0].num_leaves;
    return DMSA_OK;
}
// srand(seed); one rand() per leaf in depth-first order (helpers.h:86-94) -- the generator is a sequential recurrence, so the draws
// are made on the host (O(leaves)) and only the pick runs on the device; leaves sp->pick (index into the raw cloud per leaf)
int sp_grid_pick(dmsa_ctx* ctx, int64_t leaves, uint32_t seed) {
    StaticState* sp = ctx->sp;
    std::vector<int32_t> rnd((size_t)leaves);
    glibc_rand_fill(seed, rnd.data(), (size_t)leaves);
    HIPCHK(sp->rnd.ensure((size_t)leaves * 4));
    HIPCHK(sp->pick.ensure((size_t)leaves * 4));
    HIPCHK(hipMemcpy(sp->rnd.p, rnd.data(), (size_t)leaves * 4, hipMemcpyHostToDevice));  // rnd is a local: synchronous copy
    launch_leaf_pick(sp->leaf_start.as<int32_t>(), sp->idx_s.as<uint32_t>(), sp->rnd.as<int32_t>(), (int)leaves, sp->pick.as<int32_t>(), ctx->stream);
    HIPCHK(hipGetLastError());
    return DMSA_OK;
}

}  // namespace

extern "C" {

int dmsa_radius_exists(dmsa_ctx* ctx, const float* cloud_xyz, int64_t n_cloud, const float* query_xyz, int64_t n_query, float radius, uint8_t* flag_out) {
    if (!ctx || n_query < 0 || (n_query > 0 && (!query_xyz || !flag_out))) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    CHK(sp_build_grid(ctx, cloud_xyz, n_cloud, radius));
    CHK(sp_query(ctx, query_xyz, n_query, radius * radius));
    if (n_query > 0) HIPCHK(hipMemcpyAsync(flag_out, ctx->sp->flags.p, (size_t)n_query, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return DMSA_OK;
}

int dmsa_select_static_points(dmsa_ctx* ctx, const dmsa_static_select_problem* p, float* static_xyz_out, int32_t* static_id_out, int64_t capacity,
                              int32_t* overlap_per_keyframe, dmsa_static_select_result* res) {
    if (!ctx || !p || !res || p->num_keyframes < 0 || p->num_window < 0 || (p->num_keyframes > 0 && (!p->frame_offset || !p->keyframe_ids))) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    const int K = p->num_keyframes;
    const int64_t n = K > 0 ? p->frame_offset[K] : 0;
    *res = dmsa_static_select_result{};
    res->min_related_key_id = -1;
    if (n > 0 && (!p->key_xyz || !p->key_normal || !p->key_ring)) return DMSA_ERR_INVALID;
    if (n > (int64_t)0x7FFFFFF0) return DMSA_ERR_INVALID;
    std::vector<int32_t> at((size_t)K + 1, 0);
    if (n > 0) {
        // std::pow(1.0f * minGridSize, 2): float argument, integer exponent -> double -> back to float (DmsaSlam.h:295)
        const float sqrdMaxDist = (float)std::pow((double)(1.0f * p->min_grid_size), 2);
        CHK(sp_build_grid(ctx, p->window_xyz, p->num_window, p->min_grid_size));
        CHK(sp_query(ctx, p->key_xyz, n, sqrdMaxDist));
        StaticState* sp = ctx->sp;
        HIPCHK(sp->normal.ensure((size_t)n * 16));
        HIPCHK(sp->ring.ensure((size_t)n * 4));
        HIPCHK(sp->sel.ensure((size_t)n * 4));
        HIPCHK(sp->scan.ensure((size_t)n * 4));
        HIPCHK(sp->scan_tmp.ensure(scan_temp_bytes((size_t)n)));
        HIPCHK(sp->out_xyz.ensure((size_t)n * 16));
        HIPCHK(sp->out_id.ensure((size_t)n * 4));
        HIPCHK(sp->offsets.ensure(((size_t)K + 1) * 8 + ((size_t)K + 1) * 4));
        HIPCHK(hipMemcpyAsync(sp->normal.p, p->key_normal, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(sp->ring.p, p->key_ring, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(sp->offsets.p, p->frame_offset, ((size_t)K + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        launch_static_flags(sp->query.as<float4>(), sp->normal.as<float4>(), sp->flags.as<uint8_t>(), n, p->cur_pos[0], p->cur_pos[1], p->cur_pos[2],
                            sp->sel.as<int32_t>(), ctx->stream);
        HIPCHK(exclusive_scan_i32(sp->scan_tmp.p, sp->scan_tmp.cap, sp->sel.as<int32_t>(), sp->scan.as<int32_t>(), (size_t)n, ctx->stream));
        launch_static_scatter(sp->query.as<float4>(), sp->ring.as<int32_t>(), sp->sel.as<int32_t>(), sp->scan.as<int32_t>(), n, sp->out_xyz.as<float4>(),
                              sp->out_id.as<int32_t>(), ctx->stream);
        int32_t* d_at = reinterpret_cast<int32_t*>(sp->offsets.as<int64_t>() + (K + 1));
        launch_pick_offsets(sp->scan.as<int32_t>(), sp->sel.as<int32_t>(), sp->offsets.as<int64_t>(), K, n, d_at, ctx->stream);
        HIPCHK(hipMemcpyAsync(at.data(), d_at, ((size_t)K + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    // per-keyframe bookkeeping of the loop (:270-274, :333-342): the running maximum is tested after every point, so a later
    // keyframe only takes over when its count EXCEEDS the best so far
    int keyframeId = 0, maxOverlapKey = 0, minRelatedKeyId = -1;
    for (int kk = 0; kk < K; ++kk) {
        const int k = p->keyframe_ids[kk], currOverlap = at[(size_t)kk + 1] - at[(size_t)kk];
        if (overlap_per_keyframe) overlap_per_keyframe[kk] = currOverlap;
        if (currOverlap > 0 && (minRelatedKeyId < 0 || k < minRelatedKeyId)) minRelatedKeyId = k;
        if (currOverlap > maxOverlapKey) maxOverlapKey = currOverlap, keyframeId = k;
    }
    const int64_t total = at[(size_t)K];
    res->num_static = total, res->keyframe_id = keyframeId, res->min_related_key_id = minRelatedKeyId, res->max_overlap = maxOverlapKey;
    if (total > capacity) return DMSA_ERR_INVALID;
    if (total > 0) {
        if (static_xyz_out) HIPCHK(hipMemcpy(static_xyz_out, ctx->sp->out_xyz.p, (size_t)total * 16, hipMemcpyDeviceToHost));
        if (static_id_out) HIPCHK(hipMemcpy(static_id_out, ctx->sp->out_id.p, (size_t)total * 4, hipMemcpyDeviceToHost));
    }
    return DMSA_OK;
}

int dmsa_get_overlap(dmsa_ctx* ctx, const float* pc1_xyz, int64_t n1, const float* pc2_xyz, int64_t n2, float max_dist_overlap, float* overlap_out,
                     int64_t* num_corresp_out) {
    if (!ctx || n1 < 0 || n2 < 0) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    unsigned long long nCorresp = 0;
    float overlap = 0.0f;
    if (n1 > 0 && n2 > 0) {  // :380-381
        CHK(sp_build_grid(ctx, pc1_xyz, n1, max_dist_overlap));
        CHK(sp_query(ctx, pc2_xyz, n2, max_dist_overlap * max_dist_overlap));
        StaticState* sp = ctx->sp;
        unsigned long long* d_cnt = reinterpret_cast<unsigned long long*>(sp->small.as<char>() + 128);
        launch_count_flags(sp->flags.as<uint8_t>(), n2, d_cnt, ctx->stream);
        HIPCHK(hipMemcpyAsync(&nCorresp, d_cnt, sizeof(nCorresp), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        overlap = static_cast<float>((int)nCorresp) / static_cast<float>(n2);  // :412
    }
    if (overlap_out) *overlap_out = overlap;
    if (num_corresp_out) *num_corresp_out = (int64_t)nCorresp;
    return DMSA_OK;
}

int dmsa_random_grid_downsampling(dmsa_ctx* ctx, const float* xyz, int64_t n, float grid_size, uint32_t seed, int32_t* picked_index_out, int64_t capacity,
                                  int64_t* num_out) {
    if (!ctx || n < 0 || !(grid_size > 0.0f) || n > (int64_t)0x7FFFFFF0) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    if (num_out) *num_out = 0;
    if (n == 0) return DMSA_OK;
    StaticState* sp = sp_state(ctx);
    if (!sp) return DMSA_ERR_NOMEM;
    CHK(sp_grid_upload(ctx, xyz, n));
    int64_t leaves = 0;
    CHK(sp_grid_leaves(ctx, n, grid_size, &leaves));  // octree.getLeafCount()
    if (num_out) *num_out = leaves;
    if (leaves > capacity) return DMSA_ERR_INVALID;
    if (leaves == 0 || !picked_index_out) return DMSA_OK;
    CHK(sp_grid_pick(ctx, leaves, seed));
    HIPCHK(hipMemcpyAsync(picked_index_out, sp->pick.p, (size_t)leaves * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return DMSA_OK;
}

int dmsa_preprocess_scan(dmsa_ctx* ctx, const float* raw_xyz, int64_t n, const dmsa_preprocess_config* cfg, float* xyz_out, int32_t* src_index_out,
                         int64_t capacity, int64_t* num_out, float* grid_size_out) {
    if (num_out) *num_out = 0;
    if (!ctx || !cfg || n < 0 || (n > 0 && !raw_xyz) || cfg->max_num_points_per_scan < 0 || capacity < 0 || n > (int64_t)0x7FFFFFF0) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    static const float kGrids[4] = {0.4f, 0.3f, 0.2f, 0.15f};  // DmsaSlam.h:572-592
    if (grid_size_out) *grid_size_out = kGrids[0];
    if (n == 0) return DMSA_OK;
    StaticState* sp = sp_state(ctx);
    if (!sp) return DMSA_ERR_NOMEM;
    CHK(sp_grid_upload(ctx, raw_xyz, n));
    int64_t m = 0;
    float grid = kGrids[0];
    for (int pass = 0; pass < 4; ++pass) {  // every pass filters the RAW scan again; the last one run is the one kept
        if (pass > 0 && !(m < (int64_t)cfg->max_num_points_per_scan)) break;
        grid = kGrids[pass];
        CHK(sp_grid_leaves(ctx, n, grid, &m));
    }
    if (grid_size_out) *grid_size_out = grid;
    if (m == 0) return DMSA_OK;
    CHK(sp_grid_pick(ctx, m, cfg->seed));
    const int mi = (int)m;
    // the leaf machinery is done with code / idx / code_s / idx_s / head / incl: reuse them for the range sort and the compaction
    uint32_t* range_bits = sp->code.as<uint32_t>();
    uint32_t* iota = sp->idx.as<uint32_t>();
    uint32_t* sorted_bits = sp->code_s.as<uint32_t>();
    int32_t* sel = sp->head.as<int32_t>();
    int32_t* scan = sp->incl.as<int32_t>();
    HIPCHK(sp->out_xyz.ensure((size_t)m * 16));
    HIPCHK(sp->out_id.ensure((size_t)m * 4));
    HIPCHK(sp->small.ensure(256));
    int32_t* d_total = sp->small.as<int32_t>() + 48;
    launch_scan_ranges(sp->cloud.as<float4>(), sp->pick.as<int32_t>(), mi, range_bits, iota, ctx->stream);
    HIPCHK(sort_pairs_u32_u32(sp->sort_tmp.p, sp->sort_tmp.cap, range_bits, sorted_bits, iota, sp->idx_s.as<uint32_t>(), (size_t)m, 32u, ctx->stream));
    const int thres_pos = std::min((int)cfg->max_num_points_per_scan, mi - 1);  // :609
    launch_scan_range_gate(range_bits, sorted_bits, mi, thres_pos, cfg->min_dist_ds, cfg->min_dist, sel, ctx->stream);
    HIPCHK(exclusive_scan_i32(sp->scan_tmp.p, sp->scan_tmp.cap, sel, scan, (size_t)m, ctx->stream));
    launch_scan_emit(sp->cloud.as<float4>(), sp->pick.as<int32_t>(), sel, scan, mi, cfg->lidar_to_imu, sp->out_xyz.as<float4>(), sp->out_id.as<int32_t>(), d_total,
                     ctx->stream);
    HIPCHK(hipGetLastError());
    int32_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, d_total, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(sync_spin(ctx->stream));
    if (num_out) *num_out = total;
    if (total > capacity) return DMSA_ERR_INVALID;
    if (total > 0) {
        if (xyz_out) HIPCHK(hipMemcpyAsync(xyz_out, sp->out_xyz.p, (size_t)total * 16, hipMemcpyDeviceToHost, ctx->stream));
        if (src_index_out) HIPCHK(hipMemcpyAsync(src_index_out, sp->out_id.p, (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return DMSA_OK;
}

// include/dmsa_window_setup.h: the one per-point step of the window setup (the rest is host arithmetic in window_setup.cpp)
int dmsa_traj_tform_indices(dmsa_ctx* ctx, const double* point_stamps, int64_t n, double t0, const double* traj_time, int32_t n_total, int32_t* tform_idx_out) {
    if (!ctx || n < 0 || n_total < 1 || !traj_time || (n > 0 && (!point_stamps || !tform_idx_out)) || n > (int64_t)0x7FFFFFF0) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    if (n == 0) return DMSA_OK;
    StaticState* sp = sp_state(ctx);
    if (!sp) return DMSA_ERR_NOMEM;
    HIPCHK(sp->cloud.ensure((size_t)n * 8));
    HIPCHK(sp->query.ensure((size_t)n_total * 8));
    HIPCHK(sp->out_id.ensure((size_t)n * 4));
    HIPCHK(hipMemcpyAsync(sp->cloud.p, point_stamps, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(sp->query.p, traj_time, (size_t)n_total * 8, hipMemcpyHostToDevice, ctx->stream));
    launch_tform_indices(sp->cloud.as<double>(), n, t0, sp->query.as<double>(), n_total, sp->out_id.as<int32_t>(), ctx->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(tform_idx_out, sp->out_id.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return DMSA_OK;
}

// include/dmsa_wire_formats.h: the PointCloud2 decoder (the text side lives in wire_formats.cpp)
int dmsa_decode_pointcloud2(dmsa_ctx* ctx, const dmsa_pointcloud2* msg, int32_t sensor, float* xyz_out, double* stamp_out, int32_t* id_out) {
    if (!ctx || !msg || sensor < DMSA_SENSOR_HESAI || sensor > DMSA_SENSOR_UNKNOWN) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    const uint64_t n64 = (uint64_t)msg->height * msg->width;
    if (n64 == 0) return DMSA_OK;
    if (n64 > 0x7FFFFFF0ull || !msg->data || !msg->field_offsets || msg->num_fields < 3 || !xyz_out || !stamp_out || !id_out) return DMSA_ERR_INVALID;
    if (n64 * msg->point_step > msg->data_bytes || n64 * msg->point_step > 0xFFFFFFFFull) return DMSA_ERR_INVALID;
    // which fields the sensor type reads (dmsa_slam_ros.cpp:411-481): {stamp field, its size, ring field, its size}, -1 = none
    static const int kFields[8][4] = {{4, 8, 5, 2}, {4, 4, 6, 1}, {5, 8, 4, 2}, {5, 4, 4, 2}, {6, 8, -1, 0}, {6, 8, -1, 0}, {8, 4, 11, 1}, {-1, 0, -1, 0}};
    const int* fs = kFields[sensor];
    PointCloud2Fields f{msg->field_offsets[0], msg->field_offsets[1], msg->field_offsets[2], 0, 0};
    auto inside = [&](uint32_t off, uint32_t size) { return (uint64_t)off + size <= msg->point_step; };
    if (!inside(f.x, 4) || !inside(f.y, 4) || !inside(f.z, 4)) return DMSA_ERR_INVALID;
    if (fs[0] >= 0) {
        if ((uint32_t)fs[0] >= msg->num_fields || !inside(msg->field_offsets[fs[0]], (uint32_t)fs[1])) return DMSA_ERR_INVALID;
        f.stamp = msg->field_offsets[fs[0]];
    }
    if (fs[2] >= 0) {
        if ((uint32_t)fs[2] >= msg->num_fields || !inside(msg->field_offsets[fs[2]], (uint32_t)fs[3])) return DMSA_ERR_INVALID;
        f.ring = msg->field_offsets[fs[2]];
    }
    StaticState* sp = sp_state(ctx);
    if (!sp) return DMSA_ERR_NOMEM;
    const size_t n = (size_t)n64, bytes = n * msg->point_step;
    HIPCHK(sp->query.ensure(bytes));
    HIPCHK(sp->cloud.ensure(n * 16));
    HIPCHK(sp->code.ensure(n * 8));
    HIPCHK(sp->out_id.ensure(n * 4));
    HIPCHK(hipMemcpyAsync(sp->query.p, msg->data, bytes, hipMemcpyHostToDevice, ctx->stream));
    launch_decode_pointcloud2(sp->query.as<uint8_t>(), (uint32_t)n, msg->point_step, f, sensor, msg->stamp_msg, msg->delta_t_pcs, sp->cloud.as<float4>(),
                              sp->code.as<double>(), sp->out_id.as<int32_t>(), ctx->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(xyz_out, sp->cloud.p, n * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(stamp_out, sp->code.p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(id_out, sp->out_id.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return DMSA_OK;
}

// include/dmsa_keyframe_cloud.h
namespace {
// normals of the n points in sp->cloud (device) into sp->normal; neighbour lists into sp->sel when wanted
int sp_normals(dmsa_ctx* ctx, int64_t n, int k, float cell_hint, const float* viewpoint, bool want_nn) {
    StaticState* sp = ctx->sp;
    HIPCHK(sp->normal.ensure((size_t)n * 16));
    if (want_nn) HIPCHK(sp->sel.ensure((size_t)n * (size_t)k * 4));
    CHK(sp_build_grid_device(ctx, n, cell_hint));
    launch_knn_normals(sp->cloud.as<float4>(), n, k, sp->grid, 1.001 * (double)cell_hint, sp->pts_sorted.as<float4>(), sp->idx_s.as<uint32_t>(), sp->code_s.p,
                       sp->key32, sp->table.as<CellHashEntry>(), sp->table_mask, sp->num_finite, viewpoint[0], viewpoint[1], viewpoint[2], sp->normal.as<float4>(),
                       want_nn ? sp->sel.as<int32_t>() : nullptr, ctx->stream);
    HIPCHK(hipGetLastError());
    return DMSA_OK;
}
}  // namespace

int dmsa_update_normals(dmsa_ctx* ctx, const float* xyz, int64_t n, int32_t k, float cell_hint, const float viewpoint[3], float* normal_out, int32_t* nn_index_out) {
    if (!ctx || n < 0 || (n > 0 && (!xyz || !normal_out)) || k < 1 || k > 8 || !(cell_hint > 0.0f) || !viewpoint || n > (int64_t)0x7FFFFFF0 / 8) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    if (n == 0) return DMSA_OK;
    StaticState* sp = sp_state(ctx);
    if (!sp) return DMSA_ERR_NOMEM;
    HIPCHK(sp->cloud.ensure((size_t)n * 16));
    HIPCHK(hipMemcpyAsync(sp->cloud.p, xyz, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
    CHK(sp_normals(ctx, n, k, cell_hint, viewpoint, nn_index_out != nullptr));
    HIPCHK(hipMemcpyAsync(normal_out, sp->normal.p, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
    if (nn_index_out) HIPCHK(hipMemcpyAsync(nn_index_out, sp->sel.p, (size_t)n * (size_t)k * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return DMSA_OK;
}

int dmsa_make_keyframe_cloud(dmsa_ctx* ctx, const float* global_xyz, const int32_t* ids, int64_t n, float min_grid_size, uint32_t seed, const double pos0[3],
                             const double orient0[3], float* xyz_local_out, float* normal_out, int32_t* ring_out, int32_t* src_index_out, int64_t capacity,
                             int64_t* num_out) {
    if (num_out) *num_out = 0;
    if (!ctx || n < 0 || !(min_grid_size > 0.0f) || !pos0 || !orient0 || capacity < 0 || n > (int64_t)0x7FFFFFF0 / 8) return DMSA_ERR_INVALID;
    if ((global_xyz == nullptr) != (ids == nullptr)) return DMSA_ERR_INVALID;  // both from the host or both resident
    CHK(set_device(ctx));
    if (n == 0) return DMSA_OK;
    StaticState* sp = sp_state(ctx);
    if (!sp) return DMSA_ERR_NOMEM;
    // randomGridDownsampling(trajIn.globalPoints, keyframeCloudFiltered, trajIn.minGridSize) (:505)
    CHK(sp_grid_upload(ctx, global_xyz, n));
    int64_t m = 0;
    CHK(sp_grid_leaves(ctx, n, min_grid_size, &m));
    if (num_out) *num_out = m;
    if (m > capacity) return DMSA_ERR_INVALID;
    if (m == 0) return DMSA_OK;
    CHK(sp_grid_pick(ctx, m, seed));
    // currWorldPose = Translations.col(0).cast<float>(), currRotInv = axang2rotm(Orientations.col(0)).transpose().cast<float>() (:511-512)
    const dmsa::Mat3 R = dmsa::so3_exp({orient0[0], orient0[1], orient0[2]});
    float rinv[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) rinv[3 * r + c] = (float)R(c, r);
    HIPCHK(sp->ring.ensure((size_t)n * 4));
    HIPCHK(sp->out_xyz.ensure((size_t)m * 16));
    HIPCHK(sp->out_id.ensure((size_t)m * 4));
    if (ids)
        HIPCHK(hipMemcpyAsync(sp->ring.p, ids, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    else
        HIPCHK(hipMemcpyAsync(sp->ring.p, ctx->d_ring.p, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    launch_to_keyframe_frame(sp->cloud.as<float4>(), sp->ring.as<int32_t>(), sp->pick.as<int32_t>(), (int)m, rinv, (float)pos0[0], (float)pos0[1], (float)pos0[2],
                             sp->out_xyz.as<float4>(), sp->out_id.as<int32_t>(), ctx->stream);
    if (src_index_out) HIPCHK(hipMemcpyAsync(src_index_out, sp->pick.p, (size_t)m * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (ring_out) HIPCHK(hipMemcpyAsync(ring_out, sp->out_id.p, (size_t)m * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (xyz_local_out) HIPCHK(hipMemcpyAsync(xyz_local_out, sp->out_xyz.p, (size_t)m * 16, hipMemcpyDeviceToHost, ctx->stream));
    // updateNormals(keyframeCloud_imu) (:526): k = 6, viewpoint = origin; the local cloud becomes the grid's cloud
    HIPCHK(sp->cloud.ensure((size_t)m * 16));
    HIPCHK(hipMemcpyAsync(sp->cloud.p, sp->out_xyz.p, (size_t)m * 16, hipMemcpyDeviceToDevice, ctx->stream));
    const float origin[3] = {0.0f, 0.0f, 0.0f};
    CHK(sp_normals(ctx, m, 6, 2.0f * min_grid_size, origin, false));
    if (normal_out) HIPCHK(hipMemcpyAsync(normal_out, sp->normal.p, (size_t)m * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return DMSA_OK;
}

}  // extern "C"

