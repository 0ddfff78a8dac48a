// dmsa_api.cpp — the C ABI of include/dmsa_hip.h on top of the context: stage-level entry points (parity tests, benchmark) and the whole
// optimizeSet calls.  Creation / uploads: context.cpp; the loop: optimize_loop.cpp; the rows around the hot path: next_rows_api.cpp.
#include "dmsa_ctx.h"

extern "C" {


int dmsa_centralize(dmsa_ctx* ctx) {
    if (!ctx || ctx->model == MODEL_NONE) return DMSA_ERR_INVALID;
    if (ctx->model == MODEL_KEYFRAMES) return DMSA_OK;  // MapManagement::centralize returns immediately (MapManagement.h:73-79)
    if (ctx->centralized) return DMSA_OK;  // already in the centred frame: a second shift would lose the origin
    CHK(set_device(ctx));
    WindowHost& w = ctx->win;  // ContinuousTrajectory.h:75-88
    w.origin = {w.ctrl.rel_t[0], w.ctrl.rel_t[1], w.ctrl.rel_t[2]};
    w.ctrl.rel_t[0] = w.ctrl.rel_t[1] = w.ctrl.rel_t[2] = 0.0;
    w.ctrl.relative_to_global();
    launch_shift_points(ctx->d_local.as<float4>() + ctx->N, ctx->S, (float)w.origin.x, (float)w.origin.y, (float)w.origin.z, -1.0f, ctx->stream);
    HIPCHK(hipGetLastError());
    ctx->centralized = true;
    return DMSA_OK;
}

int dmsa_decentralize(dmsa_ctx* ctx) {
    if (!ctx || ctx->model == MODEL_NONE) return DMSA_ERR_INVALID;
    if (ctx->model == MODEL_KEYFRAMES) return DMSA_OK;
    if (!ctx->centralized) return DMSA_OK;  // nothing to undo
    CHK(set_device(ctx));
    WindowHost& w = ctx->win;  // ContinuousTrajectory.h:89-100
    w.ctrl.global_to_relative();
    w.ctrl.rel_t[0] = w.origin.x, w.ctrl.rel_t[1] = w.origin.y, w.ctrl.rel_t[2] = w.origin.z;
    w.ctrl.relative_to_global();
    launch_shift_points(ctx->d_local.as<float4>() + ctx->N, ctx->S, (float)w.origin.x, (float)w.origin.y, (float)w.origin.z, 1.0f, ctx->stream);
    HIPCHK(hipGetLastError());
    ctx->centralized = false;
    return DMSA_OK;
}

int dmsa_get_params(dmsa_ctx* ctx, double* params, int32_t* P) {
    if (!ctx || ctx->model == MODEL_NONE) return DMSA_ERR_INVALID;
    if (P) *P = num_params(ctx);
    if (params) chain(ctx).get_params(params);
    return DMSA_OK;
}
int dmsa_set_params(dmsa_ctx* ctx, const double* params) {
    if (!ctx || ctx->model == MODEL_NONE || !params) return DMSA_ERR_INVALID;
    host_set_params(ctx, params);
    return DMSA_OK;
}

int dmsa_additional_errors(dmsa_ctx* ctx, double* rows_out, int32_t capacity, int32_t* num_out) {
    if (!ctx || ctx->model == MODEL_NONE || !num_out) return DMSA_ERR_INVALID;
    const int a = num_extra_rows(ctx);
    *num_out = a;
    if (a == 0) return DMSA_OK;
    if (!rows_out || capacity < a) return DMSA_ERR_INVALID;
    chain(ctx).relative_to_global();
    if (ctx->model == MODEL_WINDOW)
        ctx->win.imu_rows(rows_out);
    else
        ctx->key.additional_rows(rows_out);
    return DMSA_OK;
}

int dmsa_num_table_rows(dmsa_ctx* ctx, int32_t* n_rows) {
    if (!ctx || ctx->model == MODEL_NONE || !n_rows) return DMSA_ERR_INVALID;
    *n_rows = ctx->rows - 1;
    return DMSA_OK;
}

// copy B tables to the host without the trailing identity row
static int download_tables(dmsa_ctx* ctx, int B, float* out) {
    const int rows = ctx->rows;
    std::vector<float> tmp((size_t)B * rows * 12);
    HIPCHK(hipMemcpyAsync(tmp.data(), ctx->d_tables.p, tmp.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < B; ++b) std::memcpy(out + (size_t)b * (rows - 1) * 12, tmp.data() + (size_t)b * rows * 12, (size_t)(rows - 1) * 48);
    return DMSA_OK;
}

int dmsa_pose_tables(dmsa_ctx* ctx, int32_t B, const double* params, float* tables_out) {
    if (!ctx || ctx->model == MODEL_NONE || B <= 0 || !params) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    const int P = num_params(ctx);
    std::vector<double> globs;
    for (int b = 0; b < B; ++b) {
        host_set_params(ctx, params + (size_t)b * P);
        append_glob(chain(ctx), globs);
    }
    CHK(build_tables(ctx, B, globs));
    if (tables_out) CHK(download_tables(ctx, B, tables_out));
    return DMSA_OK;
}

int dmsa_set_pose_tables(dmsa_ctx* ctx, int32_t B, const float* tables) {
    if (!ctx || ctx->model == MODEL_NONE || B <= 0 || !tables) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    const int rows = ctx->rows;
    std::vector<float> tmp((size_t)B * rows * 12);
    const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    for (int b = 0; b < B; ++b) {
        std::memcpy(tmp.data() + (size_t)b * rows * 12, tables + (size_t)b * (rows - 1) * 12, (size_t)(rows - 1) * 48);
        std::memcpy(tmp.data() + ((size_t)b * rows + rows - 1) * 12, I, sizeof(I));
    }
    HIPCHK(ctx->d_tables.ensure(tmp.size() * 4));
    HIPCHK(hipMemcpyAsync(ctx->d_tables.p, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->batch = B;
    ctx->tablesT_batch = 0;
    return DMSA_OK;
}

int dmsa_transform_points(dmsa_ctx* ctx, int32_t b, float* xyz_out) {
    if (!ctx || ctx->model == MODEL_NONE || b < 0 || b >= ctx->batch) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    CHK(transform_points(ctx, b));
    if (xyz_out) {
        HIPCHK(hipMemcpyAsync(xyz_out, ctx->d_global.p, (size_t)ctx->n * 16, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return DMSA_OK;
}

// stage-level calls enqueue device-side waits as well: a wait that gave up fails the call (and leaves clean counters behind); only the whole
// calls (optimize) start over by themselves
static int stage_sync_check(dmsa_ctx* ctx, int rc) {
    std::string what;
    if (ctx->dbg.device_sync != 0 && sync_wait_timed_out(ctx, &what)) {
        ctx->err = "a device-side stream dependency timed out (" + what + "): a tool that serialises kernels across queues (hardware counter collection)? "
                   "run with DMSA_DEBUG=device_sync=0" + (rc == DMSA_OK ? std::string() : " | " + ctx->err);
        return DMSA_ERR_HIP;
    }
    return rc;
}
int dmsa_build_gaussians(dmsa_ctx* ctx, const dmsa_settings* s, int32_t* M_out, int64_t* Mm_out) {
    if (!ctx || ctx->model == MODEL_NONE || !s) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    ctx->wait_seq = 0, ctx->voxel_calls = 0;
    const int rc = build_gaussians(ctx, *s);
    CHK(stage_sync_check(ctx, rc));
    if (M_out) *M_out = ctx->M;
    if (Mm_out) *Mm_out = ctx->Mm;
    return DMSA_OK;
}

int dmsa_eval_residuals(dmsa_ctx* ctx, double* e_out) {
    if (!ctx || ctx->model == MODEL_NONE || !ctx->gaussians_valid || ctx->batch <= 0 || ctx->M <= 0) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    ctx->wait_seq = 0;
    const int rc = run_residuals(ctx, ctx->batch, nullptr);
    CHK(stage_sync_check(ctx, rc));
    if (e_out) {
        HIPCHK(hipMemcpy2DAsync(e_out, (size_t)ctx->M * 8, ctx->d_E.p, (size_t)ctx->ldE * 8, (size_t)ctx->M * 8, (size_t)ctx->batch, hipMemcpyDeviceToHost,
                                ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    drain_timers(ctx);
    return DMSA_OK;
}

int dmsa_normal_equations(dmsa_ctx* ctx, int32_t P, int32_t a, const double* extra_rows, double h, double lambda, double* H_out, double* g_out) {
    if (!ctx || !ctx->gaussians_valid || ctx->batch != P + 1 || (a != ctx->extra_rows && a != 0)) return DMSA_ERR_INVALID;
    if (ctx->E_is_jacobian) {  // P > 64 turns the residual batch into [J | e0] in place: evaluate the residuals again first
        ctx->err = "dmsa_normal_equations: the residual batch was already consumed (call dmsa_eval_residuals again)";
        return DMSA_ERR_INVALID;
    }
    CHK(set_device(ctx));
    ctx->E_is_jacobian = P > 64;
    int rowsE = ctx->M;
    if (a > 0 && extra_rows) {
        if (a != ctx->extra_rows) return DMSA_ERR_INVALID;
        HIPCHK(hipMemcpy2DAsync(ctx->d_E.as<double>() + ctx->M, (size_t)ctx->ldE * 8, extra_rows, (size_t)a * 8, (size_t)a * 8, (size_t)(P + 1),
                                hipMemcpyHostToDevice, ctx->stream));
        rowsE += a;
    }
    std::vector<double> Hp((size_t)(P + 1) * (P + 1));
    {
        ScopedTimer tm(ctx, T_NORMAL);
        HIPCHK(ctx->d_ne_partial.ensure((size_t)normal_equations_partial_doubles(rowsE, P) * 8));
        HIPCHK(ctx->d_Hp.ensure(Hp.size() * 8));
        launch_normal_equations(ctx->d_E.as<double>(), ctx->ldE, rowsE, P, 1.0 / h, ctx->d_ne_partial.as<double>(), ctx->d_Hp.as<double>(), ctx->stream);
    }
    HIPCHK(hipMemcpyAsync(Hp.data(), ctx->d_Hp.p, Hp.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    drain_timers(ctx);
    const int n1 = P + 1;
    if (H_out)
        for (int j = 0; j < P; ++j)
            for (int i = 0; i < P; ++i) H_out[(size_t)j * P + i] = Hp[(size_t)j * n1 + i] + (i == j ? lambda : 0.0);
    if (g_out)
        for (int i = 0; i < P; ++i) g_out[i] = Hp[(size_t)P * n1 + i];
    return DMSA_OK;
}

int dmsa_get_voxel_level(dmsa_ctx* ctx, int32_t level, dmsa_voxel_level_info* info, uint64_t* leaf_code, uint32_t* key_xyz, int32_t* sorted_point_idx) {
    if (!ctx || !ctx->gaussians_valid || level < 0 || level > 1) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    const size_t n = (size_t)ctx->n;
    const LatticeTable& t = ctx->h_lattice[level];
    const uint64_t invalid = lattice_invalid_code(t);
    std::vector<uint64_t> code(n);
    if (ctx->key32[level]) {
        std::vector<uint32_t> c32(n);
        HIPCHK(hipMemcpy(c32.data(), ctx->code_v[level], n * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) code[i] = c32[i];
    } else {
        HIPCHK(hipMemcpy(code.data(), ctx->code_v[level], n * 8, hipMemcpyDeviceToHost));
    }
    int64_t valid = 0;
    const int nb3[3] = {t.nbits[0], t.nbits[1], t.nbits[2]};
    const int maxb = std::max(nb3[0], std::max(nb3[1], nb3[2]));
    for (size_t i = 0; i < n; ++i) {
        const bool ok = code[i] != invalid;
        valid += ok ? 1 : 0;
        uint32_t k[3] = {0, 0, 0};
        uint64_t full = UINT64_MAX;
        if (ok) {
            code[i] &= ~t.code_or;  // level tag of the merged sort
            if (t.compressed) {  // undo the compression: peel the bits off in reverse order of their insertion
                uint64_t c = code[i];
                uint32_t low[3] = {0, 0, 0};
                for (int l = 0; l < maxb; ++l)
                    for (int a = 2; a >= 0; --a)
                        if (l < nb3[a]) {
                            low[a] |= (uint32_t)(c & 1ull) << l;
                            c >>= 1;
                        }
                for (int a = 0; a < 3; ++a) k[a] = (t.key_base[a] << nb3[a]) | low[a];
            } else {
                for (int l = 0; l < t.final_depth; ++l) {
                    k[0] |= (uint32_t)((code[i] >> (3 * l + 2)) & 1ull) << l;
                    k[1] |= (uint32_t)((code[i] >> (3 * l + 1)) & 1ull) << l;
                    k[2] |= (uint32_t)((code[i] >> (3 * l)) & 1ull) << l;
                }
            }
            full = 0;
            for (int l = t.final_depth - 1; l >= 0; --l) full = (full << 3) | (((k[0] >> l) & 1u) << 2) | (((k[1] >> l) & 1u) << 1) | ((k[2] >> l) & 1u);
        }
        if (key_xyz) key_xyz[3 * i] = k[0], key_xyz[3 * i + 1] = k[1], key_xyz[3 * i + 2] = k[2];
        if (leaf_code) leaf_code[i] = full;
    }
    if (sorted_point_idx) HIPCHK(hipMemcpy(sorted_point_idx, ctx->idx_s_v[level], (size_t)valid * 4, hipMemcpyDeviceToHost));
    if (info) {
        GaussCounts h{};
        HIPCHK(hipMemcpy(&h, ctx->d_counts.p, sizeof(h), hipMemcpyDeviceToHost));
        info->resolution = ctx->level_res[level];
        for (int a = 0; a < 3; ++a) info->min_xyz[a] = t.final_mn[a];
        info->depth = t.final_depth, info->num_events = t.num_events;
        info->num_leaves = h.level[level].num_leaves, info->num_valid = valid;
    }
    return DMSA_OK;
}

int dmsa_get_gaussians(dmsa_ctx* ctx, int32_t* seg_offset, int32_t* member_idx, float* info_mats, float* weights) {
    if (!ctx || !ctx->gaussians_valid) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    const size_t M = (size_t)ctx->M;
    if (seg_offset) HIPCHK(hipMemcpy(seg_offset, ctx->d_seg_off.p, (M + 1) * 4, hipMemcpyDeviceToHost));
    if (member_idx && ctx->Mm > 0) HIPCHK(hipMemcpy(member_idx, ctx->d_memb_idx.p, (size_t)ctx->Mm * 4, hipMemcpyDeviceToHost));
    if ((info_mats || weights) && M > 0) {
        std::vector<float> tmp(M * 12);
        HIPCHK(hipMemcpy(tmp.data(), ctx->d_info12.p, M * 48, hipMemcpyDeviceToHost));
        for (size_t g = 0; g < M; ++g) {
            if (info_mats) std::memcpy(info_mats + 9 * g, tmp.data() + 12 * g, 36);
            if (weights) weights[g] = tmp[12 * g + 9];
        }
    }
    return DMSA_OK;
}

int dmsa_get_global_points(dmsa_ctx* ctx, float* xyz_out, int64_t capacity_points) {
    if (!ctx || ctx->model == MODEL_NONE || !xyz_out || capacity_points < ctx->n) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    HIPCHK(hipMemcpy(xyz_out, ctx->d_global.p, (size_t)ctx->n * 16, hipMemcpyDeviceToHost));
    if (ctx->model == MODEL_WINDOW && ctx->S > 0) {
        // static points are not moved by updateGlobalPoints; they sit (de-centralised again) in the local array
        HIPCHK(hipMemcpy(xyz_out + 4 * ctx->N, ctx->d_local.as<float4>() + ctx->N, (size_t)ctx->S * 16, hipMemcpyDeviceToHost));
        for (int64_t k = ctx->N; k < ctx->n; ++k) xyz_out[4 * k + 3] = 1.0f;
    }
    return DMSA_OK;
}

int dmsa_debug_pow_minus_one(dmsa_ctx* ctx, const int32_t* counts, int32_t count, float* out) {
    if (!ctx || !counts || !out || count < 0) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    int32_t mx = 0;
    for (int32_t i = 0; i < count; ++i) mx = std::max(mx, counts[i]);
    if (mx > (1 << 24)) return DMSA_ERR_INVALID;  // the difference table holds one powf call per count up to the largest: bounded (a Gaussian has < 2^24 members)
    CHK(upload_powm1_codes(ctx, (int64_t)mx + 1));
    DevBuf d_n, d_o;
    HIPCHK(d_n.ensure((size_t)count * 4 + 16));
    HIPCHK(d_o.ensure((size_t)count * 4 + 16));
    HIPCHK(hipMemcpy(d_n.p, counts, (size_t)count * 4, hipMemcpyHostToDevice));
    launch_pow_minus_one(d_n.as<int32_t>(), count, ctx->d_pow_codes.as<uint32_t>(), (int)std::min<int64_t>(ctx->pow_n, INT32_MAX), d_o.as<float>(), ctx->stream);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(out, d_o.p, (size_t)count * 4, hipMemcpyDeviceToHost));
    d_n.release(), d_o.release();
    return DMSA_OK;
}
int dmsa_debug_limit_covariance(dmsa_ctx* ctx, const float* cov9, int64_t count, float* out9, float* evals3, float* V9, int32_t* iterations, int32_t* info) {
    if (!ctx || !cov9 || !out9 || count < 0) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    DevBuf d_in, d_out, d_ev, d_v, d_it, d_info;
    const size_t n = (size_t)count;
    HIPCHK(d_in.ensure(n * 36 + 16));
    HIPCHK(d_out.ensure(n * 36 + 16));
    HIPCHK(d_ev.ensure(n * 12 + 16));
    HIPCHK(d_v.ensure(n * 36 + 16));
    HIPCHK(d_it.ensure(n * 4 + 16));
    HIPCHK(d_info.ensure(n * 4 + 16));
    HIPCHK(hipMemcpy(d_in.p, cov9, n * 36, hipMemcpyHostToDevice));
    launch_debug_limit_covariance(d_in.as<float>(), count, d_out.as<float>(), d_ev.as<float>(), d_v.as<float>(), d_it.as<int32_t>(), d_info.as<int32_t>(), ctx->stream);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(out9, d_out.p, n * 36, hipMemcpyDeviceToHost));
    if (evals3) HIPCHK(hipMemcpy(evals3, d_ev.p, n * 12, hipMemcpyDeviceToHost));
    if (V9) HIPCHK(hipMemcpy(V9, d_v.p, n * 36, hipMemcpyDeviceToHost));
    if (iterations) HIPCHK(hipMemcpy(iterations, d_it.p, n * 4, hipMemcpyDeviceToHost));
    if (info) HIPCHK(hipMemcpy(info, d_info.p, n * 4, hipMemcpyDeviceToHost));
    d_in.release(), d_out.release(), d_ev.release(), d_v.release(), d_it.release(), d_info.release();
    return DMSA_OK;
}
int dmsa_get_debug_counters(dmsa_ctx* ctx, dmsa_debug_counters* out) {
    if (!ctx || !out) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    unsigned long long st[3] = {(unsigned long long)ctx->skip_pairs, 0, 0};
    if (ctx->d_skip_stats.p && ctx->skip_stats_evals > 0) {
        std::vector<unsigned long long> per((size_t)ctx->skip_stats_evals * 2);
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipMemcpy(per.data(), ctx->d_skip_stats.p, per.size() * 8, hipMemcpyDeviceToHost));
        for (size_t k = 0; k < per.size(); k += 2) st[1] += per[k], st[2] += per[k + 1];
    }
    out->sync_retries = ctx->sync_retries, out->speculation_retries = ctx->speculation_retries;
    out->skip_pairs = (int64_t)st[0], out->skip_pairs_equal = (int64_t)st[1], out->skip_mismatches = (int64_t)st[2];
    unsigned long long sp[2] = {0, 0};
    if (ctx->d_split_stats.p) {
        unsigned long long stripes[128];
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream2));
        HIPCHK(hipMemcpy(stripes, ctx->d_split_stats.p, sizeof(stripes), hipMemcpyDeviceToHost));
        for (int k = 0; k < 64; ++k) sp[0] += stripes[2 * k], sp[1] += stripes[2 * k + 1];
    }
    out->split_blocks = (int64_t)sp[0], out->split_blocks_skipped = (int64_t)sp[1];
    unsigned long long changed = 0;
    if (ctx->d_coh_count.p) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipMemcpy(&changed, ctx->d_coh_count.p, 8, hipMemcpyDeviceToHost));
    }
    out->voxel_codes_compared = ctx->coh_compared, out->voxel_codes_changed = (int64_t)changed, out->voxel_lattice_changes = ctx->coh_lattice_changes;
    out->lattice_hints_held = ctx->lattice_hints_held, out->lattice_replays = ctx->lattice_replays;
    out->small_voxel_launches = ctx->small_voxel_launches, out->small_voxel_fallbacks = ctx->small_voxel_fallbacks;
    return DMSA_OK;
}

int dmsa_get_timing(dmsa_ctx* ctx, dmsa_timing* t, int32_t reset) {
    if (!ctx) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    drain_timers(ctx);
    if (t) {
        t->residual_kernel_ms = ctx->t_ms[T_RESIDUAL], t->residual_launches = ctx->residual_launches, t->residual_evaluations = ctx->residual_evals;
        t->residual_algorithmic_bytes = ctx->residual_bytes;
        t->residual_unit_bytes = ctx->residual_unit_bytes;
        t->voxelize_ms = ctx->t_ms[T_VOXEL], t->gaussian_fit_ms = ctx->t_ms[T_FIT], t->pose_table_ms = ctx->t_ms[T_TABLE];
        t->normal_eq_ms = ctx->t_ms[T_NORMAL], t->total_ms = ctx->t_ms[T_TOTAL];
    }
    if (reset) {
        for (double& v : ctx->t_ms) v = 0.0;
        ctx->residual_launches = 0, ctx->residual_evals = 0, ctx->residual_bytes = 0.0, ctx->residual_unit_bytes = 0.0;
    }
    return DMSA_OK;
}

int dmsa_sort_pairs(dmsa_ctx* ctx, const uint32_t* keys, const uint32_t* values, int64_t n, uint32_t end_bit, uint32_t* keys_sorted, uint32_t* values_sorted) {
    if (!ctx || n < 0 || end_bit > 32 || (n > 0 && (!keys || !values || !keys_sorted || !values_sorted))) return DMSA_ERR_INVALID;
    if (n == 0) return DMSA_OK;
    HIPCHK(hipSetDevice(ctx->device));
    DevBuf kin, vin, kout, vout, tmp;
    const size_t bytes = (size_t)n * 4;
    for (DevBuf* b : {&kin, &vin, &kout, &vout}) HIPCHK(b->ensure(bytes));
    HIPCHK(tmp.ensure(sort_pairs_u32_workspace_bytes((size_t)n)));
    HIPCHK(hipMemcpyAsync(kin.p, keys, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(vin.p, values, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(sort_pairs_u32_onesweep(tmp.p, tmp.cap, kin.as<uint32_t>(), kout.as<uint32_t>(), vin.as<uint32_t>(), vout.as<uint32_t>(), (size_t)n, end_bit, ctx->stream));
    HIPCHK(hipMemcpyAsync(keys_sorted, kout.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(values_sorted, vout.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return DMSA_OK;
}

int dmsa_sort_pairs64(dmsa_ctx* ctx, const uint64_t* keys, const uint32_t* values, int64_t n, uint32_t end_bit, uint64_t* keys_sorted, uint32_t* values_sorted) {
    if (!ctx || n < 0 || end_bit > 64 || (n > 0 && (!keys || !values || !keys_sorted || !values_sorted))) return DMSA_ERR_INVALID;
    if (n == 0) return DMSA_OK;
    HIPCHK(hipSetDevice(ctx->device));
    DevBuf kin, vin, kout, vout, tmp;
    HIPCHK(kin.ensure((size_t)n * 8));
    HIPCHK(kout.ensure((size_t)n * 8));
    HIPCHK(vin.ensure((size_t)n * 4));
    HIPCHK(vout.ensure((size_t)n * 4));
    HIPCHK(tmp.ensure(sort_pairs_temp_bytes((size_t)n)));
    HIPCHK(hipMemcpyAsync(kin.p, keys, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(vin.p, values, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(sort_pairs_u64_u32(tmp.p, tmp.cap, kin.as<uint64_t>(), kout.as<uint64_t>(), vin.as<uint32_t>(), vout.as<uint32_t>(), (size_t)n, end_bit, ctx->stream));
    HIPCHK(hipMemcpyAsync(keys_sorted, kout.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(values_sorted, vout.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (DevBuf* b : {&kin, &vin, &kout, &vout, &tmp}) b->release();
    return DMSA_OK;
}

int dmsa_scan_i32(dmsa_ctx* ctx, const int32_t* in, int64_t n, int32_t inclusive, int32_t* out) {
    if (!ctx || n < 0 || (n > 0 && (!in || !out))) return DMSA_ERR_INVALID;
    if (n == 0) return DMSA_OK;
    HIPCHK(hipSetDevice(ctx->device));
    DevBuf din, dout, tmp;
    HIPCHK(din.ensure((size_t)n * 4));
    HIPCHK(dout.ensure((size_t)n * 4));
    HIPCHK(tmp.ensure(scan_temp_bytes((size_t)n)));
    HIPCHK(hipMemcpyAsync(din.p, in, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    if (inclusive)
        HIPCHK(inclusive_scan_i32(tmp.p, tmp.cap, din.as<int32_t>(), dout.as<int32_t>(), (size_t)n, ctx->stream));
    else
        HIPCHK(exclusive_scan_i32(tmp.p, tmp.cap, din.as<int32_t>(), dout.as<int32_t>(), (size_t)n, ctx->stream));
    HIPCHK(hipMemcpyAsync(out, dout.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (DevBuf* b : {&din, &dout, &tmp}) b->release();
    return DMSA_OK;
}

int dmsa_leaf_segments(dmsa_ctx* ctx, const uint32_t* codes_sorted, int64_t n, uint32_t code_bits, int32_t* leaf_of_pos, int32_t* leaf_start, int32_t* num_leaves) {
    if (!ctx || n < 0 || code_bits > 31 || !num_leaves || (n > 0 && (!codes_sorted || !leaf_of_pos || !leaf_start))) return DMSA_ERR_INVALID;
    *num_leaves = 0;
    if (n == 0) return DMSA_OK;
    HIPCHK(hipSetDevice(ctx->device));
    DevBuf code, incl, start, tab, cnt, state;
    HIPCHK(code.ensure((size_t)n * 4));
    HIPCHK(incl.ensure((size_t)n * 4));
    HIPCHK(start.ensure((size_t)(n + 1) * 4));
    HIPCHK(tab.ensure(sizeof(LatticeTable)));
    HIPCHK(cnt.ensure(sizeof(LevelCounts)));
    HIPCHK(state.ensure(8 * (size_t)(1 + leaf_segment_tiles(n))));
    LatticeTable t{};
    t.compressed = 1, t.total_bits = (int)code_bits, t.code_or = 0;
    HIPCHK(hipMemcpyAsync(tab.p, &t, sizeof(t), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(code.p, codes_sorted, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(cnt.p, 0, sizeof(LevelCounts), ctx->stream));
    HIPCHK(hipMemsetAsync(state.p, 0, state.cap, ctx->stream));
    HIPCHK(hipMemsetAsync(incl.p, 0, (size_t)n * 4, ctx->stream));
    // two calls on the same state: the second must ignore what the first left behind (epochs, running ticket)
    const uint32_t tiles = (uint32_t)leaf_segment_tiles(n);
    for (uint32_t call = 0; call < 2; ++call)
        launch_leaf_segments(code.p, true, n, tab.as<LatticeTable>(), incl.as<int32_t>(), start.as<int32_t>(), cnt.as<LevelCounts>(), state.as<unsigned long long>(),
                             call + 1, call * tiles, ctx->stream);
    LevelCounts hc{};
    HIPCHK(hipMemcpyAsync(&hc, cnt.p, sizeof(hc), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(leaf_of_pos, incl.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *num_leaves = hc.num_leaves;
    if (hc.num_leaves > 0) HIPCHK(hipMemcpy(leaf_start, start.p, (size_t)(hc.num_leaves + 1) * 4, hipMemcpyDeviceToHost));
    return DMSA_OK;
}

int dmsa_serial_fallback_sums(dmsa_ctx* ctx, int32_t reset, uint64_t* count) {
    if (!ctx || !count) return DMSA_ERR_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *count = serial_fallback_sums(reset != 0);
    return DMSA_OK;
}

int dmsa_lm_solve(const double* H_damped, const double* g, int32_t P, double alpha, int32_t threads, double* step) {
    if (!H_damped || !g || !step || P < 1) return DMSA_ERR_INVALID;
    if (threads <= 1) {
        lm_solve(H_damped, g, P, alpha, step, nullptr);
        return DMSA_OK;
    }
    const ParallelRun par = [&](const std::function<void(int, int)>& fn) {
        std::vector<std::thread> th;
        for (int t = 0; t < threads; ++t) th.emplace_back([&fn, t, threads]() { fn(t, threads); });
        for (auto& x : th) x.join();
    };
    lm_solve(H_damped, g, P, alpha, step, &par);
    return DMSA_OK;
}

int dmsa_lm_solve_device(dmsa_ctx* ctx, const double* H_damped, const double* g, int32_t P, double alpha, double max_step, double* step, int32_t* nan_out) {
    if (!ctx || !H_damped || !g || !step || P < 1 || P > kLoopPanelMaxP) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    const int n1 = P + 1;
    std::vector<double> Hp((size_t)n1 * n1, 0.0);  // the layout the normal-equation kernels leave: H | g in the last column (and row)
    for (int j = 0; j < P; ++j)
        for (int i = 0; i < P; ++i) Hp[(size_t)j * n1 + i] = H_damped[(size_t)j * P + i];
    for (int i = 0; i < P; ++i) Hp[(size_t)P * n1 + i] = g[i], Hp[(size_t)i * n1 + P] = g[i];
    DevBuf dHp, dstep, dflags;
    HIPCHK(dHp.ensure(Hp.size() * 8));
    HIPCHK(dstep.ensure((size_t)P * 8 + 8));
    HIPCHK(dflags.ensure(sizeof(LoopFlags)));
    HIPCHK(hipMemcpyAsync(dHp.p, Hp.data(), Hp.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(dflags.p, 0, sizeof(LoopFlags), ctx->stream));
    HIPCHK(hipMemsetAsync(dstep.p, 0, (size_t)P * 8, ctx->stream));
    CHK(device_lm_step(ctx, dHp.as<double>(), P, 0.0, alpha, max_step, dstep.as<double>(), dflags.as<LoopFlags>()));
    LoopFlags hf{};
    HIPCHK(hipMemcpyAsync(step, dstep.p, (size_t)P * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(&hf, dflags.p, sizeof(hf), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (nan_out) *nan_out = hf.nan;
    return DMSA_OK;
}

int dmsa_detmath_eval(dmsa_ctx* ctx, int32_t fn, const double* x, const double* y, int64_t n, double* out) {
    if (!ctx || !x || !out || n < 0 || fn < 0 || fn > 3 || (fn == 3 && !y)) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    DevBuf dx, dy, dout;
    HIPCHK(dx.ensure((size_t)n * 8 + 8));
    HIPCHK(dy.ensure((size_t)n * 8 + 8));
    HIPCHK(dout.ensure((size_t)n * 8 + 8));
    HIPCHK(hipMemcpyAsync(dx.p, x, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    if (y) HIPCHK(hipMemcpyAsync(dy.p, y, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    launch_detmath_eval(fn, dx.as<double>(), dy.as<double>(), n, dout.as<double>(), ctx->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dout.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return DMSA_OK;
}
int dmsa_get_trace(dmsa_ctx* ctx, dmsa_iter_trace* out, int32_t capacity) {
    if (!ctx || !out || capacity < 0) return DMSA_ERR_INVALID;
    const int n = std::min<int>(capacity, (int)ctx->trace.size());
    for (int i = 0; i < n; ++i) out[i] = ctx->trace[(size_t)i];
    return n;
}

int dmsa_synchronize(dmsa_ctx* ctx) {
    if (!ctx) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return DMSA_OK;
}

int dmsa_optimize_window(dmsa_ctx* ctx, dmsa_window_problem* p, const dmsa_settings* s, dmsa_report* rep) {
    if (!ctx || !p || !s) return DMSA_ERR_INVALID;
    const bool trace = ctx->dbg.trace_time != 0;
    const auto t0 = std::chrono::steady_clock::now();
    CHK(dmsa_window_upload(ctx, p));
    const auto t1 = std::chrono::steady_clock::now();
    CHK(optimize(ctx, *s, rep));
    const auto t2 = std::chrono::steady_clock::now();
    write_back_poses(ctx->win.ctrl, p->rel_orient, p->rel_transl);
    if (trace)
        std::fprintf(stderr, "[dmsa] optimize_window: upload %.2f ms, optimize %.2f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
                     std::chrono::duration<double, std::milli>(t2 - t1).count());
    return DMSA_OK;
}

int dmsa_optimize_resident(dmsa_ctx* ctx, const dmsa_settings* s, dmsa_report* rep) {
    if (!ctx || !s || ctx->model == MODEL_NONE) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    return optimize(ctx, *s, rep);
}

int dmsa_adaptive_step_size(dmsa_ctx* ctx, double* params, const double* step, double error0, int32_t* best_k) {
    if (!ctx || !params || !step || !best_k) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    return adaptive_step_size(ctx, params, step, error0, best_k);
}

int dmsa_get_poses(dmsa_ctx* ctx, double* rel_orient, double* rel_transl) {
    if (!ctx || ctx->model == MODEL_NONE || !rel_orient || !rel_transl) return DMSA_ERR_INVALID;
    write_back_poses(chain(ctx), rel_orient, rel_transl);
    return DMSA_OK;
}

int dmsa_optimize_keyframes(dmsa_ctx* ctx, dmsa_keyframe_problem* p, const dmsa_settings* s, dmsa_report* rep) {
    if (!ctx || !p || !s) return DMSA_ERR_INVALID;
    CHK(dmsa_keyframes_upload(ctx, p));
    CHK(optimize(ctx, *s, rep));
    write_back_poses(ctx->key.frames, p->rel_orient, p->rel_transl);
    return DMSA_OK;
}


}  // extern "C"
