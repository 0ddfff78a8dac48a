// optimize_loop.cpp — DmsaOptimizer::optimizeSet (DmsaOptimizer.h:54-150): pose tables and residual batches, the device-resident loop
// (default) and the host-driven loop (host-built pose tables, debug switch device_loop = 0, sets too large for the chain kernels' LDS).
#include "dmsa_ctx.h"

// ---- pose tables ------------------------------------------------------------------------------------------
// `globs`: B x (C or F) x 6 doubles (axis-angle | translation) of the GLOBAL poses of every evaluation in the batch.
int build_tables(dmsa_ctx* ctx, int B, const std::vector<double>& globs, hipStream_t stream) {
    if (stream == nullptr) stream = ctx->stream;
    if (ctx->tables_pending && stream == ctx->stream) {  // an earlier batch's tables may still be in flight on the second stream
        if (ctx->tables_dev_sync)
            enqueue_wait(ctx, SYNC_TABLES, ctx->stream);
        else
            HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_tables, 0));
        ctx->tables_pending = false;
    }
    ScopedTimer tm(ctx, T_TABLE);
    const int rows = ctx->rows;
    HIPCHK(ctx->d_tables.ensure((size_t)B * rows * 48));
    const int np = ctx->model == MODEL_WINDOW ? ctx->win.ctrl.n : ctx->key.frames.n;
    if (ctx->flags & DMSA_FLAG_POSE_TABLE_HOST) {
        ctx->h_tables.resize((size_t)B * rows * 12);
        auto build_range = [&](int b0, int b1) {
            PoseChain tmp;
            tmp.resize(np);
            for (int b = b0; b < b1; ++b) {
                for (int k = 0; k < np; ++k)
                    for (int c = 0; c < 3; ++c) {
                        tmp.glob_o[3 * k + c] = globs[((size_t)b * np + k) * 6 + c];
                        tmp.glob_t[3 * k + c] = globs[((size_t)b * np + k) * 6 + 3 + c];
                    }
                float* T = &ctx->h_tables[(size_t)b * rows * 12];
                if (ctx->model == MODEL_WINDOW)
                    window_dense_table(tmp, ctx->win.stamps, ctx->win.fh, ctx->win.traj_time, T);
                else
                    keyframe_table(tmp, T);
                float* id = T + (size_t)(rows - 1) * 12;  // identity row used by static points
                const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
                std::memcpy(id, I, sizeof(I));
            }
        };
        // every table is a pure function of its control poses: build the tables of a batch on several host threads
        if ((size_t)B * rows < 4096) {
            build_range(0, B);
        } else {
            workers(ctx).run_all([&](int t, int nt) { build_range((int)((int64_t)B * t / nt), (int)((int64_t)B * (t + 1) / nt)); });
        }
        HIPCHK(hipMemcpyAsync(ctx->d_tables.p, ctx->h_tables.data(), ctx->h_tables.size() * 4, hipMemcpyHostToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream));  // h_tables is reused by the next batch
    } else {
        HIPCHK(ctx->d_ctrl.ensure(globs.size() * 8));
        constexpr int kPinSlots = 4;  // at least one stream synchronisation separates reuse of a slot (4 syncs per iteration)
        if (globs.size() > ctx->h_pin_slot) {
            HIPCHK(hipStreamSynchronize(ctx->stream));
            if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
            ctx->h_pin = nullptr;
            ctx->h_pin_slot = globs.size() + globs.size() / 2 + 64;
            HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_pin), ctx->h_pin_slot * kPinSlots * sizeof(double), hipHostMallocDefault));
        }
        double* slot = ctx->h_pin + (size_t)ctx->h_pin_next * ctx->h_pin_slot;
        ctx->h_pin_next = (ctx->h_pin_next + 1) % kPinSlots;
        std::memcpy(slot, globs.data(), globs.size() * 8);
        HIPCHK(hipMemcpyAsync(ctx->d_ctrl.p, slot, globs.size() * 8, hipMemcpyHostToDevice, stream));
        // default path: the correspondence kernels read the tables transposed ([row][evaluation][12]); batches are written both ways at once
        float* tT = nullptr;
        if (B > 1) {
            HIPCHK(ctx->d_tablesT.ensure((size_t)B * rows * 48));
            tT = ctx->d_tablesT.as<float>();
        }
        if (ctx->model == MODEL_WINDOW)
            launch_window_pose_tables(ctx->d_ctrl.as<double>(), ctx->d_stamps.as<double>(), ctx->d_fhw.as<double>(), ctx->d_trajtime.as<double>(), B,
                                      np, rows - 1, ctx->d_tables.as<float>(), tT, stream);
        else
            launch_keyframe_pose_tables(ctx->d_ctrl.as<double>(), B, np, ctx->d_tables.as<float>(), tT, stream);
        ctx->batch = B;
        ctx->tablesT_batch = tT ? B : 0;
        return DMSA_OK;
    }
    ctx->batch = B;
    ctx->tablesT_batch = 0;  // the transposed copy (default path) no longer matches
    return DMSA_OK;
}

void append_glob(const PoseChain& c, std::vector<double>& out) {
    for (int k = 0; k < c.n; ++k) {
        for (int a = 0; a < 3; ++a) out.push_back(c.glob_o[3 * k + a]);
        for (int a = 0; a < 3; ++a) out.push_back(c.glob_t[3 * k + a]);
    }
}

// one forward evaluation's host part for the CURRENT chain state: record global poses, compute additional rows
void host_eval(dmsa_ctx* ctx, std::vector<double>& globs, std::vector<double>& extra) {
    append_glob(chain(ctx), globs);
    const int a = num_extra_rows(ctx);
    if (a > 0) {
        const size_t at = extra.size();
        extra.resize(at + a);
        if (ctx->model == MODEL_WINDOW)
            ctx->win.imu_rows(&extra[at]);  // runs global_to_relative like updateImuError
        else
            ctx->key.additional_rows(&extra[at]);
    }
    ctx->evaluations += 1;
}
// setPoseParameters + the chain update of updateGlobalPoints for both models
void host_set_params(dmsa_ctx* ctx, const double* p) {
    chain(ctx).set_params(p);
    chain(ctx).relative_to_global();
}

int transform_points(dmsa_ctx* ctx, int b) {
    const float4* table = ctx->d_tables.as<float4>() + (size_t)b * ctx->rows * 3;
    ctx->base_table = reinterpret_cast<const float*>(table);  // the fit re-derives the global coordinates of the members from this table
    if (ctx->model == MODEL_KEYFRAMES)
        launch_transform_normals(ctx->d_local.as<float4>(), ctx->d_nlocal.as<float4>(), table, ctx->d_global.as<float4>(), ctx->d_nglobal.as<float4>(),
                                 ctx->n, ctx->stream);
    else
        launch_transform(ctx->d_local.as<float4>(), table, ctx->d_global.as<float4>(), ctx->n, ctx->stream);
    HIPCHK(hipGetLastError());
    return DMSA_OK;
}


// ---- residual batches ----------------------------------------------------------------------------------------
int ensure_E(dmsa_ctx* ctx, int B) {
    const int a = num_extra_rows(ctx);
    ctx->extra_rows = a;
    const int64_t ld = (((int64_t)ctx->M + a) + 31) / 32 * 32;
    ctx->ldE = ld;
    HIPCHK(ctx->d_E.ensure((size_t)B * ld * 8));
    return DMSA_OK;
}
int run_residuals(dmsa_ctx* ctx, int B, const std::vector<double>* extra, const double* d_extra, const uint32_t* rot_same, const int2* row_range) {
    CHK(ensure_E(ctx, B));
    if (ctx->tables_pending) {  // the pose tables of this batch were built on another stream (and k_size_classes did not wait for them)
        if (ctx->tables_dev_sync)
            enqueue_wait(ctx, SYNC_TABLES, ctx->stream);
        else
            HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_tables, 0));
        ctx->tables_pending = false;
    }
    const int a = ctx->extra_rows;
    if (a > 0 && d_extra != nullptr)  // device loop: the chain kernels left the additional rows of the batch in device memory
        launch_loop_scatter_extra(d_extra, B, a, ctx->d_E.as<double>(), ctx->ldE, ctx->M, ctx->stream);
    if (a > 0 && extra != nullptr) {
        // additional rows (IMU / gravity / odometry) go below the Gaussian rows of every evaluation, through a pinned ring like the
        // control poses: no host synchronisation, and the copy runs ahead of the correspondence kernels
        constexpr int kPinSlots = 4;  // at least one stream synchronisation separates reuse of a slot
        const size_t need = (size_t)a * B;
        if (need > ctx->h_xpin_slot) {
            HIPCHK(hipStreamSynchronize(ctx->stream));
            if (ctx->h_xpin) (void)hipHostFree(ctx->h_xpin);
            ctx->h_xpin = nullptr;
            ctx->h_xpin_slot = need + need / 2 + 64;
            HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_xpin), ctx->h_xpin_slot * kPinSlots * sizeof(double), hipHostMallocDefault));
        }
        double* slot = ctx->h_xpin + (size_t)ctx->h_xpin_next * ctx->h_xpin_slot;
        ctx->h_xpin_next = (ctx->h_xpin_next + 1) % kPinSlots;
        std::memcpy(slot, extra->data(), need * sizeof(double));
        HIPCHK(hipMemcpy2DAsync(ctx->d_E.as<double>() + ctx->M, (size_t)ctx->ldE * 8, slot, (size_t)a * 8, (size_t)a * 8, (size_t)B, hipMemcpyHostToDevice,
                                ctx->stream));
    }
    if (!ctx->order_valid) {
        ctx->err = "residuals: no size-class order (build the Gaussians first)";
        return DMSA_ERR_INVALID;
    }
    {
        // reference-order sums (default path): lane = evaluation on transposed pose tables
        HIPCHK(ctx->d_tablesT.ensure((size_t)B * ctx->rows * 48));
        ScopedTimer tm(ctx, T_RESIDUAL);
        if (ctx->tablesT_batch != B) launch_transpose_tables(ctx->d_tables.as<float>(), ctx->rows, B, ctx->d_tablesT.as<float>(), ctx->stream);
        const bool two = ctx->serial_two_streams && ctx->serial_counts.n_long > 0;
        const bool three = two && ctx->serial_three_streams;
        // The latency tier keeps `stream` and is launched first: its workgroups must get their CUs before the thousands of workgroups of
        // the other tiers fill the chip.  (Giving `stream` to the throughput tier in the Jacobian batch, which that tier bounds, so that it
        // starts without the ~12 us fork delay: 1050 -> 990 it/s -- the latency tier then queues behind everybody else.)
        hipStream_t s_long = ctx->stream;
        hipStream_t s_mid = two ? ctx->stream2 : ctx->stream;
        hipStream_t s_small = three ? ctx->stream3 : s_mid;
        // fork and join of the tier streams: counters in device memory (loop_kernels.h) instead of events, whose barrier packets cost
        // 8-10 us per record / wait on `stream`; debug switch device_sync = 0 keeps the events
        const bool dev_sync = two && ctx->dbg.device_sync != 0;
        const int2* gauss_rows = row_range ? ctx->d_gauss_rows.as<int2>() : nullptr;
        uint32_t* d_sync = nullptr;
        if (dev_sync) {
            d_sync = ctx->d_sync.as<uint32_t>();  // zeroed when the context was created
            // the fork signal is given by the latency tier itself once all its workgroups are placed (serial_kernels.hip); its launch is
            // enqueued before the waits (the order that rules out a deadlock on shared hardware queues)
            ctx->sync_sig[SYNC_TIER_FORK] += 1;
        } else if (two) {
            HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
            HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
            if (three) HIPCHK(hipStreamWaitEvent(ctx->stream3, ctx->ev_fork, 0));
        }
        // the second pass of the longest Gaussians shared out to helper workgroups (serial_kernels.hip: k_residuals_chain, LongHelp)
        LongSplit split_store;
        const LongSplit* split = nullptr;
        // Measured (round 6): where a FEW long Gaussians end a batch (the rosette window: 2 - 3 of them) the helpers take 10 % off the iteration; the
        // bench window's 110 long Gaussians keep ~220 compute units busy side by side and the helpers of the longest ones gain nothing there
        // (1235 -> 1200-1218 it/s), nor for the keyframe sets (4 long Gaussians, B = 187 and 9: 985 -> 953).  Rule: window model, at most 16 long Gaussians (= the Gaussians that get helper blocks), B <= 32;
        // long_split >= 2 = that many members as the threshold, everywhere (experiments).
        const bool split_auto = ctx->dbg.long_split == 1 && ctx->serial_counts.n_long <= 16 && B <= 32 && ctx->model == MODEL_WINDOW;
        if ((split_auto || ctx->dbg.long_split >= 2) && ctx->serial_counts.n_long > 0 && ctx->dbg.serial_tree != 0) {
            const SerialShape shp = serial_shape(B);
            const size_t items = (size_t)ctx->serial_counts.n_long * shp.nsub_long;
            split_store.helpers = 8;
            split_store.lead_gaussians = 16;
            // few long Gaussians (rosette window, keyframe sets): every one of them may hand over; many (the bench window has 110): the longest
            split_store.min_members = ctx->dbg.long_split >= 2 ? ctx->dbg.long_split : 0;  // (auto: every Gaussian of the latency tier)
            const size_t H = (size_t)split_store.helpers;
            if (items > ctx->long_split_items) {  // layout by CAPACITY: the tickets keep their place (and their zeros) when the item count changes between batches
                const size_t cap = items + items / 2 + 16;
                HIPCHK(ctx->d_long_split.ensure(cap * (8 + 48 * 4 + H * 16 * 12) + 64));
                HIPCHK(hipMemsetAsync(ctx->d_long_split.p, 0, ctx->d_long_split.cap, ctx->stream));  // tickets at zero, flags below every epoch
                ctx->long_split_items = cap;
            }
            const size_t cap = ctx->long_split_items;
            char* w = ctx->d_long_split.as<char>();
            split_store.partial = reinterpret_cast<double*>(w), w += cap * H * 16 * 8;
            split_store.means = reinterpret_cast<float*>(w), w += cap * 48 * 4;
            split_store.partial_key = reinterpret_cast<int*>(w), w += cap * H * 16 * 4;
            split_store.done = reinterpret_cast<uint32_t*>(w), w += cap * 4;
            split_store.ready = reinterpret_cast<uint32_t*>(w);
            split_store.timed_out = ctx->sync_timed_out();
            ctx->long_split_epoch += 1;
            if (ctx->long_split_epoch == 0) ctx->long_split_epoch = 1;
            split_store.epoch = ctx->long_split_epoch;
            split = &split_store;
        }
        if (dev_sync) {
            // latency tier first (with the signal), then the waits in front of the other tiers
            launch_residuals_serial(ctx->d_memb_local.as<float4>(), ctx->d_seg_off.as<int32_t>(), ctx->d_info12.as<float>(), ctx->d_tablesT.as<float>(), B,
                                    ctx->d_order.as<uint32_t>(), ctx->serial_counts, ctx->d_E.as<double>(), ctx->ldE, s_long, s_mid, s_small, ctx->dbg.serial_tree,
                                    d_sync + SYNC_TIER_FORK, 1, rot_same, row_range, gauss_rows, split);
            enqueue_wait(ctx, SYNC_TIER_FORK, ctx->stream2);
            if (three) enqueue_wait(ctx, SYNC_TIER_FORK, ctx->stream3);
            launch_residuals_serial(ctx->d_memb_local.as<float4>(), ctx->d_seg_off.as<int32_t>(), ctx->d_info12.as<float>(), ctx->d_tablesT.as<float>(), B,
                                    ctx->d_order.as<uint32_t>(), ctx->serial_counts, ctx->d_E.as<double>(), ctx->ldE, s_long, s_mid, s_small, ctx->dbg.serial_tree, nullptr, 6,
                                    rot_same, row_range, gauss_rows);
        } else {
            launch_residuals_serial(ctx->d_memb_local.as<float4>(), ctx->d_seg_off.as<int32_t>(), ctx->d_info12.as<float>(), ctx->d_tablesT.as<float>(), B,
                                    ctx->d_order.as<uint32_t>(), ctx->serial_counts, ctx->d_E.as<double>(), ctx->ldE, s_long, s_mid, s_small, ctx->dbg.serial_tree,
                                    nullptr, 7, rot_same, row_range, gauss_rows, split);
        }
        if (dev_sync) {
            launch_sync_signal(d_sync + SYNC_TIER_JOIN, ctx->stream2);
            if (three) launch_sync_signal(d_sync + SYNC_TIER_JOIN, ctx->stream3);
            ctx->sync_sig[SYNC_TIER_JOIN] += three ? 2 : 1;
            enqueue_wait(ctx, SYNC_TIER_JOIN, ctx->stream);
        } else {
            // joins: the stream that finishes first is waited for first (its wait is through while `stream` still works)
            if (three) {
                HIPCHK(hipEventRecord(ctx->ev_join3, ctx->stream3));
                HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join3, 0));
            }
            if (two) {
                HIPCHK(hipEventRecord(ctx->ev_join, ctx->stream2));
                HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
            }
        }
    }
    ctx->E_is_jacobian = false;
    ctx->residual_launches += 1;
    ctx->residual_evals += B;
    ctx->residual_bytes += 16.0 * (double)ctx->Mm + 48.0 * ctx->M + (double)B * (48.0 * ctx->rows + 8.0 * ctx->M);
    ctx->residual_unit_bytes += (double)B * (16.0 * (double)ctx->Mm + 56.0 * ctx->M + 48.0 * ctx->rows);
    HIPCHK(hipGetLastError());
    return DMSA_OK;
}


// adaptiveStepSize (DmsaOptimizer.h:152-182) on the resident problem and its current Gaussians, host-driven: nine trial evaluations at
// raw + 0.1 k step in one batch; `paramVec` (in: raw) becomes the arg-min if it beats error0 (strict '<'), *bestK its k (0: none -- the set is
// then left at the LAST trial, raw + 0.9 step, like the reference's object, :160-165).  Used by the host-driven loop and by the C ABI's
// dmsa_adaptive_step_size (the reference's method is public, DmsaOptimizer.h:152).
static int host_adaptive_step_size(dmsa_ctx* ctx, int P, int rowsE, std::vector<double>& paramVec, const std::vector<double>& step, double error0, int* bestK_out) {
    std::vector<double> globs, extra, test((size_t)P);
    int bestK = 0;
        globs.clear(), extra.clear();
        if (ctx->model == MODEL_KEYFRAMES && P >= 48) {
            // like the Jacobian batch: the keyframe model carries nothing from one evaluation to the next, so the nine trial chains are
            // built side by side; the chain is left where the serial loop leaves it (last trial evaluated)
            const int a = num_extra_rows(ctx);
            const size_t gsz = (size_t)chain(ctx).n * 6;
            globs.resize(9 * gsz);
            extra.resize((size_t)9 * a);
            const KeyframeHost base = ctx->key;
            workers(ctx).run_all([&](int t, int nthr) {
                KeyframeHost kh = base;
                std::vector<double> tp((size_t)P), g;
                for (int k = 1 + t; k < 10; k += nthr) {
                    for (int i = 0; i < P; ++i) tp[(size_t)i] = paramVec[(size_t)i] + 0.1 * (double)k * step[(size_t)i];
                    kh.frames.set_params(tp.data());
                    kh.frames.relative_to_global();
                    g.clear();
                    append_glob(kh.frames, g);
                    std::copy(g.begin(), g.end(), globs.begin() + (size_t)(k - 1) * gsz);
                    if (a > 0) kh.additional_rows(&extra[(size_t)(k - 1) * a]);
                }
            });
            ctx->evaluations += 9;
            for (int i = 0; i < P; ++i) test[(size_t)i] = paramVec[(size_t)i] + 0.1 * 9.0 * step[(size_t)i];
            host_set_params(ctx, test.data());
        } else {
            for (int k = 1; k < 10; ++k) {
                for (int i = 0; i < P; ++i) test[(size_t)i] = paramVec[(size_t)i] + 0.1 * (double)k * step[(size_t)i];
                host_set_params(ctx, test.data());
                host_eval(ctx, globs, extra);
            }
        }
        g_tl.mark("trial chains");
        CHK(build_tables(ctx, 9, globs));
        CHK(run_residuals(ctx, 9, &extra));
        double* errs = ctx->h_rb->errs;  // pinned
        {
            ScopedTimer tm(ctx, T_NORMAL);
            HIPCHK(ctx->d_sq_partial.ensure((size_t)squared_sums_blocked_partial_doubles(rowsE, P, 9) * 8));
            HIPCHK(ctx->d_sq_out.ensure(16 * 8));
            launch_squared_sums_blocked(ctx->d_E.as<double>(), ctx->ldE, rowsE, P, 9, ctx->d_sq_partial.as<double>(), ctx->d_sq_out.as<double>(), ctx->stream);
        }
        HIPCHK(hipMemcpyAsync(errs, ctx->d_sq_out.p, 9 * 8, hipMemcpyDeviceToHost, ctx->stream));
        g_tl.mark("line search enq");
        HIPCHK(sync_spin(ctx->stream));  // sync #4
        g_tl.mark("sync#4 wait");
        drain_timers(ctx);
        double minError = error0;
        bestK = 0;
        const std::vector<double> raw = paramVec;
        for (int k = 1; k < 10; ++k)
            if (errs[k - 1] < minError) {
                for (int i = 0; i < P; ++i) paramVec[(size_t)i] = raw[(size_t)i] + 0.1 * (double)k * step[(size_t)i];
                minError = errs[k - 1], bestK = k;
            }
    *bestK_out = bestK;
    return DMSA_OK;
}

// C ABI seam of adaptiveStepSize (include/dmsa_hip.h: dmsa_adaptive_step_size)
int adaptive_step_size(dmsa_ctx* ctx, double* params, const double* step, double error0, int32_t* best_k) {
    if (ctx->model == MODEL_NONE || !ctx->gaussians_valid || !ctx->order_valid) {
        ctx->err = "adaptive_step_size: no Gaussians (dmsa_build_gaussians or an optimize call first)";
        return DMSA_ERR_INVALID;
    }
    const int P = num_params(ctx);
    std::vector<double> pv(params, params + P), st(step, step + P);
    int k = 0;
    CHK(host_adaptive_step_size(ctx, P, ctx->M + ctx->extra_rows, pv, st, error0, &k));
    std::copy(pv.begin(), pv.end(), params);
    *best_k = k;
    return DMSA_OK;
}

// ---- the optimizeSet loop (DmsaOptimizer.h:54-150) -------------------------------------------------------------
static int optimize_impl(dmsa_ctx* ctx, const dmsa_settings& s, dmsa_report* rep);
// A failure inside the loop (HIP error, lattice deeper than 21 levels, allocation) must not leave the resident problem in the centred
// frame: the static points were shifted in place and the window origin lives only in the context.
static int optimize_device_loop(dmsa_ctx* ctx, const dmsa_settings& s, dmsa_report* rep);
// What a whole call may have to start over from (a device-side wait that gave up means a consumer ran before its producers: nothing of
// that run can be trusted): the host model state, and the static points as they were -- centralize / decentralize shifts them in float
// and is no exact round trip, so a second run on the shifted-back points would not be the run the caller asked for.
struct CallSnapshot {
    bool taken = false;
    WindowHost win;
    KeyframeHost key;
    bool centralized = false;
};
static int take_snapshot(dmsa_ctx* ctx, const dmsa_settings& s, CallSnapshot& snap) {
    snap.win = ctx->win, snap.key = ctx->key, snap.centralized = ctx->centralized;
    if (ctx->model == MODEL_WINDOW && s.use_centralization && ctx->S > 0) {
        HIPCHK(ctx->d_static_keep.ensure((size_t)ctx->S * 16));
        HIPCHK(hipMemcpyAsync(ctx->d_static_keep.p, ctx->d_local.as<float4>() + ctx->N, (size_t)ctx->S * 16, hipMemcpyDeviceToDevice, ctx->stream));
    }
    snap.taken = true;
    return DMSA_OK;
}
static int restore_snapshot(dmsa_ctx* ctx, const dmsa_settings& s, const CallSnapshot& snap) {
    ctx->win = snap.win, ctx->key = snap.key, ctx->centralized = snap.centralized;
    if (ctx->model == MODEL_WINDOW && s.use_centralization && ctx->S > 0) {
        HIPCHK(hipMemcpyAsync(ctx->d_local.as<float4>() + ctx->N, ctx->d_static_keep.p, (size_t)ctx->S * 16, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    ctx->gaussians_valid = false, ctx->order_valid = false, ctx->aabb_fresh = false;
    return DMSA_OK;
}
int optimize(dmsa_ctx* ctx, const dmsa_settings& s, dmsa_report* rep) {
    // the loop state lives on the device (one host wait per iteration); the host-driven loop remains for host-built pose tables, for the
    // debug switch device_loop = 0 and for sets whose chain state does not fit the chain kernels' LDS (hundreds of keyframes)
    const bool device_loop = !(ctx->flags & DMSA_FLAG_POSE_TABLE_HOST) && ctx->device_loop && loop_chain_fits(ctx->loop_model);
    auto run = [&]() {
        ctx->wait_seq = 0, ctx->voxel_calls = 0;
        int rc = device_loop ? optimize_device_loop(ctx, s, rep) : optimize_impl(ctx, s, rep);
        if (rc != DMSA_OK && ctx->centralized) {  // a failure inside the loop must not leave the resident problem in the centred frame
            const std::string err = ctx->err;
            (void)dmsa_decentralize(ctx);
            ctx->err = err;
        }
        return rc;
    };
    CallSnapshot snap;
    if (ctx->dbg.device_sync != 0) CHK(take_snapshot(ctx, s, snap));
    int rc = run();
    // The flag is looked at on EVERY exit: a call that failed for another reason after a wait gave up (or after a signal was counted but
    // never enqueued) would otherwise hand a stale flag and mismatched targets to the next, healthy call.
    std::string what;
    if (ctx->dbg.device_sync != 0 && sync_wait_timed_out(ctx, &what)) {
        // Counters in device memory need the producer's queue to make progress while the consumer's wait spins; a tool that serialises
        // kernels across queues in an order of its own (hardware counter collection) breaks that.  The call is run again from the state it
        // started from with event dependencies, and the context keeps them from now on.
        const std::string first = rc == DMSA_OK ? std::string() : " (first attempt: " + ctx->err + ")";
        ctx->dbg.device_sync = 0;
        ctx->sync_retries += 1;
        CHK(restore_snapshot(ctx, s, snap));
        rc = run();
        const std::string warn = "warning: a device-side stream dependency timed out (" + what + "); the call was run again with event dependencies, which this context "
                                 "uses from now on (DMSA_DEBUG=device_sync=0 selects them from the start)" + first;
        ctx->err = rc == DMSA_OK ? warn : ctx->err + " | " + warn;
    }
    return rc;
}
static int optimize_impl(dmsa_ctx* ctx, const dmsa_settings& s, dmsa_report* rep) {
    ScopedTimer total(ctx, T_TOTAL);
    const bool fixed = (ctx->flags & DMSA_FLAG_FIXED_ITERS) != 0;
    const int P = num_params(ctx);
    std::vector<double> paramVec((size_t)P), origin((size_t)P), loop((size_t)P), step((size_t)P), test((size_t)P), globs, extra;
    std::vector<double> Hp((size_t)(P + 1) * (P + 1)), H((size_t)P * P), g((size_t)P);
    int stop = DMSA_STOP_NUM_ITER, iters = 0, bestK = 0;
    double error0 = 0.0, stepNorm = 0.0;
    ctx->evaluations = 0;
    ctx->trace.clear();
    const double increment = 1.0 * std::sqrt((double)std::numeric_limits<float>::epsilon());
    const double one_div_incr = 1.0 / increment;

    if (s.use_centralization) CHK(dmsa_centralize(ctx));
    HIPCHK(ctx->d_tables.ensure((size_t)(P + 1) * ctx->rows * 48));  // never reallocated while kernels read it
    for (int iter = 0; iter < s.num_iter; ++iter) {
        ++iters;
        g_tl.on = ctx->dbg.host_timeline != 0, g_tl.reset(), g_tl.mark("start");
        chain(ctx).get_params(paramVec.data());  // :72
        // :75 updateGlobalPoints (the window model re-chains here, the keyframe model did in setPoseParameters)
        if (ctx->model == MODEL_WINDOW) chain(ctx).relative_to_global();
        globs.clear();
        append_glob(chain(ctx), globs);
        CHK(build_tables(ctx, 1, globs));
        CHK(transform_points(ctx, 0));
        g_tl.mark("table0+transform enq");
        // Host part of evaluation 0 (:99) and of the P forward-difference evaluations of calcNumericJacobian (:199-232):
        // one batch of 1+P pose tables.
        auto jacobian_batch = [&]() -> int {
            globs.clear(), extra.clear();
            host_eval(ctx, globs, extra);
            chain(ctx).get_params(origin.data());  // :204 (after updateImuError's global2relative round trip)
            if (ctx->model == MODEL_KEYFRAMES && P >= 48) {
                // The keyframe model carries no state from one evaluation to the next (setPoseParameters rewrites every
                // relative pose and re-chains, MapManagement.h:197-202), so the P perturbed chains (O(F) exp/log each) are
                // built by a few host threads; results are identical to the serial order.
                const int a = num_extra_rows(ctx);
                const size_t gsz = (size_t)chain(ctx).n * 6;
                globs.resize((size_t)(1 + P) * gsz);
                extra.resize((size_t)(1 + P) * a);
                const KeyframeHost base = ctx->key;
                workers(ctx).run_all([&](int t, int nthr) {
                    KeyframeHost kh = base;
                    std::vector<double> lp(origin), g;
                    for (int k = t; k < P; k += nthr) {
                        lp = origin;
                        lp[(size_t)k] += increment;
                        kh.frames.set_params(lp.data());
                        kh.frames.relative_to_global();
                        g.clear();
                        append_glob(kh.frames, g);
                        std::copy(g.begin(), g.end(), globs.begin() + (size_t)(1 + k) * gsz);
                        if (a > 0) kh.additional_rows(&extra[(size_t)(1 + k) * a]);
                    }
                });
                ctx->evaluations += P;
                // leave the chain where the serial loop would: last perturbation evaluated, then parameters restored
                loop = origin;
                loop[(size_t)(P - 1)] += increment;
                host_set_params(ctx, loop.data());
            } else {
                for (int k = 0; k < P; ++k) {
                    loop = origin;
                    loop[(size_t)k] += increment;
                    host_set_params(ctx, loop.data());
                    host_eval(ctx, globs, extra);
                }
            }
            chain(ctx).set_params(origin.data());  // :231
            // The tables of the batch (and, on the default path, their transposed copy) depend on nothing the GPU is busy with: they go
            // to another stream, beside the voxelisation, instead of between the fit and the correspondence kernels.
            // On the main stream: the fit launched before and after this point reads table 0 of d_tables, which this batch rewrites (with
            // the same bits) -- in stream order that is no race.  (The device-resident loop keeps the base table in its own buffer and
            // builds the batch beside the voxelisation.)
            hipStream_t ts = ctx->stream;
            CHK(build_tables(ctx, 1 + P, globs, ts));
            HIPCHK(hipEventRecord(ctx->ev_tables, ts));
            ctx->tables_pending = ts != ctx->stream, ctx->tables_dev_sync = false;
            return DMSA_OK;
        };
        // The batch does not depend on the Gaussians, so its host math (on the parity path: 1 + P libm pose tables) and the
        // pose-table upload / kernel are issued while the GPU is still voxelising (table 0 of the batch equals the base table the
        // fit reads).  The host-side order of evaluations is the reference's either way; only the early exit below has to undo it.
        const bool overlap = ctx->overlap_batch;
        const int evals_before = ctx->evaluations;
        const PoseChain chain_before = chain(ctx);  // exact undo, incl. the pose-0 round trip updateImuError leaves behind
        if (overlap)
            CHK(build_gaussians(ctx, s, jacobian_batch));  // :78-86, :96
        else
            CHK(build_gaussians(ctx, s));
        g_tl.mark("build_gaussians (incl. sync#2)");
        ctx->trace.push_back(dmsa_iter_trace{ctx->M, ctx->M1, ctx->Mm, 0.0, 0.0, 0, 0});
        if (ctx->M < s.min_num_gaussians) {  // :89-93
            stop = DMSA_STOP_FEW_GAUSSIANS;
            if (overlap) {  // undo the speculative batch: the reference had not evaluated anything in this iteration
                ctx->evaluations = evals_before;
                chain(ctx) = chain_before;
            }
            break;
        }
        if (!overlap) CHK(jacobian_batch());
        CHK(run_residuals(ctx, 1 + P, &extra));
        const int rowsE = ctx->M + ctx->extra_rows;
        {
            ScopedTimer tm(ctx, T_NORMAL);
            HIPCHK(ctx->d_ne_partial.ensure((size_t)normal_equations_partial_doubles(rowsE, P) * 8));
            HIPCHK(ctx->d_Hp.ensure(Hp.size() * 8));
            launch_normal_equations(ctx->d_E.as<double>(), ctx->ldE, rowsE, P, one_div_incr, ctx->d_ne_partial.as<double>(), ctx->d_Hp.as<double>(), ctx->stream);
        }
        if (Hp.size() > ctx->h_Hp_cap) {
            if (ctx->h_Hp) (void)hipHostFree(ctx->h_Hp);
            ctx->h_Hp = nullptr, ctx->h_Hp_cap = 0;
            HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_Hp), Hp.size() * 8, hipHostMallocDefault));
            ctx->h_Hp_cap = Hp.size();
        }
        HIPCHK(hipMemcpyAsync(ctx->h_Hp, ctx->d_Hp.p, Hp.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
        g_tl.mark("residuals+NE enq");
        HIPCHK(sync_spin(ctx->stream));  // sync #3
        g_tl.mark("sync#3 wait");
        std::memcpy(Hp.data(), ctx->h_Hp, Hp.size() * 8);
        const int n1 = P + 1;
        for (int j = 0; j < P; ++j)
            for (int i = 0; i < P; ++i) H[(size_t)j * P + i] = Hp[(size_t)j * n1 + i];
        for (int i = 0; i < P; ++i) g[(size_t)i] = Hp[(size_t)P * n1 + i];
        error0 = Hp[(size_t)P * n1 + P];  // :101
        for (int i = 0; i < P; ++i) H[(size_t)i * P + i] += (double)s.lambda_diag;  // :110
        {   // :113, explicit inverse like the reference
            const ParallelRun par = [&](const std::function<void(int, int)>& fn) { workers(ctx).run_all(fn); };
            lm_solve(H.data(), g.data(), P, s.step_length_optim, step.data(), P >= 64 ? &par : nullptr, ctx->dbg.solve_threads);
        }
        g_tl.mark("assemble+solve");
        bool anyNan = false;
        for (double v : step) anyNan = anyNan || std::isnan(v);
        if (anyNan) {  // :116-122 setPoseParameters(paramVec); break
            chain(ctx).set_params(paramVec.data());
            if (ctx->model == MODEL_KEYFRAMES) chain(ctx).relative_to_global();
            stop = DMSA_STOP_NAN;
            break;
        }
        double mx = -std::numeric_limits<double>::infinity(), mn = std::numeric_limits<double>::infinity();
        for (double v : step) mx = std::max(mx, v), mn = std::min(mn, v);
        const double maxElem = std::max(mx, -mn);  // :125
        if (maxElem > s.max_step)
            for (double& v : step) v = (s.max_step / maxElem) * v;
        // adaptiveStepSize (:152-182): nine trial evaluations in one batch
        CHK(host_adaptive_step_size(ctx, P, rowsE, paramVec, step, error0, &bestK));
        double ss = 0.0;
        for (double v : step) ss += v * v;
        stepNorm = std::sqrt(ss);
        ctx->trace.back().error0 = error0, ctx->trace.back().step_norm = stepNorm, ctx->trace.back().best_k = bestK;
        if (bestK == 0 && !fixed) {  // :130-134 — the set is left at raw + 0.9*step (last trial), not restored
            stop = DMSA_STOP_NO_IMPROVEMENT;
            break;
        }
        // :136 setPoseParameters(paramVec): the keyframe model re-chains, the window model only rewrites the relative poses
        chain(ctx).set_params(paramVec.data());
        if (ctx->model == MODEL_KEYFRAMES) chain(ctx).relative_to_global();
        if (stepNorm < s.epsilon && !fixed) {  // :139-143
            stop = DMSA_STOP_EPSILON;
            break;
        }
    }
    g_tl.print();
    if (s.use_centralization) CHK(dmsa_decentralize(ctx));
    // :149 final updateGlobalPoints
    if (ctx->model == MODEL_WINDOW) chain(ctx).relative_to_global();
    globs.clear();
    append_glob(chain(ctx), globs);
    CHK(build_tables(ctx, 1, globs));
    CHK(transform_points(ctx, 0));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (rep) {
        rep->iterations = iters, rep->stop_reason = stop;
        rep->num_gaussians = ctx->M, rep->num_gaussians_l1 = ctx->M1, rep->num_memberships = ctx->Mm;
        rep->error0 = error0, rep->last_step_norm = stepNorm, rep->last_line_search_k = bestK;
        rep->evaluations = ctx->evaluations;
    }
    return DMSA_OK;
}

// ---- the same loop with its control state on the device (loop_kernels.h) ---------------------------------------------------------
// Per iteration the host only enqueues; its one wait is for the Gaussian counts that size the correspondence launches (sync A).  The
// stop decision of iteration i (no improvement / epsilon / NaN step) is taken on the device and reaches the host with the counts of
// iteration i + 1: every loop kernel of a stopped loop is a no-op, so the extra voxelisation that was already enqueued changes nothing.
// The LM step is solved on the device as well (one workgroup up to P = 64, the panel kernel up to P = 1024); only beyond that the
// normal equations go to the host's worker pool, which costs one more wait per iteration.
int pinned_doubles(dmsa_ctx* ctx, size_t count, double** out) {
    constexpr int kPinSlots = 4;
    if (count > ctx->h_pin_slot) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
        ctx->h_pin = nullptr;
        ctx->h_pin_slot = count + count / 2 + 64;
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_pin), ctx->h_pin_slot * kPinSlots * sizeof(double), hipHostMallocDefault));
    }
    *out = ctx->h_pin + (size_t)ctx->h_pin_next * ctx->h_pin_slot;
    ctx->h_pin_next = (ctx->h_pin_next + 1) % kPinSlots;
    return DMSA_OK;
}
int device_tables(dmsa_ctx* ctx, int B, const double* d_ctrl, float* tables, float* tablesT, hipStream_t stream, uint32_t* rot_same = nullptr) {
    const int np = ctx->loop_model.n;
    if (ctx->model == MODEL_WINDOW)
        launch_window_pose_tables(d_ctrl, ctx->d_stamps.as<double>(), ctx->d_fhw.as<double>(), ctx->d_trajtime.as<double>(), B, np, ctx->rows - 1, tables, tablesT,
                                  stream, rot_same);
    else
        launch_keyframe_pose_tables(d_ctrl, B, np, tables, tablesT, stream, rot_same);
    HIPCHK(hipGetLastError());
    return DMSA_OK;
}

// :107-128 on the device: one workgroup for P <= 64, column-block workgroups handing panels to each other beyond
int device_lm_step(dmsa_ctx* ctx, const double* d_Hp, int P, double lambda, double alpha, double max_step, double* d_step, LoopFlags* d_flags) {
    if (P <= kLoopSolveMaxP) {
        launch_loop_lm_step(d_Hp, P, lambda, alpha, max_step, d_step, d_flags, ctx->stream);
    } else {
        const size_t bytes = loop_panel_solve_doubles(P) * 8;
        if (bytes > ctx->d_panel_work.cap || P != ctx->panel_P) {
            // a new size moves the epoch-tagged words (hand-over flags, record counter) onto what used to be data: start from zeros
            HIPCHK(ctx->d_panel_work.ensure(bytes));
            HIPCHK(hipMemsetAsync(ctx->d_panel_work.p, 0, ctx->d_panel_work.cap, ctx->stream));
            ctx->panel_epoch = 0, ctx->panel_P = P;
        }
        ctx->panel_epoch += 1;
        if (ctx->panel_epoch == 0) ctx->panel_epoch = 1;
        if (ctx->dbg.lm_stream != 0 && loop_lm_stream_fits(P))
            launch_loop_lm_stream(d_Hp, P, lambda, alpha, max_step, ctx->d_panel_work.as<double>(), ctx->panel_epoch, d_step, d_flags, ctx->stream);
        else
            launch_loop_lm_panels(d_Hp, P, lambda, alpha, max_step, ctx->d_panel_work.as<double>(), ctx->panel_epoch, d_step, d_flags, ctx->stream);
    }
    HIPCHK(hipGetLastError());
    return DMSA_OK;
}

static int optimize_device_loop(dmsa_ctx* ctx, const dmsa_settings& s, dmsa_report* rep) {
    ScopedTimer total(ctx, T_TOTAL);
    // debug switch trace_time: where a call's time outside its iterations goes (host clock, microseconds since the call began)
    const auto call_t0 = std::chrono::steady_clock::now();
    std::string call_trace;
    auto mark = [&](const char* what) {
        if (ctx->dbg.trace_time == 0) return;
        char buf[160];
        std::snprintf(buf, sizeof(buf), " %s %.0f", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - call_t0).count());
        call_trace += buf;
    };
    const bool fixed = (ctx->flags & DMSA_FLAG_FIXED_ITERS) != 0;
    const LoopModel& m = ctx->loop_model;
    const int P = m.P, n = m.n, a = m.extra > 0 ? m.extra : 0;
    const int num_iter = std::max(0, s.num_iter);
    int stop = DMSA_STOP_NUM_ITER, iters = 0, bestK = 0;
    double error0 = 0.0, stepNorm = 0.0;
    ctx->evaluations = 0;
    ctx->trace.clear();
    const double increment = 1.0 * std::sqrt((double)std::numeric_limits<float>::epsilon());
    const double one_div_incr = 1.0 / increment;

    if (s.use_centralization) CHK(dmsa_centralize(ctx));
    // device buffers of the loop
    const size_t st = loop_state_doubles(n);
    HIPCHK(ctx->d_loop_state.ensure(3 * st * 8));
    HIPCHK(ctx->d_loop_vec.ensure((size_t)2 * P * 8 + 64));
    HIPCHK(ctx->d_ctrl0.ensure((size_t)n * 6 * 8));
    HIPCHK(ctx->d_ctrl.ensure((size_t)(1 + P) * n * 6 * 8));
    HIPCHK(ctx->d_table0.ensure((size_t)ctx->rows * 48));
    HIPCHK(ctx->d_tables.ensure((size_t)(P + 1) * ctx->rows * 48));  // never reallocated while kernels read it
    HIPCHK(ctx->d_tablesT.ensure((size_t)(P + 1) * ctx->rows * 48));
    HIPCHK(ctx->d_loop_extra.ensure((size_t)(1 + P + 9) * std::max(a, 1) * 8));
    HIPCHK(ctx->d_rot_same.ensure((size_t)(1 + P) * 4));
    // Pairs (Gaussian, evaluation) whose pose-table rows all have evaluation 0's bits are not computed (serial_kernels.h).  Worth it where
    // a Gaussian's evaluations fill several sub-batches of sixteen lanes, i.e. in the keyframe pass; the one consumer that then has to know
    // is the Jacobian column kernel of the matrix-core normal equations (P > 64).
    const int skip_mode = P > kLoopSolveMaxP ? ctx->dbg.eval_skip : 0;
    if (skip_mode != 0) {
        HIPCHK(ctx->d_row_range.ensure((size_t)(1 + P) * 8));
        if ((size_t)P * 16 > ctx->d_skip_stats.cap) {  // per evaluation: pairs left out, pairs that differed under eval_skip = 2
            HIPCHK(ctx->d_skip_stats.ensure((size_t)P * 16));
            HIPCHK(hipMemsetAsync(ctx->d_skip_stats.p, 0, ctx->d_skip_stats.cap, ctx->stream));
            ctx->skip_stats_evals = P;
        }
    }
    // sized for at least 256 iterations from the first call on: a call with more iterations than the one before must not pay a hipMalloc /
    // hipHostMalloc inside optimizeSet (0.4 ms, seen as 2.5 % of a 20-iteration call that followed a 5-iteration one)
    const int iter_cap = std::max(num_iter + 1, 256);
    HIPCHK(ctx->d_loop_iter.ensure(sizeof(LoopFlags) + (size_t)iter_cap * sizeof(IterResult)));
    HIPCHK(ctx->d_Hp.ensure((size_t)(P + 1) * (P + 1) * 8));
    HIPCHK(ctx->d_sq_out.ensure(16 * 8));
    if (num_iter + 1 > ctx->h_results_cap) {
        if (ctx->h_results) (void)hipHostFree(ctx->h_results);
        ctx->h_results = nullptr, ctx->h_results_cap = 0;
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_results), (size_t)(iter_cap + 16) * sizeof(IterResult), hipHostMallocDefault));
        ctx->h_results_cap = iter_cap + 16;
    }
    std::memset(ctx->h_results, 0, (size_t)(num_iter + 1) * sizeof(IterResult));
    double* S0 = ctx->d_loop_state.as<double>();
    double* S1 = S0 + st;
    double* S2 = S1 + st;
    double* d_param = ctx->d_loop_vec.as<double>();
    double* d_step = d_param + P;
    double* d_extra_jac = ctx->d_loop_extra.as<double>();
    double* d_extra_trial = d_extra_jac + (size_t)(1 + P) * a;
    LoopFlags* d_flags = ctx->d_loop_iter.as<LoopFlags>();
    IterResult* d_results = reinterpret_cast<IterResult*>(d_flags + 1);
    // seed: the host chain as centralize() left it
    {
        PoseChain& c = chain(ctx);
        double* pin = nullptr;
        CHK(pinned_doubles(ctx, st, &pin));
        std::copy(c.rel_o.begin(), c.rel_o.end(), pin);
        std::copy(c.rel_t.begin(), c.rel_t.end(), pin + 3 * n);
        std::copy(c.glob_o.begin(), c.glob_o.end(), pin + 6 * n);
        std::copy(c.glob_t.begin(), c.glob_t.end(), pin + 9 * n);
        HIPCHK(hipMemcpyAsync(S0, pin, st * 8, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemsetAsync(ctx->d_loop_iter.p, 0, sizeof(LoopFlags) + (size_t)(num_iter + 1) * sizeof(IterResult), ctx->stream));
    }
    mark("seeded");
    // debug switch gap_stamps: one-thread kernels write the device's wall clock; 2 = eight stamps per iteration, printed as a table
    long long* iter_stamps = nullptr;
    if (ctx->dbg.gap_stamps != 0) {
        HIPCHK(ctx->d_gap_stamps.ensure((size_t)(16 + 8 * (ctx->dbg.gap_stamps >= 2 ? num_iter : 0)) * 8));
        if (ctx->dbg.gap_stamps >= 2) {
            iter_stamps = ctx->d_gap_stamps.as<long long>() + 16;
            HIPCHK(hipMemsetAsync(iter_stamps, 0, (size_t)8 * num_iter * 8, ctx->stream));
        }
    }
    std::vector<double> Hp, H, g, step;
    if (P > kLoopPanelMaxP) Hp.resize((size_t)(P + 1) * (P + 1)), H.resize((size_t)P * P), g.resize((size_t)P), step.resize((size_t)P);
    // what the report says about the Gaussians belongs to the last iteration that really ran
    int last_M = 0, last_M1 = 0;
    int64_t last_Mm = 0;
    int nan_evals = 0;
    hipStream_t side = ctx->dual_stream ? ctx->stream3 : ctx->stream;
    for (int iter = 0; iter < num_iter; ++iter) {
        g_tl.on = ctx->dbg.host_timeline != 0, g_tl.reset(), g_tl.mark("start");
        // :72-75 parameters, chain, base table, global points
        // the Jacobian chains on the side stream start when the state of the iteration start is in place: signalled by k_loop_begin /
        // the previous k_loop_finish themselves (dev_sync.h) -- no event on the main stream
        const bool dev_sync = ctx->dbg.device_sync != 0 && side != ctx->stream;
        if (iter_stamps) {
            launch_stamp(iter_stamps + 8 * iter + 0, ctx->stream);  // the iteration begins
            ctx->stamp_voxel = iter_stamps + 8 * iter + 1, ctx->stamp_fit = iter_stamps + 8 * iter + 2;
        }
        if (iter == 0) {  // later iterations: done by loop_finish
            launch_loop_begin(m, S0, d_param, ctx->d_ctrl0.as<double>(), d_flags, ctx->stream, dev_sync ? ctx->sync_counter(SYNC_LOOP_STATE) : nullptr);
            if (dev_sync) ctx->sync_sig[SYNC_LOOP_STATE] += 1;
        }
        {
            ScopedTimer tm(ctx, T_TABLE);
            CHK(device_tables(ctx, 1, ctx->d_ctrl0.as<double>(), ctx->d_table0.as<float>(), nullptr, ctx->stream));
        }
        ctx->base_table = ctx->d_table0.as<float>();
        {
            ScopedTimer tm(ctx, T_VOXEL);
            launch_transform_aabb(ctx->d_local.as<float4>(), ctx->model == MODEL_KEYFRAMES ? ctx->d_nlocal.as<float4>() : nullptr, ctx->d_table0.as<float4>(),
                                  ctx->d_global.as<float4>(), ctx->model == MODEL_KEYFRAMES ? ctx->d_nglobal.as<float4>() : nullptr, ctx->n,
                                  ctx->d_aabb.as<float>(), ctx->d_counts.p, sizeof(GaussCounts) + sizeof(SerialCounts), ctx->stream);
            ctx->aabb_fresh = true;
        }
        // :99, :199-232 the 1 + P chains, rows and pose tables of the Jacobian batch: beside the voxelisation, they need nothing from it
        if (dev_sync) {
            enqueue_wait(ctx, SYNC_LOOP_STATE, side);
        } else if (side != ctx->stream) {
            HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
            HIPCHK(hipStreamWaitEvent(side, ctx->ev_fork, 0));
        }
        launch_loop_chain(m, 0, S0, S1, d_param, d_step, increment, ctx->d_ctrl.as<double>(), d_extra_jac, d_flags, side);
        // with the flags that let the correspondence kernels share the rotated coordinates among the translation differences (serial_kernels.hip)
        uint32_t* rot_same = ctx->dbg.shared_rotations != 0 ? ctx->d_rot_same.as<uint32_t>() : nullptr;
        CHK(device_tables(ctx, 1 + P, ctx->d_ctrl.as<double>(), ctx->d_tables.as<float>(), ctx->d_tablesT.as<float>(), side, rot_same));
        ctx->batch = 1 + P, ctx->tablesT_batch = 1 + P;
        if (skip_mode != 0)
            launch_eval_row_ranges(m.model, ctx->d_ctrl.as<double>(), 1 + P, n, ctx->d_stamps.as<double>(), ctx->d_trajtime.as<double>(), ctx->rows - 1,
                                   ctx->d_row_range.as<int2>(), side);
        if (dev_sync) {  // the main stream picks the tables up in k_size_classes (or, if that kernel is not launched, in run_residuals)
            launch_sync_signal(ctx->sync_counter(SYNC_TABLES), side);
            ctx->sync_sig[SYNC_TABLES] += 1;
        } else {
            HIPCHK(hipEventRecord(ctx->ev_tables, side));
        }
        ctx->tables_pending = side != ctx->stream, ctx->tables_dev_sync = dev_sync;
        g_tl.mark("begin+batch enq");
        // :78-96; the previous iteration's result rides on the read-back of the counts
        if (iter > 0) {
            ctx->rb_extra_src = d_results + (iter - 1), ctx->rb_extra_dst = ctx->h_results + (iter - 1), ctx->rb_extra_bytes = sizeof(IterResult);
        } else {
            ctx->rb_extra_bytes = 0;
        }
        const int rc = build_gaussians(ctx, s);
        ctx->rb_extra_bytes = 0;
        CHK(rc);
        drain_timers(ctx);  // everything the previous iteration timed has completed
        g_tl.mark("build_gaussians (incl. sync A)");
        if (iter == 0) mark("first-counts");
        if (iter == 1) mark("second-counts");
        if (iter > 1 && ctx->dbg.trace_time >= 2) {  // every iteration's: the host's iteration period is the device's
            char what[96];
            std::snprintf(what, sizeof(what), "[M %d Mm %lld grown %lld] counts", ctx->M, (long long)ctx->Mm, DevBuf::reallocations());
            mark(what);
        }
        if (iter > 0 && ctx->h_results[iter - 1].stop != 0) break;  // the loop ended in the previous iteration: this one never started
        ++iters;
        last_M = ctx->M, last_M1 = ctx->M1, last_Mm = ctx->Mm;
        ctx->trace.push_back(dmsa_iter_trace{ctx->M, ctx->M1, ctx->Mm, 0.0, 0.0, 0, 0});
        if (ctx->M < s.min_num_gaussians) {  // :89-93 -- nothing of this iteration has touched the state the next call starts from (S0)
            stop = DMSA_STOP_FEW_GAUSSIANS;
            break;
        }
        ctx->evaluations += 1 + P;
        if (skip_mode != 0) ctx->skip_pairs += (int64_t)ctx->M * P;
        CHK(run_residuals(ctx, 1 + P, nullptr, d_extra_jac, rot_same, skip_mode == 1 ? ctx->d_row_range.as<int2>() : nullptr));
        const int rowsE = ctx->M + ctx->extra_rows;
        const bool stamps = ctx->dbg.gap_stamps != 0;
        if (stamps) {
            launch_stamp(ctx->d_gap_stamps.as<long long>() + 0, ctx->stream);  // the Jacobian batch has joined
            if (iter_stamps) launch_stamp(iter_stamps + 8 * iter + 3, ctx->stream);
        }
        {
            ScopedTimer tm(ctx, T_NORMAL);
            HIPCHK(ctx->d_ne_partial.ensure((size_t)normal_equations_partial_doubles(rowsE, P) * 8));
            EvalSkip skip;
            if (skip_mode != 0)
                skip.row_range = ctx->d_row_range.as<int2>(), skip.gauss_rows = ctx->d_gauss_rows.as<int2>(), skip.M = ctx->M, skip.check = skip_mode == 2 ? 1 : 0,
                skip.stats = (ctx->dbg.skip_stats != 0 || ctx->dbg.eval_skip == 2) ? ctx->d_skip_stats.as<unsigned long long>() : nullptr;
            // P <= 64: the block sums stay unreduced, the solve kernel adds them while it loads the matrix
            launch_normal_equations(ctx->d_E.as<double>(), ctx->ldE, rowsE, P, one_div_incr, ctx->d_ne_partial.as<double>(), ctx->d_Hp.as<double>(), ctx->stream,
                                    P > kLoopSolveMaxP, skip_mode != 0 ? &skip : nullptr);
        }
        if (stamps) launch_stamp(ctx->d_gap_stamps.as<long long>() + 1, ctx->stream);  // normal equations done
        if (iter_stamps) launch_stamp(iter_stamps + 8 * iter + 4, ctx->stream);
        bool host_nan = false;
        double* d_error0 = ctx->d_Hp.as<double>() + (size_t)P * (P + 1) + P;  // e0^T e0, element (P, P) of Hp
        if (P <= kLoopSolveMaxP) {
            const NormalEqPartials q = normal_equations_partials(rowsE, P);
            launch_loop_lm_step_partials(ctx->d_ne_partial.as<double>(), q.nsplit, q.nt, P, (double)s.lambda_diag, s.step_length_optim, s.max_step, d_step, d_flags,
                                         d_error0, ctx->stream);
        } else if (P <= kLoopPanelMaxP) {
            // :107-128 on the device
            CHK(device_lm_step(ctx, ctx->d_Hp.as<double>(), P, (double)s.lambda_diag, s.step_length_optim, s.max_step, d_step, d_flags));
        } else {
            if (Hp.size() > ctx->h_Hp_cap) {
                if (ctx->h_Hp) (void)hipHostFree(ctx->h_Hp);
                ctx->h_Hp = nullptr, ctx->h_Hp_cap = 0;
                HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_Hp), Hp.size() * 8, hipHostMallocDefault));
                ctx->h_Hp_cap = Hp.size();
            }
            HIPCHK(hipMemcpyAsync(ctx->h_Hp, ctx->d_Hp.p, Hp.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
            g_tl.mark("residuals+NE enq");
            HIPCHK(sync_spin(ctx->stream));  // sync B (P > 64 only)
            g_tl.mark("sync B wait");
            std::memcpy(Hp.data(), ctx->h_Hp, Hp.size() * 8);
            const int n1 = P + 1;
            for (int j = 0; j < P; ++j)
                for (int i = 0; i < P; ++i) H[(size_t)j * P + i] = Hp[(size_t)j * n1 + i];
            for (int i = 0; i < P; ++i) g[(size_t)i] = Hp[(size_t)P * n1 + i];
            for (int i = 0; i < P; ++i) H[(size_t)i * P + i] += (double)s.lambda_diag;  // :110
            const ParallelRun par = [&](const std::function<void(int, int)>& fn) { workers(ctx).run_all(fn); };
            lm_solve(H.data(), g.data(), P, s.step_length_optim, step.data(), &par, ctx->dbg.solve_threads);  // :113
            for (double v : step) host_nan = host_nan || std::isnan(v);
            double* pin = nullptr;
            CHK(pinned_doubles(ctx, (size_t)P, &pin));
            std::memcpy(pin, step.data(), (size_t)P * 8);
            HIPCHK(hipMemcpyAsync(d_step, pin, (size_t)P * 8, hipMemcpyHostToDevice, ctx->stream));
            launch_loop_step_finish(P, s.max_step, d_step, d_flags, ctx->stream);  // NaN test, clamp
            g_tl.mark("solve");
        }
        if (stamps) launch_stamp(ctx->d_gap_stamps.as<long long>() + 2, ctx->stream);  // LM step done
        if (iter_stamps) launch_stamp(iter_stamps + 8 * iter + 5, ctx->stream);
        // :152-182 nine trials, :130-143 decision
        // A window with IMU rows: the rows (updateImuError: a dozen Floater-Hormann evaluations per row) are half of the chain kernel's time and nobody
        // reads them before the squared sums -- the main stream computes the control poses only (part 1), the side stream the rows and the state the
        // last trial leaves (part 2) beside the pose tables and the trial batch.  26 + 4 us -> 11 us in front of the trial batch of a config-2 window.
        const bool rows_aside = ctx->dbg.trial_rows_aside != 0 && a > 0 && ctx->model == MODEL_WINDOW && dev_sync;
        if (rows_aside) {
            CHK(ensure_E(ctx, 9));  // (the scatter below needs ldE; the Jacobian batch's E is at least as large)
            launch_loop_chain(m, 1, S1, S2, d_param, d_step, increment, ctx->d_ctrl.as<double>(), d_extra_trial, d_flags, ctx->stream, 1,
                              ctx->sync_counter(SYNC_TRIAL_STEP));
            ctx->sync_sig[SYNC_TRIAL_STEP] += 1;
            enqueue_wait(ctx, SYNC_TRIAL_STEP, side);
            launch_loop_chain(m, 1, S1, S2, d_param, d_step, increment, ctx->d_ctrl.as<double>(), d_extra_trial, d_flags, side, 2);
            launch_loop_scatter_extra(d_extra_trial, 9, a, ctx->d_E.as<double>(), ctx->ldE, ctx->M, side);
            launch_sync_signal(ctx->sync_counter(SYNC_TRIAL_ROWS), side);
            ctx->sync_sig[SYNC_TRIAL_ROWS] += 1;
        } else {
            launch_loop_chain(m, 1, S1, S2, d_param, d_step, increment, ctx->d_ctrl.as<double>(), d_extra_trial, d_flags, ctx->stream);
        }
        {
            ScopedTimer tm(ctx, T_TABLE);
            CHK(device_tables(ctx, 9, ctx->d_ctrl.as<double>(), ctx->d_tables.as<float>(), ctx->d_tablesT.as<float>(), ctx->stream));
            ctx->batch = 9, ctx->tablesT_batch = 9;
        }
        if (stamps) launch_stamp(ctx->d_gap_stamps.as<long long>() + 3, ctx->stream);  // trial chains and pose tables done
        if (iter_stamps) launch_stamp(iter_stamps + 8 * iter + 6, ctx->stream);
        if (!host_nan) ctx->evaluations += 9;
        nan_evals = host_nan ? 0 : 9;
        CHK(run_residuals(ctx, 9, nullptr, rows_aside ? nullptr : d_extra_trial));
        {
            ScopedTimer tm(ctx, T_NORMAL);
            HIPCHK(ctx->d_sq_partial.ensure((size_t)squared_sums_blocked_partial_doubles(rowsE, P, 9) * 8));
            DevSync rows_in;  // the side stream's rows: waited for by the kernel that reads them (a one-wave wait kernel in front of it costs 3 us)
            if (rows_aside)
                rows_in.wait_counter = ctx->sync_counter(SYNC_TRIAL_ROWS), rows_in.wait_target = ctx->sync_sig[SYNC_TRIAL_ROWS], rows_in.timed_out = ctx->sync_timed_out();
            launch_squared_sums_blocked(ctx->d_E.as<double>(), ctx->ldE, rowsE, P, 9, ctx->d_sq_partial.as<double>(), nullptr, ctx->stream,
                                        rows_aside ? &rows_in : nullptr);  // block sums only
        }
        {
            const bool sig = ctx->dbg.device_sync != 0 && side != ctx->stream;  // for the next iteration's chains, if there is one (a signal nobody waits for is harmless)
            launch_loop_finish(m, S1, S2, S0, d_param, d_step, d_error0, ctx->d_sq_partial.as<double>(), normal_equations_partials(rowsE, P).nsplit, fixed ? 1 : 0,
                               s.epsilon, d_results + iter, d_flags, ctx->d_ctrl0.as<double>(), iter + 1 < num_iter ? 1 : 0, ctx->stream,
                               sig ? ctx->sync_counter(SYNC_LOOP_STATE) : nullptr);
            if (sig) ctx->sync_sig[SYNC_LOOP_STATE] += 1;
        }
        if (iter_stamps) launch_stamp(iter_stamps + 8 * iter + 7, ctx->stream);  // the iteration's last kernel (k_loop_finish) is done
        HIPCHK(hipGetLastError());
        g_tl.mark("iteration enq");
        if (host_nan) break;  // the device takes the same decision; nothing more to enqueue
    }
    g_tl.print();
    mark("all-enqueued");
    // final state and the results not yet seen
    std::vector<double> fin(st);
    {
        double* pin = nullptr;
        CHK(pinned_doubles(ctx, st, &pin));
        HIPCHK(hipMemcpyAsync(pin, S0, st * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (iters > 0) HIPCHK(hipMemcpyAsync(ctx->h_results, d_results, (size_t)iters * sizeof(IterResult), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        std::copy(pin, pin + st, fin.begin());
    }
    mark("state-back");
    drain_timers(ctx);
    ctx->stamp_voxel = ctx->stamp_fit = nullptr;
    if (iter_stamps && iters > 0) {
        std::vector<long long> t((size_t)8 * iters);
        HIPCHK(hipMemcpy(t.data(), iter_stamps, t.size() * 8, hipMemcpyDeviceToHost));
        std::fprintf(stderr, "[gap_stamps] per iteration, us on the device's wall clock (each stamp kernel adds ~3 us): voxelisation | fit | Jacobian batch | normal equations | LM step | "
                             "trial chains + tables | trial batch + decision | whole iteration\n");
        for (int i = 0; i < iters; ++i) {
            const long long* r = t.data() + 8 * i;
            std::fprintf(stderr, "[gap_stamps] %3d:", i);
            for (int k = 1; k < 8; ++k) std::fprintf(stderr, " %6.1f", (r[k] - r[k - 1]) * 0.01);
            std::fprintf(stderr, " | %6.1f\n", i + 1 < iters ? (t[(size_t)8 * (i + 1)] - r[0]) * 0.01 : (r[7] - r[0]) * 0.01);
        }
    }
    if (ctx->dbg.gap_stamps != 0 && ctx->d_gap_stamps.p && iters > 0) {
        long long t[4] = {0, 0, 0, 0};
        HIPCHK(hipMemcpy(t, ctx->d_gap_stamps.p, sizeof(t), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "[gap_stamps] last iteration, P = %d, %d rows: Jacobian batch joined -> normal equations done %.1f us -> LM step done %.1f us -> trial chains + tables done %.1f us "
                             "(device wall clock, unprofiled; each stamp kernel adds its own ~3 us)\n", P, ctx->M + ctx->extra_rows, (t[1] - t[0]) * 0.01, (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01);
    }
    {
        PoseChain& c = chain(ctx);
        std::copy(fin.begin(), fin.begin() + 3 * n, c.rel_o.begin());
        std::copy(fin.begin() + 3 * n, fin.begin() + 6 * n, c.rel_t.begin());
        std::copy(fin.begin() + 6 * n, fin.begin() + 9 * n, c.glob_o.begin());
        std::copy(fin.begin() + 9 * n, fin.begin() + 12 * n, c.glob_t.begin());
    }
    for (int i = 0; i < iters && i < (int)ctx->trace.size(); ++i) {
        const IterResult& r = ctx->h_results[i];
        const bool ran = !(stop == DMSA_STOP_FEW_GAUSSIANS && i == iters - 1);  // the aborted iteration has no step
        if (!ran) break;
        error0 = r.error0;
        if (r.stop == DMSA_STOP_NAN) {  // :116-122: left before the line search, nothing else of this iteration is recorded
            stop = r.stop;
            ctx->evaluations -= nan_evals;
            break;
        }
        ctx->trace[(size_t)i].error0 = r.error0, ctx->trace[(size_t)i].step_norm = r.step_norm, ctx->trace[(size_t)i].best_k = r.best_k;
        stepNorm = r.step_norm, bestK = r.best_k;
        if (r.stop != 0) stop = r.stop;
    }
    ctx->M = last_M, ctx->M1 = last_M1, ctx->Mm = last_Mm;
    if (s.use_centralization) CHK(dmsa_decentralize(ctx));
    // :149 final updateGlobalPoints
    if (ctx->model == MODEL_WINDOW) chain(ctx).relative_to_global();
    std::vector<double> globs;
    append_glob(chain(ctx), globs);
    CHK(build_tables(ctx, 1, globs));
    CHK(transform_points(ctx, 0));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    mark("final-points");
    if (ctx->dbg.trace_time != 0) std::fprintf(stderr, "[call] %d iterations, us since the call began:%s\n", iters, call_trace.c_str());
    if (rep) {
        rep->iterations = iters, rep->stop_reason = stop;
        rep->num_gaussians = ctx->M, rep->num_gaussians_l1 = ctx->M1, rep->num_memberships = ctx->Mm;
        rep->error0 = error0, rep->last_step_norm = stepNorm, rep->last_line_search_k = bestK;
        rep->evaluations = ctx->evaluations;
    }
    return DMSA_OK;
}

