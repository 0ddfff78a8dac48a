// context.cpp — lifetime of a dmsa_ctx: creation (debug switches), destruction, device buffers, uploads of the two problem models, timers.
#include "dmsa_ctx.h"

HostTimeline g_tl;

WorkerPool& workers(dmsa_ctx* ctx) {
    if (!ctx->pool) {
        const unsigned want = (unsigned)std::max(1, ctx->dbg.host_threads);  // host pose tables, perturbed keyframe chains, upload packing, host solve
        ctx->pool = new WorkerPool((int)std::min(want, std::max(2u, std::thread::hardware_concurrency())));
    }
    return *ctx->pool;
}
hipEvent_t get_event(dmsa_ctx* ctx) {
    if (!ctx->free_events.empty()) {
        hipEvent_t e = ctx->free_events.back();
        ctx->free_events.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
// fold finished event pairs into the accumulators (call after a stream synchronisation)
void drain_timers(dmsa_ctx* ctx) {
    // pairs whose closing event has not completed yet stay pending (the device-resident loop drains without a full synchronisation)
    std::vector<EventPair> later;
    for (auto& ev : ctx->pending) {
        if (hipEventQuery(ev.b) == hipErrorNotReady) {
            later.push_back(ev);
            continue;
        }
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) ctx->t_ms[ev.slot] += (double)ms;
        ctx->free_events.push_back(ev.a), ctx->free_events.push_back(ev.b);
    }
    (void)hipGetLastError();  // hipErrorNotReady is recorded as the thread's last error
    ctx->pending.swap(later);
}

// Host synchronisation on the critical path of an iteration: polling the stream avoids the ~20-30 us wake-up latency of a
// blocking hipStreamSynchronize (there are four such points per iteration).
hipError_t sync_spin(hipStream_t stream) {
    hipError_t e;
    while ((e = hipStreamQuery(stream)) == hipErrorNotReady) {
    }
    (void)hipGetLastError();  // hipErrorNotReady is recorded as the thread's last error: do not leave it for other HIP users (torch)
    return e;
}

int set_device(dmsa_ctx* ctx) {
    HIPCHK(hipSetDevice(ctx->device));
    return DMSA_OK;
}

int num_params(const dmsa_ctx* ctx) { return ctx->model == MODEL_WINDOW ? ctx->win.ctrl.num_params() : ctx->key.frames.num_params(); }
PoseChain& chain(dmsa_ctx* ctx) { return ctx->model == MODEL_WINDOW ? ctx->win.ctrl : ctx->key.frames; }
int num_extra_rows(const dmsa_ctx* ctx) { return ctx->model == MODEL_WINDOW ? ctx->win.num_extra_rows() : ctx->key.num_extra_rows(); }


// numPointsPerSet.cast<float>().array().pow(-1) (Gaussians.h:172) is libm's powf(n, -1.0f) per coefficient (Eigen 3.4.0's scalar_pow_op
// has no packet path), and powf is not correctly rounded: for some n it is one ulp away from 1.0f / n.  The device divides and applies
// what THIS machine's libm says differs, two bits per member count n: 0 same bits, 1 one ulp above, 2 one ulp below, 3 further away
// (never seen; such a count takes the division and the context reports it).  The table is computed once per process and extended on
// demand; a context uploads the range its point count allows.
namespace {
std::mutex g_pow_mutex;
std::vector<uint32_t> g_pow_codes;  // 16 counts per word
int64_t g_pow_n = 0;                // counts covered
int64_t g_pow_far = 0;              // counts whose powf is more than one ulp from the division
}  // namespace
int upload_powm1_codes(dmsa_ctx* ctx, int64_t counts) {
    counts = (counts + 15) / 16 * 16;
    if (counts <= ctx->pow_n) return DMSA_OK;
    std::vector<uint32_t> words;
    {
        std::lock_guard<std::mutex> lock(g_pow_mutex);
        if (counts > g_pow_n) {
            const int64_t from = g_pow_n;
            g_pow_codes.resize((size_t)(counts / 16), 0u);
            std::atomic<int64_t> far{0};
            workers(ctx).run_all([&](int t, int nt) {
                const int64_t w0 = from / 16 + (counts / 16 - from / 16) * t / nt, w1 = from / 16 + (counts / 16 - from / 16) * (t + 1) / nt;
                volatile float minus_one = -1.0f;  // (volatile: no folding of the call into a division)
                for (int64_t w = w0; w < w1; ++w) {
                    uint32_t word = 0;
                    for (int k = 0; k < 16; ++k) {
                        const int64_t nn = 16 * w + k;
                        if (nn == 0) continue;
                        const float x = (float)nn, a = powf(x, minus_one), d = 1.0f / x;
                        int32_t ia, id;
                        std::memcpy(&ia, &a, 4), std::memcpy(&id, &d, 4);
                        const uint32_t code = ia == id ? 0u : (ia == id + 1 ? 1u : (ia == id - 1 ? 2u : 3u));
                        if (code == 3u) far += 1;
                        word |= code << (2 * k);
                    }
                    g_pow_codes[(size_t)w] = word;
                }
            });
            g_pow_n = counts, g_pow_far += far.load();
        }
        words.assign(g_pow_codes.begin(), g_pow_codes.begin() + counts / 16);
        if (g_pow_far > 0) ctx->err = "warning: this libm's powf(n, -1) is more than one ulp away from 1 / n for some n; those counts use the division";
    }
    HIPCHK(ctx->d_pow_codes.ensure(words.size() * 4));
    HIPCHK(hipMemcpy(ctx->d_pow_codes.p, words.data(), words.size() * 4, hipMemcpyHostToDevice));
    ctx->pow_n = counts;
    return DMSA_OK;
}
// allocate everything whose size depends only on the point count
int alloc_point_buffers(dmsa_ctx* ctx) {
    const size_t n = (size_t)ctx->n;
    HIPCHK(ctx->d_global.ensure(n * 16));
    const size_t nb = (n + kAabbBlock - 1) / kAabbBlock;
    HIPCHK(ctx->d_aabb.ensure(nb * 8 * sizeof(float)));
    HIPCHK(ctx->d_lattice.ensure(2 * sizeof(LatticeTable)));
    for (int l = 0; l < 2; ++l) {
        if (l == 0) {  // both levels live in ONE array of 2n entries (level 1 behind level 0): they are sorted together
            HIPCHK(ctx->d_code[0].ensure(2 * n * 8));
            HIPCHK(ctx->d_idx[0].ensure(2 * n * 4));
            HIPCHK(ctx->d_code_s[0].ensure(2 * n * 8));
            HIPCHK(ctx->d_idx_s[0].ensure(2 * n * 4));
        }
        HIPCHK(ctx->d_leaf_incl[l].ensure(n * 4));
        HIPCHK(ctx->d_leaf_start[l].ensure((n + 1) * 4));
    }
    for (int l = 0; l < 2; ++l) {
        HIPCHK(ctx->d_head[l].ensure(n * 4));
        HIPCHK(ctx->d_slot_acc[l].ensure(2 * n * 4));
        HIPCHK(ctx->d_slot_cnt[l].ensure(2 * n * 4));
        HIPCHK(ctx->d_gauss_of_slot[l].ensure(2 * n * 4));
        HIPCHK(ctx->d_memb_of_slot[l].ensure(2 * n * 4));
        HIPCHK(ctx->d_pslot_of_slot[l].ensure(2 * n * 4));
        HIPCHK(ctx->d_sort_tmp[l].ensure(sort_pairs_temp_bytes(2 * n)));
        HIPCHK(ctx->d_scan_tmp[l].ensure(scan_temp_bytes(2 * n)));
    }
    HIPCHK(ctx->d_counts.ensure(sizeof(GaussCounts) + sizeof(SerialCounts)));  // read back together
    // memberships: every point belongs to at most one set per resolution
    HIPCHK(ctx->d_memb_local.ensure(2 * n * 16));
    HIPCHK(ctx->d_memb_idx.ensure(2 * n * 4));
    HIPCHK(ctx->d_memb_g.ensure(2 * n * 4));
    HIPCHK(ctx->d_seg_off.ensure((2 * n + 2) * 4));
    // sets have >= 2 members (two distinct ids) -- except the second half of a splitSet, which may keep a single member when
    // min_num_points_per_set <= 1: size for one set per membership
    HIPCHK(ctx->d_info12.ensure((2 * n + 16) * 48));
    // one entry per Gaussian, and M can approach 2n (see d_info12 above)
    HIPCHK(ctx->d_order.ensure((2 * n + 16) * 4));
    HIPCHK(ctx->d_fit_sums.ensure((2 * n + 16) * 6 * 4));
    CHK(upload_powm1_codes(ctx, (int64_t)n + 1));
    HIPCHK(ctx->d_memb_q.ensure(3 * (2 * n + 16) * 4));
    HIPCHK(ctx->d_gauss_rows.ensure((2 * n + 16) * 8));
    HIPCHK(ctx->d_pad_off.ensure((2 * n + 2) * 4));
    return DMSA_OK;
}


// The constants of the problem model the device-resident loop reads (IMU factors / gravity and odometry measurements) and the kernel
// argument that points at them.  Called by the upload entry points after the host model (ctx->win / ctx->key) is initialised.
int upload_loop_model(dmsa_ctx* ctx) {
    auto put = [&](DevBuf& buf, const void* src, size_t bytes) -> int {
        HIPCHK(buf.ensure(bytes + 16));
        if (bytes) HIPCHK(hipMemcpy(buf.p, src, bytes, hipMemcpyHostToDevice));
        return DMSA_OK;
    };
    LoopModel m{};
    if (ctx->model == MODEL_WINDOW) {
        const WindowHost& w = ctx->win;
        m.model = 1, m.n = w.ctrl.n, m.P = w.ctrl.num_params(), m.extra = w.num_extra_rows();
        m.stamps = ctx->d_stamps.as<double>(), m.fhw = ctx->d_fhw.as<double>(), m.traj_time = ctx->d_trajtime.as<double>();
        m.imu = w.imu_consts();
        m.imu.param_indices = nullptr, m.imu.preint_rot = m.imu.preint_pos = m.imu.preint_vel = m.imu.cov_inv = nullptr;
        if (w.use_imu) {
            CHK(put(ctx->d_imu_idx, w.param_indices.data(), w.param_indices.size() * 4));
            CHK(put(ctx->d_imu_rot, w.preint_rot.data(), w.preint_rot.size() * 8));
            CHK(put(ctx->d_imu_pos, w.preint_pos.data(), w.preint_pos.size() * 8));
            CHK(put(ctx->d_imu_vel, w.preint_vel.data(), w.preint_vel.size() * 8));
            CHK(put(ctx->d_imu_cov, w.cov_inv.data(), w.cov_inv.size() * 8));
            m.imu.param_indices = ctx->d_imu_idx.as<int>(), m.imu.preint_rot = ctx->d_imu_rot.as<double>(), m.imu.preint_pos = ctx->d_imu_pos.as<double>();
            m.imu.preint_vel = ctx->d_imu_vel.as<double>(), m.imu.cov_inv = ctx->d_imu_cov.as<double>();
        }
    } else {
        const KeyframeHost& k = ctx->key;
        m.model = 2, m.n = k.frames.n, m.P = k.frames.num_params(), m.extra = k.num_extra_rows();
        m.key = k.row_consts();
        m.key.measured_gravity = nullptr, m.key.gravity_plausible = nullptr, m.key.odom_transl = nullptr, m.key.odom_orient_mat = nullptr;
        if (k.use_gravity) {
            CHK(put(ctx->d_key_grav, k.measured_gravity.data(), k.measured_gravity.size() * 8));
            CHK(put(ctx->d_key_plaus, k.gravity_plausible.data(), k.gravity_plausible.size() * 4));
            m.key.measured_gravity = ctx->d_key_grav.as<double>(), m.key.gravity_plausible = ctx->d_key_plaus.as<int>();
        }
        if (k.use_odometry) {
            CHK(put(ctx->d_key_odom_t, k.odom_transl.data(), k.odom_transl.size() * 8));
            CHK(put(ctx->d_key_odom_R, k.odom_orient_mat.data(), k.odom_orient_mat.size() * 8));
            m.key.odom_transl = ctx->d_key_odom_t.as<double>(), m.key.odom_orient_mat = ctx->d_key_odom_R.as<double>();
        }
    }
    ctx->loop_model = m;
    return DMSA_OK;
}

int upload_common(dmsa_ctx* ctx) {
    CHK(alloc_point_buffers(ctx));
    ctx->gaussians_valid = false;
    ctx->centralized = false;
    ctx->batch = 0;
    ctx->base_table = nullptr;
    ctx->depth_guess[0] = ctx->depth_guess[1] = -1;
    ctx->fit_guess_valid = false;
    return DMSA_OK;
}


void enqueue_wait(dmsa_ctx* ctx, int slot, hipStream_t stream) {
    ctx->wait_seq += 1;
    uint32_t target = ctx->sync_sig[slot];
    int spins = 1 << 23;
    if (ctx->dbg.sync_fault > 0 && ctx->wait_seq == ctx->dbg.sync_fault) target += 1, spins = 1 << 12;  // test hook: a signal that never comes
    launch_sync_wait(ctx->sync_counter(slot), target, ctx->sync_timed_out(), stream, spins);
}
bool sync_wait_timed_out(dmsa_ctx* ctx, std::string* what) {
    if (!ctx->d_sync.p) return false;
    int32_t t[3] = {0, 0, 0};
    const bool read = hipDeviceSynchronize() == hipSuccess && hipMemcpy(t, ctx->sync_timed_out(), sizeof(t), hipMemcpyDeviceToHost) == hipSuccess;
    if (read && t[0] == 0) return false;
    if (what) *what = read ? "waited for " + std::to_string(t[1]) + ", counter at " + std::to_string(t[2]) : std::string("flag unreadable");
    // start over: nothing is in flight after the synchronisation above
    (void)hipMemset(ctx->d_sync.p, 0, SYNC_SLOTS * 4);
    (void)hipDeviceSynchronize();
    for (uint32_t& v : ctx->sync_sig) v = 0;
    ctx->tables_pending = false;
    return true;
}
// ---- the parts of an upload that do not depend on how the points arrive (flat arrays: below; strided PCL containers: aos_upload.cpp) ----
int ensure_stage(dmsa_ctx* ctx, size_t bytes) {
    if (bytes > ctx->h_stage_cap) {
        HIPCHK(hipStreamSynchronize(ctx->stream));  // the old area may still feed a copy
        if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
        ctx->h_stage = nullptr, ctx->h_stage_cap = 0;
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_stage), bytes + bytes / 8, hipHostMallocDefault));
        ctx->h_stage_cap = bytes + bytes / 8;
    }
    return DMSA_OK;
}
int window_upload_begin(dmsa_ctx* ctx, const dmsa_window_problem* p, int64_t N, int64_t S) {
    CHK(set_device(ctx));
    if (!(p->min_grid_size > 0.0f)) {
        ctx->err = "invalid window problem (min_grid_size <= 0)";
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
    ctx->N = N, ctx->S = S, ctx->n = N + S;
    ctx->rows = p->n_total + 1;
    HIPCHK(ctx->d_local.ensure((size_t)ctx->n * 16 + 16));
    HIPCHK(ctx->d_ring.ensure((size_t)ctx->n * 4 + 16));
    return DMSA_OK;
}
int window_upload_finish(dmsa_ctx* ctx, const dmsa_window_problem* p) {
    const int C = ctx->win.ctrl.n;
    HIPCHK(ctx->d_stamps.ensure((size_t)C * 8));
    HIPCHK(ctx->d_fhw.ensure((size_t)C * 8));
    HIPCHK(ctx->d_trajtime.ensure((size_t)p->n_total * 8));
    HIPCHK(hipMemcpy(ctx->d_stamps.p, ctx->win.stamps.data(), (size_t)C * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->d_fhw.p, ctx->win.fh.w.data(), (size_t)C * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->d_trajtime.p, ctx->win.traj_time.data(), (size_t)p->n_total * 8, hipMemcpyHostToDevice));
    ctx->win.ctrl.relative_to_global();
    ctx->min_grid_size = p->min_grid_size;
    CHK(upload_loop_model(ctx));
    return upload_common(ctx);
}
int keyframes_upload_begin(dmsa_ctx* ctx, const dmsa_keyframe_problem* p, int64_t n_points) {
    if (!p->rel_orient || !p->rel_transl || !(p->min_grid_size > 0.0f)) {
        ctx->err = "invalid keyframe problem (null pose arrays or min_grid_size <= 0)";
        return DMSA_ERR_INVALID;
    }
    CHK(set_device(ctx));
    if (!ctx->key.init(*p)) return DMSA_ERR_INVALID;
    ctx->model = MODEL_KEYFRAMES;
    ctx->n = n_points, ctx->N = ctx->n, ctx->S = 0;
    ctx->rows = p->num_frames + 1;
    const size_t n = (size_t)ctx->n;
    HIPCHK(ctx->d_local.ensure(n * 16 + 16));
    HIPCHK(ctx->d_nlocal.ensure(n * 16 + 16));
    HIPCHK(ctx->d_nglobal.ensure(n * 16 + 16));
    HIPCHK(ctx->d_ring.ensure(n * 4 + 16));
    return DMSA_OK;
}
int keyframes_upload_finish(dmsa_ctx* ctx, const dmsa_keyframe_problem* p) {
    ctx->min_grid_size = p->min_grid_size;
    CHK(upload_loop_model(ctx));
    return upload_common(ctx);
}
void write_back_poses(const PoseChain& c, double* rel_o, double* rel_t) {
    std::copy(c.rel_o.begin(), c.rel_o.end(), rel_o);
    std::copy(c.rel_t.begin(), c.rel_t.end(), rel_t);
}


extern "C" {

void dmsa_default_debug_options(dmsa_debug_options* o) {
    if (!o) return;
    o->device_loop = 1, o->dual_stream = 1, o->serial_streams = 3, o->merge_sort = 0, o->key_compress = 1, o->fused_segments = 1, o->sort_prehist = 0;
    o->overlap_batch = 1, o->serial_tree = 1, o->host_threads = 16, o->solve_threads = 12, o->host_timeline = 0, o->trace_time = 0, o->fused_leaf_scan = 1, o->device_sync = 1, o->shared_rotations = 1;
    o->eval_skip = 1, o->sync_fault = 0, o->speculation_fault = 0, o->voxel_coherence = 0, o->lm_stream = 1, o->stream_priority = 0, o->gap_stamps = 0, o->lattice_hint = 1, o->fit_classes = 7, o->eigen_l1_bytes = 32 * 1024, o->small_threshold = 0, o->skip_stats = 0;
    o->small_voxel = 0, o->long_split = 1, o->long_log2 = 0, o->sort_items = 0, o->trial_rows_aside = 1;
}
// DMSA_DEBUG="name=value,name=value": the one environment variable of the library (include/dmsa_debug.h)
static void apply_debug_env(dmsa_debug_options* o) {
    const char* e = std::getenv("DMSA_DEBUG");
    if (!e) return;
    struct Field {
        const char* name;
        int32_t* v;
    } fields[] = {{"device_loop", &o->device_loop},     {"dual_stream", &o->dual_stream},   {"serial_streams", &o->serial_streams}, {"merge_sort", &o->merge_sort},
                  {"key_compress", &o->key_compress},   {"fused_segments", &o->fused_segments}, {"sort_prehist", &o->sort_prehist},
                  {"overlap_batch", &o->overlap_batch}, {"serial_tree", &o->serial_tree},   {"host_threads", &o->host_threads},     {"solve_threads", &o->solve_threads},
                  {"host_timeline", &o->host_timeline}, {"trace_time", &o->trace_time},     {"fused_leaf_scan", &o->fused_leaf_scan}, {"device_sync", &o->device_sync},
                  {"shared_rotations", &o->shared_rotations}, {"eval_skip", &o->eval_skip}, {"sync_fault", &o->sync_fault}, {"speculation_fault", &o->speculation_fault}, {"voxel_coherence", &o->voxel_coherence}, {"lm_stream", &o->lm_stream}, {"stream_priority", &o->stream_priority}, {"gap_stamps", &o->gap_stamps}, {"lattice_hint", &o->lattice_hint}, {"fit_classes", &o->fit_classes}, {"eigen_l1_bytes", &o->eigen_l1_bytes}, {"small_threshold", &o->small_threshold}, {"skip_stats", &o->skip_stats}, {"small_voxel", &o->small_voxel}, {"long_split", &o->long_split}, {"long_log2", &o->long_log2}, {"sort_items", &o->sort_items}, {"trial_rows_aside", &o->trial_rows_aside}};
    std::string text(e);
    size_t at = 0;
    while (at < text.size()) {
        size_t end = text.find(',', at);
        if (end == std::string::npos) end = text.size();
        const std::string item = text.substr(at, end - at);
        const size_t eq = item.find('=');
        if (eq != std::string::npos) {
            const std::string name = item.substr(0, eq);
            bool known = false;
            for (auto& f : fields)
                if (name == f.name) *f.v = std::atoi(item.c_str() + eq + 1), known = true;
            if (!known) std::fprintf(stderr, "[dmsa] DMSA_DEBUG: unknown switch '%s' ignored\n", name.c_str());
        }
        at = end + 1;
    }
}
int dmsa_create(int device, uint32_t flags, dmsa_ctx** out) { return dmsa_create_ex(device, flags, nullptr, out); }
int dmsa_create_ex(int device, uint32_t flags, const dmsa_debug_options* options, dmsa_ctx** out) {
    return dmsa_create_ex2(device, flags, options, (uint32_t)sizeof(dmsa_debug_options), out);
}
int dmsa_create_ex2(int device, uint32_t flags, const dmsa_debug_options* options, uint32_t options_bytes, dmsa_ctx** out) {
    if (!out) return DMSA_ERR_INVALID;
    dmsa_debug_options dbg;
    dmsa_default_debug_options(&dbg);
    if (options) {
        // the struct is append-only since round 6: an older caller's shorter struct sets its leading fields, the rest keeps the defaults
        if (options_bytes < 4 || options_bytes % 4 != 0 || options_bytes > sizeof(dmsa_debug_options)) return DMSA_ERR_INVALID;
        std::memcpy(&dbg, options, options_bytes);
    }
    apply_debug_env(&dbg);
    dbg.serial_streams = std::max(1, std::min(3, dbg.serial_streams)), dbg.serial_tree = std::max(0, std::min(3, dbg.serial_tree));
    dbg.host_threads = std::max(1, std::min(64, dbg.host_threads)), dbg.solve_threads = std::max(1, std::min(16, dbg.solve_threads));
    if (dbg.eigen_l1_bytes < 4096) dbg.eigen_l1_bytes = 32 * 1024;  // (Eigen's own default when cpuid reports nothing)
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return DMSA_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return DMSA_ERR_NO_DEVICE;
    dmsa_ctx* ctx = new (std::nothrow) dmsa_ctx();
    if (!ctx) return DMSA_ERR_NOMEM;
    if (flags & ~(DMSA_FLAG_POSE_TABLE_HOST | DMSA_FLAG_FIXED_ITERS | DMSA_FLAG_MIRROR_SUMS | DMSA_FLAG_STAGE_TIMERS)) {
        delete ctx;
        return DMSA_ERR_INVALID;  // e.g. 0x10, the opt-in fast sums of rounds 1-4: retired, the library has ONE summation order (the reference's)
    }
    ctx->device = device, ctx->flags = flags;
    ctx->dbg = dbg;
    if (dbg.sort_items != 0) sort_set_items_override(dbg.sort_items);
    ctx->compress_keys = dbg.key_compress != 0, ctx->overlap_batch = dbg.overlap_batch != 0, ctx->device_loop = dbg.device_loop != 0;
    ctx->fused_segments = dbg.fused_segments != 0, ctx->prehist = dbg.sort_prehist != 0, ctx->dual_stream = dbg.dual_stream != 0;
    ctx->merge_sort = dbg.merge_sort < 0 ? -1 : (dbg.merge_sort != 0 ? 1 : 0);
    ctx->serial_two_streams = dbg.serial_streams != 1, ctx->serial_three_streams = dbg.serial_streams >= 3;
    // stream_priority: bit 0 / 1 / 2 = main / second / third stream at the device's highest priority (the wave dispatcher then serves that
    // queue first when kernels of several streams compete for the CUs)
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    auto make_stream = [&](hipStream_t* st, int bit) {
        return hipStreamCreateWithPriority(st, hipStreamNonBlocking, ((dbg.stream_priority >> bit) & 1) ? prio_greatest : prio_least / 2 + prio_greatest / 2);
    };
    if (make_stream(&ctx->stream, 0) != hipSuccess || make_stream(&ctx->stream2, 1) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_scan0, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_counts, hipEventDisableTiming) != hipSuccess ||
        make_stream(&ctx->stream3, 2) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_join3, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_tables, hipEventDisableTiming) != hipSuccess) {
        delete ctx;
        return DMSA_ERR_HIP;
    }
    if (hipHostMalloc(reinterpret_cast<void**>(&ctx->h_rb), sizeof(dmsa_ctx::Readback), hipHostMallocDefault) != hipSuccess) {
        delete ctx;
        return DMSA_ERR_NOMEM;
    }
    std::memset(ctx->h_rb, 0, sizeof(dmsa_ctx::Readback));
    ctx->h_lattice = ctx->h_rb->lattice;
    // counters of the device-side stream dependencies (loop_kernels.h): zero BEFORE any stream of this context can look at them -- the
    // three streams are not ordered among themselves, and a recycled allocation still holds the counts of the context that freed it
    if (ctx->d_sync.ensure(SYNC_SLOTS * 4) != hipSuccess || hipMemsetAsync(ctx->d_sync.p, 0, SYNC_SLOTS * 4, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
        dmsa_destroy(ctx);
        return DMSA_ERR_HIP;
    }
    *out = ctx;
    return DMSA_OK;
}

void dmsa_destroy(dmsa_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    drain_timers(ctx);
    for (hipEvent_t e : ctx->free_events) (void)hipEventDestroy(e);
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    if (ctx->h_xpin) (void)hipHostFree(ctx->h_xpin);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->h_rb) (void)hipHostFree(ctx->h_rb);
    if (ctx->h_Hp) (void)hipHostFree(ctx->h_Hp);
    if (ctx->h_results) (void)hipHostFree(ctx->h_results);
    DevBuf* bufs[] = {&ctx->d_local, &ctx->d_nlocal, &ctx->d_ring, &ctx->d_global, &ctx->d_nglobal, &ctx->d_tables, &ctx->d_ctrl, &ctx->d_stamps,
                      &ctx->d_fhw, &ctx->d_trajtime, &ctx->d_aabb, &ctx->d_lattice, &ctx->d_code[0], &ctx->d_code[1], &ctx->d_idx[0], &ctx->d_idx[1],
                      &ctx->d_code_s[0], &ctx->d_code_s[1], &ctx->d_idx_s[0], &ctx->d_idx_s[1], &ctx->d_leaf_incl[0], &ctx->d_leaf_incl[1],
                      &ctx->d_leaf_start[0], &ctx->d_leaf_start[1], &ctx->d_counts, &ctx->d_memb_local, &ctx->d_memb_idx, &ctx->d_memb_g, &ctx->d_seg_off,
                      &ctx->d_info12, &ctx->d_order, &ctx->d_fit_sums, &ctx->d_pow_codes, &ctx->d_memb_q, &ctx->d_tablesT, &ctx->d_pad_off, &ctx->d_E, &ctx->d_ne_partial, &ctx->d_Hp, &ctx->d_sq_partial, &ctx->d_sq_out};
    for (DevBuf* b : bufs) b->release();
    if (ctx->sp) {
        for (DevBuf* b : ctx->sp->all) b->release();
        delete ctx->sp;
    }
    delete ctx->pool;
    for (int l = 0; l < 2; ++l)
        for (DevBuf* b : {&ctx->d_head[l], &ctx->d_slot_acc[l], &ctx->d_slot_cnt[l], &ctx->d_gauss_of_slot[l], &ctx->d_memb_of_slot[l], &ctx->d_pslot_of_slot[l], &ctx->d_pos_slot_rank[l],
                          &ctx->d_nsorted[l], &ctx->d_pair_d[l], &ctx->d_sort_tmp[l], &ctx->d_scan_tmp[l]})
            b->release();
    (void)hipStreamSynchronize(ctx->stream2);
    (void)hipEventDestroy(ctx->ev_fork), (void)hipEventDestroy(ctx->ev_scan0), (void)hipEventDestroy(ctx->ev_join), (void)hipEventDestroy(ctx->ev_counts);
    (void)hipStreamSynchronize(ctx->stream3);
    (void)hipEventDestroy(ctx->ev_join3), (void)hipEventDestroy(ctx->ev_tables);
    (void)hipStreamDestroy(ctx->stream3);
    (void)hipStreamDestroy(ctx->stream2);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* dmsa_last_error(const dmsa_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

void dmsa_default_settings(dmsa_settings* s) {
    if (!s) return;
    s->num_iter = 15, s->epsilon = 1e-5, s->use_analytic_jacobi = 0, s->step_length_optim = 0.05, s->max_step = 0.01, s->gauss_split = 0;
    s->grid_size_1_factor = 2.0f, s->grid_size_2_factor = 5.0f, s->min_num_points_per_set = 6, s->min_num_gaussians = 30;
    s->lambda_diag = 0.00001f, s->use_centralization = 1;
}

int dmsa_window_upload(dmsa_ctx* ctx, const dmsa_window_problem* p) {
    if (!ctx || !p || p->num_points < 0 || p->num_static < 0) return DMSA_ERR_INVALID;
    if ((p->num_points > 0 && (!p->xyz_local || !p->tform_idx || !p->ring_id)) || (p->num_static > 0 && (!p->xyz_static || !p->ring_id_static))) {
        ctx->err = "invalid window problem (null point arrays)";
        return DMSA_ERR_INVALID;
    }
    CHK(window_upload_begin(ctx, p, p->num_points, p->num_static));
    const size_t n = (size_t)ctx->n;
    // local points: (x, y, z, row index); static points ride along with the identity row.  Packed into pinned memory by a few
    // host threads (the user's arrays are pageable), then one DMA per array.
    CHK(ensure_stage(ctx, n * 20 + 64));
    float* loc = reinterpret_cast<float*>(ctx->h_stage);
    int32_t* ring = reinterpret_cast<int32_t*>(ctx->h_stage + n * 16);
    const int32_t id_row = p->n_total;
    const int64_t N = ctx->N, S = ctx->S;
    std::atomic<bool> bad_row{false};
    auto pack = [&](int64_t i0, int64_t i1) {
        for (int64_t i = i0; i < i1; ++i) {
            if (i < N) {
                const int32_t row = p->tform_idx[i];
                if (row < 0 || row >= p->n_total) {
                    bad_row = true;
                    return;
                }
                loc[4 * i] = p->xyz_local[4 * i], loc[4 * i + 1] = p->xyz_local[4 * i + 1], loc[4 * i + 2] = p->xyz_local[4 * i + 2];
                std::memcpy(&loc[4 * i + 3], &row, 4);
                ring[i] = p->ring_id[i];
            } else {
                const int64_t k = i - N;
                loc[4 * i] = p->xyz_static[4 * k], loc[4 * i + 1] = p->xyz_static[4 * k + 1], loc[4 * i + 2] = p->xyz_static[4 * k + 2];
                std::memcpy(&loc[4 * i + 3], &id_row, 4);
                ring[i] = p->ring_id_static[k];
            }
        }
    };
    if (n < 131072) {
        pack(0, N + S);
    } else {
        workers(ctx).run_all([&](int t, int nt) { pack((N + S) * t / nt, (N + S) * (t + 1) / nt); });
    }
    if (bad_row) {
        ctx->err = "tform_idx out of range";
        return DMSA_ERR_INVALID;
    }
    HIPCHK(hipMemcpyAsync(ctx->d_local.p, loc, n * 16, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_ring.p, ring, n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));  // the staging buffer is reused by the next upload
    return window_upload_finish(ctx, p);
}

int dmsa_keyframes_upload(dmsa_ctx* ctx, const dmsa_keyframe_problem* p) {
    if (!ctx || !p || p->num_frames < 2) return DMSA_ERR_INVALID;
    if (!p->frame_offset || !p->xyz_local || !p->normal_local || !p->ring_id) {
        ctx->err = "invalid keyframe problem (null point arrays)";
        return DMSA_ERR_INVALID;
    }
    if (p->frame_offset[0] != 0) {
        ctx->err = "invalid keyframe problem (frame_offset[0] != 0)";
        return DMSA_ERR_INVALID;
    }
    for (int k = 0; k < p->num_frames; ++k)
        if (p->frame_offset[k + 1] < p->frame_offset[k]) {
            ctx->err = "invalid keyframe problem (frame_offset not non-decreasing)";
            return DMSA_ERR_INVALID;
        }
    const int F = p->num_frames;
    CHK(keyframes_upload_begin(ctx, p, p->frame_offset[F]));
    const size_t n = (size_t)ctx->n;
    std::vector<float> loc(n * 4);
    for (int k = 0; k < F; ++k)
        for (int64_t i = p->frame_offset[k]; i < p->frame_offset[k + 1]; ++i) {
            loc[4 * i] = p->xyz_local[4 * i], loc[4 * i + 1] = p->xyz_local[4 * i + 1], loc[4 * i + 2] = p->xyz_local[4 * i + 2];
            const int32_t row = k;
            std::memcpy(&loc[4 * i + 3], &row, 4);
        }
    HIPCHK(hipMemcpy(ctx->d_local.p, loc.data(), n * 16, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->d_nlocal.p, p->normal_local, n * 16, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->d_ring.p, p->ring_id, n * 4, hipMemcpyHostToDevice));
    return keyframes_upload_finish(ctx, p);
}


}  // extern "C"
