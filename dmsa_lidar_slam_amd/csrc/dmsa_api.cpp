// dmsa_api.cpp — the C ABI of libdmsa_hip.so (include/dmsa_hip.h) and the host-side optimizeSet loop.
//
// Host/device split: the control flow of DmsaOptimizer::optimizeSet (DmsaOptimizer.h:54-150),
// the control-pose chain, parameter vectors, additional error rows and the P x P solve run here in double;
// every O(#points) stage is a HIP kernel on device-resident data (dmsa_kernels.hip).  Per iteration only pose
// tables / control poses go to the device and (P+1)^2 + 9 doubles plus a few counters come back.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dmsa_hip.h"
#include "../../include/dmsa_static_points.h"
#include "../../include/dmsa_window_setup.h"
#include "../../include/dmsa_wire_formats.h"
#include "../../include/dmsa_keyframe_cloud.h"
#include "device_prims.h"
#include "radix_sort_dev.h"
#include "dmsa_kernels.h"
#include "host_math.h"
#include "loop_kernels.h"
#include "serial_kernels.h"
#include "static_kernels.h"

using namespace dmsa;

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }  // every buffer of a context is released with it, whether or not dmsa_destroy lists it
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr, cap = 0;
        const size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr, cap = 0;
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

enum Model { MODEL_NONE = 0, MODEL_WINDOW = 1, MODEL_KEYFRAMES = 2 };

enum TimerSlot { T_RESIDUAL = 0, T_VOXEL, T_FIT, T_TABLE, T_NORMAL, T_TOTAL, T_COUNT };

struct EventPair {
    hipEvent_t a, b;
    int slot;
};

// scratch of the static-point functions (include/dmsa_static_points.h); allocated on first use, independent of the resident problem
struct StaticState {
    DevBuf cloud, query, normal, ring, code, idx, code_s, idx_s, pts_sorted, table, flags, sel, scan, sort_tmp, scan_tmp, out_xyz, out_id, offsets, small,
        aabb, lattice, head, incl, leaf_start, counts, rnd, pick;
    DevBuf* all[27] = {&cloud, &query, &normal, &ring, &code, &idx, &code_s, &idx_s, &pts_sorted, &table, &flags, &sel, &scan, &sort_tmp, &scan_tmp, &out_xyz,
                       &out_id, &offsets, &small, &aabb, &lattice, &head, &incl, &leaf_start, &counts, &rnd, &pick};
    // the cell grid currently built over `cloud`
    CellGrid grid{};
    int64_t n_cloud = 0;
    uint32_t num_finite = 0;
    bool key32 = false;
    uint32_t table_mask = 0;
};

// A few persistent host threads for the O(#poses x #evaluations) host math (perturbed pose chains of the keyframe pass, host pose
// tables of the parity path, packing of an upload).  Spawning threads per batch cost ~0.5 ms per iteration; the workers sleep on a
// condition variable between batches.
class WorkerPool {
public:
    explicit WorkerPool(int n) {
        for (int t = 0; t < n; ++t) threads_.emplace_back([this, t]() { run(t); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& th : threads_) th.join();
    }
    int size() const { return (int)threads_.size(); }
    // fn(worker_index, num_workers) on every worker; returns when all are done
    void run_all(const std::function<void(int, int)>& fn) {
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn, pending_ = (int)threads_.size(), ++generation_;
        }
        cv_.notify_all();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this]() { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    void run(int t) {
        long seen = 0;
        while (true) {
            const std::function<void(int, int)>* fn = nullptr;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&]() { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_, fn = fn_;
            }
            (*fn)(t, (int)threads_.size());
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int, int)>* fn_ = nullptr;
    int pending_ = 0;
    long generation_ = 0;
    bool stop_ = false;
};

}  // namespace

struct dmsa_ctx {
    int device = 0;
    uint32_t flags = 0;
    hipStream_t stream = nullptr, stream2 = nullptr;  // stream2 carries the second voxel level only
    hipStream_t stream3 = nullptr;                    // the short tier of the correspondence kernels (DMSA_SERIAL_STREAMS=2: with the throughput tier on stream2, =1: everything on `stream`)
    hipEvent_t ev_join3 = nullptr, ev_tables = nullptr;
    bool tables_pending = false;  // the current batch's pose tables were enqueued on stream2 (ev_tables marks their end)
    int tablesT_batch = 0;        // d_tablesT holds the transposed tables of a batch of this size (0: stale)
    bool serial_three_streams = true;
    hipEvent_t ev_fork = nullptr, ev_scan0 = nullptr /* end of the size classes: the read-back stream waits for it */, ev_join = nullptr, ev_counts = nullptr;
    bool dual_stream = true;  // DMSA_DUAL_STREAM=0: both levels on `stream`
    std::string err;

    Model model = MODEL_NONE;
    int64_t n = 0, N = 0, S = 0;  // total points, moving points, static points
    int rows = 0;                 // pose-table rows incl. the identity row (n_total+1 or F+1)
    WindowHost win;
    KeyframeHost key;
    bool centralized = false;
    float min_grid_size = 0.3f;

    // device-resident problem
    DevBuf d_local, d_nlocal, d_ring, d_global, d_nglobal;
    // pose tables of the current batch
    DevBuf d_tables, d_ctrl, d_stamps, d_fhw, d_trajtime;
    int batch = 0;
    // pinned staging ring for the per-batch control poses (so the H2D copy needs no host synchronisation)
    double* h_pin = nullptr;
    size_t h_pin_slot = 0;  // doubles per slot
    int h_pin_next = 0;
    double* h_xpin = nullptr;  // the same for the additional rows of a batch
    size_t h_xpin_slot = 0;
    int h_xpin_next = 0;
    std::vector<float> h_tables;
    // pinned staging of the point upload (packed on several host threads, then one DMA per array)
    char* h_stage = nullptr;
    size_t h_stage_cap = 0;
    // voxelisation
    // per-resolution scratch: the two voxelisations of an iteration run concurrently on `stream` and `stream2`
    DevBuf d_aabb, d_lattice, d_code[2], d_idx[2], d_code_s[2], d_idx_s[2], d_head[2], d_leaf_incl[2], d_leaf_start[2], d_slot_acc[2], d_slot_cnt[2],
        d_gauss_of_slot[2], d_memb_of_slot[2], d_pslot_of_slot[2], d_pos_slot_rank[2], d_nsorted[2], d_pair_d[2], d_sort_tmp[2], d_scan_tmp[2], d_counts;
    // Small device->host read-backs land in PINNED memory: an async copy into pageable memory blocks the host for 20-30 us.
    struct Readback {
        LatticeTable lattice[2];
        GaussCounts g;
        TileCounts t;
        SerialCounts sc;  // d_counts holds the three structs back to back
        double errs[16];
    };
    Readback* h_rb = nullptr;  // hipHostMalloc
    double* h_Hp = nullptr;     // pinned (P+1)^2 read-back of the normal equations
    size_t h_Hp_cap = 0;
    LatticeTable* h_lattice = nullptr;  // = h_rb->lattice
    DevBuf d_seg_state[2];           // look-back state of k_leaf_segments (ticket counter + one word per tile), zeroed when allocated
    uint32_t seg_epoch[2] = {0, 0}, seg_ticket[2] = {0, 0};
    bool prehist = false;            // DMSA_SORT_PREHIST=1: the key kernels count the sort digits (measured 1.5 % slower than the sort's own histogram pass)
    bool fused_segments = true;      // DMSA_FUSED_SEGMENTS=0: head flags / library scan / leaf starts as three kernels
    bool key32[2] = {false, false};  // leaf codes are 32-bit (both levels share the width: they are sorted together)
    // level views into the shared code / index arrays (level 1 starts n entries behind level 0)
    void* code_v[2] = {nullptr, nullptr};
    void* code_s_v[2] = {nullptr, nullptr};
    uint32_t* idx_v[2] = {nullptr, nullptr};
    uint32_t* idx_s_v[2] = {nullptr, nullptr};
    int depth_guess[2] = {-1, -1};   // tree depths of the previous voxelisation (speculation: saves one host sync)
    int bits_guess[2] = {-1, -1};    // leaf-code widths of the previous voxelisation
    bool compress_keys = true;       // drop the constant high key bits before sorting (DMSA_KEY_COMPRESS=0 disables)
    bool overlap_batch = true;       // host math of the Jacobian batch while the GPU voxelises (DMSA_OVERLAP_BATCH=0 disables)
    bool device_loop = true;         // DMSA_DEVICE_LOOP=0: drive the default path's loop from the host as rounds 1-2 did
    int merge_sort = -1;             // -1: by size; DMSA_MERGE_SORT=0/1 forces two sorts / one sort of both levels
    double level_res[2] = {0, 0};
    // Gaussians
    DevBuf d_memb_local, d_memb_idx, d_memb_g, d_seg_off, d_info12, d_wg_seg;
    DevBuf d_order;  // reference-order path: Gaussians by descending size class
    DevBuf d_tablesT;                                          // pose tables of the current batch, transposed ([row][evaluation][12])
    bool order_valid = false;
    DevBuf d_fit_sums;           // six centred product sums per Gaussian (fit kernels -> finish kernel)
    bool fit_guess_valid = false;  // serial_counts of the previous voxelisation may size this one's speculative fit launches
    SerialCounts serial_counts{0, 0, 0, 0};
    bool E_is_jacobian = false;  // the matrix-core normal equations (P > 64) rewrote the residual batch as the columns of [J | e0]
    bool aabb_fresh = false;  // d_aabb / the zeroed counters belong to the current d_global (launch_transform_aabb ran last)
    const float* base_table = nullptr;  // the pose table d_global was computed with (the fit re-derives the members' global coordinates from it)
    // ---- device-resident optimizeSet loop (loop_kernels.h) ----
    LoopModel loop_model{};      // built at upload: device pointers to the model's constants
    DevBuf d_imu_idx, d_imu_rot, d_imu_pos, d_imu_vel, d_imu_cov;           // window model, IMU factor rows
    DevBuf d_key_grav, d_key_plaus, d_key_odom_t, d_key_odom_R;             // keyframe model, gravity / odometry rows
    DevBuf d_loop_state;   // three chain states: start of the iteration, after the Jacobian batch, after the line search
    DevBuf d_loop_vec;     // paramVec[P] | step[P]
    DevBuf d_ctrl0;        // global poses of the base table (n x 6)
    DevBuf d_table0;       // the base pose table (own buffer: the Jacobian batch's tables are written beside the fit that still reads it)
    DevBuf d_loop_extra;   // additional rows of the two batches: [1+P][a] | [9][a]
    DevBuf d_loop_iter;    // LoopFlags | IterResult[num_iter]
    DevBuf d_panel_work;   // scratch of the blocked device solve for P > 64 (published panels, inverse, hand-over flags)
    uint32_t panel_epoch = 0;
    IterResult* h_results = nullptr;  // pinned
    int h_results_cap = 0;
    // one extra device->host copy riding on the counts read-back of build_gaussians (the previous iteration's IterResult)
    const void* rb_extra_src = nullptr;
    void* rb_extra_dst = nullptr;
    size_t rb_extra_bytes = 0;
    bool serial_two_streams = true;  // DMSA_SERIAL_STREAMS=1: all tiers of the reference-order correspondence kernels on one stream
    DevBuf d_memb_tile, d_tiles, d_tile_rows, d_fallback, d_pad_off;
    int num_tiles = 0, num_fallback = 0, tile_max_rows = 0, tile_max_gauss = 0;
    bool use_tiles = true;  // DMSA_K4_TILES=0 selects the streaming kernel
    bool tiles_usable = true;  // false when a tile references more pose rows than the tiled kernels' LDS holds (very long windows)
    int M = 0, M1 = 0;
    int64_t Mm = 0;
    int num_wg = 0;
    int cfg_num_wg = 768, cfg_big_n = 512;  // correspondence-kernel launch shape (DMSA_K4_WGS / DMSA_K4_BIG override)
    bool gaussians_valid = false;
    // residual batches
    DevBuf d_E, d_ne_partial, d_Hp, d_sq_partial, d_sq_out;
    int64_t ldE = 0;
    int extra_rows = 0;
    // timing
    std::vector<EventPair> pending;
    std::vector<hipEvent_t> free_events;
    double t_ms[T_COUNT] = {0, 0, 0, 0, 0, 0};
    int64_t residual_launches = 0, residual_evals = 0;
    double residual_bytes = 0.0, residual_unit_bytes = 0.0;
    int evaluations = 0;
    std::vector<dmsa_iter_trace> trace;
    StaticState* sp = nullptr;
    WorkerPool* pool = nullptr;  // created on first use
};

namespace {

#define HIPCHK(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess) {                                                                               \
            ctx->err = std::string(#expr) + ": " + hipGetErrorString(_e);                                     \
            return DMSA_ERR_HIP;                                                                              \
        }                                                                                                     \
    } while (0)

#define CHK(expr)                  \
    do {                           \
        int _rc = (expr);          \
        if (_rc != DMSA_OK) return _rc; \
    } while (0)

WorkerPool& workers(dmsa_ctx* ctx) {
    if (!ctx->pool) {
        unsigned want = 16;  // DMSA_HOST_THREADS overrides (host pose tables of the parity path, perturbed keyframe chains, upload packing)
        if (const char* e = std::getenv("DMSA_HOST_THREADS")) want = (unsigned)std::max(1, std::atoi(e));
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
struct ScopedTimer {
    dmsa_ctx* ctx;
    EventPair ev;
    bool on;
    ScopedTimer(dmsa_ctx* c, int slot) : ctx(c) {
        // the correspondence kernel is always timed (roofline contract); other stages only on request
        on = slot == T_RESIDUAL || (c->flags & DMSA_FLAG_STAGE_TIMERS) != 0;
        if (!on) return;
        ev.a = get_event(c), ev.b = get_event(c), ev.slot = slot;
        (void)hipEventRecord(ev.a, c->stream);
    }
    ~ScopedTimer() {
        if (!on) return;
        (void)hipEventRecord(ev.b, ctx->stream);
        ctx->pending.push_back(ev);
    }
};
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
    HIPCHK(ctx->d_counts.ensure(sizeof(GaussCounts) + sizeof(TileCounts) + sizeof(SerialCounts)));  // read back together
    // memberships: every point belongs to at most one set per resolution
    HIPCHK(ctx->d_memb_local.ensure(2 * n * 16));
    HIPCHK(ctx->d_memb_idx.ensure(2 * n * 4));
    HIPCHK(ctx->d_memb_g.ensure(2 * n * 4));
    HIPCHK(ctx->d_seg_off.ensure((2 * n + 2) * 4));
    // sets have >= 2 members (two distinct ids) -- except the second half of a splitSet, which may keep a single member when
    // min_num_points_per_set <= 1: size for one set per membership
    HIPCHK(ctx->d_info12.ensure((2 * n + 16) * 48));
    HIPCHK(ctx->d_wg_seg.ensure(4096 * 4));
    if (ctx->flags & DMSA_FLAG_MIRROR_SUMS) {
        // one entry per Gaussian, and M can approach 2n (see d_info12 above)
        HIPCHK(ctx->d_order.ensure((2 * n + 16) * 4));
        HIPCHK(ctx->d_fit_sums.ensure((2 * n + 16) * 6 * 8));
    }
    // tiled correspondence kernels: windows of 3T/4 members plus own-tile Gaussians (> T/4 members each) and their
    // neighbours: tiles <= (4/3 + 8)*Mm/T + 1 with Mm <= 2n; one row list of `rows` entries per tile
    const size_t max_tiles = 41 * n / (size_t)tile_points() + 64;  // windows + own tiles + one head per kTileGauss Gaussians (M <= n)
    HIPCHK(ctx->d_memb_tile.ensure(tile_slot_capacity(n) * 16));
    HIPCHK(ctx->d_pad_off.ensure((2 * n + 2) * 4));
    HIPCHK(ctx->d_tiles.ensure(max_tiles * sizeof(TileDesc)));
    HIPCHK(ctx->d_tile_rows.ensure(max_tiles * (size_t)ctx->rows * 4));
    HIPCHK(ctx->d_fallback.ensure((2 * n / (size_t)tile_points() + 16) * 8));  // single-Gaussian tiles: > T members each
    return DMSA_OK;
}

// ---- pose tables ------------------------------------------------------------------------------------------
// `globs`: B x (C or F) x 6 doubles (axis-angle | translation) of the GLOBAL poses of every evaluation in the batch.
int build_tables(dmsa_ctx* ctx, int B, const std::vector<double>& globs, hipStream_t stream = nullptr) {
    if (stream == nullptr) stream = ctx->stream;
    if (ctx->tables_pending && stream == ctx->stream) {  // an earlier batch's tables may still be in flight on the second stream
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
    if (ctx->h_results) (void)hipHostFree(ctx->h_results);
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
        if ((ctx->flags & DMSA_FLAG_MIRROR_SUMS) && B > 1) {
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

// DMSA_HOST_TIMELINE=1: host-side time stamps of the last iteration's phases (where the host enqueues, where it waits)
struct HostTimeline {
    bool on = std::getenv("DMSA_HOST_TIMELINE") != nullptr;
    std::vector<std::pair<const char*, std::chrono::steady_clock::time_point>> marks;
    void reset() { marks.clear(); }
    void mark(const char* what) {
        if (on) marks.emplace_back(what, std::chrono::steady_clock::now());
    }
    void print() const {
        if (!on || marks.size() < 2) return;
        std::fprintf(stderr, "[host timeline]");
        for (size_t i = 1; i < marks.size(); ++i)
            std::fprintf(stderr, " %s %.0f |", marks[i].first, std::chrono::duration<double, std::micro>(marks[i].second - marks[i - 1].second).count());
        std::fprintf(stderr, " total %.0f us\n", std::chrono::duration<double, std::micro>(marks.back().second - marks.front().second).count());
    }
};
HostTimeline g_tl;

// ---- Gaussians (DmsaOptimizer.h:78-96) ---------------------------------------------------------------------
// `overlap` (optional) runs on the host after every voxelisation kernel has been enqueued and before the counts are read
// back: host work placed there hides behind the GPU.
int build_gaussians(dmsa_ctx* ctx, const dmsa_settings& s, const std::function<int()>& overlap = nullptr, bool allow_speculation = true,
                    bool allow_compression = true) {
    const int64_t n = ctx->n;
    ctx->gaussians_valid = false;
    ctx->order_valid = false;
    ctx->M = 0, ctx->M1 = 0, ctx->Mm = 0;
    const bool lvl_on[2] = {s.grid_size_1_factor > std::numeric_limits<float>::min(), s.grid_size_2_factor > std::numeric_limits<float>::min()};
    // createGaussianSets(set, factor * minGridSize, ...): float product, widened to double by the octree constructor
    ctx->level_res[0] = (double)(s.grid_size_1_factor * ctx->min_grid_size);
    ctx->level_res[1] = (double)(s.grid_size_2_factor * ctx->min_grid_size);
    if (!lvl_on[0]) ctx->level_res[0] = ctx->level_res[1];
    if (!lvl_on[1]) ctx->level_res[1] = ctx->level_res[0];
    const bool compress = ctx->compress_keys && allow_compression;
    const bool speculate = allow_speculation && ctx->depth_guess[0] >= 0 && ctx->depth_guess[1] >= 0 && ctx->depth_guess[0] < 20 && ctx->depth_guess[1] < 20 &&
                           (!compress || (ctx->bits_guess[0] >= 0 && ctx->bits_guess[1] >= 0));
    // the key kernels count the digits of the sort that follows (own sort, 32-bit codes): no clearing kernel, no histogram pass
    const bool prehist = sort_is_onesweep() && ctx->prehist;
    {
        ScopedTimer tm(ctx, T_VOXEL);
        const int nb = (int)((n + kAabbBlock - 1) / kAabbBlock);
        if (!ctx->aabb_fresh)  // the device loop's fused transform already left the block bounds and cleared the counters
            launch_block_aabb(ctx->d_global.as<float4>(), n, ctx->d_aabb.as<float>(), ctx->d_counts.p, sizeof(GaussCounts) + sizeof(TileCounts) + sizeof(SerialCounts),
                              ctx->stream);
        ctx->aabb_fresh = false;
        launch_lattice(ctx->d_global.as<float4>(), n, ctx->d_aabb.as<float>(), nb, ctx->level_res[0], ctx->level_res[1], compress,
                       ctx->d_lattice.as<LatticeTable>(), prehist ? ctx->d_sort_tmp[0].p : nullptr, prehist ? ctx->d_sort_tmp[1].p : nullptr, ctx->stream);
        if (!speculate) {  // sync #1: tree depths select the radix-sort bit range (speculation reads them with the counts instead)
            HIPCHK(hipMemcpyAsync(ctx->h_lattice, ctx->d_lattice.p, 2 * sizeof(LatticeTable), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(sync_spin(ctx->stream));
        }
    }
    // The sort only needs an UPPER bound of the tree depth.  From the second iteration on the previous depths are used
    // without waiting for the lattice kernel; the true depths arrive with the counts (sync #2) and a too-small guess (the
    // bounding box doubled between two iterations) re-runs the voxelisation synchronously.
    int sort_depth[2], sort_bits[2];
    for (int l = 0; l < 2; ++l) {
        sort_depth[l] = speculate ? ctx->depth_guess[l] : ctx->h_lattice[l].final_depth;
        // width of the leaf codes: all 3*depth bits, or (compressed) only the bits that vary over the points
        sort_bits[l] = !compress ? 3 * sort_depth[l] : (speculate ? ctx->bits_guess[l] : ctx->h_lattice[l].total_bits);
        if (!speculate && lvl_on[l] && ctx->h_lattice[l].status != 0) return ctx->h_lattice[l].status;
    }
    GaussCounts* counts = ctx->d_counts.as<GaussCounts>();
    const bool tiles_on = ctx->use_tiles && !(ctx->flags & DMSA_FLAG_MIRROR_SUMS);
    const bool split = s.gauss_split != 0 && ctx->model == MODEL_KEYFRAMES;
    if (split)
        for (int l = 0; l < 2; ++l) {
            HIPCHK(ctx->d_pos_slot_rank[l].ensure((size_t)n * 4));
            HIPCHK(ctx->d_nsorted[l].ensure((size_t)n * 16));
            HIPCHK(ctx->d_pair_d[l].ensure(split_scratch_bytes(n)));
        }
    // The two resolutions are independent until their member lists are appended (level 1 starts at level 0's totals):
    // level 0 runs on `stream`, level 1 on `stream2`; their launches are enqueued stage by stage so that both streams fill.
    const bool two = ctx->dual_stream && lvl_on[0] && lvl_on[1];
    hipStream_t st[2] = {ctx->stream, two ? ctx->stream2 : ctx->stream};
    bool k32v[2] = {false, false};
    // Both resolutions are keyed into one array of 2n (code, point) pairs -- level 1 carries a tag bit above the widest code --
    // and sorted by ONE radix sort: half the launches, twice the parallelism per pass, and the sorted halves are the two levels.
    const int tag_bit = std::max(sort_bits[0], sort_bits[1]) + 1;  // bit `sort_bits` is the marker of non-finite points
    const unsigned end_bit = (unsigned)(tag_bit + 1);
    const bool k32 = end_bit <= 32;
    {
        const size_t ksz = k32 ? 4 : 8;
        for (int l = 0; l < 2; ++l) {
            k32v[l] = ctx->key32[l] = k32;
            ctx->code_v[l] = ctx->d_code[0].as<char>() + (size_t)l * n * ksz, ctx->code_s_v[l] = ctx->d_code_s[0].as<char>() + (size_t)l * n * ksz;
            ctx->idx_v[l] = ctx->d_idx[0].as<uint32_t>() + (size_t)l * n, ctx->idx_s_v[l] = ctx->d_idx_s[0].as<uint32_t>() + (size_t)l * n;
        }
    }
    // Small clouds (keyframe sets: 3 x 10^5 points) are launch-bound: one sort of 2n pairs on one stream.  Large clouds (the window:
    // 1.5 x 10^6) keep the two levels on two streams with one sort each (a sort only looks at the bits below its end bit, so the
    // tag is inert there).
    const bool merged = ctx->merge_sort < 0 ? n <= (int64_t)(1 << 20) : ctx->merge_sort != 0;
    const bool prepared = prehist && k32;
    SortPlan plan[2];  // merged: one sort of 2n pairs in workspace 0, both key kernels count into its header
    for (int l = 0; l < 2; ++l)
        plan[l] = merged ? sort_pairs_u32_plan(ctx->d_sort_tmp[0].p, (size_t)(2 * n), end_bit)
                         : sort_pairs_u32_plan(ctx->d_sort_tmp[l].p, (size_t)n, (unsigned)(sort_bits[l] + 1));
    auto stage_keys = [&](int l, hipStream_t stream) {  // a disabled level is keyed with the other level's lattice (level_res is aliased) and ignored later
        SortPlan pl = plan[l];
        if (merged && l == 1) pl.state_words = 0;  // the look-back words of the common sort are cleared once
        launch_voxel_keys(ctx->d_global.as<float4>(), n, ctx->d_lattice.as<LatticeTable>() + l, ctx->level_res[l], ctx->code_v[l], k32, ctx->idx_v[l],
                          l == 0 ? 0ull : (1ull << tag_bit), prepared ? &pl : nullptr, stream);
    };
    auto stage_sort_both = [&]() -> int {
        stage_keys(0, ctx->stream), stage_keys(1, ctx->stream);
        if (k32 && prepared)
            HIPCHK(sort_pairs_u32_onesweep(ctx->d_sort_tmp[0].p, ctx->d_sort_tmp[0].cap, ctx->d_code[0].as<uint32_t>(), ctx->d_code_s[0].as<uint32_t>(),
                                           ctx->d_idx[0].as<uint32_t>(), ctx->d_idx_s[0].as<uint32_t>(), (size_t)(2 * n), end_bit, ctx->stream, true));
        else if (k32)
            HIPCHK(sort_pairs_u32_u32(ctx->d_sort_tmp[0].p, ctx->d_sort_tmp[0].cap, ctx->d_code[0].as<uint32_t>(), ctx->d_code_s[0].as<uint32_t>(),
                                      ctx->d_idx[0].as<uint32_t>(), ctx->d_idx_s[0].as<uint32_t>(), (size_t)(2 * n), end_bit, ctx->stream));
        else
            HIPCHK(sort_pairs_u64_u32(ctx->d_sort_tmp[0].p, ctx->d_sort_tmp[0].cap, ctx->d_code[0].as<uint64_t>(), ctx->d_code_s[0].as<uint64_t>(),
                                      ctx->d_idx[0].as<uint32_t>(), ctx->d_idx_s[0].as<uint32_t>(), (size_t)(2 * n), end_bit, ctx->stream));
        return DMSA_OK;
    };
    auto stage_sort = [&](int l) -> int {
        stage_keys(l, st[l]);
        const unsigned eb = (unsigned)(sort_bits[l] + 1);
        if (k32 && prepared)
            HIPCHK(sort_pairs_u32_onesweep(ctx->d_sort_tmp[l].p, ctx->d_sort_tmp[l].cap, (const uint32_t*)ctx->code_v[l], (uint32_t*)ctx->code_s_v[l], ctx->idx_v[l],
                                           ctx->idx_s_v[l], (size_t)n, eb, st[l], true));
        else if (k32)
            HIPCHK(sort_pairs_u32_u32(ctx->d_sort_tmp[l].p, ctx->d_sort_tmp[l].cap, (const uint32_t*)ctx->code_v[l], (uint32_t*)ctx->code_s_v[l], ctx->idx_v[l],
                                      ctx->idx_s_v[l], (size_t)n, eb, st[l]));
        else
            HIPCHK(sort_pairs_u64_u32(ctx->d_sort_tmp[l].p, ctx->d_sort_tmp[l].cap, (const uint64_t*)ctx->code_v[l], (uint64_t*)ctx->code_s_v[l], ctx->idx_v[l],
                                      ctx->idx_s_v[l], (size_t)n, eb, st[l]));
        return DMSA_OK;
    };
    auto stage_leaves = [&](int l) -> int {
        const LatticeTable* tab = ctx->d_lattice.as<LatticeTable>() + l;
        const bool k32 = k32v[l];
        if (ctx->fused_segments) {
            // one single-pass kernel; its look-back state is never cleared (epoch-tagged words, running ticket counter)
            const size_t need = 8 * (size_t)(1 + leaf_segment_tiles(n));
            if (need > ctx->d_seg_state[l].cap) {
                HIPCHK(ctx->d_seg_state[l].ensure(need));
                HIPCHK(hipMemsetAsync(ctx->d_seg_state[l].p, 0, ctx->d_seg_state[l].cap, st[l]));
                ctx->seg_epoch[l] = 0, ctx->seg_ticket[l] = 0;
            }
            ctx->seg_epoch[l] += 1;
            if (ctx->seg_epoch[l] == 0) ctx->seg_epoch[l] = 1;
            launch_leaf_segments(ctx->code_s_v[l], k32, n, tab, ctx->d_leaf_incl[l].as<int32_t>(), ctx->d_leaf_start[l].as<int32_t>(), &counts->level[l],
                                 ctx->d_seg_state[l].as<unsigned long long>(), ctx->seg_epoch[l], ctx->seg_ticket[l], st[l]);
            ctx->seg_ticket[l] += (uint32_t)leaf_segment_tiles(n);
        } else {
            launch_head_flags(ctx->code_s_v[l], k32, n, tab, ctx->d_head[l].as<int32_t>(), st[l]);
            HIPCHK(inclusive_scan_i32(ctx->d_scan_tmp[l].p, ctx->d_scan_tmp[l].cap, ctx->d_head[l].as<int32_t>(), ctx->d_leaf_incl[l].as<int32_t>(), (size_t)n, st[l]));
            launch_leaf_starts(ctx->d_head[l].as<int32_t>(), ctx->d_leaf_incl[l].as<int32_t>(), ctx->code_s_v[l], k32, tab, n,
                               ctx->d_leaf_start[l].as<int32_t>(), &counts->level[l], st[l]);
        }
        launch_leaf_accept(ctx->d_leaf_start[l].as<int32_t>(), ctx->idx_s_v[l], ctx->d_ring.as<int32_t>(), &counts->level[l],
                           s.min_num_points_per_set, n, ctx->d_slot_acc[l].as<int32_t>(), ctx->d_slot_cnt[l].as<int32_t>(), st[l]);
        if (split)
            launch_leaf_split(ctx->d_leaf_incl[l].as<int32_t>(), ctx->d_leaf_start[l].as<int32_t>(), ctx->idx_s_v[l], ctx->d_ring.as<int32_t>(),
                              ctx->d_nglobal.as<float4>(), &counts->level[l], s.min_num_points_per_set, n, ctx->d_nsorted[l].as<float4>(),
                              ctx->d_pair_d[l].as<unsigned long long>(), ctx->d_slot_acc[l].as<int32_t>(), ctx->d_slot_cnt[l].as<int32_t>(),
                              ctx->d_pos_slot_rank[l].as<int32_t>(), st[l]);
        launch_leaf_scan(ctx->d_slot_acc[l].as<int32_t>(), ctx->d_slot_cnt[l].as<int32_t>(), ctx->d_gauss_of_slot[l].as<int32_t>(), ctx->d_memb_of_slot[l].as<int32_t>(),
                         ctx->d_pslot_of_slot[l].as<int32_t>(), &counts->level[l], st[l]);
        return DMSA_OK;
    };
    auto stage_gather = [&](int l, hipStream_t gs) {
        const LatticeTable* tab = ctx->d_lattice.as<LatticeTable>() + l;
        launch_gather_members(ctx->d_leaf_incl[l].as<int32_t>(), ctx->d_leaf_start[l].as<int32_t>(), ctx->idx_s_v[l],
                              ctx->code_s_v[l], k32v[l], tab, ctx->d_slot_acc[l].as<int32_t>(), ctx->d_gauss_of_slot[l].as<int32_t>(),
                              ctx->d_memb_of_slot[l].as<int32_t>(), split ? ctx->d_pos_slot_rank[l].as<int32_t>() : nullptr, ctx->d_local.as<float4>(),
                              ctx->d_slot_cnt[l].as<int32_t>(), counts, l, n, ctx->d_memb_local.as<float4>(), ctx->d_memb_idx.as<int32_t>(),
                              ctx->d_memb_g.as<int32_t>(), ctx->d_seg_off.as<int32_t>(), ctx->d_pslot_of_slot[l].as<int32_t>(), ctx->d_pad_off.as<int32_t>(), gs);
    };
    {
        ScopedTimer tm(ctx, T_VOXEL);
        if (merged) CHK(stage_sort_both());
        if (two) {
            HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
            HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
        }
        if (!merged)
            for (int l = 0; l < 2; ++l)
                if (lvl_on[l]) CHK(stage_sort(l));
        for (int l = 0; l < 2; ++l) {
            if (!lvl_on[l]) continue;
            CHK(stage_leaves(l));
        }
        // Both gathers on the first stream (level 1 appends behind level 0's totals anyway): the level-1 chain ends with its leaf scan,
        // long before level 0's gather is through, so the wait below finds its event signalled -- a join at the END of a stream costs
        // ~20 us of cross-queue signalling in front of everything that follows.
        if (two) HIPCHK(hipEventRecord(ctx->ev_join, ctx->stream2));
        if (lvl_on[0]) stage_gather(0, ctx->stream);
        if (two) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        if (lvl_on[1]) stage_gather(1, ctx->stream);
    }
    if (!tiles_on) {
        ScopedTimer tm(ctx, T_FIT);
        for (int l = 0; l < 2; ++l)
            if (lvl_on[l])
                launch_gauss_fit(ctx->d_seg_off.as<int32_t>(), ctx->d_memb_idx.as<int32_t>(), ctx->d_global.as<float4>(), counts, l, ctx->d_info12.as<float>(),
                                 (ctx->flags & DMSA_FLAG_MIRROR_SUMS) != 0, ctx->stream);
    }
    // Tiles of whole Gaussians for the fit and the correspondence kernel; their counts travel with M / Mm.
    TileCounts htc{};
    if (tiles_on) {
        ScopedTimer tm(ctx, T_FIT);
        launch_build_tiles(ctx->d_seg_off.as<int32_t>(), counts, ctx->d_memb_local.as<float4>(), ctx->d_memb_g.as<int32_t>(), ctx->rows, ctx->d_tiles.as<TileDesc>(),
                           reinterpret_cast<TileCounts*>(ctx->d_counts.as<GaussCounts>() + 1), ctx->d_fallback.as<int2>(), ctx->d_memb_tile.as<float4>(), ctx->d_tile_rows.as<int32_t>(),
                           ctx->d_pad_off.as<int32_t>(), ctx->stream);
    }
    const bool classes_on = !tiles_on && (ctx->flags & DMSA_FLAG_MIRROR_SUMS) != 0;
    if (classes_on)  // size classes of the reference-order correspondence kernels: needs only seg_off, so it runs before the read-back
        launch_size_classes(ctx->d_seg_off.as<int32_t>(), counts, ctx->d_order.as<uint32_t>(),
                            reinterpret_cast<SerialCounts*>(ctx->d_counts.as<char>() + sizeof(GaussCounts) + sizeof(TileCounts)), ctx->stream);
    // The read-back of the counts runs on the third stream: a device-to-host copy ends with a system-scope release that holds up the
    // stream it is on for ~20 us, and the fit behind it does not need to wait for that.
    hipStream_t rb = ctx->dual_stream ? ctx->stream3 : ctx->stream;
    if (rb != ctx->stream) {
        HIPCHK(hipEventRecord(ctx->ev_scan0, ctx->stream));
        HIPCHK(hipStreamWaitEvent(rb, ctx->ev_scan0, 0));
    }
    HIPCHK(hipMemcpyAsync(&ctx->h_rb->g, ctx->d_counts.p, sizeof(GaussCounts) + sizeof(TileCounts) + sizeof(SerialCounts), hipMemcpyDeviceToHost, rb));
    HIPCHK(hipMemcpyAsync(ctx->h_lattice, ctx->d_lattice.p, 2 * sizeof(LatticeTable), hipMemcpyDeviceToHost, rb));  // incl. out_of_range
    if (ctx->rb_extra_bytes)  // device loop: the previous iteration's stop decision travels with the counts
        HIPCHK(hipMemcpyAsync(ctx->rb_extra_dst, ctx->rb_extra_src, ctx->rb_extra_bytes, hipMemcpyDeviceToHost, rb));
    // The fit does not need the counts on the host (fixed grids, device-side tile counts): with the LDS table sized for ALL pose rows
    // it is enqueued right behind the read-back, so the GPU keeps working while the host waits for M (sync #2 waits on an event
    // recorded BEFORE the fit, not on the stream).
    const bool early_fit = tiles_on && (size_t)(ctx->rows + 1) * 48 <= 56 * 1024;
    HIPCHK(hipEventRecord(ctx->ev_counts, rb));
    // Default path: the fit (oracle's tree order) is enqueued BEHIND the read-back as well, with the previous iteration's class
    // counts (+ margin) as grids -- the kernels take the true ranges from device memory, surplus workgroups exit, and whatever the
    // guess missed is launched after sync #2.  The three classes run side by side on two streams (each is latency-bound on its own).
    const int32_t* d_sc = reinterpret_cast<const int32_t*>(ctx->d_counts.as<char>() + sizeof(GaussCounts) + sizeof(TileCounts));
    const float* fit_table = ctx->base_table ? ctx->base_table : ctx->d_tables.as<float>();
    int fit_launched[3] = {0, 0, 0}, finish_launched = 0;
    auto launch_fit = [&](const int first[3], const int tasks[3], int finish_gauss) -> int {
        ScopedTimer tm(ctx, T_FIT);
        static const bool merged_fit = std::getenv("DMSA_FIT_MERGED") == nullptr || std::atoi(std::getenv("DMSA_FIT_MERGED")) != 0;
        if (merged_fit) {
            // one launch for the three size classes and the rebalancing weights: no fork to a second stream, no join (DMSA_FIT_MERGED=0: the
            // classes as three kernels on two streams, the weights behind the long class)
            launch_gauss_fit_all(ctx->d_memb_local.as<float4>(), ctx->d_seg_off.as<int32_t>(), fit_table, ctx->d_order.as<uint32_t>(), d_sc, first, tasks,
                                 ctx->d_fit_sums.as<double>(), counts, ctx->d_info12.as<float>(), true, ctx->stream);
        } else {
            const bool two = ctx->dual_stream && (tasks[1] > 0 || tasks[2] > 0);
            if (two) {
                HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
                HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
            }
            hipStream_t s2 = two ? ctx->stream2 : ctx->stream;
            launch_gauss_fit_tree(ctx->d_memb_local.as<float4>(), ctx->d_seg_off.as<int32_t>(), fit_table, ctx->d_order.as<uint32_t>(), d_sc, 0, first[0], tasks[0],
                                  ctx->d_fit_sums.as<double>(), ctx->stream);
            launch_gauss_fit_tree(ctx->d_memb_local.as<float4>(), ctx->d_seg_off.as<int32_t>(), fit_table, ctx->d_order.as<uint32_t>(), d_sc, 1, first[1], tasks[1],
                                  ctx->d_fit_sums.as<double>(), s2);
            launch_gauss_fit_tree(ctx->d_memb_local.as<float4>(), ctx->d_seg_off.as<int32_t>(), fit_table, ctx->d_order.as<uint32_t>(), d_sc, 2, first[2], tasks[2],
                                  ctx->d_fit_sums.as<double>(), s2);
            launch_rebalancing_weights(ctx->d_seg_off.as<int32_t>(), counts, ctx->d_info12.as<float>(), true, ctx->stream);
            if (two) {
                HIPCHK(hipEventRecord(ctx->ev_join, ctx->stream2));
                HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
            }
        }
        launch_gauss_fit_finish(ctx->d_seg_off.as<int32_t>(), counts, ctx->d_fit_sums.as<double>(), finish_gauss, ctx->d_info12.as<float>(), ctx->stream);
        HIPCHK(hipGetLastError());
        return DMSA_OK;
    };
    if (classes_on && ctx->fit_guess_valid) {
        const SerialCounts& pg = ctx->serial_counts;  // previous iteration
        const int first[3] = {0, 0, 0};
        auto grow = [](int v) { return v + v / 8 + 16; };
        fit_launched[0] = grow(pg.n_long), fit_launched[1] = grow(pg.n_chain - pg.n_long), fit_launched[2] = grow(pg.n_small);
        finish_launched = grow(pg.n_chain + pg.n_small);
        CHK(launch_fit(first, fit_launched, finish_launched));
    }
    if (early_fit) {
        ScopedTimer tm(ctx, T_FIT);
        launch_fit_tiled(ctx->d_memb_tile.as<float4>(), ctx->d_seg_off.as<int32_t>(), ctx->d_tables.as<float>(), ctx->rows + 1, ctx->d_tiles.as<TileDesc>(),
                         reinterpret_cast<TileCounts*>(ctx->d_counts.as<GaussCounts>() + 1), ctx->d_fallback.as<int2>(), ctx->d_tile_rows.as<int32_t>(),
                         ctx->d_info12.as<float>(), ctx->stream);
        launch_rebalancing_weights(ctx->d_seg_off.as<int32_t>(), counts, ctx->d_info12.as<float>(), false, ctx->stream);
    }
    g_tl.mark("voxel enq");
    if (overlap) CHK(overlap());
    g_tl.mark("jacobian batch host+enq");
    {  // sync #2: M sizes every later launch
        hipError_t e;
        while ((e = hipEventQuery(ctx->ev_counts)) == hipErrorNotReady) {
        }
        (void)hipGetLastError();  // see sync_spin
        HIPCHK(e);
    }
    g_tl.mark("sync#2 wait");
    const GaussCounts h = ctx->h_rb->g;
    htc = ctx->h_rb->t;
    for (int l = 0; l < 2; ++l) {
        if (lvl_on[l] && ctx->h_lattice[l].status != 0) return ctx->h_lattice[l].status;
        const int true_bits = compress ? ctx->h_lattice[l].total_bits : 3 * ctx->h_lattice[l].final_depth;
        if (lvl_on[l] && compress && ctx->h_lattice[l].out_of_range) {
            ctx->depth_guess[0] = ctx->depth_guess[1] = -1;
            return build_gaussians(ctx, s, nullptr, false, false);  // a key left the predicted range: redo with full-width codes
        }
        if (speculate && lvl_on[l] && (ctx->h_lattice[l].final_depth > sort_depth[l] || true_bits > sort_bits[l])) {
            ctx->depth_guess[0] = ctx->depth_guess[1] = -1;
            return build_gaussians(ctx, s, nullptr, false, allow_compression);  // mis-speculated: redo (overlap work already ran)
        }
        ctx->depth_guess[l] = ctx->h_lattice[l].final_depth;
        ctx->bits_guess[l] = ctx->h_lattice[l].total_bits;
    }
    ctx->num_tiles = htc.num_tiles, ctx->num_fallback = htc.num_fallback, ctx->tile_max_rows = htc.max_rows, ctx->tile_max_gauss = htc.max_gauss;
    ctx->tiles_usable = !tiles_on || tiled_kernels_fit(htc.max_rows, htc.max_gauss);
    {
        ScopedTimer tm(ctx, T_FIT);
        if (tiles_on && !early_fit && ctx->num_tiles > 0 && !ctx->tiles_usable) {
            // tiles that reference more pose rows than fit in LDS: wave-per-set fit on the gathered members instead
            for (int l = 0; l < 2; ++l)
                if (lvl_on[l])
                    launch_gauss_fit(ctx->d_seg_off.as<int32_t>(), ctx->d_memb_idx.as<int32_t>(), ctx->d_global.as<float4>(), counts, l, ctx->d_info12.as<float>(), false,
                                     ctx->stream);
        } else if (tiles_on && !early_fit && ctx->num_tiles > 0)
            launch_fit_tiled(ctx->d_memb_tile.as<float4>(), ctx->d_seg_off.as<int32_t>(), ctx->d_tables.as<float>(), ctx->tile_max_rows,
                             ctx->d_tiles.as<TileDesc>(), reinterpret_cast<TileCounts*>(ctx->d_counts.as<GaussCounts>() + 1), ctx->d_fallback.as<int2>(), ctx->d_tile_rows.as<int32_t>(),
                             ctx->d_info12.as<float>(), ctx->stream);
        const int M_all = h.level[0].num_gauss + h.level[1].num_gauss;
        if (classes_on) {
            // whatever the pre-sync launches did not cover (first iteration, or a class that grew by more than the margin)
            ctx->serial_counts = ctx->h_rb->sc;
            const SerialCounts& sc = ctx->serial_counts;
            const int want[3] = {sc.n_long, sc.n_chain - sc.n_long, sc.n_small};
            int rest[3], any = 0;
            for (int c = 0; c < 3; ++c) rest[c] = std::max(0, want[c] - fit_launched[c]), any += rest[c];
            if (M_all > 0 && (any > 0 || finish_launched < M_all)) CHK(launch_fit(fit_launched, rest, M_all));
            ctx->fit_guess_valid = M_all > 0;
            ctx->order_valid = true;
        }
        if (!early_fit && !classes_on)
            launch_rebalancing_weights(ctx->d_seg_off.as<int32_t>(), counts, ctx->d_info12.as<float>(), (ctx->flags & DMSA_FLAG_MIRROR_SUMS) != 0, ctx->stream);
    }
    HIPCHK(hipGetLastError());
    ctx->M1 = h.level[0].num_gauss;
    ctx->M = h.level[0].num_gauss + h.level[1].num_gauss;
    ctx->Mm = (int64_t)h.level[0].num_memb + h.level[1].num_memb;
    if (ctx->M > 0) {
        // enough workgroups to fill 256 CUs several times over, but never more workgroups than Gaussians
        int wg = ctx->cfg_num_wg;
        if (wg > ctx->M) wg = ctx->M;
        ctx->num_wg = wg;
        // the workgroup partition only feeds the streaming / parity correspondence kernels
        if (!classes_on && (!tiles_on || ctx->num_tiles == 0 || !ctx->tiles_usable)) launch_segment_partition(ctx->d_seg_off.as<int32_t>(), ctx->M, wg, ctx->d_wg_seg.as<int32_t>(), ctx->stream);
    }
    ctx->gaussians_valid = true;
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
int run_residuals(dmsa_ctx* ctx, int B, const std::vector<double>* extra, const double* d_extra = nullptr) {
    CHK(ensure_E(ctx, B));
    if (ctx->tables_pending) {  // the pose tables of this batch were built on the second stream
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
    const bool tiles_on = ctx->use_tiles && !(ctx->flags & DMSA_FLAG_MIRROR_SUMS) && ctx->num_tiles > 0 && ctx->tiles_usable;
    if (tiles_on) {
        ScopedTimer tm(ctx, T_RESIDUAL);
        launch_residuals_tiled(ctx->d_memb_tile.as<float4>(), ctx->d_memb_local.as<float4>(), ctx->d_seg_off.as<int32_t>(), ctx->d_info12.as<float>(),
                               ctx->d_tables.as<float>(), ctx->rows, ctx->M, B, ctx->d_tiles.as<TileDesc>(), ctx->d_tile_rows.as<int32_t>(), ctx->num_tiles,
                               ctx->tile_max_rows, ctx->tile_max_gauss, ctx->d_fallback.as<int2>(), ctx->num_fallback, ctx->cfg_big_n, ctx->d_E.as<double>(), ctx->ldE,
                               ctx->stream);
    } else if ((ctx->flags & DMSA_FLAG_MIRROR_SUMS) && ctx->order_valid) {
        // reference-order sums (default path): lane = evaluation on transposed pose tables
        HIPCHK(ctx->d_tablesT.ensure((size_t)B * ctx->rows * 48));
        ScopedTimer tm(ctx, T_RESIDUAL);
        if (ctx->tablesT_batch != B) launch_transpose_tables(ctx->d_tables.as<float>(), ctx->rows, B, ctx->d_tablesT.as<float>(), ctx->stream);
        const bool two = ctx->serial_two_streams && ctx->serial_counts.n_long > 0;
        if (two) {  // the latency tier keeps `stream`; the throughput tiers run beside it on stream2
            HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
            HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
        }
        const bool three = two && ctx->serial_three_streams;
        if (three) HIPCHK(hipStreamWaitEvent(ctx->stream3, ctx->ev_fork, 0));
        launch_residuals_serial(ctx->d_memb_local.as<float4>(), ctx->d_seg_off.as<int32_t>(), ctx->d_info12.as<float>(), ctx->d_tablesT.as<float>(), B,
                                ctx->d_order.as<uint32_t>(), ctx->serial_counts, ctx->d_E.as<double>(), ctx->ldE, ctx->stream, two ? ctx->stream2 : ctx->stream,
                                three ? ctx->stream3 : (two ? ctx->stream2 : ctx->stream));
        if (two) {
            HIPCHK(hipEventRecord(ctx->ev_join, ctx->stream2));
            HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        }
        if (three) {
            HIPCHK(hipEventRecord(ctx->ev_join3, ctx->stream3));
            HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join3, 0));
        }
    } else {
        ScopedTimer tm(ctx, T_RESIDUAL);
        launch_residuals(ctx->d_memb_local.as<float4>(), ctx->d_seg_off.as<int32_t>(), ctx->d_info12.as<float>(), ctx->d_tables.as<float>(), ctx->rows, ctx->M,
                         B, ctx->d_wg_seg.as<int32_t>(), ctx->num_wg, ctx->cfg_big_n, ctx->d_E.as<double>(), ctx->ldE, ctx->stream, false);
    }
    ctx->E_is_jacobian = false;
    ctx->residual_launches += 1;
    ctx->residual_evals += B;
    ctx->residual_bytes += 16.0 * (double)ctx->Mm + 48.0 * ctx->M + (double)B * (48.0 * ctx->rows + 8.0 * ctx->M);
    ctx->residual_unit_bytes += (double)B * (16.0 * (double)ctx->Mm + 56.0 * ctx->M + 48.0 * ctx->rows);
    HIPCHK(hipGetLastError());
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

// ---- the optimizeSet loop (DmsaOptimizer.h:54-150) -------------------------------------------------------------
int optimize_impl(dmsa_ctx* ctx, const dmsa_settings& s, dmsa_report* rep);
// A failure inside the loop (HIP error, lattice deeper than 21 levels, allocation) must not leave the resident problem in the centred
// frame: the static points were shifted in place and the window origin lives only in the context.
int optimize_device_loop(dmsa_ctx* ctx, const dmsa_settings& s, dmsa_report* rep);
int optimize(dmsa_ctx* ctx, const dmsa_settings& s, dmsa_report* rep) {
    // default path: the loop state lives on the device (one host wait per iteration); the host-driven loop remains for the opt-in fast
    // sums and for host-built pose tables
    const bool device_loop = (ctx->flags & DMSA_FLAG_MIRROR_SUMS) && !(ctx->flags & DMSA_FLAG_POSE_TABLE_HOST) && ctx->device_loop;
    const int rc = device_loop ? optimize_device_loop(ctx, s, rep) : optimize_impl(ctx, s, rep);
    if (rc != DMSA_OK && ctx->centralized) {
        const std::string err = ctx->err;
        (void)dmsa_decentralize(ctx);
        ctx->err = err;
    }
    return rc;
}
int optimize_impl(dmsa_ctx* ctx, const dmsa_settings& s, dmsa_report* rep) {
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
        g_tl.reset(), g_tl.mark("start");
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
            ctx->tables_pending = ts != ctx->stream;
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
        if (ctx->flags & DMSA_FLAG_MIRROR_SUMS)
        {   // :113, explicit inverse like the reference
            const ParallelRun par = [&](const std::function<void(int, int)>& fn) { workers(ctx).run_all(fn); };
            lm_solve(H.data(), g.data(), P, s.step_length_optim, step.data(), P >= 64 ? &par : nullptr);
        }
        else
            lm_solve_lu(H.data(), g.data(), P, s.step_length_optim, step.data());
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
            const bool blocked = (ctx->flags & DMSA_FLAG_MIRROR_SUMS) != 0;
            HIPCHK(ctx->d_sq_partial.ensure((size_t)std::max(squared_sums_partial_doubles(rowsE, 9), squared_sums_blocked_partial_doubles(rowsE, P, 9)) * 8));
            HIPCHK(ctx->d_sq_out.ensure(16 * 8));
            if (blocked)
                launch_squared_sums_blocked(ctx->d_E.as<double>(), ctx->ldE, rowsE, P, 9, ctx->d_sq_partial.as<double>(), ctx->d_sq_out.as<double>(), ctx->stream);
            else
                launch_squared_sums(ctx->d_E.as<double>(), ctx->ldE, rowsE, 9, ctx->d_sq_partial.as<double>(), ctx->d_sq_out.as<double>(), ctx->stream);
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
// P > 64 (keyframe sets) still solves the normal equations on the host's worker pool: one more wait per iteration.
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
int device_tables(dmsa_ctx* ctx, int B, const double* d_ctrl, float* tables, float* tablesT, hipStream_t stream) {
    const int np = ctx->loop_model.n;
    if (ctx->model == MODEL_WINDOW)
        launch_window_pose_tables(d_ctrl, ctx->d_stamps.as<double>(), ctx->d_fhw.as<double>(), ctx->d_trajtime.as<double>(), B, np, ctx->rows - 1, tables, tablesT,
                                  stream);
    else
        launch_keyframe_pose_tables(d_ctrl, B, np, tables, tablesT, stream);
    HIPCHK(hipGetLastError());
    return DMSA_OK;
}

// :107-128 on the device: one workgroup for P <= 64, column-block workgroups handing panels to each other beyond
int device_lm_step(dmsa_ctx* ctx, const double* d_Hp, int P, double lambda, double alpha, double max_step, double* d_step, LoopFlags* d_flags) {
    if (P <= kLoopSolveMaxP) {
        launch_loop_lm_step(d_Hp, P, lambda, alpha, max_step, d_step, d_flags, ctx->stream);
    } else {
        const size_t bytes = loop_panel_solve_doubles(P) * 8;
        if (bytes > ctx->d_panel_work.cap) {
            HIPCHK(ctx->d_panel_work.ensure(bytes));
            HIPCHK(hipMemsetAsync(ctx->d_panel_work.p, 0, ctx->d_panel_work.cap, ctx->stream));
            ctx->panel_epoch = 0;
        }
        ctx->panel_epoch += 1;
        if (ctx->panel_epoch == 0) ctx->panel_epoch = 1;
        launch_loop_lm_panels(d_Hp, P, lambda, alpha, max_step, ctx->d_panel_work.as<double>(), ctx->panel_epoch, d_step, d_flags, ctx->stream);
    }
    HIPCHK(hipGetLastError());
    return DMSA_OK;
}

int optimize_device_loop(dmsa_ctx* ctx, const dmsa_settings& s, dmsa_report* rep) {
    ScopedTimer total(ctx, T_TOTAL);
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
    HIPCHK(ctx->d_loop_iter.ensure(sizeof(LoopFlags) + (size_t)(num_iter + 1) * sizeof(IterResult)));
    HIPCHK(ctx->d_Hp.ensure((size_t)(P + 1) * (P + 1) * 8));
    HIPCHK(ctx->d_sq_out.ensure(16 * 8));
    if (num_iter + 1 > ctx->h_results_cap) {
        if (ctx->h_results) (void)hipHostFree(ctx->h_results);
        ctx->h_results = nullptr, ctx->h_results_cap = 0;
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_results), (size_t)(num_iter + 17) * sizeof(IterResult), hipHostMallocDefault));
        ctx->h_results_cap = num_iter + 17;
    }
    std::memset(ctx->h_results, 0, (size_t)ctx->h_results_cap * sizeof(IterResult));
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
    std::vector<double> Hp, H, g, step;
    if (P > kLoopPanelMaxP) Hp.resize((size_t)(P + 1) * (P + 1)), H.resize((size_t)P * P), g.resize((size_t)P), step.resize((size_t)P);
    // what the report says about the Gaussians belongs to the last iteration that really ran
    int last_M = 0, last_M1 = 0;
    int64_t last_Mm = 0;
    int nan_evals = 0;
    hipStream_t side = ctx->dual_stream ? ctx->stream3 : ctx->stream;
    for (int iter = 0; iter < num_iter; ++iter) {
        g_tl.reset(), g_tl.mark("start");
        // :72-75 parameters, chain, base table, global points
        if (iter == 0) launch_loop_begin(m, S0, d_param, ctx->d_ctrl0.as<double>(), d_flags, ctx->stream);  // later iterations: done by loop_finish
        {
            ScopedTimer tm(ctx, T_TABLE);
            CHK(device_tables(ctx, 1, ctx->d_ctrl0.as<double>(), ctx->d_table0.as<float>(), nullptr, ctx->stream));
        }
        ctx->base_table = ctx->d_table0.as<float>();
        {
            ScopedTimer tm(ctx, T_VOXEL);
            launch_transform_aabb(ctx->d_local.as<float4>(), ctx->model == MODEL_KEYFRAMES ? ctx->d_nlocal.as<float4>() : nullptr, ctx->d_table0.as<float4>(),
                                  ctx->d_global.as<float4>(), ctx->model == MODEL_KEYFRAMES ? ctx->d_nglobal.as<float4>() : nullptr, ctx->n,
                                  ctx->d_aabb.as<float>(), ctx->d_counts.p, sizeof(GaussCounts) + sizeof(TileCounts) + sizeof(SerialCounts), ctx->stream);
            ctx->aabb_fresh = true;
        }
        // :99, :199-232 the 1 + P chains, rows and pose tables of the Jacobian batch: beside the voxelisation, they need nothing from it
        if (side != ctx->stream) {
            HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
            HIPCHK(hipStreamWaitEvent(side, ctx->ev_fork, 0));
        }
        launch_loop_chain(m, 0, S0, S1, d_param, d_step, increment, ctx->d_ctrl.as<double>(), d_extra_jac, d_flags, side);
        CHK(device_tables(ctx, 1 + P, ctx->d_ctrl.as<double>(), ctx->d_tables.as<float>(), ctx->d_tablesT.as<float>(), side));
        ctx->batch = 1 + P, ctx->tablesT_batch = 1 + P;
        HIPCHK(hipEventRecord(ctx->ev_tables, side));
        ctx->tables_pending = side != ctx->stream;
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
        if (iter > 0 && ctx->h_results[iter - 1].stop != 0) break;  // the loop ended in the previous iteration: this one never started
        ++iters;
        last_M = ctx->M, last_M1 = ctx->M1, last_Mm = ctx->Mm;
        ctx->trace.push_back(dmsa_iter_trace{ctx->M, ctx->M1, ctx->Mm, 0.0, 0.0, 0, 0});
        if (ctx->M < s.min_num_gaussians) {  // :89-93 -- nothing of this iteration has touched the state the next call starts from (S0)
            stop = DMSA_STOP_FEW_GAUSSIANS;
            break;
        }
        ctx->evaluations += 1 + P;
        CHK(run_residuals(ctx, 1 + P, nullptr, d_extra_jac));
        const int rowsE = ctx->M + ctx->extra_rows;
        {
            ScopedTimer tm(ctx, T_NORMAL);
            HIPCHK(ctx->d_ne_partial.ensure((size_t)normal_equations_partial_doubles(rowsE, P) * 8));
            // P <= 64: the block sums stay unreduced, the solve kernel adds them while it loads the matrix
            launch_normal_equations(ctx->d_E.as<double>(), ctx->ldE, rowsE, P, one_div_incr, ctx->d_ne_partial.as<double>(), ctx->d_Hp.as<double>(), ctx->stream,
                                    P > kLoopSolveMaxP);
        }
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
            lm_solve(H.data(), g.data(), P, s.step_length_optim, step.data(), &par);  // :113
            for (double v : step) host_nan = host_nan || std::isnan(v);
            double* pin = nullptr;
            CHK(pinned_doubles(ctx, (size_t)P, &pin));
            std::memcpy(pin, step.data(), (size_t)P * 8);
            HIPCHK(hipMemcpyAsync(d_step, pin, (size_t)P * 8, hipMemcpyHostToDevice, ctx->stream));
            launch_loop_step_finish(P, s.max_step, d_step, d_flags, ctx->stream);  // NaN test, clamp
            g_tl.mark("solve");
        }
        // :152-182 nine trials, :130-143 decision
        launch_loop_chain(m, 1, S1, S2, d_param, d_step, increment, ctx->d_ctrl.as<double>(), d_extra_trial, d_flags, ctx->stream);
        {
            ScopedTimer tm(ctx, T_TABLE);
            CHK(device_tables(ctx, 9, ctx->d_ctrl.as<double>(), ctx->d_tables.as<float>(), ctx->d_tablesT.as<float>(), ctx->stream));
            ctx->batch = 9, ctx->tablesT_batch = 9;
        }
        if (!host_nan) ctx->evaluations += 9;
        nan_evals = host_nan ? 0 : 9;
        CHK(run_residuals(ctx, 9, nullptr, d_extra_trial));
        {
            ScopedTimer tm(ctx, T_NORMAL);
            HIPCHK(ctx->d_sq_partial.ensure((size_t)std::max(squared_sums_partial_doubles(rowsE, 9), squared_sums_blocked_partial_doubles(rowsE, P, 9)) * 8));
            launch_squared_sums_blocked(ctx->d_E.as<double>(), ctx->ldE, rowsE, P, 9, ctx->d_sq_partial.as<double>(), nullptr, ctx->stream);  // block sums only
        }
        launch_loop_finish(m, S1, S2, S0, d_param, d_step, d_error0, ctx->d_sq_partial.as<double>(), normal_equations_partials(rowsE, P).nsplit, fixed ? 1 : 0,
                           s.epsilon, d_results + iter, d_flags, ctx->d_ctrl0.as<double>(), iter + 1 < num_iter ? 1 : 0, ctx->stream);
        HIPCHK(hipGetLastError());
        g_tl.mark("iteration enq");
        if (host_nan) break;  // the device takes the same decision; nothing more to enqueue
    }
    g_tl.print();
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
    drain_timers(ctx);
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
    if (rep) {
        rep->iterations = iters, rep->stop_reason = stop;
        rep->num_gaussians = ctx->M, rep->num_gaussians_l1 = ctx->M1, rep->num_memberships = ctx->Mm;
        rep->error0 = error0, rep->last_step_norm = stepNorm, rep->last_line_search_k = bestK;
        rep->evaluations = ctx->evaluations;
    }
    return DMSA_OK;
}

void write_back_poses(const PoseChain& c, double* rel_o, double* rel_t) {
    std::copy(c.rel_o.begin(), c.rel_o.end(), rel_o);
    std::copy(c.rel_t.begin(), c.rel_t.end(), rel_t);
}

}  // namespace

// ====================================================================================================================
extern "C" {

int dmsa_create(int device, uint32_t flags, dmsa_ctx** out) {
    if (!out) return DMSA_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return DMSA_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return DMSA_ERR_NO_DEVICE;
    dmsa_ctx* ctx = new (std::nothrow) dmsa_ctx();
    if (!ctx) return DMSA_ERR_NOMEM;
    // the reference's summation order is the default; DMSA_FLAG_FAST_SUMS opts out (internally the default is the MIRROR_SUMS bit)
    flags = (flags & DMSA_FLAG_FAST_SUMS) ? (flags & ~DMSA_FLAG_MIRROR_SUMS) : (flags | DMSA_FLAG_MIRROR_SUMS);
    ctx->device = device, ctx->flags = flags;
    if (const char* e = std::getenv("DMSA_K4_WGS")) ctx->cfg_num_wg = std::max(1, std::min(4000, std::atoi(e)));
    if (const char* e = std::getenv("DMSA_K4_BIG")) ctx->cfg_big_n = std::max(1, std::atoi(e));
    if (const char* e = std::getenv("DMSA_K4_TILES")) ctx->use_tiles = std::atoi(e) != 0;
    if (const char* e = std::getenv("DMSA_KEY_COMPRESS")) ctx->compress_keys = std::atoi(e) != 0;
    if (const char* e = std::getenv("DMSA_OVERLAP_BATCH")) ctx->overlap_batch = std::atoi(e) != 0;
    if (const char* e = std::getenv("DMSA_DEVICE_LOOP")) ctx->device_loop = std::atoi(e) != 0;
    if (const char* e = std::getenv("DMSA_FUSED_SEGMENTS")) ctx->fused_segments = std::atoi(e) != 0;
    if (const char* e = std::getenv("DMSA_SORT_PREHIST")) ctx->prehist = std::atoi(e) != 0;
    if (const char* e = std::getenv("DMSA_DUAL_STREAM")) ctx->dual_stream = std::atoi(e) != 0;
    if (const char* e = std::getenv("DMSA_MERGE_SORT")) ctx->merge_sort = std::atoi(e) != 0 ? 1 : 0;
    if (const char* e = std::getenv("DMSA_SERIAL_STREAMS")) ctx->serial_two_streams = std::atoi(e) != 1, ctx->serial_three_streams = std::atoi(e) >= 3;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_scan0, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_counts, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_join3, hipEventDisableTiming) != hipSuccess ||
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
    DevBuf* bufs[] = {&ctx->d_local, &ctx->d_nlocal, &ctx->d_ring, &ctx->d_global, &ctx->d_nglobal, &ctx->d_tables, &ctx->d_ctrl, &ctx->d_stamps,
                      &ctx->d_fhw, &ctx->d_trajtime, &ctx->d_aabb, &ctx->d_lattice, &ctx->d_code[0], &ctx->d_code[1], &ctx->d_idx[0], &ctx->d_idx[1],
                      &ctx->d_code_s[0], &ctx->d_code_s[1], &ctx->d_idx_s[0], &ctx->d_idx_s[1], &ctx->d_leaf_incl[0], &ctx->d_leaf_incl[1],
                      &ctx->d_leaf_start[0], &ctx->d_leaf_start[1], &ctx->d_counts, &ctx->d_memb_local, &ctx->d_memb_idx, &ctx->d_memb_g, &ctx->d_seg_off,
                      &ctx->d_info12, &ctx->d_wg_seg, &ctx->d_order, &ctx->d_fit_sums, &ctx->d_tablesT, &ctx->d_memb_tile, &ctx->d_tiles, &ctx->d_tile_rows, &ctx->d_fallback, &ctx->d_pad_off, &ctx->d_E, &ctx->d_ne_partial, &ctx->d_Hp, &ctx->d_sq_partial, &ctx->d_sq_out};
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
    CHK(set_device(ctx));
    if ((p->num_points > 0 && (!p->xyz_local || !p->tform_idx || !p->ring_id)) || (p->num_static > 0 && (!p->xyz_static || !p->ring_id_static)) ||
        !(p->min_grid_size > 0.0f)) {
        ctx->err = "invalid window problem (null point arrays or min_grid_size <= 0)";
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
    ctx->N = p->num_points, ctx->S = p->num_static, ctx->n = ctx->N + ctx->S;
    ctx->rows = p->n_total + 1;
    const size_t n = (size_t)ctx->n;
    // local points: (x, y, z, row index); static points ride along with the identity row.  Packed into pinned memory by a few
    // host threads (the user's arrays are pageable), then one DMA per array.
    const size_t stage_bytes = n * 20 + 64;
    if (stage_bytes > ctx->h_stage_cap) {
        if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
        ctx->h_stage = nullptr, ctx->h_stage_cap = 0;
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_stage), stage_bytes + stage_bytes / 8, hipHostMallocDefault));
        ctx->h_stage_cap = stage_bytes + stage_bytes / 8;
    }
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
    HIPCHK(ctx->d_local.ensure(n * 16 + 16));
    HIPCHK(ctx->d_ring.ensure(n * 4 + 16));
    HIPCHK(hipMemcpyAsync(ctx->d_local.p, loc, n * 16, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_ring.p, ring, n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));  // the staging buffer is reused by the next upload
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

int dmsa_keyframes_upload(dmsa_ctx* ctx, const dmsa_keyframe_problem* p) {
    if (!ctx || !p || p->num_frames < 2) return DMSA_ERR_INVALID;
    if (!p->frame_offset || !p->xyz_local || !p->normal_local || !p->ring_id || !p->rel_orient || !p->rel_transl || !(p->min_grid_size > 0.0f)) {
        ctx->err = "invalid keyframe problem (null arrays or min_grid_size <= 0)";
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
    CHK(set_device(ctx));
    if (!ctx->key.init(*p)) return DMSA_ERR_INVALID;
    ctx->model = MODEL_KEYFRAMES;
    const int F = p->num_frames;
    ctx->n = p->frame_offset[F], ctx->N = ctx->n, ctx->S = 0;
    ctx->rows = F + 1;
    const size_t n = (size_t)ctx->n;
    std::vector<float> loc(n * 4);
    for (int k = 0; k < F; ++k)
        for (int64_t i = p->frame_offset[k]; i < p->frame_offset[k + 1]; ++i) {
            loc[4 * i] = p->xyz_local[4 * i], loc[4 * i + 1] = p->xyz_local[4 * i + 1], loc[4 * i + 2] = p->xyz_local[4 * i + 2];
            const int32_t row = k;
            std::memcpy(&loc[4 * i + 3], &row, 4);
        }
    HIPCHK(ctx->d_local.ensure(n * 16 + 16));
    HIPCHK(ctx->d_nlocal.ensure(n * 16 + 16));
    HIPCHK(ctx->d_nglobal.ensure(n * 16 + 16));
    HIPCHK(ctx->d_ring.ensure(n * 4 + 16));
    HIPCHK(hipMemcpy(ctx->d_local.p, loc.data(), n * 16, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->d_nlocal.p, p->normal_local, n * 16, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->d_ring.p, p->ring_id, n * 4, hipMemcpyHostToDevice));
    ctx->min_grid_size = p->min_grid_size;
    CHK(upload_loop_model(ctx));
    return upload_common(ctx);
}

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

int dmsa_build_gaussians(dmsa_ctx* ctx, const dmsa_settings* s, int32_t* M_out, int64_t* Mm_out) {
    if (!ctx || ctx->model == MODEL_NONE || !s) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    CHK(build_gaussians(ctx, *s));
    if (M_out) *M_out = ctx->M;
    if (Mm_out) *Mm_out = ctx->Mm;
    return DMSA_OK;
}

int dmsa_eval_residuals(dmsa_ctx* ctx, double* e_out) {
    if (!ctx || ctx->model == MODEL_NONE || !ctx->gaussians_valid || ctx->batch <= 0 || ctx->M <= 0) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    CHK(run_residuals(ctx, ctx->batch, nullptr));
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
    static const bool trace = std::getenv("DMSA_TRACE_TIME") != nullptr;
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

int dmsa_get_poses(dmsa_ctx* ctx, double* rel_orient, double* rel_transl) {
    if (!ctx || ctx->model == MODEL_NONE || !rel_orient || !rel_transl) return DMSA_ERR_INVALID;
    write_back_poses(chain(ctx), rel_orient, rel_transl);
    return DMSA_OK;
}

// Debug-only (not declared in include/dmsa_hip.h): phase cycle counters of k_residuals_tiles, -DDMSA_PHASE_CLOCKS builds.
int dmsa_debug_phase_clocks(dmsa_ctx* ctx, long long* out, int capacity_tiles) {
    static DevBuf buf;
    if (!ctx) return DMSA_ERR_INVALID;
    CHK(set_device(ctx));
    const size_t bytes = (size_t)capacity_tiles * 64 * sizeof(long long);
    if (out == nullptr) {  // arm
        HIPCHK(buf.ensure(bytes));
        HIPCHK(hipMemset(buf.p, 0, bytes));
        set_phase_clock_buffer(buf.as<long long>());
        return DMSA_OK;
    }
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, buf.p, bytes, hipMemcpyDeviceToHost));
    return ctx->num_tiles;
}

int dmsa_optimize_keyframes(dmsa_ctx* ctx, dmsa_keyframe_problem* p, const dmsa_settings* s, dmsa_report* rep) {
    if (!ctx || !p || !s) return DMSA_ERR_INVALID;
    CHK(dmsa_keyframes_upload(ctx, p));
    CHK(optimize(ctx, *s, rep));
    write_back_poses(ctx->key.frames, p->rel_orient, p->rel_transl);
    return DMSA_OK;
}

}  // extern "C"

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
