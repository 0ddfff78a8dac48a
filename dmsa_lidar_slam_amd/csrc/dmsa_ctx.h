// dmsa_ctx.h — the context of libdmsa_hip.so and what its translation units share (internal; nothing here is part of the ABI).
//
//   context.cpp           create / destroy, debug switches, uploads, device buffers, timers
//   voxelize_driver.cpp   createGaussianSets at both resolutions + fit (DmsaOptimizer.h:78-96): the launch sequence, streams, speculation
//   optimize_loop.cpp     optimizeSet (DmsaOptimizer.h:54-150): the device-resident loop and the host-driven loop
//   dmsa_api.cpp          the C ABI of include/dmsa_hip.h (stage-level entry points, whole calls)
//   next_rows_api.cpp     the C ABI of the rows around the hot path (static points, preProcess, window setup, wire formats, keyframe clouds)
#pragma once
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
#include "../../include/dmsa_debug.h"
#include "../../include/dmsa_static_points.h"
#include "../../include/dmsa_window_setup.h"
#include "../../include/dmsa_wire_formats.h"
#include "../../include/dmsa_keyframe_cloud.h"
#include "../../include/dmsa_aos.h"
#include "device_prims.h"
#include "radix_sort_dev.h"
#include "dmsa_kernels.h"
#include "host_math.h"
#include "loop_kernels.h"
#include "serial_kernels.h"
#include "static_kernels.h"

using namespace dmsa;  // kernels / host math of this library



struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }  // every buffer of a context is released with it, whether or not dmsa_destroy lists it
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        ++reallocations();  // (hipFree waits for the device: a buffer that grows inside the loop costs an iteration its overlap)
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
    static long long& reallocations() {  // process-wide count of growing ensure() calls (debug switch trace_time prints it per iteration)
        static long long n = 0;
        return n;
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


// include/dmsa_window_ring.h: the window's scans resident in HBM (slot s holds up to `cap` points at offset s * cap)
struct WindowRing {
    int num_scans = 0;  // 0: no ring
    int64_t cap = 0;
    int head = 0, filled = 0;  // next slot to overwrite, slots in use
    std::vector<int64_t> count;
    DevBuf xyz, stamp, id;
};

struct dmsa_ctx {
    int device = 0;
    uint32_t flags = 0;
    WindowRing ring;
    dmsa_debug_options dbg{};  // include/dmsa_debug.h: fixed at dmsa_create(_ex); the fields below that mirror it are set from it there
    hipStream_t stream = nullptr, stream2 = nullptr;  // stream2 carries the second voxel level only
    hipStream_t stream3 = nullptr;                    // the short tier of the correspondence kernels (debug switch serial_streams: 2 = with the throughput tier on stream2, 1 = everything on `stream`)
    hipEvent_t ev_join3 = nullptr, ev_tables = nullptr;
    bool tables_pending = false;  // the current batch's pose tables were enqueued on stream2 (ev_tables marks their end)
    int tablesT_batch = 0;        // d_tablesT holds the transposed tables of a batch of this size (0: stale)
    bool serial_three_streams = true;
    hipEvent_t ev_fork = nullptr, ev_scan0 = nullptr /* end of the size classes: the read-back stream waits for it */, ev_join = nullptr, ev_counts = nullptr;
    bool dual_stream = true;  // debug switch dual_stream = 0: both levels on `stream`
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
        SerialCounts sc;  // d_counts holds the two structs back to back
        double errs[16];
    };
    Readback* h_rb = nullptr;  // hipHostMalloc
    double* h_Hp = nullptr;     // pinned (P+1)^2 read-back of the normal equations
    size_t h_Hp_cap = 0;
    LatticeTable* h_lattice = nullptr;  // = h_rb->lattice
    DevBuf d_seg_state[2];           // look-back state of k_leaf_segments (ticket counter + one word per tile), zeroed when allocated
    uint32_t seg_epoch[2] = {0, 0}, seg_ticket[2] = {0, 0};
    DevBuf d_fin_state[2];           // the same for k_leaf_finalize (six words per tile)
    uint32_t fin_epoch[2] = {0, 0}, fin_ticket[2] = {0, 0};
    bool prehist = false;            // debug switch sort_prehist: the key kernels count the sort digits (measured 1.5 % slower than the sort's own histogram pass)
    bool fused_segments = true;      // debug switch fused_segments = 0: head flags / library scan / leaf starts as three kernels
    bool key32[2] = {false, false};  // leaf codes are 32-bit (both levels share the width: they are sorted together)
    // level views into the shared code / index arrays (level 1 starts n entries behind level 0)
    void* code_v[2] = {nullptr, nullptr};
    void* code_s_v[2] = {nullptr, nullptr};
    uint32_t* idx_v[2] = {nullptr, nullptr};
    uint32_t* idx_s_v[2] = {nullptr, nullptr};
    int depth_guess[2] = {-1, -1};   // tree depths of the previous voxelisation (speculation: saves one host sync)
    int bits_guess[2] = {-1, -1};    // leaf-code widths of the previous voxelisation
    bool compress_keys = true;       // drop the constant high key bits before sorting (debug switch key_compress)
    bool overlap_batch = true;       // host math of the Jacobian batch while the GPU voxelises (debug switch overlap_batch)
    bool device_loop = true;         // debug switch device_loop = 0: drive the default path's loop from the host as rounds 1-2 did
    int merge_sort = -1;             // -1: by size; debug switch merge_sort = 0 / 1 forces two sorts / one sort of both levels
    double level_res[2] = {0, 0};
    // Gaussians
    DevBuf d_memb_local, d_memb_idx, d_memb_g, d_seg_off, d_info12;
    DevBuf d_sv_stamps;  // debug switch gap_stamps = 3: phase stamps of k_voxel_small
    DevBuf d_long_split;  // scratch of the latency tier's wide second pass (serial_kernels.h: LongSplit)
    size_t long_split_items = 0;
    uint32_t long_split_epoch = 0;
    DevBuf d_order;  // reference-order path: Gaussians by descending size class
    DevBuf d_tablesT;                                          // pose tables of the current batch, transposed ([row][evaluation][12])
    bool order_valid = false;
    DevBuf d_gap_stamps;         // debug switch gap_stamps: 16 wall-clock slots
    bool lattice_hint_valid = false;  // d_lattice holds the tables of an earlier voxelisation of this context (k_lattice verifies them before it replays)
    int64_t lattice_hints_held = 0, lattice_replays = 0;
    DevBuf d_fit_sums;           // six centred product sums per Gaussian (fit kernels -> finish kernel)
    DevBuf d_memb_q;             // fit: global x | y | z of the members of Gaussians too long for the fit's LDS chunk, [3][2n + 16] floats
    DevBuf d_pow_codes;          // two bits per member count n: how libm's powf(n, -1) differs from 1.0f / n (context.cpp: upload_powm1_codes)
    int64_t pow_n = 0;           // counts covered by d_pow_codes
    DevBuf d_gauss_rows;         // (smallest, largest) pose-table row among the members of every Gaussian, identity row excluded (fit kernel)
    DevBuf d_row_range;          // per evaluation of the Jacobian batch: (first, last) pose-table row that can differ from evaluation 0's (loop chain + pose-table kernels)
    DevBuf d_skip_stats;         // per evaluation of the Jacobian batch: pairs (Gaussian, evaluation) not computed, pairs that differed under eval_skip = 2
    int skip_stats_evals = 0;
    DevBuf d_code_prev[2], d_coh_count;  // debug switch voxel_coherence: last voxelisation's leaf codes per level, and a counter of changed codes
    LatticeTable coh_lattice[2]{};
    bool coh_valid[2] = {false, false}, coh_key32[2] = {false, false}, coh_pending[2] = {false, false}, coh_count_zeroed = false;
    int64_t coh_n[2] = {0, 0};
    int64_t coh_compared = 0, coh_lattice_changes = 0;
    DevBuf d_split_stats;        // splitSet search: 64 x 64 pair blocks looked at / skipped by the cone bound (k_split_pairs)
    int64_t skip_pairs = 0;      // pairs the eval_skip logic looked at since the context was created
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
    int panel_P = -1;       // parameter count the scratch is laid out for
    DevBuf d_rot_same;     // per evaluation of the Jacobian batch: 1 = its control rotations are evaluation 0's bit for bit (pose-table kernels)
    DevBuf d_sync;         // counters of the device-side stream dependencies (dev_sync.h: SyncSlot), zeroed when the context is created
    uint32_t sync_sig[SYNC_SLOTS] = {};  // signals enqueued so far per slot = the value a wait enqueued now has to see
    bool tables_dev_sync = false;        // tables_pending is to be resolved through SYNC_TABLES, not ev_tables
    int wait_seq = 0;                    // waits enqueued by the current whole call (debug switch sync_fault withholds the signal of one of them)
    long long* stamp_voxel = nullptr;    // debug switch gap_stamps = 2: where build_gaussians stamps "voxelisation done" / "fit done"
    long long* stamp_fit = nullptr;
    int voxel_calls = 0;                 // voxelisations of the current whole call (debug switch speculation_fault plants a wrong guess in one of them)
    int small_voxel_launches = 0, small_voxel_fallbacks = 0;  // voxelisations on the one-launch path; voxelisations the small path (small_voxel.hip) handed back to the general path (codes wider than 32 bits)
    int sync_retries = 0, speculation_retries = 0;  // since the context was created: calls re-run with events after a wait timed out; voxelisations re-run after a wrong guess
    DevBuf d_aos_raw, d_aos_idx;         // include/dmsa_aos.h: the caller's strided clouds as they lie in memory, and their per-point indices
    DevBuf d_static_keep;                // the static points as uploaded (a call that has to start over restores them: centralize / decentralize is no exact round trip)
    uint32_t* sync_counter(int slot) const { return d_sync.as<uint32_t>() + slot; }
    int32_t* sync_timed_out() const { return d_sync.as<int32_t>() + SYNC_TIMED_OUT; }
    IterResult* h_results = nullptr;  // pinned
    int h_results_cap = 0;
    // one extra device->host copy riding on the counts read-back of build_gaussians (the previous iteration's IterResult)
    const void* rb_extra_src = nullptr;
    void* rb_extra_dst = nullptr;
    size_t rb_extra_bytes = 0;
    bool serial_two_streams = true;  // debug switch serial_streams = 1: all tiers of the reference-order correspondence kernels on one stream
    DevBuf d_pad_off;            // (written by the member gather: prefix of the member counts rounded up to 8; unused since the tiled kernels are gone)
    int M = 0, M1 = 0;
    int64_t Mm = 0;
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


// ---- shared helpers (context.cpp) ----
WorkerPool& workers(dmsa_ctx* ctx);
hipEvent_t get_event(dmsa_ctx* ctx);
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
void drain_timers(dmsa_ctx* ctx);
hipError_t sync_spin(hipStream_t stream);
int set_device(dmsa_ctx* ctx);
// a one-wave wait on `stream` for everything signalled on `slot` so far (dev_sync.h)
void enqueue_wait(dmsa_ctx* ctx, int slot, hipStream_t stream);
// Did a device-side wait of the call that just ended give up?  Synchronises the device, clears the flag and the counters (whatever the
// call's own status was: a stale flag or a half-counted signal must not reach the next call).  `what` receives the counter values.
bool sync_wait_timed_out(dmsa_ctx* ctx, std::string* what);
int num_params(const dmsa_ctx* ctx);
PoseChain& chain(dmsa_ctx* ctx);
int num_extra_rows(const dmsa_ctx* ctx);
int alloc_point_buffers(dmsa_ctx* ctx);
int upload_powm1_codes(dmsa_ctx* ctx, int64_t counts);
int upload_loop_model(dmsa_ctx* ctx);
int upload_common(dmsa_ctx* ctx);
void write_back_poses(const PoseChain& c, double* rel_o, double* rel_t);
int ensure_stage(dmsa_ctx* ctx, size_t bytes);  // the pinned staging area of the uploads holds at least `bytes`
int window_upload_begin(dmsa_ctx* ctx, const dmsa_window_problem* p, int64_t N, int64_t S);  // host model, sizes, point buffers (points themselves: the caller)
int window_upload_finish(dmsa_ctx* ctx, const dmsa_window_problem* p);
int keyframes_upload_begin(dmsa_ctx* ctx, const dmsa_keyframe_problem* p, int64_t n_points);
int keyframes_upload_finish(dmsa_ctx* ctx, const dmsa_keyframe_problem* p);
struct HostTimeline {
    bool on = false;  // dmsa_debug_options::host_timeline of the context that optimises
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
extern HostTimeline g_tl;
// ---- voxelize_driver.cpp ----
int build_gaussians(dmsa_ctx* ctx, const dmsa_settings& s, const std::function<int()>& overlap = nullptr, bool allow_speculation = true,
                    bool allow_compression = true, bool allow_small = true);
// ---- optimize_loop.cpp ----
int build_tables(dmsa_ctx* ctx, int B, const std::vector<double>& globs, hipStream_t stream = nullptr);
void append_glob(const PoseChain& c, std::vector<double>& out);
void host_set_params(dmsa_ctx* ctx, const double* p);
int transform_points(dmsa_ctx* ctx, int b);
int ensure_E(dmsa_ctx* ctx, int B);
int run_residuals(dmsa_ctx* ctx, int B, const std::vector<double>* extra, const double* d_extra = nullptr, const uint32_t* rot_same = nullptr,
                  const int2* row_range = nullptr /* Jacobian batch of the device loop: pairs equal to evaluation 0 are left out (serial_kernels.h) */);
int device_lm_step(dmsa_ctx* ctx, const double* d_Hp, int P, double lambda, double alpha, double max_step, double* d_step, LoopFlags* d_flags);
int optimize(dmsa_ctx* ctx, const dmsa_settings& s, dmsa_report* rep);
int adaptive_step_size(dmsa_ctx* ctx, double* params, const double* step, double error0, int32_t* best_k);
