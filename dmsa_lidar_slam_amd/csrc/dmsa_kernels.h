// dmsa_kernels.h — launchers of the hand-written gfx950 kernels of the DMSA hot path.
// All kernels are compiled with -ffp-contract=off: the float/double operation sequences that feed voxel keys and
// parity checks must not be fused into FMAs (the reference is built -O1 without -march).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "dev_sync.h"

namespace dmsa {

constexpr int kMaxLatticeEvents = 64;   // bounding-box doublings; depth <= 21 bounds this by 21 + first define
constexpr int kAabbBlock = 1024;        // points per AABB block of the lattice pre-pass
constexpr uint32_t kStaticRowFlag = 0;  // static points use row index == n_rows (identity row appended to every table)

// Final state of PCL's incremental bounding box, as epochs: points inserted at epoch e were keyed with mn[e] at
// depth[e]; suffix_shift[e] is what later re-rootings add to their keys.
struct LatticeTable {
    int32_t num_events;
    int32_t final_depth;
    int32_t status;      // 0 ok, DMSA_ERR_DEPTH if deeper than 21 levels
    int32_t defined;
    int64_t first_idx;   // first finite point
    int64_t event_idx[kMaxLatticeEvents + 1];   // point index that triggered event e (ascending, repeats allowed)
    double mn[kMaxLatticeEvents + 1][3];
    uint32_t suffix_shift[kMaxLatticeEvents + 1][3];
    int32_t depth[kMaxLatticeEvents + 1];
    double final_mn[3];
    // Leaf-code compression: over all finite points the final key of axis a lies in [key_lo[a], key_hi[a]]; the bits above
    // nbits[a] are the same for every point (key_base[a] = common prefix), so the depth-first code only needs the low
    // nbits[a] bits of each axis, interleaved level by level (order-preserving).  total_bits = sum of nbits.
    int32_t nbits[3];
    int32_t total_bits;
    uint32_t key_base[3];
    int32_t compressed;    // 1: codes are the compressed form, invalid marker = 1 << total_bits; 0: full 3*depth-bit codes
    int32_t out_of_range;  // set by k_voxel_keys if a key left [key_base << nbits, ...] (compression must be redone off)
    int32_t pad3;
    // Level tag OR'ed into every code of this resolution (written by k_voxel_keys): both resolutions are sorted in ONE radix
    // sort of 2n pairs, the tag bit sits above the widest code, so the first n sorted entries are level 0 and the rest level 1.
    uint64_t code_or;
};
__host__ __device__ inline uint64_t lattice_invalid_code(const LatticeTable& t) {
    return (1ull << (t.compressed ? t.total_bits : 3 * t.final_depth)) | t.code_or;
}

// per-level device scalars produced by the segmentation stage
struct LevelCounts {
    int32_t num_leaves;
    int32_t num_gauss;     // accepted point sets of this level
    int32_t num_memb;      // their total membership
    int32_t pad;
};

struct GaussCounts {       // both levels, read back once per iteration
    LevelCounts level[2];
    float weight_mean;
    int32_t pad[3];
};

// ---- K2 for small point sets: both resolutions from the lattice to the member lists in ONE launch (small_voxel.hip) --------------
struct SmallVoxelArgs {
    const float4* global;      // transformed points (n)
    const float4* local;       // local points, .w = pose-table row
    const int32_t* ring;       // getIdOfPoint per point
    int n;
    int min_pts;               // min_num_points_per_set
    LatticeTable* tables;      // [2], written by k_lattice in front of this launch
    double res[2];
    uint32_t* code[2];         // leaf codes per point, unsorted / sorted, and the point indices (dmsa_get_voxel_level reads them)
    uint32_t* idx[2];
    uint32_t* code_s[2];
    uint32_t* idx_s[2];
    GaussCounts* counts;       // level[l] = leaves, Gaussians, members; pad[l] = 1: level l needs the general path (codes wider than 32 bits)
    float4* memb_local;
    int32_t* memb_idx;
    int32_t* memb_g;
    int32_t* seg_off;
    uint32_t* sync;            // dev_sync.h counter: level 0's totals are published
    uint32_t sync_target;
    int32_t* timed_out;
    long long* stamps;         // debug (gap_stamps = 3), may be null: [2][16] device wall-clock stamps (100 MHz) of the kernel's phases
};
int small_voxel_max_points();
void launch_voxel_small(const SmallVoxelArgs& a, hipStream_t s);

// ---- K0: rigid transforms -------------------------------------------------------------------------------
// global[i] = T[row(i)] * local[i]  (Matrix4f*Vector4f order), local.w carries the row index as int bits.
void launch_transform(const float4* local, const float4* table, float4* global, int64_t n, hipStream_t s);
// keyframe model: also rotate normals
void launch_transform_normals(const float4* local, const float4* nlocal, const float4* table, float4* global, float4* nglobal, int64_t n, hipStream_t s);
// static points: xyz += sign * origin (float), ContinuousTrajectory::centralize/decentralize
void launch_shift_points(float4* pts, int64_t n, float ox, float oy, float oz, float sign, hipStream_t s);

// ---- K1: dense pose tables from global control poses (device variant) ------------------------------------
// ctrl: B x C x 6 doubles (axis-angle, translation) ; stamps C ; fh_w C ; traj_time n_t ; out: B x (n_t+1) x 12 floats
void launch_window_pose_tables(const double* ctrl, const double* stamps, const double* fh_w, const double* traj_time, int B, int C, int n_t,
                               float* tables, float* tablesT /* may be null: the same rows as [row][B][12] */, hipStream_t s,
                               uint32_t* rot_same = nullptr /* [B], may be null: 1 where all control rotations of an evaluation equal evaluation 0's bit for bit */);
// frames: B x F x 6 doubles global poses -> B x (F+1) x 12
void launch_keyframe_pose_tables(const double* frames, int B, int F, float* tables, float* tablesT, hipStream_t s, uint32_t* rot_same = nullptr);

// include/dmsa_detmath.h evaluated on the device (parity tests): fn 0 sin, 1 cos, 2 acos, 3 atan2(y, x)
void launch_detmath_eval(int fn, const double* x, const double* y, int64_t n, double* out, hipStream_t s);

// ---- K2: PCL-exact voxel lattice + keys -----------------------------------------------------------------
// updateGlobalPoints and the block bounds of the voxelisation that follows in one pass over the points (nlocal / nglobal null for the window model)
void launch_transform_aabb(const float4* local, const float4* nlocal, const float4* table, float4* global, float4* nglobal, int64_t n, float* aabb, void* zero,
                           size_t zero_bytes, hipStream_t s);
// also clears `zero_bytes` (a multiple of 4) at `zero`: the counters of the iteration
void launch_block_aabb(const float4* global, int64_t n, float* aabb /* nb x 8 */, void* zero, size_t zero_bytes, hipStream_t s);
// sort_header0/1 (may be null): sort headers (radix_sort_dev.h) cleared on the side, for key kernels that count the sort digits
void launch_lattice(const float4* global, int64_t n, const float* aabb, int nb, double res0, double res1, bool compress, LatticeTable* tables /* [2] */,
                    void* sort_header0, void* sort_header1, hipStream_t s, uint32_t* done = nullptr /* dev_sync.h counter: += 2 when both tables are written */,
                    bool use_hint = false /* `tables` still hold the events of the previous voxelisation of (nearly) the same points: they are verified in
                                             parallel first and the sequential replay only runs if they no longer hold; LatticeTable::pad3 = 1 says they held */);
// leaf codes are 32-bit (key32) when 3*depth + 1 <= 32, else 64-bit; the buffers are sized for 64-bit keys either way
// `sort` (may be null; 32-bit keys only): the plan of the sort that follows -- the kernel then clears its look-back words and adds the
// digit histograms of the keys it writes to its (cleared) header, and the sort is started with prepared = true
struct SortPlan;
void launch_voxel_keys(const float4* global, int64_t n, LatticeTable* table, double res, void* code, bool key32, uint32_t* idx, uint64_t code_or,
                       const SortPlan* sort, hipStream_t s);
// ---- segmentation of the sorted (code, idx) arrays -------------------------------------------------------
void launch_head_flags(const void* code_sorted, bool key32, int64_t n, const LatticeTable* table, int32_t* head, hipStream_t s);
// the three steps head flags / inclusive scan / leaf starts in one single-pass kernel; `state` = 8 * (1 + leaf_segment_tiles(n)) bytes,
// zeroed once when allocated; epoch > 0 differs from call to call, ticket_base = sum of the tile counts of all earlier calls on this state
int leaf_segment_tiles(int64_t n);
void launch_leaf_segments(const void* code_sorted, bool key32, int64_t n, const LatticeTable* table, int32_t* leaf_incl, int32_t* leaf_start, LevelCounts* counts,
                          unsigned long long* state, uint32_t epoch, uint32_t ticket_base, hipStream_t s);
void launch_leaf_starts(const int32_t* head, const int32_t* leaf_of_pos, const void* code_sorted, bool key32, const LatticeTable* table, int64_t n,
                        int32_t* leaf_start, LevelCounts* counts, hipStream_t s);
void launch_leaf_accept(const int32_t* leaf_start, const uint32_t* idx_sorted, const int32_t* ring, const LevelCounts* counts, int min_pts,
                        int64_t capacity /* leaves */, int32_t* slot_acc /* 2 per leaf */, int32_t* slot_cnt, hipStream_t s);
// keyframe pass: splitSet on accepted leaves; rewrites slot_acc/slot_cnt and fills per-position (set, rank)
size_t split_scratch_bytes(int64_t n);  // pair_best[n] + task list + counter
void launch_leaf_split(const int32_t* leaf_incl, const int32_t* leaf_start, const uint32_t* idx_sorted, const int32_t* ring, const float4* nglobal,
                       const LevelCounts* counts, int min_pts, int64_t n, float4* nsorted /* n */, unsigned long long* pair_best /* split_scratch_bytes(n) */,
                       int32_t* slot_acc, int32_t* slot_cnt, int32_t* pos_slot_rank, hipStream_t s,
                       unsigned long long* block_stats = nullptr /* [2] += 64 x 64 pair blocks looked at / skipped by the cone bound (debug counters) */);
void launch_leaf_scan(const int32_t* slot_acc, const int32_t* slot_cnt, int32_t* gauss_of_slot, int32_t* memb_of_slot,
                      int32_t* pslot_of_slot /* prefix of the member counts rounded up to 8: tile slots */, LevelCounts* counts, hipStream_t s);
// launch_leaf_scan as a multi-workgroup single-pass kernel.  `state`: leaf_finalize_state_bytes(n) bytes zeroed once when allocated;
// epoch / ticket_base as for launch_leaf_segments (tiles per call: leaf_finalize_tiles(n)).
int leaf_finalize_tiles(int64_t n);
size_t leaf_finalize_state_bytes(int64_t n);
void launch_leaf_finalize(const int32_t* slot_acc, const int32_t* slot_cnt, int64_t n /* leaf capacity */, int32_t* gauss_of_slot, int32_t* memb_of_slot,
                          int32_t* pslot_of_slot, LevelCounts* counts, unsigned long long* state, uint32_t epoch, uint32_t ticket_base, hipStream_t s);
// debug switch voxel_coherence: count the points whose leaf code differs from `prev` (if compare), then prev <- now
void launch_count_code_changes(const void* now, void* prev, bool key32, int64_t n, bool compare, unsigned long long* count, hipStream_t s);
void launch_gather_members(const int32_t* leaf_of_pos, const int32_t* leaf_start, const uint32_t* idx_sorted, const void* code_sorted, bool key32,
                           const LatticeTable* table, const int32_t* slot_acc, const int32_t* gauss_of_slot, const int32_t* memb_of_slot,
                           const int32_t* pos_slot_rank /* or null */, const float4* local, const int32_t* slot_cnt, const GaussCounts* counts, int level,
                           int64_t n, float4* memb_local, int32_t* memb_idx, int32_t* memb_g, int32_t* seg_off, const int32_t* pslot_of_slot,
                           int32_t* pad_off /* M+1 tile-slot offsets */, hipStream_t s);
// ---- K3: Gaussian fit -------------------------------------------------------------------------------------
// The fit's float reductions in Eigen 3.4's own order (Gaussians.h:146-147, :172-176; oracle: Gaussians::addPointSet): column means as
// linear vectorised reductions, centred products as the chains of the blocked product (depth blocks from eigen_l1_bytes), weights
// through pow(-1) of the counts as this machine's libm computes it (pow_codes).  `order` = Gaussians by descending size class, `sc` = the DEVICE copy of
// SerialCounts (class ranges); classes 0 / 1 / 2 = long / middle / short with 16 / 4 / 1 waves per Gaussian; tasks[c] Gaussians of class c
// starting at first[c] within the class; with_weights: one more workgroup computes the rebalancing weights.  Writes the six centred
// product sums (floats) per Gaussian (launch_gauss_fit_finish turns them into information matrices, max_gauss >= M threads) and, if asked,
// gauss_rows[g] = (smallest, largest) pose-table row among the members of Gaussian g, the identity row `id_row` of the static points
// left out ((INT_MAX, -1): static points only) -- what the correspondence kernels need to tell which evaluations of a Jacobian batch
// can differ from evaluation 0 for that Gaussian.
void launch_gauss_fit_all(const float4* memb_local, const int32_t* seg_off, const float* table0, const uint32_t* order, const int32_t* sc, const int first[3],
                          const int tasks[3], float* sums, GaussCounts* counts, float* info12, bool with_weights, int id_row, int2* gauss_rows, int eigen_l1_bytes,
                          const uint32_t* pow_codes, int pow_n, float* memb_q /* [3][q_stride] scratch */, size_t q_stride, hipStream_t s);
void launch_debug_limit_covariance(const float* cov9, int64_t count, float* out9, float* evals3, float* V9, int32_t* iters, int32_t* info, hipStream_t s);  // test hook (dmsa_debug_limit_covariance)
void launch_pow_minus_one(const int32_t* n, int count, const uint32_t* pow_codes, int pow_n, float* out, hipStream_t s);  // test hook (dmsa_debug_pow_minus_one)
void launch_gauss_fit_finish(const int32_t* seg_off, const GaussCounts* counts, const float* sums, int max_gauss, float* info12, hipStream_t s);
// ---- K5: normal equations + squared-error sums -------------------------------------------------------------
// Hp = [J | e0]^T [J | e0] of size (P+1)^2, col-major, J.col(k) = inv_h * (E[k+1] - E[0]) over `rows` rows
// reduce = false leaves the block sums in `partial` for a consumer that adds them itself (loop_kernels.hip): element (i, j) of Hp is the sum
// over sp < nsplit, in that order, of partial[((sp * nt + j / 32) * nt + i / 32) * 1024 + (j % 32) * 32 + i % 32]
struct NormalEqPartials {
    int nsplit, nt;
};
NormalEqPartials normal_equations_partials(int rows, int P);
// which (Gaussian, evaluation) pairs of a Jacobian batch were left out by the correspondence kernels (serial_kernels.h): pair (r, k + 1)
// with row_range[k + 1] and gauss_rows[r] disjoint has the residual of evaluation 0 (P > 64 only: the column kernel puts it in place)
struct EvalSkip {
    const int2* row_range = nullptr;      // [1 + P]
    const int2* gauss_rows = nullptr;     // [M]
    int M = 0;                            // rows below M are Gaussians; additional rows are always computed
    int check = 0;                        // 1: the pairs WERE computed (debug switch eval_skip = 2) -- compare them with evaluation 0 instead
    unsigned long long* stats = nullptr;  // [P][2] += per evaluation k + 1: pairs left out (check: that could have been), pairs that differed (check only)
};
void launch_normal_equations(const double* E, int64_t ldE, int rows, int P, double inv_h, double* partial, double* Hp, hipStream_t s, bool reduce = true,
                             const EvalSkip* skip = nullptr);
int normal_equations_partial_doubles(int rows, int P);
// the squared sums of the nine trials in the blocked row order of the normal equations (bit-identical to the oracle); out == nullptr leaves the block
// sums of evaluation b at partial[b * nsplit + sp] (nsplit as normal_equations_partials) for a consumer that adds them in order
void launch_squared_sums_blocked(const double* E, int64_t ldE, int rows, int P, int B, double* partial, double* out, hipStream_t s,
                                 const DevSync* sy = nullptr /* a dependency every workgroup waits for before it reads E (dev_sync.h) */);
int squared_sums_blocked_partial_doubles(int rows, int P, int B);

}  // namespace dmsa
