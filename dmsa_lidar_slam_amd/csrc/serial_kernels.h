// serial_kernels.h — correspondence kernels of the DEFAULT (reference-order) path.
//
// DmsaOptimizer::updateErrorTerms (DmsaOptimizer.h:234-273) sums every Gaussian with two explicit serial loops: a float mean in
// member order (:247-254) and float Mahalanobis terms added to a double in member order (:259-264).  Rounding makes both order
// dependent, and the numeric Jacobian amplifies 1e-7 differences into millimetres, so the library's default path
// keeps the reference's order bit for bit.  What is parallel: the B evaluations of a batch (lane = evaluation: every lane runs the
// same chain on its own pose table), the Gaussians, and everything AROUND a chain (transform + quadratic form of each member).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "dev_sync.h"
#include "dmsa_kernels.h"

namespace dmsa {

// Gaussians ordered for the serial-order kernels: [0, n_chain) long ones by descending size class (pipelined kernel),
// [n_chain, n_chain + n_small) the short ones by descending size (lane-per-evaluation kernel).
struct SerialCounts {
    int32_t n_chain, n_small, max_members;
    int32_t n_long;  // the first n_long entries of the order: Gaussians of the latency tier (>= 2^12 members)
};
int serial_small_threshold();  // members; Gaussians up to this size go to the lane-per-evaluation kernel
// one workgroup: counting sort of the M = counts->level[0..1].num_gauss Gaussians by size class
void launch_size_classes(const int32_t* seg_off, const GaussCounts* counts, uint32_t* order /* M */, SerialCounts* out, hipStream_t s,
                         const DevSync& sy = DevSync() /* waits / signals of the stream dependencies around the kernel (dev_sync.h) */,
                         int small_threshold = 0 /* 0: serial_small_threshold() */,
                         int long_log2 = 0 /* Gaussians of >= 2^long_log2 members go to the latency tier; 0: the built-in 12 */);
// tables [B][rows][12] -> tablesT [rows][B][12]: the B evaluations of one pose row are contiguous (lane = evaluation reads coalesce)
void launch_transpose_tables(const float* tables, int rows, int B, float* tablesT, hipStream_t s);
// updateErrorTerms for B pose tables, bit-identical to the reference's serial loops.  E[b * ldE + g] = sqrt(|sum|).
// The latency tier runs on s_long, the throughput tier on s_rest, the short tier on s_small (pass a stream twice to serialise tiers); the caller joins
// the streams.
// scratch of the latency tier's helper workgroups (k_residuals_chain: the second pass of the longest Gaussians shared out), owned by the caller;
// means == null: every workgroup does its own second pass
struct LongSplit {
    float* means = nullptr;      // [items][3][16]
    uint32_t* ready = nullptr;   // [items], any value but `epoch`
    double* partial = nullptr;   // [items][helpers][16]
    int* partial_key = nullptr;  // [items][helpers][16]
    uint32_t* done = nullptr;    // [items], zero between launches
    int32_t* timed_out = nullptr;  // dev_sync.h: a wait that gave up
    uint32_t epoch = 1;          // a value no earlier launch on this scratch used
    int helpers = 8;
    int min_members = 8192;      // Gaussians from this size on hand their second pass to helpers
    int lead_gaussians = 16;     // helpers are launched for this many Gaussians from the front of the (descending) order
};
void launch_residuals_serial(const float4* memb_local, const int32_t* seg_off, const float* info12, const float* tablesT, int B, const uint32_t* order,
                             const SerialCounts& sc, double* E, int64_t ldE, hipStream_t s_long, hipStream_t s_rest, hipStream_t s_small,
                             int tree_mode = 1 /* dmsa_debug_options::serial_tree */,
                             uint32_t* start_signal = nullptr /* counter the latency tier adds one to once all its workgroups are placed (loop_kernels.h: launch_sync_wait) */,
                             int tiers = 7 /* bit 0: latency tier, bit 1: throughput tier, bit 2: short tier */,
                             const uint32_t* rot_same = nullptr /* [B] from the pose-table kernels: evaluations whose rotations are evaluation 0's */,
                             const int2* row_range = nullptr /* [B] launch_eval_row_ranges; with gauss_rows: (Gaussian, evaluation) pairs whose rows all equal
                                                                evaluation 0's are NOT computed and their E entries are left untouched */,
                             const int2* gauss_rows = nullptr /* [M] launch_gauss_fit_all */,
                             const LongSplit* split = nullptr /* the latency tier's second pass on many compute units */);
// row_range[b] = (first, last) pose-table row of evaluation b whose INPUTS differ from evaluation 0's bit for bit -- (INT_MAX, -1) if
// none, (0, INT_MAX) for b = 0 -- from the global poses of the batch, ctrl[B][np][6].  model 2 (keyframes): row k is a function of pose k
// alone (MapManagement.h:120-149).  model 1 (window): the rotation of dense pose j is a function of the two control rotations around it,
// its translation of ALL control translations (Floater-Hormann), ContinuousTrajectory.h:189-226.  Same inputs, same instructions, same
// bits: a conservative test (a row whose inputs differ may still round to the same floats).
void launch_eval_row_ranges(int model, const double* ctrl, int B, int np, const double* stamps, const double* traj_time, int n_t, int2* row_range, hipStream_t s);
// LDS / shape parameters chosen for a batch of B evaluations (exposed for the bench's roofline notes and the tests)
struct SerialShape {
    int nsub_long, Bs_long;  // latency tier: evaluation sub-batches per Gaussian, evaluations per sub-batch
    int nsub, Bs;            // throughput tier
    int lanes, nsub_small;   // lane-per-evaluation kernel: lanes per (Gaussian, sub-batch), sub-batches
};
SerialShape serial_shape(int B);
// Introspection: how many (Gaussian, evaluation sub-batch) workgroups of the chain tiers failed the exactness test of the parallel second
// pass and were summed again member by member since the last reset (synchronises the device).
unsigned long long serial_fallback_sums(bool reset);

}  // namespace dmsa
