/*
 * dmsa_debug.h — debug / experiment switches of a dmsa_ctx.
 *
 * Production code never needs this header: dmsa_create() (dmsa_hip.h) uses the defaults listed here.  The switches select between
 * implementations that produce THE SAME RESULTS (tests run both sides of each); they exist for A/B timing, for bisecting, and for the
 * parity tests that exercise a fallback on purpose.  They are fixed when the context is created:
 *
 *   dmsa_create_ex(device, flags, &options, &ctx)     from code, or
 *   DMSA_DEBUG="name=value,name=value" in the environment: read ONCE by dmsa_create / dmsa_create_ex, overrides fields by name
 *                                                      (profiling scripts).  It is the only environment variable the library reads.
 */
#ifndef DMSA_DEBUG_H
#define DMSA_DEBUG_H

#include "dmsa_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dmsa_debug_options {
    int32_t device_loop;     /* 1   optimizeSet's control state lives on the device (csrc/loop_kernels.hip); 0: the host drives the loop
                                     (three waits per iteration, as in rounds 1-2)                                                        */
    int32_t dual_stream;     /* 1   the two voxel resolutions and the count read-back on their own streams; 0: everything on one stream   */
    int32_t serial_streams;  /* 3   streams the three tiers of the reference-order correspondence kernels run on (1, 2 or 3)              */
    int32_t merge_sort;      /* 0   one radix sort per voxel level, on two streams; 1: both levels in ONE sort of 2 n pairs (level 1 tagged above the
                                     widest code); -1: merged up to 2^20 points (the rule of rounds 2-4, from before the stream dependencies moved
                                     to device counters: round 5 measured separate sorts 1-2.5 % faster at every size, profiles/r05_ab_merge_sort.txt) */
    int32_t key_compress;    /* 1   drop the key bits that are equal for all points before sorting; 0: full 3 x depth bit leaf codes
                                     (64-bit codes, sorted as two 32-bit halves, from depth 11 on -- also the fallback of a mis-predicted range) */
    int32_t fused_segments;  /* 1   head flags + scan + leaf starts in one single-pass kernel; 0: three kernels (csrc/radix_sort.hip's scan)  */
    int32_t sort_prehist;    /* 0   the key kernels count the sort digits (measured 1.5 % slower than the sort's own histogram pass)     */
    int32_t overlap_batch;   /* 1   host-driven loop only: host math of the Jacobian batch while the GPU voxelises                        */
    int32_t serial_tree;     /* 1   double sum of the chain tiers: 1 parallel reduction when the exactness test allows it,
                                     0 always the member-order chain, 2 both and keep the chain's result (test hook).  Where the test fails
                                     for a Gaussian of the latency tier, only the BLOCKS of its member list whose own bounds fail are chained
                                     (csrc/serial_kernels.hip); 3 (test hook): every block through that chain                               */
    int32_t host_threads;    /* 16  worker threads of the context (upload packing, host-built pose tables, host solve)                   */
    int32_t solve_threads;   /* 12  of which the blocked host LM solve uses at most this many                                            */
    int32_t host_timeline;   /* 0   print host-side time stamps of the last iteration of every optimize call to stderr                   */
    int32_t trace_time;      /* 0   print upload / optimize wall time of dmsa_optimize_window to stderr                                  */
    int32_t fused_leaf_scan; /* 1   slot scans of the accepted leaves as a multi-workgroup single-pass kernel; 0: the                
                                     single-workgroup k_leaf_scan                                                                         */
    int32_t device_sync;     /* 1   fork / join of the three tier streams through counters in device memory (one-wave signal / wait
                                     kernels); 0: hipEventRecord / hipStreamWaitEvent                                                   */
    int32_t shared_rotations; /* 1  forward differences of translation parameters share the rotated member coordinates of evaluation 0 in
                                     the second pass of the chain tiers (serial_kernels.hip); 0: every evaluation transforms on its own     */
    int32_t eval_skip;       /* 1   a (Gaussian, evaluation) pair of the Jacobian batch whose pose-table rows all have evaluation 0's bits is not
                                     computed: its residual IS evaluation 0's (csrc/serial_kernels.hip); 0: every pair is computed;
                                     2: every pair is computed and the pairs 1 would have skipped are compared (dmsa_eval_skip_stats)       */
    int32_t sync_fault;      /* 0   test hook: k > 0 withholds the signal of the k-th device-side wait of every whole call -- the wait
                                     gives up, the library restores the state of the call's start and runs it again with events            */
    int32_t speculation_fault; /* 0 test hook: k > 0 plants a tree depth one too small in the k-th voxelisation of every whole call, so the
                                     speculative sort width is wrong and the voxelisation runs again                                       */
    int32_t voxel_coherence; /* 0   1: count, per voxelisation and level, the points whose leaf code differs from the previous voxelisation's of
                                     the same context (dmsa_debug_counters: how much of last iteration's sorted order would survive)       */
    int32_t lm_stream;       /* 1   LM solve for 64 < P <= 192 as a stream of pivot-step records (one wave per 8 columns, csrc/loop_kernels.hip:
                                     k_loop_lm_stream); 0: column-block workgroups handing panels over (k_loop_lm_panels).  Same bits.        */
    int32_t stream_priority; /* 0   bit 0 / 1 / 2: the main / second / third stream of the context is created at the device's highest priority
                                     (which of the concurrent kernels of an iteration the wave dispatcher serves first)                      */
    int32_t gap_stamps;      /* 0   1: one-thread kernels write the device's wall clock in front of, between and behind the kernels of the normal
                                     equations and the LM solve of every iteration; the gaps of the LAST iteration of a call are printed to stderr
                                     -- what the kernel trace of a profiler cannot tell: whether the holes it shows between those kernels
                                     exist when no profiler slows the host's launches.  2: also eight stamps per iteration (begin, voxelisation,
                                     fit, Jacobian batch, normal equations, LM step, trial chains + tables, end), printed as a table             */
    int32_t lattice_hint;    /* 1   k_lattice first checks, in parallel, whether the bounding-box growth events of the previous voxelisation of this
                                     context still hold for the moved points (same result as the replay, proved per launch); 0: always the
                                     sequential replay of PCL's adoptBoundingBoxToPoint                                                    */
    int32_t fit_classes;     /* 7   PROFILING ONLY (results are wrong unless 7): bit 0 / 1 / 2 = the Gaussian fit runs its long / middle / short
                                     size class -- how much of k_gauss_fit_all's time belongs to which class                              */
    int32_t eigen_l1_bytes;  /* 32768  NOT an A/B switch: the L1 data cache size of the machine the REFERENCE runs on.  Eigen sizes the depth
                                     blocks kc of centered^T * centered (Gaussians.h:147) from it at run time (32768 -> 680, 49152 -> 1016), and
                                     the float sum of a Gaussian with more members than kc depends on kc.  Set it to that machine's L1d to
                                     reproduce its bits; Gaussians up to kc members do not depend on it.                                  */
    int32_t small_threshold; /* 0   members up to which a Gaussian goes to the lane-per-evaluation correspondence kernel (and the fit's one-wave class);
                                     0 = the built-in rule (previous voxelisation: < 2000 Gaussians nobody, < 6000 up to 32 members, else 256), else 1 .. 256.
                                     Every tier computes the same bits.                                                                  */
    int32_t skip_stats;      /* 0   1: k_jacobian_columns counts, per evaluation, the (Gaussian, evaluation) pairs eval_skip left out and (eval_skip = 2)
                                     the ones that differed after all -> counters skip_pairs_equal / skip_mismatches; k_split_pairs counts the pair
                                     blocks it looked at and skipped -> split_blocks / split_blocks_skipped.  Off by default: atomics on the
                                     path of every iteration (two per wave and evaluation in k_jacobian_columns)                          */
    /* ---- appended in round 6; from here on the struct is APPEND-ONLY (no field is removed or moved) and dmsa_create_ex2 takes its size ---- */
    int32_t small_voxel;     /* 1   point sets of at most 32 768 points (the reference's everyday windows: 5 scans x <= 3000 points + static points) are
                                     voxelised by ONE launch -- a workgroup per resolution keeps its (code, point) pairs in registers from the leaf codes to
                                     the member lists (csrc/small_voxel.hip) -- instead of ten dependent kernels per level; 0: always the general path.
                                     Same bits.                                                                                              */
    int32_t long_split;      /* 1   latency tier (Gaussians of >= 4096 members) of a window with at most 16 of them, B <= 32: the workgroup of a (Gaussian,
                                     sub-batch) ends with its float chain, leaves the means in device memory, and 8 HELPER workgroups per item -- blocks at the
                                     end of the same launch -- sum a slice of the members each (the parallel second pass; the last one to arrive tests the
                                     exactness bounds and runs the member-by-member chain itself if they fail).  >= 2: that many members as the threshold, in
                                     every batch of every model (experiments); 0: every workgroup does its own second pass.  Same bits.             */
    int32_t long_log2;       /* 0   Gaussians of at least 2^long_log2 members go to the latency tier of the correspondence kernels (10 waves per workgroup,
                                     helpers); 0 = the built-in rule (12), else 9 .. 20.  Every tier computes the same bits.                          */
    int32_t sort_items;      /* 0   EXPERIMENTS ONLY, process-wide: pairs per thread of a tile of the onesweep sort (512 threads): 2, 4, 8 or 16; 0 = by
                                     size (2 up to 2^16 pairs, 4 up to 2^18, else 16).  Same bits.                                           */
    int32_t trial_rows_aside; /* 1  window with IMU rows, device loop: the nine trial chains of the line search write their control poses on the main stream
                                     and compute their additional rows on the side stream, beside the trial batch (0: one kernel, in front of it).  Same bits. */
} dmsa_debug_options;

/* what the switches above leave behind, since the context was created */
typedef struct dmsa_debug_counters {
    int64_t sync_retries;         /* whole calls run again with event dependencies after a device-side wait gave up           */
    int64_t speculation_retries;  /* voxelisations run again because the speculated sort width was too small                  */
    int64_t skip_pairs;           /* (Gaussian, evaluation) pairs of the Jacobian batches the eval_skip logic looked at       */
    int64_t skip_pairs_equal;     /* ... of which had the pose-table rows of evaluation 0 (eval_skip = 1: were not computed) */
    int64_t skip_mismatches;      /* eval_skip = 2 only: such pairs whose computed residual differed from evaluation 0's      */
    int64_t split_blocks;         /* splitSet search (gauss_split): blocks of 64 positions x 64 partners looked at ...       */
    int64_t split_blocks_skipped; /* ... and skipped because their normals cannot be within 0.5 of anti-parallel       */
    int64_t voxel_codes_compared; /* voxel_coherence = 1: (point, level) pairs compared with the previous voxelisation            */
    int64_t voxel_codes_changed;  /* ... whose leaf code changed                                                               */
    int64_t lattice_hints_held;   /* lattice_hint: (voxelisation, level) pairs whose previous growth events were verified instead of replayed ... */
    int64_t lattice_replays;      /* ... and pairs that went through the sequential replay                                       */
    int64_t voxel_lattice_changes;/* ... voxelisations (per level) whose lattice (origin, depth, code bits) differed from the previous one's:
                                     every code of that level counts as changed                                                 */
    /* appended in round 6 (append-only like the options) */
    int64_t small_voxel_launches; /* small_voxel: voxelisations that went through the one-launch path (csrc/small_voxel.hip) ...        */
    int64_t small_voxel_fallbacks;/* ... and how many of them it handed back to the general path (leaf codes wider than 32 bits) */
} dmsa_debug_counters;
int dmsa_get_debug_counters(dmsa_ctx* ctx, dmsa_debug_counters* out);
/* Test hook: pow(-1) of `count` member counts as the Gaussian fit computes it on the device (Gaussians.h:172: libm's powf(n, -1.0f) through
 * the difference table of the host's libm; counts[] and out[] are host arrays; counts above 2^24 are refused). */
int dmsa_debug_pow_minus_one(dmsa_ctx* ctx, const int32_t* counts, int32_t count, float* out);
/* Test hook: Gaussians::limitCovariance (Gaussians.h:181-201) as the fit computes it on the device, on `count` column-major 3 x 3 matrices (host arrays):
 * out9 = V max(D, 1e-4) V^-1; optionally the EigenSolver<Matrix3f> behind it -- eigenvalues().real() in the order of the Schur form's diagonal,
 * eigenvectors().real() column-major, Francis QR steps, info (0 Success, 1 NumericalIssue, 2 NoConvergence).  NULL = not wanted. */
int dmsa_debug_limit_covariance(dmsa_ctx* ctx, const float* cov9, int64_t count, float* out9, float* evals3, float* V9, int32_t* iterations, int32_t* info);
/* Test hooks of csrc/radix_sort.hip on caller data (host arrays): the stable sort of (u64 key, u32 value) pairs on the key bits
 * [0, end_bit) that wide leaf codes go through, and the inclusive (1) / exclusive (0) prefix scan of an int32 array. */
int dmsa_sort_pairs64(dmsa_ctx* ctx, const uint64_t* keys, const uint32_t* values, int64_t n, uint32_t end_bit, uint64_t* keys_sorted, uint32_t* values_sorted);
int dmsa_scan_i32(dmsa_ctx* ctx, const int32_t* in, int64_t n, int32_t inclusive, int32_t* out);

void dmsa_default_debug_options(dmsa_debug_options* o);
/* dmsa_create with explicit switches (`options` may be NULL = defaults); DMSA_DEBUG still overrides by name.  `options` must be THIS header's
 * struct; a caller that may have been built against an older header passes the size of the struct it knows to dmsa_create_ex2. */
int dmsa_create_ex(int device, uint32_t flags, const dmsa_debug_options* options, dmsa_ctx** out);
/* The same with the size of the caller's struct in bytes (a multiple of 4, at least 4): only that many leading bytes are read, the fields
 * behind them keep their defaults; a size larger than the library's struct is refused (DMSA_ERR_INVALID). */
int dmsa_create_ex2(int device, uint32_t flags, const dmsa_debug_options* options, uint32_t options_bytes, dmsa_ctx** out);

#ifdef __cplusplus
}
#endif
#endif /* DMSA_DEBUG_H */
