// loop_kernels.h — the O(#poses) part of optimizeSet on the device (DmsaOptimizer.h:54-182), so that an iteration needs no host round
// trip between its two evaluation batches: parameter vectors of the Jacobian batch and of the line search, the pose chains
// (ConsecutivePoses.h:26-67), the additional error rows (ContinuousTrajectory.h:603-663, MapManagement.h:210-252), the LM step for
// P <= 64 (DmsaOptimizer.h:107-128) and the arg-min of adaptiveStepSize (:152-182).  Same arithmetic as host_math.cpp: both compile
// pose_math.h / dmsa_detmath.h with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "dev_sync.h"
#include "pose_math.h"

namespace dmsa {

// what the kernels need to know about the problem model (passed by value; the pointers are device pointers)
struct LoopModel {
    int model;   // 1 window (ContinuousTrajectory), 2 keyframes (MapManagement)
    int n;       // control poses / keyframes
    int P;       // 6 (n - 1)
    int extra;   // additional error rows per evaluation
    const double* stamps;     // window: n control stamps
    const double* fhw;        // window: Floater-Hormann weights
    const double* traj_time;  // window: dense time grid
    ImuConsts imu;            // window
    KeyframeRowConsts key;    // keyframes
};

// chain state of the problem: rel_o[3n] rel_t[3n] glob_o[3n] glob_t[3n]
inline size_t loop_state_doubles(int n) { return 12 * (size_t)n; }

// per-iteration record written by the device loop (pinned read-back one sync late)
struct IterResult {
    int32_t stop;     // dmsa_stop_reason decided in this iteration (0: go on)
    int32_t best_k;
    double error0, step_norm;
};
// device-side control words
struct LoopFlags {
    int32_t stop;     // != 0: the loop has ended; every later loop kernel is a no-op
    int32_t nan;      // the LM step of the current iteration holds a NaN (DmsaOptimizer.h:116-122)
    int32_t pad[2];
};

// the chain kernels hold one chain state in LDS: false when the set has too many poses for that (optimize() then drives the loop from the host)
bool loop_chain_fits(const LoopModel& m);
// iteration start (:72-75): paramVec = getPoseParameters(); window model: relative2global; ctrl0 = global poses (n x 6: axis-angle | translation)
void launch_loop_begin(const LoopModel& m, double* state0, double* paramVec, double* ctrl0, LoopFlags* flags, hipStream_t s,
                       uint32_t* state_ready = nullptr /* dev_sync.h: signalled when the state of the iteration start is in place */);
// mode 0: the 1 + P evaluations of calcNumericJacobian (:199-232) from state_in (after loop_begin) -> ctrl[1+P][n][6], extra[1+P][a], state_out
// mode 1: the 9 trials of adaptiveStepSize (:152-182), paramVec + 0.1 k step, from state_in (after the Jacobian batch) -> ctrl[9][n][6], extra[9][a], state_out
void launch_loop_chain(const LoopModel& m, int mode, const double* state_in, double* state_out, const double* paramVec, const double* step, double increment,
                       double* ctrl, double* extra, const LoopFlags* flags, hipStream_t s,
                       int part = 0 /* mode 1 only: 1 = control poses only, 2 = additional rows and state_out only (loop_kernels.hip) */,
                       uint32_t* start_signal = nullptr /* dev_sync.h: raised when the kernel starts */);
// additional rows of a batch below the Gaussian rows of the residual batch: E[b * ldE + M + r] = extra[b * a + r]
void launch_loop_scatter_extra(const double* extra, int B, int a, double* E, int64_t ldE, int M, hipStream_t s);
// LM step (:107-128) for P <= 64 from Hp = [J | e0]^T [J | e0] ((P+1)^2, column-major): H + lambda I, Gauss-Jordan inverse with partial
// pivoting, step = (-alpha H^-1) g, NaN test, clamp to max_step
constexpr int kLoopSolveMaxP = 64;
void launch_loop_lm_step(const double* Hp, int P, double lambda, double alpha, double max_step, double* step, LoopFlags* flags, hipStream_t s);
// the same from the block sums of the normal-equation kernel (launch_normal_equations(..., reduce = false)): the reduction rides along;
// *error0_out receives e0^T e0 (element (P, P))
void launch_loop_lm_step_partials(const double* partial, int nsplit, int nt, int P, double lambda, double alpha, double max_step, double* step, LoopFlags* flags,
                                  double* error0_out, hipStream_t s);
// The same step for 64 < P <= 1024 on ceil(2P / 8) workgroups (column blocks of [A | I], panels of 8 pivot columns handed from owner to
// owner).  `work`: loop_panel_solve_doubles(P) doubles of device scratch, zeroed once when allocated; `epoch` > 0 differs from call to
// call on the same scratch (published panels and counters carry it, nothing is cleared between calls).
constexpr int kLoopPanelMaxP = 1024;
constexpr int kLoopStreamMaxP = 192;   // k_loop_lm_stream: up to three rows per lane (four would spill)
size_t loop_panel_solve_doubles(int P);
void launch_loop_lm_panels(const double* Hp, int P, double lambda, double alpha, double max_step, double* work, unsigned int epoch, double* step,
                           LoopFlags* flags, hipStream_t s);
// The same step for 64 < P <= kLoopStreamMaxP as a stream of pivot-step records (k_loop_lm_stream + k_loop_lm_stream_tail): one wave per 16
// columns of [A | I], the factorisation stays inside a workgroup for 64 steps at a time.  Same scratch, same epoch rule.
bool loop_lm_stream_fits(int P);
void launch_loop_lm_stream(const double* Hp, int P, double lambda, double alpha, double max_step, double* work, unsigned int epoch, double* step,
                           LoopFlags* flags, hipStream_t s);
// the same tail (NaN test, clamp) for a step the host solved (P > 64)
void launch_loop_step_finish(int P, double max_step, double* step, LoopFlags* flags, hipStream_t s);
// end of the iteration (:130-143): arg-min over the trial errors, setPoseParameters, stop decisions; state_jac / state_trial are the
// states after the two batches, state0 receives the state the next iteration starts from
// With chain_next != 0 and the loop going on it also does the next iteration's launch_loop_begin (paramVec, window re-chain, ctrl0), so
// begin is launched for the first iteration only; chain_next = 0 after the last iteration (decentralize() reads the last trial's poses).
// error0: e0^T e0 of the iteration; trial_errs: the nine e^T e, or (trial_nsplit > 0) their block sums [9][trial_nsplit], added here in order.
void launch_loop_finish(const LoopModel& m, const double* state_jac, const double* state_trial, double* state0, double* paramVec, const double* step,
                        const double* error0, const double* trial_errs, int trial_nsplit, int fixed_iters, double epsilon, IterResult* result, LoopFlags* flags,
                        double* ctrl0, int chain_next, hipStream_t s, uint32_t* state_ready = nullptr /* as for launch_loop_begin */);

}  // namespace dmsa
