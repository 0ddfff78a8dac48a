// dev_sync.h — stream dependencies carried by counters in device memory instead of HIP events.
//
// A hipEventRecord / hipStreamWaitEvent pair costs 8-10 us on each of the two streams (barrier packets serialise the queue around
// them); an iteration of optimizeSet crosses streams eleven times.  Here the producer adds one to a counter when its results are
// visible and the consumer's stream holds a one-wave kernel (or the first thread of an existing small kernel) that spins until the
// counter has reached the value the host expects.  Counters only grow; comparisons are wrap-safe.
//
// Rules that make this deadlock-free: (1) every signal is ENQUEUED before the wait that needs it -- hardware queues are FIFO, so a
// wait can only ever sit in front of packets that were enqueued after its signal, also when streams share a hardware queue;
// (2) waits are single waves (a spinning grid could occupy the chip the producer needs); (3) every wait is bounded: it gives up after
// about ten seconds, flags the context, and the host returns DMSA_ERR_HIP -- never a hung GPU.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace dmsa {

// what a kernel does for the dependencies around it (null pointers: nothing)
struct DevSync {
    const uint32_t* wait_counter = nullptr;  // spin until *wait_counter has reached wait_target, before the kernel's own work
    uint32_t wait_target = 0;
    uint32_t* signal_counter = nullptr;      // add one once the kernel's results are visible device-wide
    int32_t* timed_out = nullptr;            // [0] = 1, [1] = target, [2] = value seen when a wait gives up
};

// slots of dmsa_ctx::d_sync (uint32 words)
enum SyncSlot : int {
    SYNC_TIER_FORK = 0,   // latency tier placed -> the other tier streams start
    SYNC_TIER_JOIN = 1,   // tier streams done -> main stream goes on
    SYNC_TIMED_OUT = 2,   // three words, see DevSync::timed_out
    SYNC_LOOP_STATE = 5,  // k_loop_begin / k_loop_finish -> Jacobian chains on the side stream
    SYNC_TABLES = 6,      // pose tables of the Jacobian batch (side stream) -> main stream
    SYNC_CLASSES = 7,     // k_size_classes -> read-back of the counts
    SYNC_LATTICE = 8,     // k_lattice -> level 1 of the voxelisation on its own stream
    SYNC_LEVEL1 = 9,      // level 1 leaves done -> its member gather on the main stream
    SYNC_SMALL_L0 = 10,   // small_voxel.hip: level 0's totals -> the workgroup of level 1 (inside one kernel)
    SYNC_TRIAL_STEP = 11, // the trial chains' control-pose kernel runs (= the LM step is done) -> their additional rows on the side stream
    SYNC_TRIAL_ROWS = 12, // those rows are in E -> the squared sums on the main stream
    SYNC_SLOTS = 16
};

#if defined(__HIPCC__)
__device__ __forceinline__ void dev_sync_signal(uint32_t* counter) {
    __threadfence();  // release at agent scope: what this kernel wrote so far is visible to every later reader
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
constexpr int kSyncMaxSpins = 1 << 23;  // ~10 s: three orders of magnitude above the longest legitimate wait (one iteration)
__device__ __forceinline__ void dev_sync_wait(const uint32_t* counter, uint32_t target, int32_t* timed_out, int max_spins = kSyncMaxSpins) {
    uint32_t v = 0;
    for (int spin = 0; spin < max_spins; ++spin) {
        v = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int32_t)(v - target) >= 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            return;
        }
        if (spin < 4096)  // poll tightly at first (a join is usually microseconds away), then back off
            __builtin_amdgcn_s_sleep(1);
        else
            __builtin_amdgcn_s_sleep(8);
    }
    if (timed_out != nullptr) timed_out[0] = 1, timed_out[1] = (int32_t)target, timed_out[2] = (int32_t)v;
}
// the two halves as they sit in a kernel: thread 0 waits before the first barrier, thread 0 signals behind a last barrier
__device__ __forceinline__ void dev_sync_enter(const DevSync& sy) {
    if (sy.wait_counter != nullptr && threadIdx.x == 0) dev_sync_wait(sy.wait_counter, sy.wait_target, sy.timed_out);
}
__device__ __forceinline__ void dev_sync_leave(const DevSync& sy) {
    if (sy.signal_counter != nullptr) {
        __syncthreads();
        if (threadIdx.x == 0) dev_sync_signal(sy.signal_counter);
    }
}
#endif

// one-wave kernels for dependencies no existing kernel can carry (loop_kernels.hip)
void launch_sync_signal(uint32_t* counter, hipStream_t s);
void launch_stamp(long long* slot, hipStream_t s);  // debug switch gap_stamps: wall_clock64() of the device at this point of the stream
void launch_sync_wait(const uint32_t* counter, uint32_t target, int32_t* timed_out, hipStream_t s, int max_spins = 1 << 23);

}  // namespace dmsa
