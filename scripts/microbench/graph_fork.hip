// Fork / join of two independent 100-us kernels: streams + events vs the same pattern captured into a hipGraph.  Do the branches of a
// graph run side by side, and what do the fork and the join cost?
//   hipcc --offload-arch=gfx950 -O2 -o graph_fork graph_fork.hip && ./graph_fork
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_spin(long long cycles, int* p) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (threadIdx.x == 0) p[blockIdx.x] += 1;
}
int main() {
    int* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ef, ej; CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    const long long spin = 200000;  // ~100 us at 2 GHz
    const int reps = 50;
    auto pattern = [&](bool fork) {
        for (int i = 0; i < reps; ++i) {
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, 2000, d);  // root
            if (fork) { (void)hipEventRecord(ef, s1); (void)hipStreamWaitEvent(s2, ef, 0); }
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, spin, d);
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, fork ? s2 : s1, spin, d);
            if (fork) { (void)hipEventRecord(ej, s2); (void)hipStreamWaitEvent(s1, ej, 0); }
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, 2000, d);  // join
        }
    };
    auto time_it = [&](const char* what, auto&& fn) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            fn();
            CK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep == 2) printf("%-52s %8.1f us per root + 2 x 100 us + join\n", what, us / reps);
        }
        return 0;
    };
    time_it("one stream (serial)", [&] { pattern(false); });
    time_it("two streams, fork and join by events", [&] { pattern(true); });
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
    pattern(true);
    CK(hipStreamEndCapture(s1, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    time_it("the fork / join pattern as a captured graph", [&] { (void)hipGraphLaunch(ge, s1); });
    return 0;
}
