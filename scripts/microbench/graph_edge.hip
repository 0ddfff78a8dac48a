// What does a cross-stream dependency cost?  A chain A -> B -> A -> B ... of tiny kernels alternating between two streams, (1) with
// hipEventRecord / hipStreamWaitEvent, (2) the same chain captured into a hipGraph, (3) all kernels on one stream (no cross-stream edge).
//   hipcc --offload-arch=gfx950 -O2 -o graph_edge graph_edge.hip && ./graph_edge
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_tiny(int* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1; }
int main() {
    int* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ev[2]; CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    const int hops = 200;
    auto chain = [&](hipStream_t a, hipStream_t b) {
        for (int i = 0; i < hops; ++i) {
            hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, a, d);
            if (a != b) { (void)hipEventRecord(ev[0], a); (void)hipStreamWaitEvent(b, ev[0], 0); }
            hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, b, d);
            if (a != b) { (void)hipEventRecord(ev[1], b); (void)hipStreamWaitEvent(a, ev[1], 0); }
        }
    };
    auto time_it = [&](const char* what, auto&& fn) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            fn();
            CK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep == 2) printf("%-44s %8.1f us per pair of kernels\n", what, us / hops);
        }
        return 0;
    };
    time_it("one stream", [&] { chain(s1, s1); });
    time_it("two streams, event record / wait per hop", [&] { chain(s1, s2); });
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
    chain(s1, s2);
    CK(hipStreamEndCapture(s1, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    time_it("the same chain as a captured graph", [&] { (void)hipGraphLaunch(ge, s1); });
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
    chain(s1, s1);
    CK(hipStreamEndCapture(s1, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    time_it("one-stream chain as a captured graph", [&] { (void)hipGraphLaunch(ge, s1); });
    return 0;
}
