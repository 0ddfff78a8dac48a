// Does hipExtAnyOrderLaunch overlap kernels of ONE stream on gfx950?  (hip_ext.h says the flag is "not supported on GFX9xx" for the
// module-launch variant.)  Three spin kernels of ~100 us on few workgroups each: launched normally they take 3 x 100 us, with the second
// and third launched any-order they should take ~100 us, and a kernel launched normally behind them must still wait for all three.
//   hipcc --offload-arch=gfx950 -O2 -o anyorder anyorder.hip && ./anyorder
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void spin(long long cycles, int* out, int tag) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(out, tag);
}
__global__ void reader(const int* in, int* seen) { *seen = *in; }

int main() {
    int *d_out, *d_seen;
    hipMalloc(&d_out, 4), hipMalloc(&d_seen, 4);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const long long cyc = 10000;  // wall_clock64 ticks at 100 MHz: 100 us
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemsetAsync(d_out, 0, 4, s);
            hipStreamSynchronize(s);
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s, cyc, d_out, 1);
            if (mode == 0) {
                hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s, cyc, d_out, 10);
                hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s, cyc, d_out, 100);
            } else {
                hipExtLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, d_out, 10);
                hipExtLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, d_out, 100);
            }
            hipLaunchKernelGGL(reader, dim3(1), dim3(1), 0, s, d_out, d_seen);  // ordinary launch: must see all three tags
            hipStreamSynchronize(s);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            int seen = 0;
            hipMemcpy(&seen, d_seen, 4, hipMemcpyDeviceToHost);
            std::printf("%s rep %d: %.0f us, reader saw %d (111 = ordered behind all three)\n", mode ? "any-order" : "in-order ", rep, us, seen);
        }
    }
    return 0;
}
