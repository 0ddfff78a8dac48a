// Micro-benchmarks that price the design choices of the serial-order residual kernel on gfx950:
// dependent-add latency (fp32, fp32+DPP, fp64), independent issue rate of one wave and of several waves per SIMD,
// packed fp32, and an LDS-fed chain (ds_read + add) like the chainer wave's inner loop.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off valu_issue.hip -o valu_issue && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_ITER 4096

template <int CH>
__global__ void k_add_f32(float* out, float x, long long* cyc) {
    float a[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) a[c] = (float)threadIdx.x + c;
    long long t0 = clock64();
    for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(x));
    }
    long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int CH>
__global__ void k_add_f32_dpp(float* out, float x, long long* cyc) {
    float a[CH];
    float v = x + threadIdx.x;
#pragma unroll
    for (int c = 0; c < CH; ++c) a[c] = (float)threadIdx.x + c;
    long long t0 = clock64();
    for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) asm volatile("v_add_f32_dpp %0, %1, %0 row_ror:3 row_mask:0xf bank_mask:0xf" : "+v"(a[c]) : "v"(v));
    }
    long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int CH>
__global__ void k_add_f64(double* out, double x, long long* cyc) {
    double a[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) a[c] = (double)threadIdx.x + c;
    long long t0 = clock64();
    for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[c]) : "v"(x));
    }
    long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// cvt f32->f64 then dependent f64 add (the reference's `double += float term`)
__global__ void k_cvt_add_f64(double* out, float x, long long* cyc) {
    double a = (double)threadIdx.x;
    float t = x + threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < N_ITER; ++i) {
        double d;
        asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(t));
        asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(d));
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int CH>
__global__ void k_pk_add_f32(float* out, float x, long long* cyc) {
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v a[CH];
    float2v xx = {x, x};
#pragma unroll
    for (int c = 0; c < CH; ++c) a[c] = float2v{(float)threadIdx.x + c, 1.0f};
    long long t0 = clock64();
    for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(xx));
    }
    long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += a[c].x + a[c].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// chain fed from LDS: per step one ds_read of WIDTH floats + WIDTH dependent adds on one accumulator
template <int WIDTH>
__global__ void k_lds_chain(float* out, long long* cyc) {
    __shared__ float s_q[64 * 4 * 64];  // 64 KB
    for (int i = threadIdx.x; i < 64 * 4 * 64; i += blockDim.x) s_q[i] = 1.0f / (1 + (i & 1023));
    __syncthreads();
    if (threadIdx.x >= 64) return;
    float acc = 0.f;
    long long t0 = clock64();
    for (int rep = 0; rep < 16; ++rep) {
        if (WIDTH == 1) {
#pragma unroll 16
            for (int j = 0; j < 256; ++j) acc = acc + s_q[j * 64 + threadIdx.x];
        } else {
            const float4* q4 = reinterpret_cast<const float4*>(s_q);
#pragma unroll 8
            for (int j = 0; j < 64; ++j) {
                const float4 v = q4[j * 64 + threadIdx.x];
                acc = acc + v.x, acc = acc + v.y, acc = acc + v.z, acc = acc + v.w;
            }
        }
    }
    long long t1 = clock64();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;  // 16 * 256 chain steps
}
// the same with a double accumulator fed by floats (pass 2) or by doubles
template <int MODE>  // 0: float4 -> cvt + add; 1: double2 -> add
__global__ void k_lds_chain_f64(double* out, long long* cyc) {
    __shared__ double s_q[32 * 4 * 64];  // 64 KB
    for (int i = threadIdx.x; i < 32 * 4 * 64; i += blockDim.x) s_q[i] = 1.0 / (1 + (i & 1023));
    __syncthreads();
    if (threadIdx.x >= 64) return;
    double acc = 0.0;
    long long t0 = clock64();
    for (int rep = 0; rep < 16; ++rep) {
        if (MODE == 0) {
            const float4* q4 = reinterpret_cast<const float4*>(s_q);
#pragma unroll 8
            for (int j = 0; j < 64; ++j) {
                const float4 v = q4[j * 64 + threadIdx.x];
                acc += (double)v.x, acc += (double)v.y, acc += (double)v.z, acc += (double)v.w;
            }
        } else {
            const double2* q2 = reinterpret_cast<const double2*>(s_q);
#pragma unroll 8
            for (int j = 0; j < 128; ++j) {
                const double2 v = q2[j * 64 + threadIdx.x];
                acc += v.x, acc += v.y;
            }
        }
    }
    long long t1 = clock64();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;  // 16 * 256 chain steps
}

// replica of the chainer's pass-1 loop: rolling ring of D float4 reads, ONE read per four dependent adds, pinned by sched_barrier;
// MODE bit 0: s_setprio 3; bit 1: an s_barrier with nine idle waves after every 64 adds
template <int MODE>
__global__ void k_ring_chain(float* out, long long* cyc) {
    __shared__ float4 s_q[4096];  // 64 KB
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s_q[i] = float4{1.0f / (1 + (i & 1023)), 0.5f, 0.25f, 0.125f};
    __syncthreads();
    const bool chain = threadIdx.x < 64;
    if (chain && (MODE & 1)) __builtin_amdgcn_s_setprio(3);
    float acc = 0.f;
    constexpr int D = 8, S = 16;
    float4 r[D];
    const float4* q4 = s_q + (threadIdx.x & 63);
    if (chain)
        for (int k = 0; k < D; ++k) r[k] = q4[k * 64];
    long long t0 = clock64();
    for (int rep = 0; rep < 64; ++rep) {
        if (chain) {
            const float4* cs = q4 + (rep & 3) * 1024;
#pragma unroll
            for (int k = 0; k < S; ++k) {
                const float4 v = r[k % D];
                r[k % D] = cs[((k + D) % S) * 64];
                acc = acc + v.x, acc = acc + v.y, acc = acc + v.z, acc = acc + v.w;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (MODE & 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    long long t1 = clock64();
    if (chain) out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;  // 64 * 64 chain steps
}

template <class F>
static void run(const char* name, F launch, double steps, long long* d_cyc) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long cyc = 0;
    hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
    printf("%-46s clock64 %9lld ticks  %7.3f ticks/step   event %8.1f us  %7.2f ns/step\n", name, cyc, (double)cyc / steps, ms * 1e3, ms * 1e6 / steps);
}

int main() {
    float* d_out;
    double* d_outd;
    long long* d_cyc;
    hipMalloc(&d_out, 1 << 24), hipMalloc(&d_outd, 1 << 24), hipMalloc(&d_cyc, 64);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("device %s  clockRate %d kHz  CUs %d\n", p.name, p.clockRate, p.multiProcessorCount);
    const double n = N_ITER;
    puts("-- one wave on the chip: dependent latency / single-wave issue --");
    run("f32 add, 1 chain (dependent)", [&] { hipLaunchKernelGGL(k_add_f32<1>, 1, 64, 0, 0, d_out, 1.0f, d_cyc); }, n, d_cyc);
    run("f32 add, 2 chains", [&] { hipLaunchKernelGGL(k_add_f32<2>, 1, 64, 0, 0, d_out, 1.0f, d_cyc); }, n * 2, d_cyc);
    run("f32 add, 3 chains", [&] { hipLaunchKernelGGL(k_add_f32<3>, 1, 64, 0, 0, d_out, 1.0f, d_cyc); }, n * 3, d_cyc);
    run("f32 add, 8 chains", [&] { hipLaunchKernelGGL(k_add_f32<8>, 1, 64, 0, 0, d_out, 1.0f, d_cyc); }, n * 8, d_cyc);
    run("f32 add dpp row_ror, 1 chain", [&] { hipLaunchKernelGGL(k_add_f32_dpp<1>, 1, 64, 0, 0, d_out, 1.0f, d_cyc); }, n, d_cyc);
    run("f32 add dpp row_ror, 3 chains", [&] { hipLaunchKernelGGL(k_add_f32_dpp<3>, 1, 64, 0, 0, d_out, 1.0f, d_cyc); }, n * 3, d_cyc);
    run("f64 add, 1 chain", [&] { hipLaunchKernelGGL(k_add_f64<1>, 1, 64, 0, 0, d_outd, 1.0, d_cyc); }, n, d_cyc);
    run("f64 add, 4 chains", [&] { hipLaunchKernelGGL(k_add_f64<4>, 1, 64, 0, 0, d_outd, 1.0, d_cyc); }, n * 4, d_cyc);
    run("cvt f32->f64 + f64 add, 1 chain (per member)", [&] { hipLaunchKernelGGL(k_cvt_add_f64, 1, 64, 0, 0, d_outd, 1.0f, d_cyc); }, n, d_cyc);
    run("pk f32 add, 1 chain", [&] { hipLaunchKernelGGL(k_pk_add_f32<1>, 1, 64, 0, 0, d_out, 1.0f, d_cyc); }, n, d_cyc);
    run("pk f32 add, 8 chains", [&] { hipLaunchKernelGGL(k_pk_add_f32<8>, 1, 64, 0, 0, d_out, 1.0f, d_cyc); }, n * 8, d_cyc);
    puts("-- LDS-fed chains, one wave (per chain step = per member) --");
    run("lds b32 + f32 add chain", [&] { hipLaunchKernelGGL(k_lds_chain<1>, 1, 256, 0, 0, d_out, d_cyc); }, 16.0 * 256, d_cyc);
    run("lds b128 + 4 f32 adds chain", [&] { hipLaunchKernelGGL(k_lds_chain<4>, 1, 256, 0, 0, d_out, d_cyc); }, 16.0 * 256, d_cyc);
    run("lds b128(4 f32) + cvt + f64 add chain", [&] { hipLaunchKernelGGL(k_lds_chain_f64<0>, 1, 256, 0, 0, d_outd, d_cyc); }, 16.0 * 256, d_cyc);
    run("lds b128(2 f64) + f64 add chain", [&] { hipLaunchKernelGGL(k_lds_chain_f64<1>, 1, 256, 0, 0, d_outd, d_cyc); }, 16.0 * 256, d_cyc);
    puts("-- replica of the chainer loop (rolling ring, 1 read per 4 adds): plain / setprio 3 / + barrier per 64 adds with 9 idle waves --");
    run("ring chain, 1 wave", [&] { hipLaunchKernelGGL(k_ring_chain<0>, 1, 64, 0, 0, d_out, d_cyc); }, 4096.0, d_cyc);
    run("ring chain, 1 wave, setprio 3", [&] { hipLaunchKernelGGL(k_ring_chain<1>, 1, 64, 0, 0, d_out, d_cyc); }, 4096.0, d_cyc);
    run("ring chain, 10 waves, barrier per 64 adds", [&] { hipLaunchKernelGGL(k_ring_chain<2>, 1, 640, 0, 0, d_out, d_cyc); }, 4096.0, d_cyc);
    run("ring chain, 10 waves, barrier, setprio 3", [&] { hipLaunchKernelGGL(k_ring_chain<3>, 1, 640, 0, 0, d_out, d_cyc); }, 4096.0, d_cyc);
    run("ring chain, 10 waves, barrier, 220 workgroups", [&] { hipLaunchKernelGGL(k_ring_chain<3>, 220, 640, 0, 0, d_out, d_cyc); }, 4096.0, d_cyc);
    puts("-- LDS-fed f32 chain (b128 + 4 adds), one chain wave per workgroup, many workgroups: does the per-step time hold chip-wide? --");
    for (int wgs : {1, 64, 256, 512}) {
        char nm[96];
        snprintf(nm, sizeof nm, "lds b128 + 4 f32 adds chain, %3d workgroups x 256 threads", wgs);
        run(nm, [&] { hipLaunchKernelGGL(k_lds_chain<4>, wgs, 256, 0, 0, d_out, d_cyc); }, 16.0 * 256, d_cyc);
        snprintf(nm, sizeof nm, "lds b128(2 f64) + f64 add chain, %3d workgroups", wgs);
        run(nm, [&] { hipLaunchKernelGGL(k_lds_chain_f64<1>, wgs, 256, 0, 0, d_outd, d_cyc); }, 16.0 * 256, d_cyc);
    }
    puts("-- one workgroup (one CU): waves per SIMD sharing the VALU (ticks per instruction of ONE wave) --");
    for (int waves : {4, 8, 16}) {
        char nm[96];
        snprintf(nm, sizeof nm, "f32 add 8 chains, %2d waves in the CU", waves);
        run(nm, [&] { hipLaunchKernelGGL(k_add_f32<8>, 1, 64 * waves, 0, 0, d_out, 1.0f, d_cyc); }, n * 8, d_cyc);
        snprintf(nm, sizeof nm, "pk f32 add 8 chains, %2d waves in the CU", waves);
        run(nm, [&] { hipLaunchKernelGGL(k_pk_add_f32<8>, 1, 64 * waves, 0, 0, d_out, 1.0f, d_cyc); }, n * 8, d_cyc);
        snprintf(nm, sizeof nm, "f64 add 4 chains, %2d waves in the CU", waves);
        run(nm, [&] { hipLaunchKernelGGL(k_add_f64<4>, 1, 64 * waves, 0, 0, d_outd, 1.0, d_cyc); }, n * 4, d_cyc);
    }
    puts("-- whole chip: 2048 workgroups x 256 threads, 8 chains (event time -> lane-ops/s) --");
    run("f32 add, chip", [&] { hipLaunchKernelGGL(k_add_f32<8>, 2048, 256, 0, 0, d_out, 1.0f, d_cyc); }, n * 8, d_cyc);
    run("pk f32 add, chip", [&] { hipLaunchKernelGGL(k_pk_add_f32<8>, 2048, 256, 0, 0, d_out, 1.0f, d_cyc); }, n * 8, d_cyc);
    run("f64 add, chip", [&] { hipLaunchKernelGGL(k_add_f64<4>, 2048, 256, 0, 0, d_outd, 1.0, d_cyc); }, n * 4, d_cyc);
    return 0;
}
