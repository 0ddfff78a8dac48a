// What exactly does v_mfma_f64_16x16x4_f64 compute per output element?  Candidates for d = c + sum_k a_k b_k:
//   (A) fused chain, k ascending:   d = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, c))))
//   (B) unfused chain:               d = (((c + a0*b0) + a1*b1) + a2*b2) + a3*b3
//   (C) fused chain, k descending;   (D) products summed first (fused), then + c
// Also confirms the register layout.   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off mfma_f64_semantics.hip -o mfma_f64_semantics
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A /*16x4 row-major*/, const double* B /*4x16 row-major*/, const double* C /*16x16*/, double* D) {
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + l / 16];   // A[i][k], i = l % 16, k = l / 16
    const double b = B[(l / 16) * 16 + l % 16];  // B[k][j], k = l / 16, j = l % 16
    d4 c;
    // C / D: lane l, register r holds row ROW(l, r), column l % 16
#ifndef ROW
#define ROW(l, r) ((l) / 16 + 4 * (r))
#endif
    for (int r = 0; r < 4; ++r) c[r] = C[ROW(l, r) * 16 + l % 16];
    const d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[ROW(l, r) * 16 + l % 16] = d[r];
}
int main() {
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    long bad[4] = {0, 0, 0, 0}, total = 0;
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 64 * 8), hipMalloc(&dB, 64 * 8), hipMalloc(&dC, 256 * 8), hipMalloc(&dD, 256 * 8);
    for (int trial = 0; trial < 200; ++trial) {
        std::vector<double> A(64), B(64), C(256), D(256);
        for (auto& v : A) v = U(rng) * std::ldexp(1.0, (int)(rng() % 40) - 20);
        for (auto& v : B) v = U(rng);
        for (auto& v : C) v = U(rng) * std::ldexp(1.0, (int)(rng() % 20) - 10);
        hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice), hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice), hipMemcpy(dC, C.data(), 256 * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, 1, 64, 0, 0, dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                const double* a = &A[i * 4];
                double b[4];
                for (int kk = 0; kk < 4; ++kk) b[kk] = B[kk * 16 + j];
                const double c = C[i * 16 + j];
                const double fA = std::fma(a[3], b[3], std::fma(a[2], b[2], std::fma(a[1], b[1], std::fma(a[0], b[0], c))));
                volatile double p0 = a[0] * b[0], p1 = a[1] * b[1], p2 = a[2] * b[2], p3 = a[3] * b[3];
                const double fB = (((c + p0) + p1) + p2) + p3;
                const double fC = std::fma(a[0], b[0], std::fma(a[1], b[1], std::fma(a[2], b[2], std::fma(a[3], b[3], c))));
                const double fD = c + std::fma(a[3], b[3], std::fma(a[2], b[2], std::fma(a[1], b[1], a[0] * b[0])));
                const double got = D[i * 16 + j];
                bad[0] += std::memcmp(&got, &fA, 8) != 0, bad[1] += std::memcmp(&got, &fB, 8) != 0, bad[2] += std::memcmp(&got, &fC, 8) != 0, bad[3] += std::memcmp(&got, &fD, 8) != 0;
                ++total;
            }
    }
    printf("v_mfma_f64_16x16x4_f64 vs candidates over %ld outputs: mismatches  A(fused k asc)=%ld  B(unfused)=%ld  C(fused k desc)=%ld  D(products first)=%ld\n", total, bad[0], bad[1], bad[2], bad[3]);
    return 0;
}
