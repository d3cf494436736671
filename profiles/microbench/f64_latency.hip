// Dependent-chain latency of the f64 operations a turn of the implicit solve is made of, one wave (64 lanes) on an otherwise idle GPU:
// clocks per operation from s_memtime around N dependent repetitions.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 2048
__device__ inline double recip_refined(double y) {
    const double r0 = __builtin_amdgcn_rcp(y);
    const double f0 = __builtin_fma(-y, r0, 1.0);
    const double r1 = __builtin_fma(r0, f0, r0);
    const double f1 = __builtin_fma(-y, r1, 1.0);
    return __builtin_fma(r1, f1, r1);
}
__global__ void k(double* out, long long* clk, double a, double b) {
    double x = a + threadIdx.x * 1e-3;
    long long t0, t1;
    // 0: fma chain
    t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = __builtin_fma(x, b, a);
    t1 = clock64(); if (threadIdx.x == 0) clk[0] = t1 - t0;
    // 1: add chain
    t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = x + b;
    t1 = clock64(); if (threadIdx.x == 0) clk[1] = t1 - t0;
    // 2: full division chain
    x = 1.0 + threadIdx.x * 1e-3;
    t0 = clock64();
#pragma unroll 4
    for (int i = 0; i < N; ++i) x = a / (x + b);
    t1 = clock64(); if (threadIdx.x == 0) clk[2] = t1 - t0;
    // 3: division tail with a prepared reciprocal (divisor constant)
    const double y = 1.0 + b, r = recip_refined(y);
    t0 = clock64();
#pragma unroll 4
    for (int i = 0; i < N; ++i) { const double q0 = x * r; const double e = __builtin_fma(-y, q0, x); x = __builtin_fma(e, r, q0) + a; }
    t1 = clock64(); if (threadIdx.x == 0) clk[3] = t1 - t0;
    // 4: rcp chain
    t0 = clock64();
#pragma unroll 8
    for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rcp(x) + a;
    t1 = clock64(); if (threadIdx.x == 0) clk[4] = t1 - t0;
    // 5: f32 division chain
    float f = (float)x;
    t0 = clock64();
#pragma unroll 4
    for (int i = 0; i < N; ++i) f = (float)a / (f + (float)b);
    t1 = clock64(); if (threadIdx.x == 0) clk[5] = t1 - t0;
    // 6: cvt f64->f32->f64 chain
    t0 = clock64();
#pragma unroll 8
    for (int i = 0; i < N; ++i) x = (double)(float)(x + b);
    t1 = clock64(); if (threadIdx.x == 0) clk[6] = t1 - t0;
    // 7: compare + select chain
    t0 = clock64();
#pragma unroll 8
    for (int i = 0; i < N; ++i) x = (x < a) ? x + b : x - b;
    t1 = clock64(); if (threadIdx.x == 0) clk[7] = t1 - t0;
    // 8: LDS write + read round trip
    __shared__ double sh[64];
    t0 = clock64();
#pragma unroll 4
    for (int i = 0; i < N; ++i) { sh[threadIdx.x] = x; x = sh[(threadIdx.x + 1) & 63] + b; }
    t1 = clock64(); if (threadIdx.x == 0) clk[8] = t1 - t0;
    out[threadIdx.x] = x + f;
}
int main() {
    double* out; long long* clk;
    hipMalloc(&out, 64 * 8); hipMalloc(&clk, 16 * 8);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, clk, 1.000001, 0.999999); hipDeviceSynchronize(); }
    long long h[16]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"fma", "add", "full division (+add)", "division tail (+add)", "rcp (+add)", "f32 division (+add)", "cvt f64->f32->f64 (+add)", "compare+select (+add)", "LDS write/read (+add)"};
    for (int i = 0; i < 9; ++i) printf("%-28s %8.1f clocks per dependent repetition\n", names[i], (double)h[i] / N);
    return 0;
}
