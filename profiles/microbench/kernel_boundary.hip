// Microbenchmark: what does a kernel boundary cost on MI355X (8 XCDs, one L2 each) as a function of what the kernel did?
//   hipcc --offload-arch=gfx950 -O3 kernel_boundary.hip -o kb && ./kb
// Back-to-back launches on one stream, timed with one event pair around K launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_noop(int* p) {}
__global__ void k_read(const int* p, int* sink, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n && p[i] == 0x7fffffff) *sink = 1; }
__global__ void k_write(int* p, int n, int stride, int v) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[(size_t)i * stride] = v; }
// what a round kernel's list append does: one returning atomic per workgroup on ONE counter, then a store at the returned position
__global__ void k_append(int* counter, int* out) {
    __shared__ int base;
    if (threadIdx.x == 0) base = atomicAdd(counter, (int)blockDim.x);
    __syncthreads();
    out[(base + threadIdx.x) & 0xfffff] = base;
}
__global__ void k_append_noret(int* counter, int* out) {
    if (threadIdx.x == 0) atomicAdd(counter, 1);
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xfffff] = 1;
}
// the same signal as a plain store of a flag (no read-modify-write)
__global__ void k_flag(int* flag, int* out) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xfffff] = 1;
}
// ... and as atomics spread over 64 counters on separate cache lines
__global__ void k_append_spread(int* counters, int* out) {
    if (threadIdx.x == 0) atomicAdd(counters + 32 * (blockIdx.x & 63), 1);
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xfffff] = 1;
}
__global__ void k_chain(const int* next, int* out, int hops) { int x = threadIdx.x + blockIdx.x * blockDim.x; for (int h = 0; h < hops; ++h) x = next[x]; out[threadIdx.x + blockIdx.x * blockDim.x] = x; }
int main() {
    const int K = 2000;
    int* buf; size_t bytes = (size_t)1 << 30; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
    int* sink; CK(hipMalloc(&sink, 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 50; ++i) launch(i);
        (void)hipStreamSynchronize(s);
        (void)hipEventRecord(a, s);
        for (int i = 0; i < K; ++i) launch(i);
        (void)hipEventRecord(b, s);
        (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        printf("%-58s %7.2f us per launch\n", name, 1e3 * ms / K);
    };
    run("no-op, 1 block", [&](int) { hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, s, buf); });
    run("no-op, 1024 blocks x 256", [&](int) { hipLaunchKernelGGL(k_noop, dim3(1024), dim3(256), 0, s, buf); });
    run("read 1 MB (4096 blocks... 256k ints)", [&](int) { hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, s, (const int*)buf, sink, 262144); });
    run("write 1 int", [&](int i) { hipLaunchKernelGGL(k_write, dim3(1), dim3(64), 0, s, buf, 1, 1, i); });
    run("write 256 ints, contiguous (1 KB)", [&](int i) { hipLaunchKernelGGL(k_write, dim3(1), dim3(256), 0, s, buf, 256, 1, i); });
    run("write 32 k ints contiguous (128 KB), 128 blocks", [&](int i) { hipLaunchKernelGGL(k_write, dim3(128), dim3(256), 0, s, buf, 32768, 1, i); });
    run("write 32 k ints, stride 1 KB (32 k lines), 128 blocks", [&](int i) { hipLaunchKernelGGL(k_write, dim3(128), dim3(256), 0, s, buf, 32768, 256, i); });
    run("write 1 M ints contiguous (4 MB), 4096 blocks", [&](int i) { hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, s, buf, 1 << 20, 1, i); });
    run("write 256 k ints, stride 1 KB (256 k lines), 1024 blocks", [&](int i) { hipLaunchKernelGGL(k_write, dim3(1024), dim3(256), 0, s, buf, 262144, 256, i); });
    for (int blocks : {16, 128, 512, 2048, 8192}) {
        char nm[96]; snprintf(nm, sizeof nm, "append: 1 returning atomic per block, %4d blocks x 256", blocks);
        run(nm, [&](int) { hipLaunchKernelGGL(k_append, dim3(blocks), dim3(256), 0, s, sink, buf); });
        snprintf(nm, sizeof nm, "          non-returning atomic per block, %4d blocks x 256", blocks);
        run(nm, [&](int) { hipLaunchKernelGGL(k_append_noret, dim3(blocks), dim3(256), 0, s, sink, buf); });
    }
    for (int blocks : {2048, 8192}) {
        char nm[96]; snprintf(nm, sizeof nm, "flag store per block (no RMW), %4d blocks x 256", blocks);
        run(nm, [&](int) { hipLaunchKernelGGL(k_flag, dim3(blocks), dim3(256), 0, s, sink, buf); });
        snprintf(nm, sizeof nm, "atomics spread over 64 cache lines, %4d blocks x 256", blocks);
        run(nm, [&](int) { hipLaunchKernelGGL(k_append_spread, dim3(blocks), dim3(256), 0, s, buf + (1 << 22), buf); });
    }
    // dependent-load chain: next[x] = (x * 40503 + 17) mod M over a 256 MB table
    const int M = 1 << 26;
    std::vector<int> h(M); for (int i = 0; i < M; ++i) h[i] = (int)(((long long)i * 40503 + 17) % M);
    CK(hipMemcpy(buf, h.data(), (size_t)M * 4, hipMemcpyHostToDevice));
    int* out = buf + M;
    for (int hops : {1, 4, 8, 16, 32}) {
        char nm[96]; snprintf(nm, sizeof nm, "pointer chase, %2d dependent loads, 128 blocks x 256", hops);
        run(nm, [&](int) { hipLaunchKernelGGL(k_chain, dim3(128), dim3(256), 0, s, (const int*)buf, out, hops); });
    }
    return 0;
}
