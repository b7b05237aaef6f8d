// microbenchmark: LDS atomic add throughput by type (float / u32 / u64 / double), conflict-free lanes
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T> __global__ __launch_bounds__(256) void k(T* out, int iters) {
    __shared__ T buf[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = 0;
    __syncthreads();
    int idx = threadIdx.x;
    for (int it = 0; it < iters; it++) {
        __hip_atomic_fetch_add(&buf[(idx + it * 67) & 4095], (T)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = buf[0];
}
template <typename T> __global__ __launch_bounds__(256) void krmw(T* out, int iters) {
    __shared__ T buf[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = 0;
    __syncthreads();
    int idx = threadIdx.x;
    for (int it = 0; it < iters; it++) { int a = (idx + it * 64) & 4095; buf[a] = buf[a] + (T)1; }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = buf[0];
}
template <typename T> void run(const char* name, bool rmw) {
    T* d; hipMalloc(&d, sizeof(T) * 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int iters = 2000, blocks = 1024;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a);
        if (rmw) hipLaunchKernelGGL(krmw<T>, dim3(blocks), dim3(256), 0, 0, d, iters); else hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    double waveInstr = (double)blocks * 4 * iters;
    printf("%-14s %s: %.3f ms, %.1f cycles per wave-instruction per CU (at 2.4 GHz, 256 CUs)\n", name, rmw ? "rmw   " : "atomic", ms, ms * 1e-3 * 2.4e9 * 256 / waveInstr);
    hipFree(d);
}
int main() {
    run<float>("float", false); run<unsigned>("u32", false); run<unsigned long long>("u64", false); run<double>("double", false); run<int>("i32", false);
    run<float>("float", true); run<double>("double", true);
    return 0;
}
