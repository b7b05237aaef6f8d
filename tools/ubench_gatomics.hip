// (GPU box) throughput of no-return global float / u64 atomic adds on MI355X by access pattern: what does the memory-side atomic unit
// charge for -- wave instructions, 32-byte sectors, or lanes?  (the j-force scatter of the pair kernel.)
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/ubench_gatomics.hip -o /tmp/ubga && /tmp/ubga
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// pattern: lane l of a wave adds to element  base(wave, it) + off(l);  active lanes: l < nActive
template <typename T, int MODE> __global__ void k(T* a, unsigned nElem, int iters, int nActive, int laneStride, int arrays) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned h = wave * 2654435761u + 12345u;
    for (int it = 0; it < iters; it++) {
        h = h * 1664525u + 1013904223u;
        unsigned base;
        if (MODE == 0) base = (h >> 8) % (nElem / 64) * 64;                  // a random 64-element block per instruction
        else base = ((wave * 64u * (unsigned)iters + (unsigned)it * 64u) * 7u) % (nElem - 4096);      // a walk with local reuse
        if (lane < nActive) {
            for (int c = 0; c < arrays; c++)
                __hip_atomic_fetch_add(a + (size_t)c * nElem + base + (unsigned)(lane * laneStride) % 4096u, (T)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <typename T> int run(const char* name, int nActive, int laneStride, int arrays) {
    const unsigned nElem = 1u << 20;      // 4 MB (float) / 8 MB (u64) per array: the size of a 300k-atom force component
    T* d; CK(hipMalloc(&d, sizeof(T) * (size_t)nElem * 3)); CK(hipMemset(d, 0, sizeof(T) * (size_t)nElem * 3));
    const int iters = 64, blocks = 4096, threads = 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<T, 0>), dim3(blocks), dim3(threads), 0, 0, d, nElem, iters, nActive, laneStride, arrays);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<T, 0>), dim3(blocks), dim3(threads), 0, 0, d, nElem, iters, nActive, laneStride, arrays);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr = (double)blocks * threads / 64 * iters * arrays;
    const double bytesPerLane = sizeof(T) * laneStride;
    const double spanBytes = bytesPerLane * (nActive - 1) + sizeof(T);
    double sectors = laneStride * sizeof(T) >= 32 ? nActive : (spanBytes + 31) / 32;      // 32-byte sectors touched per instruction
    printf("%-58s %8.3f ms  %7.2f wave-instr/ns  %7.2f sectors/ns  %7.2f lane-adds/ns\n", name, ms, instr / (ms * 1e6), instr * sectors / (ms * 1e6), instr * nActive / (ms * 1e6));
    CK(hipFree(d));
    return 0;
}

int main() {
    run<float>("f32, 64 lanes contiguous (256 B, 8 sectors)", 64, 1, 1);
    run<float>("f32, 32 lanes contiguous (128 B, 4 sectors)", 32, 1, 1);
    run<float>("f32, 16 lanes contiguous (64 B, 2 sectors)", 16, 1, 1);
    run<float>("f32, 8 lanes contiguous (32 B, 1 sector)", 8, 1, 1);
    run<float>("f32, 32 lanes, one per sector (stride 8)", 32, 8, 1);
    run<float>("f32, 32 lanes, every other element (stride 2, 8 sectors)", 32, 2, 1);
    run<float>("f32, 32 lanes contiguous x 3 arrays (the tile's scatter)", 32, 1, 3);
    run<unsigned long long>("u64, 32 lanes contiguous (256 B, 8 sectors)", 32, 1, 1);
    run<unsigned long long>("u64, 32 lanes contiguous x 3 arrays (SNB_MIXED scatter)", 32, 1, 3);
    run<float>("f32, 1 lane (1 sector)", 1, 1, 1);
    return 0;
}
