// microbenchmark: issue rate of scalar vs packed fp32 VALU ops and transcendentals on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f, a4 = a0 * 1.4f, a5 = a0 * 1.5f, a6 = a0 * 1.6f, a7 = a0 * 1.7f;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float b = 1.0001f, c = 0.0001f;
    const float2v pb = {b, b}, pc = {c, c};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {   // 8 independent v_fma_f32
            a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
            a4 = __builtin_fmaf(a4, b, c); a5 = __builtin_fmaf(a5, b, c); a6 = __builtin_fmaf(a6, b, c); a7 = __builtin_fmaf(a7, b, c);
        } else if (MODE == 1) {   // 4 independent v_pk_fma_f32 (same flops as mode 0)
            p0 = __builtin_elementwise_fma(p0, pb, pc); p1 = __builtin_elementwise_fma(p1, pb, pc);
            p2 = __builtin_elementwise_fma(p2, pb, pc); p3 = __builtin_elementwise_fma(p3, pb, pc);
        } else if (MODE == 2) {   // 8 v_rsq_f32
            a0 = __builtin_amdgcn_rsqf(a0); a1 = __builtin_amdgcn_rsqf(a1); a2 = __builtin_amdgcn_rsqf(a2); a3 = __builtin_amdgcn_rsqf(a3);
            a4 = __builtin_amdgcn_rsqf(a4); a5 = __builtin_amdgcn_rsqf(a5); a6 = __builtin_amdgcn_rsqf(a6); a7 = __builtin_amdgcn_rsqf(a7);
        } else if (MODE == 3) {   // 8 v_mov_b32_dpp row_ror:1
            a0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a0), 0x121, 0xF, 0xF, true)) + c;
            a1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a1), 0x121, 0xF, 0xF, true)) + c;
            a2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a2), 0x121, 0xF, 0xF, true)) + c;
            a3 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a3), 0x121, 0xF, 0xF, true)) + c;
        } else if (MODE == 4) {   // 4 v_pk_mul_f32 + 4 v_pk_add_f32
            p0 = p0 * pb; p1 = p1 * pb; p2 = p2 * pb; p3 = p3 * pb; p0 = p0 + pc; p1 = p1 + pc; p2 = p2 + pc; p3 = p3 + pc;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE> void run(const char* name, int instrPerIter) {
    float* d; hipMalloc(&d, sizeof(float) * 256 * 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int iters = 4000, blocks = 256 * 8;   // 8 blocks/CU = 32 waves/CU = 8 waves/SIMD
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) { hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); }
    double waveInstrPerSimd = (double)blocks * 4 * iters * instrPerIter / 1024.0;
    printf("%-28s %.3f ms  -> %.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / waveInstrPerSimd);
    hipFree(d);
}
int main() {
    run<0>("v_fma_f32 x8", 8); run<1>("v_pk_fma_f32 x4", 4); run<4>("v_pk_mul+v_pk_add x8", 8); run<2>("v_rsq_f32 x8", 8); run<3>("v_mov_dpp+v_add x4", 8);
    return 0;
}
