// microbenchmark 2: cycles per wave-instruction (s_memtime = shader cycles) for the instruction classes of the pair kernel, gfx950
// build on the box: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu2.hip -o /tmp/ub2 && /tmp/ub2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP8(X) X X X X X X X X
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f, a4 = a0 * 1.4f, a5 = a0 * 1.5f, a6 = a0 * 1.6f, a7 = a0 * 1.7f;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0 * 1.01f, p5 = p1 * 1.01f, p6 = p2 * 1.01f, p7 = p3 * 1.01f;
    const v2f pb = {1.0001f, 1.0001f}, pc = {0.0001f, 0.0001f};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { asm volatile(REP8("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %2, %2, %1, %1\n") : "+v"(a0), "+v"(a1), "+v"(a2) : ); }
        else if (MODE == 1) { asm volatile(REP8("v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2\n") : "+v"(p0), "+v"(p1) : "v"(pb)); }
        else if (MODE == 2) { asm volatile(REP8("v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2\n") : "+v"(p0), "+v"(p1) : "v"(pb)); }
        else if (MODE == 3) { asm volatile(REP8("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n") : "+v"(p0), "+v"(p1) : "v"(pc)); }
        else if (MODE == 4) { asm volatile(REP8("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n") : "+v"(a0), "+v"(a1) : ); }
        else if (MODE == 5) { asm volatile(REP8("v_exp_f32 %0, %0\n v_rcp_f32 %1, %1\n") : "+v"(a0), "+v"(a1) : ); }
        else if (MODE == 6) { asm volatile(REP8("v_sub_f32_dpp %0, %0, %2 row_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_sub_f32_dpp %1, %1, %2 row_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n") : "+v"(a0), "+v"(a1) : "v"(a2)); }
        else if (MODE == 7) { asm volatile(REP8("v_add_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n") : "+v"(a0), "+v"(a1) : "v"(a2)); }
        else if (MODE == 8) { asm volatile(REP8("v_cmp_gt_f32 vcc, %0, %2\n v_cndmask_b32 %1, 0, %1, vcc\n") : "+v"(a0), "+v"(a1) : "v"(a2) : "vcc"); }
        else if (MODE == 9) {   // 8 independent pk_mul chains (no dependency stalls) interleaved
            asm volatile(REP8("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb)); }
        else if (MODE == 10) {  // 4 independent pk_fma chains
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb)); }
        else if (MODE == 11) {  // 4 independent scalar fma chains
            asm volatile(REP8("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a7)); }
        else if (MODE == 12) { asm volatile(REP8("v_mov_b32 %0, %2\n v_mov_b32 %1, %2\n") : "+v"(a0), "+v"(a1) : "v"(a2)); }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.x + p6.x + p7.x;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char* name, int instrPerIter, int wavesPerSimd) {
    float* d; long long* c; hipMalloc(&d, sizeof(float) * 256 * 4096); hipMalloc(&c, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int iters = 20000, blocks = 256 * wavesPerSimd;   // 1 block of 4 waves per CU per unit
    float ms = 0; long long hc = 0;
    for (int rep = 0; rep < 2; rep++) { hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, c, iters, 1.0f); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); }
    hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
    double perWave = (double)hc / ((double)iters * instrPerIter);
    printf("%-34s waves/SIMD %d: %.3f ms, %lld cycles in wave 0 -> %.2f cycles/instr seen by one wave, %.2f cycles/instr per SIMD (issue), clock %.2f GHz\n",
           name, wavesPerSimd, ms, hc, perWave, perWave / wavesPerSimd, hc / (ms * 1e6));
    hipFree(d); hipFree(c);
}
int main() {
    for (int w : {1, 2, 4, 8}) {
        switch (w) {
#define ALL(W) run<11>("v_fma_f32 (4 chains)", 32, W); run<10>("v_pk_fma_f32 (4 chains)", 32, W); run<9>("v_pk_mul_f32 (4 chains)", 32, W); run<3>("v_pk_add_f32 (2 chains)", 16, W); \
    run<4>("v_rsq_f32", 16, W); run<5>("v_exp_f32+v_rcp_f32", 16, W); run<6>("v_sub_f32_dpp row_ror:1", 16, W); run<7>("v_add_f32+v_mul_f32", 16, W); run<8>("v_cmp+v_cndmask", 16, W); run<12>("v_mov_b32", 16, W);
            case 1: ALL(1) break; case 2: ALL(2) break; case 4: ALL(4) break; default: ALL(8) break;
        }
    }
    return 0;
}
