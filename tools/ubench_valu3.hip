// microbenchmark 3: packed-fp32 issue cost by operand kind (VGPR / SGPR pair / op_sel broadcast / dependent chain), gfx950.
// Also run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES` to see how the counters count packed instructions.
// build on the box: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu3.hip -o /tmp/ub3 && /tmp/ub3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP8(X) X X X X X X X X
template <int MODE> __global__ __launch_bounds__(256) void k3(float* out, int iters, float seed, v2f sb) {
    float a0 = seed + threadIdx.x, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = p0 * 1.01f, p3 = p1 * 1.01f;
    const v2f pb = {1.0001f, 1.0001f};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { asm volatile(REP8("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed)); }
        else if (MODE == 1) { asm volatile(REP8("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb)); }
        else if (MODE == 2) { asm volatile(REP8("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "s"(sb)); }
        else if (MODE == 3) { asm volatile(REP8("v_pk_fma_f32 %0, %0, %1, %1\n s_nop 0\n v_pk_fma_f32 %0, %0, %1, %1\n s_nop 0\n v_pk_fma_f32 %0, %0, %1, %1\n s_nop 0\n v_pk_fma_f32 %0, %0, %1, %1\n s_nop 0\n") : "+v"(p0) : "v"(pb)); }
        else if (MODE == 4) { asm volatile(REP8("v_pk_mul_f32 %0, %0, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %1, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %2, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %3, %3, %4 op_sel_hi:[1,0]\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb)); }
        else if (MODE == 5) { asm volatile(REP8("v_pk_fma_f32 %0, %2, %0, %3\n s_nop 0\n v_pk_fma_f32 %0, %2, %0, %3\n s_nop 0\n v_pk_fma_f32 %0, %2, %0, %3\n s_nop 0\n v_pk_fma_f32 %0, %2, %0, %3\n s_nop 0\n") : "+v"(p0), "+v"(p1) : "v"(pb), "s"(sb)); }
        else if (MODE == 6) { asm volatile(REP8("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n") : "+v"(a0) : "v"(seed)); }
        else if (MODE == 7) { asm volatile(REP8("v_pk_add_f32 %0, %0, %4 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %4 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb)); }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE> void run(const char* name, int wavesPerSimd) {
    float* d; hipMalloc(&d, sizeof(float) * 256 * 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 20000, blocks = 256 * wavesPerSimd;
    float ms = 0;
    v2f sb = {1.0001f, 1.0001f};
    for (int rep = 0; rep < 2; rep++) { hipEventRecord(a); hipLaunchKernelGGL(k3<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, sb); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); }
    const double perSimd = (double)iters * 32 * wavesPerSimd;   // wave-instructions each SIMD issued (one wave of every block lands on each SIMD)
    printf("MODE %d %-44s waves/SIMD %d: %.3f ms -> %.3f ns per wave-instruction per SIMD\n", MODE, name, wavesPerSimd, ms, ms * 1e6 / perSimd);
    hipFree(d);
}
int main() {
    for (int w : {4, 8}) {
        if (w == 4) {
            run<0>("v_fma_f32 4 chains", 4); run<1>("v_pk_fma_f32 4 chains, VGPR operands", 4); run<2>("v_pk_fma_f32 4 chains, SGPR-pair operand", 4);
            run<3>("v_pk_fma_f32 1 dependent chain + s_nop 0", 4); run<4>("v_pk_mul_f32 op_sel_hi:[1,0]", 4); run<5>("v_pk_fma_f32 Horner (dependent, SGPR addend)", 4);
            run<6>("v_fma_f32 1 dependent chain", 4); run<7>("v_pk_add_f32 with neg/op_sel modifiers", 4);
        } else {
            run<0>("v_fma_f32 4 chains", 8); run<1>("v_pk_fma_f32 4 chains, VGPR operands", 8); run<2>("v_pk_fma_f32 4 chains, SGPR-pair operand", 8);
            run<3>("v_pk_fma_f32 1 dependent chain + s_nop 0", 8); run<4>("v_pk_mul_f32 op_sel_hi:[1,0]", 8); run<5>("v_pk_fma_f32 Horner (dependent, SGPR addend)", 8);
            run<6>("v_fma_f32 1 dependent chain", 8); run<7>("v_pk_add_f32 with neg/op_sel modifiers", 8);
        }
    }
    return 0;
}
