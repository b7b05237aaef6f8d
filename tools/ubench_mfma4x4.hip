// (GPU box) lane layout of v_mfma_f32_4x4x1_16b_f32, checked empirically (used by the lambda mix of k_convolveX for <= 4 subsets):
// hypothesis  A: lane l holds A[block l/4][row l%4];  B: lane l holds B[block l/4][col l%4];  D: register r of lane l = D[block l/4][row r][col l%4]
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* o) {
    const int l = threadIdx.x;
    const float a = 1.0f + (l / 4) * 10.0f + (l % 4);          // A[b][i] = 1 + 10 b + i
    const float b = 100.0f + (l / 4) * 1000.0f + (l % 4) * 7.0f; // B[b][j] = 100 + 1000 b + 7 j
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) o[l * 4 + r] = c[r];
}
int main() {
    float* d; hipMalloc(&d, 64 * 4 * sizeof(float));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
        const int b = l / 4, j = l % 4, i = r;
        const float want = (1.0f + b * 10.0f + i) * (100.0f + b * 1000.0f + j * 7.0f);
        if (h[l * 4 + r] != want) { if (bad < 8) printf("lane %d reg %d: got %g want %g\n", l, r, h[l * 4 + r], want); bad++; }
    }
    printf(bad ? "layout hypothesis WRONG (%d mismatches)\n" : "layout hypothesis confirmed (%d mismatches)\n", bad);
    return bad != 0;
}
