// ewald.hip -- classic Ewald reciprocal sum with per-subset structure factors (nonbonded method "Ewald").
//
// Replaces platforms/common/src/kernels/ewald.cc (calculateEwaldCosSinSums :9-78, calculateEwaldForces :85-142);
// arithmetic parity with platforms/reference/src/ReferenceSlicedLJCoulombIxn.cpp:256-358.  O(N*K): meant for the
// small systems this method is used for; one work-group per k-vector for the structure factors (LDS reduction per
// subset), one thread per atom for the forces.
#include "snb_internal.h"

namespace snb {

__device__ inline void sinCos(float x, float* s, float* c) { sincosf(x, s, c); }
__device__ inline void sinCos(double x, double* s, double* c) { sincos(x, s, c); }

template <typename Real> __global__ __launch_bounds__(256) void k_ewaldSums(const EwaldParams<Real> p) {
    extern __shared__ double s_sum[];      // [2*nsub]: cos sums, sin sums
    const int kv = blockIdx.x;
    const int3 m = p.kvec[kv];
    const Real kx = m.x * p.recipBox[0], ky = m.y * p.recipBox[1], kz = m.z * p.recipBox[2];
    for (int i = threadIdx.x; i < 2 * p.nsub; i += 256) s_sum[i] = 0.0;
    __syncthreads();
    // the sorted order is subset-major, so a thread's atoms (stride 256) change subset a handful of times: the sums of the current subset
    // stay in registers and reach LDS once per change (round 2 issued two ds_add_f64 per atom: 256 threads funnelled into 2 n addresses)
    int cur = -1; double accC = 0.0, accS = 0.0;
    auto flush = [&]() {
        if (cur >= 0) {
            __hip_atomic_fetch_add(&s_sum[cur], accC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&s_sum[p.nsub + cur], accS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        accC = 0.0; accS = 0.0;
    };
    for (int a = threadIdx.x; a < p.natoms; a += 256) {
        const int s = p.atomSubset[a];
        if (s < 0) continue;
        if (s != cur) { flush(); cur = s; }
        const auto v = p.posq[a];
        const Real ph = kx * v.x + ky * v.y + kz * v.z;
        Real sn, cs;
        sinCos(ph, &sn, &cs);
        accC += (double)(v.w * cs); accS += (double)(v.w * sn);
    }
    flush();
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * p.nsub; i += 256) p.cosSin[(size_t)kv * 2 * p.nsub + i] = (Real)s_sum[i];
    if (p.wantEnergy && threadIdx.x == 0) {
        const double k2 = (double)kx * kx + (double)ky * ky + (double)kz * kz;
        const double ak = exp(k2 * p.factorEwald) / k2;
        for (int j = 0; j < p.nsub; j++) {
            for (int i = 0; i < j; i++)
                atomicAdd(&SNB_SLICE_E_PARTITION(p.sliceE, p.nsub * (p.nsub + 1))[2 * (j * (j + 1) / 2 + i)], 2 * p.recipCoeff * ak * (s_sum[i] * s_sum[j] + s_sum[p.nsub + i] * s_sum[p.nsub + j]));
            atomicAdd(&SNB_SLICE_E_PARTITION(p.sliceE, p.nsub * (p.nsub + 1))[2 * (j * (j + 3) / 2)], p.recipCoeff * ak * (s_sum[j] * s_sum[j] + s_sum[p.nsub + j] * s_sum[p.nsub + j]));
        }
    }
}

template <typename Real> __global__ __launch_bounds__(256) void k_ewaldForces(const EwaldParams<Real> p) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= p.natoms) return;
    const int si = p.atomSubset[a];
    if (si < 0) return;
    const auto v = p.posq[a];
    double fx = 0, fy = 0, fz = 0;
    for (int kv = 0; kv < p.nk; kv++) {
        const int3 m = p.kvec[kv];
        const Real kx = m.x * p.recipBox[0], ky = m.y * p.recipBox[1], kz = m.z * p.recipBox[2];
        const Real ph = kx * v.x + ky * v.y + kz * v.z;
        Real sn, cs;
        sinCos(ph, &sn, &cs);
        const double k2 = (double)kx * kx + (double)ky * ky + (double)kz * kz;
        const double ak = exp(k2 * p.factorEwald) / k2;
        const Real* cS = p.cosSin + (size_t)kv * 2 * p.nsub;
        double f = 0;
        for (int j = 0; j < p.nsub; j++) {
            const int slice = si > j ? si * (si + 1) / 2 + j : j * (j + 1) / 2 + si;
            f += (double)p.lambdas[2 * slice] * ((double)cS[j] * (v.w * sn) - (double)cS[p.nsub + j] * (v.w * cs));
        }
        f *= 2 * p.recipCoeff * ak;
        fx += f * kx; fy += f * ky; fz += f * kz;
    }
    p.fpx[a] = (Real)fx; p.fpy[a] = (Real)fy; p.fpz[a] = (Real)fz;
}

template <typename Real> void launchEwald(const EwaldParams<Real>& p, hipStream_t s) {
    if (p.nk <= 0 || p.natoms <= 0) return;
    hipLaunchKernelGGL((k_ewaldSums<Real>), dim3(p.nk), dim3(256), sizeof(double) * 2 * p.nsub, s, p);
    hipLaunchKernelGGL((k_ewaldForces<Real>), dim3((p.natoms + 255) / 256), dim3(256), 0, s, p);
}
template void launchEwald<float>(const EwaldParams<float>&, hipStream_t);
template void launchEwald<double>(const EwaldParams<double>&, hipStream_t);

}  // namespace snb
