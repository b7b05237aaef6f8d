// direct.hip -- direct-space sliced pair kernel for gfx950 (MI355X), plus the per-pair list kernels
// (1-4 exceptions, Ewald exclusion corrections).
//
// What it replaces in the reference: the pair-interaction snippet platforms/common/src/kernels/coulombLennardJones.cc:1-124
// (spliced into OpenMM's 32x32 tile kernel), the exception snippet nonbondedExceptions.cc:1-25 and the exclusion
// snippet pmeExclusions.cc:1-47; arithmetic parity is with platforms/reference/src/ReferenceSlicedLJCoulombIxn.cpp:367-506,
// 571-631 and ReferenceSlicedLJCoulomb14.cpp:61-95.
//
// MI355X mapping (not the reference's): one 64-lane wavefront owns one 32-atom i-block.  Lane l holds i-atom
// (l & 31); the two 32-lane halves work on different j-atoms of the same 32-atom j-tile, so a tile is 16 steps of
// 64 pair slots.  j-atoms are staged once per tile into LDS (posq b128 + sigeps b64, stored twice so the rotated
// read index il+h+2s needs no wrap), j-forces are accumulated with LDS float atomics (ds_add_f32) and flushed with
// one global atomic per j-atom and component.  Because blocks are subset-uniform, the slice of a tile is a scalar:
// lambda scaling costs two multiplies per TILE and per-slice energies are two accumulators per tile -- there is no
// per-pair slice arithmetic at all (the reference computes the slice index and loads LAMBDA[slice] per pair).
#include "snb_internal.h"

namespace snb {

// ---- math helpers -------------------------------------------------------------------------------
__device__ inline float rsq(float x) { return __frsqrt_rn(x); }
__device__ inline double rsq(double x) { return 1.0 / sqrt(x); }
__device__ inline float fexp(float x) { return __expf(x); }
__device__ inline double fexp(double x) { return exp(x); }
// erfc(ar) given e = exp(-ar^2).  Single precision: Abramowitz & Stegun 7.1.26 (max abs error 1.5e-7), the
// same approximation the reference GPU path uses (coulombLennardJones.cc:18-23); double: libm.
__device__ inline float erfcFromExp(float ar, float e) {
    float t = __frcp_rn(1.0f + 0.3275911f * ar);
    return (0.254829592f + (-0.284496736f + (1.421413741f + (-1.453152027f + 1.061405429f * t) * t) * t) * t) * t * e;
}
__device__ inline double erfcFromExp(double ar, double) { return erfc(ar); }

__device__ inline void ldsAdd(float* p, float v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void ldsAdd(double* p, double v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void gAdd(float* p, float v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void gAdd(double* p, double v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ inline double waveSum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ inline int sliceOf(int a, int b) { return a > b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a; }

template <typename Real> __device__ inline void wrapDelta(Real& dx, Real& dy, Real& dz, const Real* box, const Real* inv) {
    // OpenMM ReferenceForce::getDeltaRPeriodic (triclinic form)
    Real s = floor(dz * inv[2] + Real(0.5)); dx -= s * box[6]; dy -= s * box[7]; dz -= s * box[8];
    s = floor(dy * inv[1] + Real(0.5)); dx -= s * box[3]; dy -= s * box[4];
    s = floor(dx * inv[0] + Real(0.5)); dx -= s * box[0];
}

// ---- the tile kernel ----------------------------------------------------------------------------
template <typename Real, int MC, bool WRAP, bool ENERGY>
__global__ __launch_bounds__(256) void k_direct(const DirectParams<Real> p) {
    using T4 = typename Vec<Real>::T4;
    using T2 = typename Vec<Real>::T2;
    __shared__ T4 s_pos[4][64];
    __shared__ T2 s_se[4][64];
    __shared__ Real s_f[4][3][64];

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * 4 + wid;
    if (item >= p.numWork) return;                       // whole wave leaves; no block-level barrier is used below
    const int I = __builtin_amdgcn_readfirstlane(p.workOrder[p.workStart + item * p.workStride]);
    const int il = lane & 31, h = lane >> 5;

    const T4 pi = p.posq[I * 32 + il];
    const T2 sei = p.sigeps[I * 32 + il];
    const int si = p.blockSubset[I];
    const Real qi = pi.w * p.k4pe;
    Real c6i = 0;
    if (MC == MC_LJPME) c6i = Real(8) * sei.x * sei.x * sei.x * sei.y;
    Real fix = 0, fiy = 0, fiz = 0;
    Real ecl = 0, elj = 0;
    int curSlice = -1;

    const int2 bt = p.blockTiles[I];
    T4* myPos = s_pos[wid];
    T2* mySe = s_se[wid];
    Real* myFx = s_f[wid][0]; Real* myFy = s_f[wid][1]; Real* myFz = s_f[wid][2];

    for (int t = bt.x; t < bt.x + bt.y; t++) {
        // ---- stage the j-tile (both halves store: entries k and k+32 hold the same atom) ----
        const int jcode = p.tileJ[t * 32 + il];
        const int4 info = p.tileInfo[t];
        const int jidx = jcode & SNB_JIDX_MASK;
        const bool jvalid = jcode >= 0;
        T4 pj; T2 sej;
        if (jvalid) {
            pj = p.posq[jidx]; sej = p.sigeps[jidx];
            if (!WRAP) {
                const int sc = (jcode >> SNB_JSHIFT_BITS) & 31;
                pj.x += p.shifts[sc * 3]; pj.y += p.shifts[sc * 3 + 1]; pj.z += p.shifts[sc * 3 + 2];
            }
        } else {
            pj.x = Real(3e9) + Real(1e6) * il; pj.y = Real(-5e9); pj.z = Real(7e9); pj.w = 0; sej.x = 0; sej.y = 0;
        }
        myPos[lane] = pj; mySe[lane] = sej;
        myFx[lane] = 0; myFy[lane] = 0; myFz[lane] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        const int sj = info.x;
        const int slice = sliceOf(si, sj);
        if (ENERGY && slice != curSlice) {
            if (curSlice >= 0) {
                double a = waveSum((double)ecl), b = waveSum((double)elj);
                if (lane == 0) { atomicAdd(&p.sliceE[2 * curSlice], a); atomicAdd(&p.sliceE[2 * curSlice + 1], b); }
            }
            ecl = 0; elj = 0; curSlice = slice;
        }
        const Real lamC = p.lambdas[2 * slice], lamL = p.lambdas[2 * slice + 1];
        const unsigned maskWord = info.y >= 0 ? p.masks[info.y * 32 + il] : 0u;
        // lambda folded into the i-side parameters once per tile (forces only need the scaled values)
        const Real qiS = qi * lamC;
        const Real epsiS = sei.y * lamL;

#pragma unroll 4
        for (int s = 0; s < 16; s++) {
            const int js = il + h + 2 * s;               // 0..62, duplicate layout => no wrap
            const T4 xj = myPos[js];
            const T2 sj2 = mySe[js];
            Real dx = pi.x - xj.x, dy = pi.y - xj.y, dz = pi.z - xj.z;
            if (WRAP) wrapDelta<Real>(dx, dy, dz, p.box, p.invBoxDiag);
            const Real r2 = dx * dx + dy * dy + dz * dz;
            const Real invR = rsq(r2);
            const Real r = r2 * invR;
            bool include = !((maskWord >> (js & 31)) & 1u);
            if (MC != MC_NOCUTOFF) include = include && (r2 < p.cutoff2);

            // Lennard-Jones (ReferenceSlicedLJCoulombIxn.cpp:390-396, 600-616)
            const Real sig = sei.x + sj2.x;
            Real s2 = sig * invR; s2 *= s2;
            const Real s6 = s2 * s2 * s2;
            Real fLJ, eLJ = 0, fC, eC = 0;
            if (ENERGY) {
                const Real es6 = sei.y * sj2.y * s6;
                fLJ = es6 * (Real(12) * s6 - Real(6));
                eLJ = es6 * (s6 - Real(1));
            } else {
                const Real es6 = epsiS * sj2.y * s6;
                fLJ = es6 * (Real(12) * s6 - Real(6));
            }
            if (MC == MC_LJPME) {
                // multiplicative grid term + potential shifts (:398-426)
                const Real dar2 = p.alphaD * p.alphaD * r2;
                const Real dar4 = dar2 * dar2, dar6 = dar4 * dar2;
                const Real invR2 = invR * invR;
                const Real c6 = c6i * (Real(8) * sj2.x * sj2.x * sj2.x * sj2.y);
                const Real coef = invR2 * invR2 * invR2 * c6;
                const Real expd = fexp(-dar2);
                const Real epre = Real(1) + dar2 + Real(0.5) * dar4;
                const Real dpre = epre + dar6 * Real(1.0 / 6.0);
                const Real fmul = Real(6) * coef * (Real(1) - expd * dpre);
                if (ENERGY) {
                    Real sg2 = sig * sig; const Real sg6 = sg2 * sg2 * sg2 * p.invCut6;
                    eLJ += coef * (Real(1) - expd * epre) + sei.y * sj2.y * (Real(1) - sg6) * sg6 - c6 * p.multShift6;
                    fLJ += fmul;
                } else
                    fLJ += fmul * lamL;
            } else if (MC != MC_NOCUTOFF) {
                if (p.useSwitch && r > p.switchDist) {   // (:380-384, 428-431); wave-uniform flag, per-lane distance
                    const Real tt = (r - p.switchDist) * p.invSwitchWidth;
                    const Real sw = Real(1) + tt * tt * tt * (Real(-10) + tt * (Real(15) - tt * Real(6)));
                    const Real dsw = tt * tt * (Real(-30) + tt * (Real(60) - tt * Real(30))) * p.invSwitchWidth;
                    Real e0 = eLJ;
                    if (!ENERGY) { const Real es6 = epsiS * sj2.y * s6; e0 = es6 * (s6 - Real(1)); }
                    fLJ = fLJ * sw - e0 * dsw * r;
                    eLJ *= sw;
                }
            }
            // Coulomb
            const Real qq = (ENERGY ? qi : qiS) * xj.w;
            if (MC == MC_EWALD || MC == MC_LJPME) {
                const Real ar = p.alpha * r;
                const Real ex = fexp(-ar * ar);
                const Real erfcv = erfcFromExp(ar, ex);
                const Real pref = qq * invR;
                fC = pref * (erfcv + ar * ex * Real(1.1283791670955126));   // 2/sqrt(pi) (:387-388)
                if (ENERGY) eC = pref * erfcv;                                // (:444)
            } else if (MC == MC_RF) {
                fC = qq * (invR - Real(2) * p.krf * r2);                      // (:609)
                if (ENERGY) eC = qq * (invR + p.krf * r2 - p.crf);            // (:619)
            } else {
                fC = qq * invR;                                               // (:611)
                if (ENERGY) eC = fC;
            }
            Real f = ENERGY ? (lamL * fLJ + lamC * fC) : (fLJ + fC);
            f *= invR * invR;
            f = include ? f : Real(0);
            if (ENERGY) { ecl += include ? eC : Real(0); elj += include ? eLJ : Real(0); }
            const Real gx = f * dx, gy = f * dy, gz = f * dz;
            fix += gx; fiy += gy; fiz += gz;
            ldsAdd(&myFx[js], -gx); ldsAdd(&myFy[js], -gy); ldsAdd(&myFz[js], -gz);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- flush j-forces: one global atomic per j-atom and component ----
        if (h == 0 && jvalid) {
            gAdd(&p.fx[jidx], myFx[il] + myFx[il + 32]);
            gAdd(&p.fy[jidx], myFy[il] + myFy[il + 32]);
            gAdd(&p.fz[jidx], myFz[il] + myFz[il + 32]);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // combine the two halves' i-forces and flush
    fix += __shfl_xor(fix, 32, 64); fiy += __shfl_xor(fiy, 32, 64); fiz += __shfl_xor(fiz, 32, 64);
    if (h == 0) {
        gAdd(&p.fx[I * 32 + il], fix); gAdd(&p.fy[I * 32 + il], fiy); gAdd(&p.fz[I * 32 + il], fiz);
    }
    if (ENERGY && curSlice >= 0) {
        double a = waveSum((double)ecl), b = waveSum((double)elj);
        if (lane == 0) { atomicAdd(&p.sliceE[2 * curSlice], a); atomicAdd(&p.sliceE[2 * curSlice + 1], b); }
    }
}

template <typename Real, int MC> static void launchDirectMC(const DirectParams<Real>& p, bool wrap, bool energy, hipStream_t s) {
    const int myItems = p.numWork;
    if (myItems <= 0) return;
    dim3 grid((myItems + 3) / 4), block(256);
    if (wrap) {
        if (energy) hipLaunchKernelGGL((k_direct<Real, MC, true, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((k_direct<Real, MC, true, false>), grid, block, 0, s, p);
    } else {
        if (energy) hipLaunchKernelGGL((k_direct<Real, MC, false, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((k_direct<Real, MC, false, false>), grid, block, 0, s, p);
    }
}

template <typename Real> void launchDirect(const DirectParams<Real>& p, int mc, bool wrap, bool energy, hipStream_t s) {
    switch (mc) {
        case MC_NOCUTOFF: launchDirectMC<Real, MC_NOCUTOFF>(p, wrap, energy, s); break;
        case MC_RF: launchDirectMC<Real, MC_RF>(p, wrap, energy, s); break;
        case MC_EWALD: launchDirectMC<Real, MC_EWALD>(p, wrap, energy, s); break;
        default: launchDirectMC<Real, MC_LJPME>(p, wrap, energy, s); break;
    }
}
template void launchDirect<float>(const DirectParams<float>&, int, bool, bool, hipStream_t);
template void launchDirect<double>(const DirectParams<double>&, int, bool, bool, hipStream_t);

// ---- 1-4 exceptions: ReferenceSlicedLJCoulomb14.cpp:61-95 ----------------------------------------
template <typename Real, bool ENERGY> __global__ void k_exceptions(const PairListParams<Real> p) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    double e0 = 0, e1 = 0; int slice = 0;
    if (k < p.n) {
        const int2 ij = p.pairs[k];
        const auto par = p.params[k];
        const auto xi = p.posq[ij.x]; const auto xj = p.posq[ij.y];
        Real dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
        if (p.periodic) { Real inv[3] = {Real(1) / p.box[0], Real(1) / p.box[4], Real(1) / p.box[8]}; wrapDelta<Real>(dx, dy, dz, p.box, inv); }
        const Real invR = rsq(dx * dx + dy * dy + dz * dz);
        Real s2 = invR * par.x; s2 *= s2;
        const Real s6 = s2 * s2 * s2;
        slice = (int)par.w;
        const Real lamC = p.lambdas[2 * slice], lamL = p.lambdas[2 * slice + 1];
        Real dEdR = lamL * par.y * (Real(12) * s6 - Real(6)) * s6 + lamC * par.z * invR;
        dEdR *= invR * invR;
        gAdd(&p.fx[ij.x], dEdR * dx); gAdd(&p.fy[ij.x], dEdR * dy); gAdd(&p.fz[ij.x], dEdR * dz);
        gAdd(&p.fx[ij.y], -dEdR * dx); gAdd(&p.fy[ij.y], -dEdR * dy); gAdd(&p.fz[ij.y], -dEdR * dz);
        if (ENERGY) { e0 = par.z * invR; e1 = par.y * (s6 - Real(1)) * s6; }
    }
    if (ENERGY && k < p.n) { atomicAdd(&p.sliceE[2 * slice], e0); atomicAdd(&p.sliceE[2 * slice + 1], e1); }
}

// ---- Ewald exclusion corrections: ReferenceSlicedLJCoulombIxn.cpp:449-506 -------------------------
template <typename Real, bool ENERGY> __global__ void k_exclusionCorrection(const PairListParams<Real> p) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= p.n) return;
    const int2 ij = p.pairs[k];
    const auto par = p.params[k];   // x = k*qi*qj, y = c6i*c6j, w = slice
    const auto xi = p.posq[ij.x]; const auto xj = p.posq[ij.y];
    Real dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
    if (p.periodic) { Real inv[3] = {Real(1) / p.box[0], Real(1) / p.box[4], Real(1) / p.box[8]}; wrapDelta<Real>(dx, dy, dz, p.box, inv); }
    const Real r2 = dx * dx + dy * dy + dz * dz;
    const Real r = sqrt(r2);
    const Real invR = Real(1) / r;
    const Real ar = p.alpha * r;
    const int slice = (int)par.w;
    const Real lamC = p.lambdas[2 * slice], lamL = p.lambdas[2 * slice + 1];
    Real f = 0; double e0 = 0, e1 = 0;
    // the erf evaluation itself is done in double even in the single-precision engine: excluded pairs are few
    // (O(N)) and erf(x) for small x is cancellation-prone in float.
    const double erfv = erf((double)ar);
    if (erfv > 1e-6) {
        const Real ex = fexp(-ar * ar);
        f = -lamC * par.x * invR * invR * invR * (Real(erfv) - ar * ex * Real(1.1283791670955126));
        if (ENERGY) e0 = -(double)par.x * (double)invR * erfv;
    } else if (ENERGY)
        e0 = -(double)p.alpha * 1.1283791670955126 * (double)par.x;
    if (p.ljpme) {
        const Real dar2 = p.alphaD * p.alphaD * r2, dar4 = dar2 * dar2, dar6 = dar4 * dar2;
        const Real invR2 = invR * invR;
        const Real expd = fexp(-dar2);
        const Real coef = par.y * invR2 * invR2 * invR2;
        if (ENERGY) e1 = coef * (Real(1) - expd * (Real(1) + dar2 + Real(0.5) * dar4));
        // reference: dEdR = -6 c6 r^-8 (...), forces[ii] -= lam*dEdR*delta  => f (applied as +f*delta on ii) = +6...
        f += lamL * Real(6) * coef * invR2 * (Real(1) - expd * (Real(1) + dar2 + Real(0.5) * dar4 + dar6 * Real(1.0 / 6.0)));
    }
    if (f != Real(0)) {
        gAdd(&p.fx[ij.x], f * dx); gAdd(&p.fy[ij.x], f * dy); gAdd(&p.fz[ij.x], f * dz);
        gAdd(&p.fx[ij.y], -f * dx); gAdd(&p.fy[ij.y], -f * dy); gAdd(&p.fz[ij.y], -f * dz);
    }
    if (ENERGY) { atomicAdd(&p.sliceE[2 * slice], e0); if (p.ljpme) atomicAdd(&p.sliceE[2 * slice + 1], e1); }
}

template <typename Real> void launchExceptions(const PairListParams<Real>& p, bool energy, hipStream_t s) {
    if (p.n <= 0) return;
    dim3 grid((p.n + 255) / 256), block(256);
    if (energy) hipLaunchKernelGGL((k_exceptions<Real, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((k_exceptions<Real, false>), grid, block, 0, s, p);
}
template <typename Real> void launchExclusionCorrection(const PairListParams<Real>& p, bool energy, hipStream_t s) {
    if (p.n <= 0) return;
    dim3 grid((p.n + 255) / 256), block(256);
    if (energy) hipLaunchKernelGGL((k_exclusionCorrection<Real, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((k_exclusionCorrection<Real, false>), grid, block, 0, s, p);
}
template void launchExceptions<float>(const PairListParams<float>&, bool, hipStream_t);
template void launchExceptions<double>(const PairListParams<double>&, bool, hipStream_t);
template void launchExclusionCorrection<float>(const PairListParams<float>&, bool, hipStream_t);
template void launchExclusionCorrection<double>(const PairListParams<double>&, bool, hipStream_t);

}  // namespace snb
