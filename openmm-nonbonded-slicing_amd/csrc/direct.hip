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
#include <hip/hip_ext.h>
#include <algorithm>
#include <map>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#ifndef SNB_DIRECT_F64_WAVES
#define SNB_DIRECT_F64_WAVES 2      // waves per SIMD the double-precision pair kernel is compiled for (256 VGPRs: no spills; measured on c5)
#endif
namespace snb {

// ---- math helpers -------------------------------------------------------------------------------
__device__ inline float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
// double: the hardware estimate (v_rsq_f64, ~2^-27) refined by two Newton steps -- about a third of the instructions of 1.0 / sqrt(x),
// which runs its own refinements for the square root and again for the division; relative error < 1e-15 (tests hold 1e-12 on forces)
__device__ inline double rsq(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
}
__device__ inline float fexp(float x) { return __expf(x); }
__device__ inline double fexp(double x) { return exp(x); }
// erfc(ar) given e = exp(-ar^2).  Single precision: Abramowitz & Stegun 7.1.26 (max abs error 1.5e-7), the
// same approximation the reference GPU path uses (coulombLennardJones.cc:18-23); double: libm.
__device__ inline float erfcFromExp(float ar, float e) {
    float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ar);
    return (0.254829592f + (-0.284496736f + (1.421413741f + (-1.453152027f + 1.061405429f * t) * t) * t) * t) * t * e;
}
__device__ inline double erfcFromExp(double ar, double) { return erfc(ar); }
// exp(-alpha^2 r^2): single precision folds log2(e) into the constant and issues one v_exp_f32
__device__ inline float expNegAlpha2R2(float a2l2e, float, float r2) { return __builtin_amdgcn_exp2f(-a2l2e * r2); }
__device__ inline double expNegAlpha2R2(double, double alpha, double r2) { return exp(-alpha * alpha * r2); }
__device__ inline double erfOf(float ar, float e) { return (double)(1.0f - erfcFromExp(ar, e)); }
__device__ inline double erfOf(double ar, double) { return erf(ar); }

__device__ inline void ldsAdd(float* p, float v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void ldsAdd(double* p, double v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void gAdd(float* p, float v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void gAdd(double* p, double v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Force accumulation of atom slot `a`.  SNB_MIXED engines (P::fixed) keep the direct-space accumulators as 64-bit fixed point, 2^32 per
// kJ/mol/nm, the way every GPU platform of the reference does (realToFixedPoint; CommonNonbondedSlicingKernels.cpp adds into
// getLongForceBuffer, pme.cc:381-389): integer sums do not depend on the order in which waves arrive, so the force of a step is
// reproducible bit for bit.  The component arrays are then arrays of 64-bit words behind the same base pointers.
__device__ inline unsigned long long toFixedForce(float v) { return (unsigned long long)(long long)(v * 4294967296.0f); }
template <typename P> __device__ inline void fAdd(const P& p, float* comp, size_t a, float v) {
    if (p.fixed) { __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(comp) + a * p.fs, toFixedForce(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    gAdd(comp + a * p.fs, v);
}
template <typename P> __device__ inline void fAdd(const P& p, double* comp, size_t a, double v) { gAdd(comp + a * p.fs, v); }
// (the packed tile kernel takes the choice as a template parameter: the run-time test cost the default path 2.7 % on c3)
template <bool FIXED, typename P> __device__ inline void fAddT(const P& p, float* comp, size_t a, float v) {
    if constexpr (FIXED) __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(comp) + a * p.fs, toFixedForce(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else gAdd(comp + a * p.fs, v);
}

__device__ inline double waveSum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Physical CU of the calling wave: XCC_ID[2:0] (hardware register 20) and the SE_ID | SH_ID | CU_ID bits 15:8 of HW_ID (register 4).
// Used only to count co-resident work-groups of one launch (SNB_CU_SLOTS entries).
__device__ inline int physicalCu() {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);      // size-1 = 3, offset 0
    const unsigned cu = __builtin_amdgcn_s_getreg((7 << 11) | (8 << 6) | 4);        // size-1 = 7, offset 8
    return (int)(((xcc & 7u) << 8) | (cu & 255u));
}

// CU-limited launch of an overlapped step (engine.hip, overlapMode): may this work-group stay on its CU?  Called by every thread (one barrier).
// The reciprocal pipeline's work-groups need what the limit leaves free IN ONE PIECE: registers are allocated as contiguous ranges per
// wave, and two resident pair work-groups picked by order of arrival sit at random two of the four positions the dispatcher filled,
// which leaves the plane kernel's 4 x 64 registers per SIMD without room on a third of the CUs (measured: it then waits for the pair
// kernel to end).  So the rule is positional: a work-group stays when each of its four waves was allocated within the lowest
// cuBaseMax + its own size registers of its SIMD (VGPR_BASE, bits 5:0 of GPR_ALLOC, in units of 8 registers) -- on an idle CU exactly the
// first cuLimit work-groups the dispatcher placed -- and everything above them stays free and contiguous.  cuBaseMax < 0: the arrival-count rule.
// cuSlots[key]: arrivals in the low half, work-groups that stayed in the high half (statistics, SNB_OVERLAP_DEBUG).
template <typename P> __device__ inline bool cuResident(const P& p) {
    __shared__ int s_base[4];
    __shared__ int s_stay;
    if ((threadIdx.x & 63) == 0) s_base[threadIdx.x >> 6] = (int)__builtin_amdgcn_s_getreg((5 << 11) | (0 << 6) | 5);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int maxBase = max(max(s_base[0], s_base[1]), max(s_base[2], s_base[3]));
        const int key = physicalCu();
        int stay, old;
        if (p.cuBaseMax >= 0) { stay = maxBase <= p.cuBaseMax ? 1 : 0; old = atomicAdd(&p.cuSlots[key], stay ? 0x10001 : 1) & 0xFFFF; }
        else { old = atomicAdd(&p.cuSlots[key], 1) & 0xFFFF; stay = old < p.cuLimit ? 1 : 0; if (stay) atomicAdd(&p.cuSlots[key], 0x10000); }
        s_stay = stay;
        if (p.cuTrace && old < 4) { p.cuTrace[(key * 4 + old) * 2] = (int)__builtin_amdgcn_s_getreg((31 << 11) | 5) ^ (stay << 31); p.cuTrace[(key * 4 + old) * 2 + 1] = (int)__builtin_amdgcn_s_getreg((31 << 11) | 6); }      // GPR_ALLOC (top bit: stayed), LDS_ALLOC
    }
    __syncthreads();
    return s_stay != 0;
}

// Work items of an overlapped step are handed out through SNB_WORK_SHARDS counters, each on a 128-byte line of its own; counter k owns items
// k, k + SNB_WORK_SHARDS, ... (the list is sorted longest first, so every counter's sequence is too).  One counter for all waves serialises
// at ~15 ns per claim (same-address device-scope atomics: 28 768 claims = 0.44 ms, measured) -- sixteen are far below that rate.
// A wave starts at a home counter and, once that one is exhausted, reads all counters in one load and moves to the next one with items left.
struct WorkClaim {
    int* ctr; int numWork, shard, lane;
    __device__ int issue() const { int v = 0; if (lane == 0) v = atomicAdd(&ctr[shard * 32], 1); return v; }      // lane 0 holds the claim
    __device__ bool anyLeft() const {      // (one load of all counters: a launch that arrives when everything is claimed leaves after one round trip)
        int left = 0;
        if (lane < SNB_WORK_SHARDS) left = (numWork - lane + SNB_WORK_SHARDS - 1) / SNB_WORK_SHARDS - __hip_atomic_load(&ctr[lane * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return __ballot(left > 0) != 0;
    }
    __device__ int resolve(int v) {      // item index of a claim issued on `shard`, moving on to other counters when that one has run out; numWork: no items left
        int idx = shard + SNB_WORK_SHARDS * __builtin_amdgcn_readfirstlane(v);
        while (idx >= numWork) {
            int left = 0;
            if (lane < SNB_WORK_SHARDS) left = (numWork - lane + SNB_WORK_SHARDS - 1) / SNB_WORK_SHARDS - __hip_atomic_load(&ctr[lane * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned m = (unsigned)__ballot(left > 0);
            if (m == 0) return numWork;
            const int from = (shard + 1) & (SNB_WORK_SHARDS - 1);
            const unsigned rot = ((m | (m << SNB_WORK_SHARDS)) >> from) & ((1u << SNB_WORK_SHARDS) - 1);
            shard = (from + __builtin_ctz(rot)) & (SNB_WORK_SHARDS - 1);
            idx = shard + SNB_WORK_SHARDS * __builtin_amdgcn_readfirstlane(issue());
        }
        return idx;
    }
};

__device__ inline int sliceOf(int a, int b) { return a > b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a; }

template <typename Real> __device__ inline void wrapDelta(Real& dx, Real& dy, Real& dz, const Real* box, const Real* inv) {
    // OpenMM ReferenceForce::getDeltaRPeriodic (triclinic form)
    Real s = floor(dz * inv[2] + Real(0.5)); dx -= s * box[6]; dy -= s * box[7]; dz -= s * box[8];
    s = floor(dy * inv[1] + Real(0.5)); dx -= s * box[3]; dy -= s * box[4];
    s = floor(dx * inv[0] + Real(0.5)); dx -= s * box[0];
}

// The sorted coordinates are box-wrapped per atom (imageOffset = wrapped - user).  Non-periodic exceptions
// (periodicExceptions == false, ReferenceSlicedLJCoulombIxn.cpp:461-464) need the user's own coordinates back.
template <typename Real> __device__ inline void unwrapDelta(Real& dx, Real& dy, Real& dz, const Real* off, int i, int j) {
    dx -= off[3 * i] - off[3 * j]; dy -= off[3 * i + 1] - off[3 * j + 1]; dz -= off[3 * i + 2] - off[3 * j + 2];
}

// ---- cross-lane helpers -------------------------------------------------------------------------
// DPP row_ror:1 -- every 16-lane row rotates by one lane (lane c receives the value of lane (c-1)&15).
__device__ inline float rowRor1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, true)); }
__device__ inline double rowRor1(double v) {
    long long b = __builtin_bit_cast(long long, v);
    int lo = (int)b, hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x121, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x121, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}

// The 16 pair steps of one tile for this lane's row.  MASKED: the tile carries an exclusion/padding mask;
// SWITCH: LJ switching function active.  Both are wave-uniform and resolved at compile time so that the common
// case (no mask, no switch) carries no test for them in the loop.
template <typename Real, int MC, bool WRAP, bool ENERGY, bool MASKED, bool SWITCH>
__device__ __forceinline__ void tileSteps(const DirectParams<Real>& p, const typename Vec<Real>::T4* rdPos, const typename Vec<Real>::T2* rdSe,
                                          const typename Vec<Real>::T4 pi, const typename Vec<Real>::T2 sei, const Real qi, const Real qiS, const Real epsiS,
                                          const Real c6i, const Real lamC, const Real lamL, const unsigned maskWord, const int c,
                                          Real& fix, Real& fiy, Real& fiz, Real& fjx, Real& fjy, Real& fjz, Real& ecl, Real& elj) {
    using T4 = typename Vec<Real>::T4;
    using T2 = typename Vec<Real>::T2;
#pragma unroll 4
        for (int s = 0; s < 16; s++) {
            const T4 xj = rdPos[-s];
            const T2 sj2 = rdSe[-s];
            Real dx = pi.x - xj.x, dy = pi.y - xj.y, dz = pi.z - xj.z;
            if (WRAP) wrapDelta<Real>(dx, dy, dz, p.box, p.invBoxDiag);
            const Real r2 = dx * dx + dy * dy + dz * dz;
            const Real invR = rsq(r2);
            const Real r = r2 * invR;
            bool include = MC == MC_NOCUTOFF ? true : (r2 < p.cutoff2);
            if (MASKED) include = include && !((maskWord >> ((c - s) & 15)) & 1u);

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
            if (MC == MC_LJPME && !ENERGY && std::is_same<Real, double>::value) {
                // forces only, double: 6 c6 [1 - e^{-x}(1 + x + x^2/2 + x^3/6)] / r^6 = 6 c6 r^2 Gd(r^2), x = (alpha_d r)^2, with
                // Gd(r^2) = alpha_d^8 e^{-x} sum_k x^k / (k+4)!  an entire function of r^2: a degree-17 polynomial in the same t as the
                // Ewald factor (~1e-12 of Gd(0)) instead of a double-precision exp and its pre-factors (engine.hip buildEwaldPoly)
                const Real t = r2 * p.ewScale - Real(1);
                Real gd = p.dispPoly[SNB_DISP_DEG_F64];
#pragma unroll
                for (int k = SNB_DISP_DEG_F64 - 1; k >= 0; k--) gd = gd * t + p.dispPoly[k];
                const Real c6 = c6i * (Real(8) * sj2.x * sj2.x * sj2.x * sj2.y);
                fLJ += Real(6) * c6 * gd * r2 * lamL;
            } else if (MC == MC_LJPME) {
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
                if (SWITCH && r > p.switchDist) {        // (:380-384, 428-431)
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
            if ((MC == MC_EWALD || MC == MC_LJPME) && !ENERGY && std::is_same<Real, double>::value) {
                // forces only: [erfc(ar)/r + 2a/sqrt(pi) e^{-(ar)^2}] = 1/r - r^2 Bt(r^2), Bt a degree-16 polynomial (~6e-12 of Bt(0)): no libm erfc / exp
                const Real t = r2 * p.ewScale - Real(1);
                Real bt = p.ewPoly[SNB_EW_DEG_F64];
#pragma unroll
                for (int k = SNB_EW_DEG_F64 - 1; k >= 0; k--) bt = bt * t + p.ewPoly[k];
                fC = qq * (invR - r2 * bt);
            } else if (MC == MC_EWALD || MC == MC_LJPME) {
                const Real ar = p.alpha * r;
                const Real ex = expNegAlpha2R2(p.alpha2l2e, p.alpha, r2);
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
            // the accumulator follows its j-slot along the row: rotate in (slot of lane c at step s = slot of lane c-1 at step s-1), then add
            fjx = rowRor1(fjx) - gx; fjy = rowRor1(fjy) - gy; fjz = rowRor1(fjz) - gz;
        }
}

// ---- packed single-precision variant of the 16 pair steps ---------------------------------------------------------
// On gfx950 v_pk_mul_f32 / v_pk_add_f32 issue at the cost of their scalar forms (measured, tools/ubench_valu.hip), so a
// lane that works on TWO j-slots at once halves the issue cost of every plain multiply/add (FMAs and transcendentals
// cost the same either way).  8 steps; at step s lane c meets slots (c-2s) [component .x] and (c-2s-1) [component .y];
// each component's j-force accumulator follows its slot with a two-lane DPP rotation per step.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ inline float rowRor8(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true)); }

// ---- the tile kernel ----------------------------------------------------------------------------
// Work item = (i-block, run of <= 8 tiles).  Lane layout: 4 DPP rows of 16 lanes; row r works on the 16x16 sub-tile
// (i-half r&1, j-half r>>1).  At step s lane c of a row meets j-slot (c-s)&15 of its j-half; the j-force accumulator
// travels WITH the j-slot by a one-lane DPP row rotation per step (v_add_f32_dpp: one VALU op per component, no LDS),
// so after 16 steps lane c holds the force on j-slot c.  j-atom data is read from LDS (staged once per tile, each
// 16-atom half stored twice so the rotated index c+16-s needs no wrap).
template <typename Real, bool ENERGY> __device__ __forceinline__ void exceptionsBody(const PairListParams<Real>& p, const int blk);
template <typename Real, bool ENERGY> __device__ __forceinline__ void exclusionAtomsBody(const PairListParams<Real>& p, const int blk);
// (as in k_directPacked: the first nListBlocks work-groups of the launch run the O(N) pair lists -- exclusion corrections, then 1-4
// exceptions -- so that their latency-bound work overlaps the tile work instead of trailing it as a 65 us launch of its own on c5)
template <typename Real, int MC, bool WRAP, bool ENERGY>
__global__ __launch_bounds__(256, (sizeof(Real) == 8 ? SNB_DIRECT_F64_WAVES : 4)) void k_direct(const DirectParams<Real> p, const PairListParams<Real> q, const int nExclBlocks, const int nListBlocks) {
    // (a CU-limited launch takes its list blocks LAST: its tile work-groups must be the first thing the dispatcher places on the idle CUs)
    const int listBlock = p.listsLast ? (int)blockIdx.x - ((int)gridDim.x - nListBlocks) : (int)blockIdx.x;
    if (listBlock >= 0 && listBlock < nListBlocks) {      // (energy steps: the list bodies reduce their slice energies in 2 S doubles of dynamic LDS)
        if (listBlock < nExclBlocks) { PairListParams<Real> qe = q; qe.n = q.nExclAtoms; exclusionAtomsBody<Real, ENERGY>(qe, listBlock); }
        else exceptionsBody<Real, ENERGY>(q, listBlock - nExclBlocks);
        return;
    }
    const int tileBlock = p.listsLast ? (int)blockIdx.x : (int)blockIdx.x - nListBlocks, nTileBlocks = gridDim.x - nListBlocks;
    if (p.stepTrace && tileBlock == 0 && threadIdx.x == 0) p.stepTrace[p.traceSlot] = (long long)wall_clock64();
    double* const sliceE = SNB_SLICE_E_PARTITION(p.sliceE, p.nsub * (p.nsub + 1));
    using T4 = typename Vec<Real>::T4;
    using T2 = typename Vec<Real>::T2;
    __shared__ T4 s_pos[4][64];
    __shared__ T2 s_se[4][64];
    // (overlapped steps: CU-limited first launch and items through the device counters, as in k_directPacked)
    if (p.cuSlots != nullptr && p.cuLimit > 0 && !cuResident(p)) return;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool dyn = p.workCounter != nullptr;
    WorkClaim wc{p.workCounter, p.numWork, (tileBlock * 4 + wid) & (SNB_WORK_SHARDS - 1), lane};
    int item = tileBlock * 4 + wid;
    if (dyn) item = wc.anyLeft() ? wc.resolve(wc.issue()) : p.numWork;
    for (; item < p.numWork; ) {   // no block-level barrier inside the loop
    int claimed = 0;
    if (dyn) claimed = wc.issue();
    const int4 wi = p.workItems[p.workStart + item * p.workStride];
    const int I = __builtin_amdgcn_readfirstlane(wi.x);
    const int tBegin = __builtin_amdgcn_readfirstlane(wi.y), tEnd = tBegin + __builtin_amdgcn_readfirstlane(wi.z);
    const int c = lane & 15, row = lane >> 4;
    const int il = 16 * (row & 1) + c;                   // my i-atom within the block
    const int jh = row >> 1;                             // my j-half
    const int stageJ = 16 * (lane >> 5) + c;             // j-atom this lane stages (entry `lane` of the doubled image)

    const T4 pi = p.posq[I * 32 + il];
    const T2 sei = p.sigeps[I * 32 + il];
    const Real qi = pi.w * p.k4pe;
    Real c6i = 0;
    if (MC == MC_LJPME) c6i = Real(8) * sei.x * sei.x * sei.x * sei.y;
    Real fix = 0, fiy = 0, fiz = 0;
    Real ecl = 0, elj = 0;
    int curSlice = -1;

    T4* myPos = s_pos[wid];
    T2* mySe = s_se[wid];
    const T4* rdPos = myPos + 32 * jh + c + 16;          // entry for step s is rdPos[-s]
    const T2* rdSe = mySe + 32 * jh + c + 16;

    // software pipeline, two tiles deep: while tile t is computed the j-atoms, mask word and lambdas of tile t+1 are in flight
    // (their addresses come from tileJ/tileInfo words loaded one tile earlier) and the tileJ/tileInfo words of tile t+2 are
    // requested, so no load at the head of a tile waits on another load
    T4 pj; T2 sej;
    auto fetch = [&](int code, T4& x, T2& se) {
        if (code != -1) {   // -1 = padding slot (image codes use bits 27..31, so the sign bit is NOT a validity flag)
            const int idx = code & SNB_JIDX_MASK;
            x = p.posq[idx]; se = p.sigeps[idx];
            if (!WRAP) {
                const int sc = (code >> SNB_JSHIFT_BITS) & 127;
                const int kx = sc / 25, ky = (sc - 25 * kx) / 5, kz = sc - 25 * kx - 5 * ky;
                const Real ka = Real(kx - 2), kb = Real(ky - 2), kc = Real(kz - 2);      // lattice image: + ka a + kb b + kc c (rows of p.box)
                x.x += ka * p.box[0] + kb * p.box[3] + kc * p.box[6]; x.y += kb * p.box[4] + kc * p.box[7]; x.z += kc * p.box[8];
            }
        } else { x.x = Real(3e9) + Real(1e6) * c; x.y = Real(-5e9); x.z = Real(7e9); x.w = 0; se.x = 0; se.y = 0; }
    };
    struct TileHead { int slice, maskIdx; };
    // energy steps: a tile whose slice is not wanted (p.sliceNeed) runs the forces-only arithmetic
    auto loadHead = [&](int t) { const int4 v = p.tileInfo[t]; return TileHead{v.x & 0xFFFF, v.y}; };      // (slice of the tile, mask index)
    auto loadMask = [&](const TileHead& h) { return (h.maskIdx >= 0) ? p.masks[h.maskIdx * 32 + il] : 0u; };
    int jcode = p.tileJ[tBegin * 32 + stageJ];
    TileHead head = loadHead(tBegin);
    int jcodeNext = -1; TileHead headNext = head;
    if (tBegin + 1 < tEnd) { jcodeNext = p.tileJ[(tBegin + 1) * 32 + stageJ]; headNext = loadHead(tBegin + 1); }
    fetch(jcode, pj, sej);
    unsigned maskPre = loadMask(head);
    int slicePre = head.slice;
    Real lamCPre = p.lambdas[2 * slicePre], lamLPre = p.lambdas[2 * slicePre + 1];

    for (int t = tBegin; t < tEnd; t++) {
        __builtin_amdgcn_wave_barrier();
        myPos[lane] = pj; mySe[lane] = sej;
        const int curCode = jcode;
        const bool hasMask = head.maskIdx >= 0;
        unsigned maskWord = maskPre;
        const int slice = slicePre;
        const Real lamC = lamCPre, lamL = lamLPre;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        jcode = jcodeNext; head = headNext;
        if (t + 1 < tEnd) {
            fetch(jcode, pj, sej);
            maskPre = loadMask(head);
            slicePre = head.slice;
            lamCPre = p.lambdas[2 * slicePre]; lamLPre = p.lambdas[2 * slicePre + 1];
        }
        if (t + 2 < tEnd) { jcodeNext = p.tileJ[(t + 2) * 32 + stageJ]; headNext = loadHead(t + 2); }

        if (ENERGY && slice != curSlice) {
            if (curSlice >= 0) {
                double a = waveSum((double)ecl), b = waveSum((double)elj);
                if (lane == 0) { atomicAdd(&sliceE[2 * curSlice], a); atomicAdd(&sliceE[2 * curSlice + 1], b); }
            }
            ecl = 0; elj = 0; curSlice = slice;
        }
        maskWord >>= 16 * jh;                            // my j-half's 16 bits
        // lambda folded into the i-side parameters once per tile (forces only need the scaled values)
        const Real qiS = qi * lamC;
        const Real epsiS = sei.y * lamL;
        Real fjx = 0, fjy = 0, fjz = 0;

        const bool tileE = ENERGY && p.sliceNeed[slice] != 0;      // (uniform)
#define SNB_TILE_STEPS(E, M, SW) tileSteps<Real, MC, WRAP, E, M, SW>(p, rdPos, rdSe, pi, sei, qi, qiS, epsiS, c6i, lamC, lamL, maskWord, c, fix, fiy, fiz, fjx, fjy, fjz, ecl, elj)
        if (p.useSwitch && MC != MC_LJPME && MC != MC_NOCUTOFF) {
            if (tileE) { if (hasMask) SNB_TILE_STEPS(ENERGY, true, true); else SNB_TILE_STEPS(ENERGY, false, true); }
            else { if (hasMask) SNB_TILE_STEPS(false, true, true); else SNB_TILE_STEPS(false, false, true); }
        } else {
            if (tileE) { if (hasMask) SNB_TILE_STEPS(ENERGY, true, false); else SNB_TILE_STEPS(ENERGY, false, false); }
            else { if (hasMask) SNB_TILE_STEPS(false, true, false); else SNB_TILE_STEPS(false, false, false); }
        }
#undef SNB_TILE_STEPS
            // rotate-then-add leaves lane c holding slot c+1: one more rotation brings every slot home
        fjx = rowRor1(fjx); fjy = rowRor1(fjy); fjz = rowRor1(fjz);
        // add the two i-halves (rows r and r^1) and flush
        fjx += __shfl_xor(fjx, 16, 64); fjy += __shfl_xor(fjy, 16, 64); fjz += __shfl_xor(fjz, 16, 64);
        if ((row & 1) == 0 && curCode != -1) {            // lanes 0-15 (j-half 0) and 32-47 (j-half 1): entry `lane` == j-slot
            const int jidx = curCode & SNB_JIDX_MASK;
            fAdd(p, p.fx, jidx, fjx); fAdd(p, p.fy, jidx, fjy); fAdd(p, p.fz, jidx, fjz);
        }
    }
    // rows r and r^2 hold the same i-atoms (different j-halves)
    fix += __shfl_xor(fix, 32, 64); fiy += __shfl_xor(fiy, 32, 64); fiz += __shfl_xor(fiz, 32, 64);
    if (row < 2) { fAdd(p, p.fx, (I * 32 + il), fix); fAdd(p, p.fy, (I * 32 + il), fiy); fAdd(p, p.fz, (I * 32 + il), fiz); }
    if (ENERGY && curSlice >= 0) {
        double a = waveSum((double)ecl), b = waveSum((double)elj);
        if (lane == 0) { atomicAdd(&sliceE[2 * curSlice], a); atomicAdd(&sliceE[2 * curSlice + 1], b); }
    }
    __builtin_amdgcn_wave_barrier();
    item = dyn ? wc.resolve(claimed) : item + nTileBlocks * 4;
    }   // work-item loop
    if (p.stepTrace && threadIdx.x == 0) p.stepTrace[p.traceSlot + 1] = (long long)wall_clock64();
}

// ---- single-precision forces-only tile kernel with packed math ---------------------------------------
// Same work items, tiles and masks as k_direct, different lane layout: lane (row r, column c) holds TWO i-atoms (c and c+16)
// as the halves of float2 registers and meets ONE j-slot per step, slot 8r + ((c - s) & 7) of j-quarter r, for 8 steps.
// All pair arithmetic issues as v_pk_*_f32 on (i_c, j) and (i_c+16, j): the j-side operands are broadcast by op_sel (no register
// shuffling), the i-side operands are packed once per work item, and one LDS read (b128 + b64) feeds two pair slots.  The
// j-force accumulator travels with its slot by a one-lane row rotation fused into the subtract (v_sub_f32_dpp); lanes c and
// c+8 carry two partial sums of the same slot and are merged by an 8-lane rotation at the end of the tile.
// Staging: entry e (0..15) of quarter r holds atom 8r + (e & 7), so the rotated read index (c & 7) + 8 - s needs no wrap.
// ENERGY: the raw (unscaled) pair energies of the tile's slice are accumulated as well -- eLJ = eps_ij s6 (s6 - 1) and the Coulomb
// energy of the method (Ewald: qq erfc(ar)/r with the A&S erfc) -- from the
// RAW i-parameters qiRaw / epsiRaw, while the forces keep using the lambda-scaled ones.  This is the kernel of every step of a force
// that asks for energy-parameter derivatives (the reference accumulates them whether or not the energy is requested, Q4).
template <int MC, bool MASKED, bool POLY, bool ENERGY, bool SWITCH>
__device__ __forceinline__ void tileStepsPacked(const DirectParams<float>& p, const float4* rdPos, const float2* rdSe, const v2f pix, const v2f piy, const v2f piz,
                                                const v2f sigi, const v2f qiS, const v2f epsiS, const v2f qiRaw, const v2f epsiRaw, const v2f c6iRaw, const float lamL, const unsigned maskA, const unsigned maskB, const int c,
                                                v2f& fix, v2f& fiy, v2f& fiz, float& fjx, float& fjy, float& fjz, v2f& ecl, v2f& elj) {
#pragma unroll 4
    for (int s = 0; s < 8; s++) {
        const float4 xj = rdPos[-s];
        const float2 sj = rdSe[-s];
        const v2f dx = pix - xj.x, dy = piy - xj.y, dz = piz - xj.z;
        const v2f r2 = dx * dx + dy * dy + dz * dz;
        const v2f invR = {__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
        // Lennard-Jones (sigeps holds sigma/2 and 2 sqrt(eps))
        v2f s2 = (sigi + sj.x) * invR; s2 = s2 * s2;
        const v2f s6 = s2 * s2 * s2;
        const v2f es6 = (epsiS * sj.y) * s6;
        v2f f = es6 * (s6 * 12.0f - 6.0f);
        v2f eLJ = {0.f, 0.f}, eC = {0.f, 0.f};
        if (ENERGY) eLJ = ((epsiRaw * sj.y) * s6) * (s6 - 1.0f);
        if (SWITCH) {
            // LJ switching function (ReferenceSlicedLJCoulombIxn.cpp:380-384, 428-431): branch-free, tt clamps to 0 below the switching distance
            const v2f r = r2 * invR;
            v2f tt = (r - p.switchDist) * p.invSwitchWidth;
            tt.x = tt.x > 0.0f ? tt.x : 0.0f; tt.y = tt.y > 0.0f ? tt.y : 0.0f;
            const v2f t2 = tt * tt;
            const v2f sw = (t2 * tt) * ((tt * -6.0f + 15.0f) * tt + -10.0f) + 1.0f;
            const v2f dsw = t2 * ((tt * -30.0f + 60.0f) * tt + -30.0f) * p.invSwitchWidth;
            f = f * sw - (es6 * (s6 - 1.0f)) * (dsw * r);          // the lambda-scaled pair energy drives the force term
            if (ENERGY) eLJ = eLJ * sw;
        }
        const v2f invR2 = invR * invR;
        if (MC == MC_LJPME) {
            // LJPME: the mesh carries the geometric-mean dispersion; inside the cutoff the real-space term corrects to Lorentz-Berthelot
            // (multiplicative grid term + potential shifts, ReferenceSlicedLJCoulombIxn.cpp:398-426)
            const v2f dar2 = r2 * (p.alphaD * p.alphaD), dar4 = dar2 * dar2;
            const float sj3 = sj.x * sj.x * sj.x;
            const v2f c6 = c6iRaw * (8.0f * sj3 * sj.y);
            const v2f coef = (invR2 * invR2 * invR2) * c6;
            const v2f ed = dar2 * -1.4426950408889634f;
            const v2f expd = {__builtin_amdgcn_exp2f(ed.x), __builtin_amdgcn_exp2f(ed.y)};
            const v2f epre = dar4 * 0.5f + dar2 + 1.0f;
            const v2f dpre = (dar4 * dar2) * (1.0f / 6.0f) + epre;
            const v2f fmul = (coef * 6.0f) * (1.0f - expd * dpre);
            f = f + fmul * lamL;
            if (ENERGY) {
                const v2f sg = sigi + sj.x;
                v2f sg2 = sg * sg; const v2f sg6 = (sg2 * sg2 * sg2) * p.invCut6;
                eLJ = eLJ + coef * (1.0f - expd * epre) + (epsiRaw * sj.y) * ((1.0f - sg6) * sg6) - c6 * p.multShift6;
            }
        }
        // Coulomb
        const v2f qq = qiS * xj.w;
        v2f qqRaw = {0.f, 0.f};
        if (ENERGY) qqRaw = qiRaw * xj.w;
        if ((MC == MC_EWALD || MC == MC_LJPME) && POLY) {
            // [erfc(ar)/r + 2a/sqrt(pi) e^{-(ar)^2}] / r^2 = 1/r^3 - Bt(r^2), Bt a degree-11 polynomial in t = r^2 * ewScale - 1: no exp, no rcp
            const v2f t = r2 * p.ewScale - 1.0f;
            v2f bt = t * p.ewPoly[11] + p.ewPoly[10];
#pragma unroll
            for (int k = 9; k >= 0; k--) bt = bt * t + p.ewPoly[k];
            f = f * invR2 + qq * (invR2 * invR - bt);
            if (ENERGY) {
                // pair ENERGY: qq (1/r - Et(r^2)), Et = erf(ar)/r a degree-13 polynomial in the same t.  The A&S erfc used before is good to
                // 1.5e-7 absolute, but its error is one-signed over most of the range and qq erfc(ar)/r is summed over 6e7 pairs: +1.0 kJ/mol
                // on the water-water slice of the 96k-atom box, whose direct (6.5e6) and reciprocal (-6.5e6) parts cancel to 788 -- 1.3e-3.
                // Degree 13 leaves a tenth of that (tools/erfc_energy_check.py; degree 11 is worse than A&S, beyond 13 float coefficients limit it).
                v2f et = t * p.ewPolyE[13] + p.ewPolyE[12];
#pragma unroll
                for (int k = 11; k >= 0; k--) et = et * t + p.ewPolyE[k];
                eC = qqRaw * (invR - et);
            }
        } else if (MC == MC_EWALD || MC == MC_LJPME) {
            const v2f r = r2 * invR;
            const v2f ar = r * p.alpha;
            const v2f e2 = r2 * (-p.alpha2l2e);
            const v2f ex = {__builtin_amdgcn_exp2f(e2.x), __builtin_amdgcn_exp2f(e2.y)};
            const v2f den = ar * 0.3275911f + 1.0f;
            const v2f tt = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
            v2f poly = tt * 1.061405429f + (-1.453152027f);
            poly = poly * tt + 1.421413741f;
            poly = poly * tt + (-0.284496736f);
            poly = poly * tt + 0.254829592f;
            const v2f erfcv = poly * tt * ex;
            f = (f + (qq * invR) * (erfcv + (ar * ex) * 1.1283791670955126f)) * invR2;
            if (ENERGY) eC = (qqRaw * invR) * erfcv;
        } else if (MC == MC_RF) {
            f = (f + qq * (invR - r2 * (2.0f * p.krf))) * invR2;
            if (ENERGY) eC = qqRaw * (invR + r2 * p.krf - p.crf);
        } else {
            f = (f + qq * invR) * invR2;
            if (ENERGY) eC = qqRaw * invR;
        }
        bool inA = MC == MC_NOCUTOFF ? true : (r2.x < p.cutoff2), inB = MC == MC_NOCUTOFF ? true : (r2.y < p.cutoff2);
        if (MASKED) { const int k = (c - s) & 7; inA = inA && !((maskA >> k) & 1u); inB = inB && !((maskB >> k) & 1u); }
        f.x = inA ? f.x : 0.0f; f.y = inB ? f.y : 0.0f;
        if (ENERGY) {
            eC.x = inA ? eC.x : 0.0f; eC.y = inB ? eC.y : 0.0f; eLJ.x = inA ? eLJ.x : 0.0f; eLJ.y = inB ? eLJ.y : 0.0f;
            ecl = ecl + eC; elj = elj + eLJ;
        }
        const v2f gx = f * dx, gy = f * dy, gz = f * dz;
        fix = fix + gx; fiy = fiy + gy; fiz = fiz + gz;
        fjx = rowRor1(fjx) - (gx.x + gx.y); fjy = rowRor1(fjy) - (gy.x + gy.y); fjz = rowRor1(fjz) - (gz.x + gz.y);
    }
}

template <typename Real, bool ENERGY> __device__ __forceinline__ void exceptionsBody(const PairListParams<Real>& p, const int blk);
template <typename Real, bool ENERGY> __device__ __forceinline__ void exclusionAtomsBody(const PairListParams<Real>& p, const int blk);

// The first nListBlocks work-groups of the launch run the O(N) pair lists (exclusion corrections, then 1-4 exceptions: latency-bound
// work that overlaps the VALU-bound tile work instead of trailing it as a launch of its own); the others loop over tile work items.
template <int MC, bool POLY, bool ENERGY, bool SWITCH, bool FIXED>
__global__ __launch_bounds__(256, 4) void k_directPacked(const DirectParams<float> p, const PairListParams<float> q, const int nExclBlocks, const int nListBlocks) {
    const int listBlock = p.listsLast ? (int)blockIdx.x - ((int)gridDim.x - nListBlocks) : (int)blockIdx.x;      // (CU-limited launch: list blocks last, see k_direct)
    if (listBlock >= 0 && listBlock < nListBlocks) {      // (energy steps: the list bodies reduce their slice energies in 2 S doubles of dynamic LDS)
        if (listBlock < nExclBlocks) { PairListParams<float> qe = q; qe.n = q.nExclAtoms; exclusionAtomsBody<float, ENERGY>(qe, listBlock); }
        else exceptionsBody<float, ENERGY>(q, listBlock - nExclBlocks);
        return;
    }
    const int tileBlock = p.listsLast ? (int)blockIdx.x : (int)blockIdx.x - nListBlocks, nTileBlocks = gridDim.x - nListBlocks;
    if (p.stepTrace && tileBlock == 0 && threadIdx.x == 0) p.stepTrace[p.traceSlot] = (long long)wall_clock64();
    double* const sliceE = SNB_SLICE_E_PARTITION(p.sliceE, p.nsub * (p.nsub + 1));
    __shared__ float4 s_pos[4][64];
    __shared__ float2 s_se[4][64];
    __shared__ float4 s_shift[128];      // lattice-image shift of every 7-bit image code (125 in use): + ka a + kb b + kc c (rows of p.box)
    if (threadIdx.x < 128) {
        const int sc = threadIdx.x;
        const int kx = sc / 25, ky = (sc - 25 * kx) / 5, kz = sc - 25 * kx - 5 * ky;
        const float ka = float(kx - 2), kb = float(ky - 2), kc = float(kz - 2);
        s_shift[sc] = sc < 125 ? make_float4(ka * p.box[0] + kb * p.box[3] + kc * p.box[6], kb * p.box[4] + kc * p.box[7], kc * p.box[8], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // CU-limited first launch of an overlapped step: work-groups beyond the limit leave at once (the items are handed out through the
    // counters, so nothing is lost) and the reciprocal pipeline's work-groups always find the registers and LDS the limit leaves free
    if (p.cuSlots != nullptr && p.cuLimit > 0) { if (!cuResident(p)) return; }
    else __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool dyn = p.workCounter != nullptr;      // (uniform) items through the device counters: every wave of either launch claims one at a time
    WorkClaim wc{p.workCounter, p.numWork, (tileBlock * 4 + wid) & (SNB_WORK_SHARDS - 1), lane};
    int item = tileBlock * 4 + wid;
    if (dyn) item = wc.anyLeft() ? wc.resolve(wc.issue()) : p.numWork;
    for (; item < p.numWork; ) {   // no block-level barrier inside the loop
    int claimed = 0;
    if (dyn) claimed = wc.issue();      // the next item, requested a whole item ahead (claims past the end are harmless)
    const int4 wi = p.workItems[p.workStart + item * p.workStride];
    const int I = __builtin_amdgcn_readfirstlane(wi.x);
    const int tBegin = __builtin_amdgcn_readfirstlane(wi.y), tEnd = tBegin + __builtin_amdgcn_readfirstlane(wi.z);
    const int c = lane & 15, row = lane >> 4;
    const int stageJ = 8 * row + (c & 7);                // j-atom this lane stages (entry c of quarter `row`)

    const float4 pa = p.posq[I * 32 + c], pb = p.posq[I * 32 + 16 + c];
    const float2 sa = p.sigeps[I * 32 + c], sb = p.sigeps[I * 32 + 16 + c];
    const v2f pix = {pa.x, pb.x}, piy = {pa.y, pb.y}, piz = {pa.z, pb.z};
    const v2f qi = {pa.w * p.k4pe, pb.w * p.k4pe}, sigi = {sa.x, sb.x}, epsi = {sa.y, sb.y};
    v2f fix = {0.f, 0.f}, fiy = {0.f, 0.f}, fiz = {0.f, 0.f};
    const v2f c6i = {8.0f * sa.x * sa.x * sa.x * sa.y, 8.0f * sb.x * sb.x * sb.x * sb.y};      // LJPME: c6 of the two i-atoms
    v2f ecl = {0.f, 0.f}, elj = {0.f, 0.f};
    int curSlice = -1; bool curNeeded = false;
    auto flushEnergy = [&]() {   // raw energies of the slice just finished: wave sum in double, one atomic per term
        if (curSlice >= 0) {
            const double a = waveSum((double)ecl.x + (double)ecl.y), b = waveSum((double)elj.x + (double)elj.y);
            if (lane == 0) { atomicAdd(&sliceE[2 * curSlice], a); atomicAdd(&sliceE[2 * curSlice + 1], b); }
        }
        ecl = {0.f, 0.f}; elj = {0.f, 0.f};
    };

    float4* myPos = s_pos[wid];
    float2* mySe = s_se[wid];
    const float4* rdPos = myPos + 16 * row + (c & 7) + 8;   // entry for step s is rdPos[-s]
    const float2* rdSe = mySe + 16 * row + (c & 7) + 8;

    // two-tile-deep software pipeline, as in k_direct
    // Software pipeline over the item's tiles with TWO register sets used alternately (the tile loop is unrolled by two).  While tile t
    // (set X) is evaluated, the atoms, masks and lambdas of tile t+1 are requested into set Y, and the list entry + header of tile t+2
    // into the jcode/head fields of X (X's own entry has been copied to `curCode` by then).  No requested register is copied, shifted
    // or converted before the trip in which it is consumed -- a copy (a loop-carried "next = loaded" move), the image shift added at
    // the load, or a uniform header moved to SGPRs behind its load each cost a full memory round trip per tile, atomics included,
    // because the wait counter is in-order.
    struct TileRegs { int jcode, slice, maskIdx; float4 pj; float2 sej; float shx, shy, shz; unsigned mA, mB; float2 lam; int need; };
    TileRegs A, B;
    int vzero; asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));      // a zero the compiler cannot see through (keeps the header load in VGPRs)
    // (the j-force atomics of a tile are issued at the start of the NEXT trip: an atomic sits behind a conditional skip the wait-count
    // pass cannot count, so any wait after it degrades to "everything"; issued first, they are a whole tile old when that wait comes)
    float pendX = 0.f, pendY = 0.f, pendZ = 0.f; int pendIdx = -1;
#ifdef SNB_EXP_NO_JATOMICS      // experiment builds only (wrong forces): the kernel without its j-force scatter
    auto flushPending = [&]() { asm volatile("" :: "v"(pendX), "v"(pendY), "v"(pendZ), "v"(pendIdx)); };
#else
    auto flushPending = [&]() { if (pendIdx >= 0) { fAddT<FIXED>(p, p.fx, pendIdx, pendX); fAddT<FIXED>(p, p.fy, pendIdx, pendY); fAddT<FIXED>(p, p.fz, pendIdx, pendZ); } };
#endif
    auto requestList = [&](TileRegs& r, int t) {                    // list entry + header of tile t
        r.jcode = p.tileJ[t * 32 + stageJ];
        const int2 v = *reinterpret_cast<const int2*>(&p.tileInfo[t + vzero]);
        r.slice = v.x; r.maskIdx = v.y;
    };
    auto requestAtoms = [&](TileRegs& r) {                          // needs r.jcode / r.slice / r.maskIdx (requested a tile earlier)
        const int code = r.jcode;
        asm volatile("" :: "v"(code), "v"(r.slice), "v"(r.maskIdx));   // the wait for the list entry belongs HERE, before anything younger is issued
        flushPending();
        const int idx = code == -1 ? 0 : (code & SNB_JIDX_MASK);
        r.pj = p.posq[idx]; r.sej = p.sigeps[idx];
        const float4 sh = s_shift[(code >> SNB_JSHIFT_BITS) & 127];      // lattice image of the entry (table filled at kernel start)
        r.shx = sh.x; r.shy = sh.y; r.shz = sh.z;
        const int mi = r.maskIdx < 0 ? 0 : r.maskIdx;              // unconditional loads: a conditionally loaded register is a phi with a copy
        r.mA = p.masks[mi * 32 + c]; r.mB = p.masks[mi * 32 + 16 + c];
        r.lam = *reinterpret_cast<const float2*>(&p.lambdas[2 * (r.slice & 0xFFFF)]);      // (the slice index comes with the tile header, written by the builder)
        if (ENERGY) r.need = p.sliceNeed[r.slice & 0xFFFF];      // energy steps: is this slice's energy wanted?
    };
    // what the staging of a tile leaves behind for its evaluation
    struct Staged { int code; bool hasMask; unsigned maskA, maskB; float lamC, lamL; int slice; bool needE; };
    auto stage = [&](TileRegs& R) {                                 // tile R -> LDS (the wait for its atoms sits here, in straight-line code)
        Staged st;
        __builtin_amdgcn_wave_barrier();
        st.code = R.jcode;
        if (st.code != -1) { myPos[lane] = make_float4(R.pj.x + R.shx, R.pj.y + R.shy, R.pj.z + R.shz, R.pj.w); mySe[lane] = R.sej; }
        else { myPos[lane] = make_float4(3e9f + 1e6f * c, -5e9f, 7e9f, 0.f); mySe[lane] = make_float2(0.f, 0.f); }      // padding slot: parked far away
        st.hasMask = R.maskIdx >= 0;
        st.maskA = st.hasMask ? R.mA >> (8 * row) : 0u; st.maskB = st.hasMask ? R.mB >> (8 * row) : 0u;   // my j-quarter's 8 bits
        st.lamC = R.lam.x; st.lamL = R.lam.y;
        st.slice = R.slice & 0xFFFF;
        st.needE = ENERGY && __builtin_amdgcn_readfirstlane(R.need) != 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        return st;
    };
    // One trip: request (unconditionally -- past the item's end the indices are clamped and the data unused, so that the number of
    // memory operations in flight is a compile-time constant and every wait can be exact), evaluate the staged tile X, scatter its
    // j-forces, then stage Y.  The wait for Y's atoms thus comes after a whole tile of arithmetic and leaves the younger requests and
    // the atomics in flight.
    auto tile = [&](TileRegs& X, TileRegs& Y, const int t, Staged& st) {
        requestAtoms(Y);      // (issues the previous tile's j-force atomics first)
        requestList(X, t + 2 < tEnd ? t + 2 : tEnd - 1);
        // (the sums of a slice nobody asked for are zero: no wave reduction, no atomics -- on derivative steps 95 % of the slice changes)
        if (ENERGY && st.slice != curSlice) { if (curNeeded) flushEnergy(); curSlice = st.slice; curNeeded = st.needE; }
        const v2f qiS = qi * st.lamC, epsiS = epsi * st.lamL;      // lambda folded into the i-side parameters once per tile
        float fjx = 0, fjy = 0, fjz = 0;
#define SNB_TILE_PACKED(M, E) tileStepsPacked<MC, M, POLY, E, SWITCH>(p, rdPos, rdSe, pix, piy, piz, sigi, qiS, epsiS, qi, epsi, c6i, st.lamL, st.maskA, st.maskB, c, fix, fiy, fiz, fjx, fjy, fjz, ecl, elj)
        if (ENERGY && st.needE) { if (st.hasMask) SNB_TILE_PACKED(true, ENERGY); else SNB_TILE_PACKED(false, ENERGY); }
        else { if (st.hasMask) SNB_TILE_PACKED(true, false); else SNB_TILE_PACKED(false, false); }      // forces only (also: energy steps, slice not wanted)
#undef SNB_TILE_PACKED
        // rotate-then-subtract leaves lane c holding slot (c+1)&7: one more rotation brings every slot home, then the two
        // partial sums of a slot (lanes c and c+8) are merged
        fjx = rowRor1(fjx); fjy = rowRor1(fjy); fjz = rowRor1(fjz);
        pendX = fjx + rowRor8(fjx); pendY = fjy + rowRor8(fjy); pendZ = fjz + rowRor8(fjz);
        const int curCode = st.code;
        pendIdx = (c < 8 && curCode != -1) ? (curCode & SNB_JIDX_MASK) : -1;      // entry c < 8 of quarter `row` == j-slot 8*row + c == the atom this lane staged
        st = stage(Y);
    };
    requestList(A, tBegin);
    B = A;
    requestList(B, tBegin + 1 < tEnd ? tBegin + 1 : tBegin);
    requestAtoms(A);
    Staged st = stage(A);
    for (int t = tBegin; t < tEnd; t += 2) {
        tile(A, B, t, st);
        if (t + 1 < tEnd) tile(B, A, t + 1, st);
    }
    flushPending();
    // the four rows hold partial sums for the same i-atoms (different j-quarters)
    float ox = fix.x, oy = fiy.x, oz = fiz.x, ux = fix.y, uy = fiy.y, uz = fiz.y;
    ox += __shfl_xor(ox, 16, 64); oy += __shfl_xor(oy, 16, 64); oz += __shfl_xor(oz, 16, 64);
    ux += __shfl_xor(ux, 16, 64); uy += __shfl_xor(uy, 16, 64); uz += __shfl_xor(uz, 16, 64);
    ox += __shfl_xor(ox, 32, 64); oy += __shfl_xor(oy, 32, 64); oz += __shfl_xor(oz, 32, 64);
    ux += __shfl_xor(ux, 32, 64); uy += __shfl_xor(uy, 32, 64); uz += __shfl_xor(uz, 32, 64);
    if (row == 0) { fAddT<FIXED>(p, p.fx, (I * 32 + c), ox); fAddT<FIXED>(p, p.fy, (I * 32 + c), oy); fAddT<FIXED>(p, p.fz, (I * 32 + c), oz); }
    if (row == 1) { fAddT<FIXED>(p, p.fx, (I * 32 + 16 + c), ux); fAddT<FIXED>(p, p.fy, (I * 32 + 16 + c), uy); fAddT<FIXED>(p, p.fz, (I * 32 + 16 + c), uz); }
    if (ENERGY) { if (curNeeded) flushEnergy(); curSlice = -1; curNeeded = false; }
    __builtin_amdgcn_wave_barrier();
    item = dyn ? wc.resolve(claimed) : item + nTileBlocks * 4;
    }   // work-item loop
    if (p.stepTrace && threadIdx.x == 0) p.stepTrace[p.traceSlot + 1] = (long long)wall_clock64();      // (plain store: the last work-group to leave writes last)
}


// evStart/evStop (both or neither): hipExtLaunchKernelGGL stamps them with the kernel's own begin and end -- the duration rocprofv3 reports,
// without the marker-packet overhead of hipEventRecord pairs around the launch.  *timed tells the caller whether a kernel took them.
// (a CU-limited launch of an overlapped step is told where its resident work-groups may sit: below (cuLimit - 1) allocations of this kernel)
#define SNB_LAUNCH_LDS(KERNEL, GRID, LDS, P, ...) do { auto p_ = (P); if (p_.cuSlots && p_.cuLimit > 0 && p_.cuBaseMax == 0x7fffffff) p_.cuBaseMax = (p_.cuLimit - 1) * vgprUnits((const void*)(KERNEL)); \
                                                    if (evStart) { hipExtLaunchKernelGGL(KERNEL, GRID, block, LDS, s, evStart, evStop, 0, p_, __VA_ARGS__); *timed = true; } \
                                                    else hipLaunchKernelGGL(KERNEL, GRID, block, LDS, s, p_, __VA_ARGS__); } while (0)
// register allocation of a kernel in the hardware's units of 8 VGPRs (accumulation registers included), looked up once per kernel
static int vgprUnits(const void* kernel) {
    static std::map<const void*, int> cache;
    auto it = cache.find(kernel);
    if (it != cache.end()) return it->second;
    hipFuncAttributes attr;
    int units = 16;
    if (hipFuncGetAttributes(&attr, kernel) == hipSuccess && attr.numRegs > 0) units = (attr.numRegs + 7) / 8;
    cache[kernel] = units;
    return units;
}
#define SNB_LAUNCH(KERNEL, GRID, ...) SNB_LAUNCH_LDS(KERNEL, GRID, 0, __VA_ARGS__)
template <typename Real, int MC> static bool launchDirectMC(const DirectParams<Real>& p, bool wrap, bool energy, const PairListParams<Real>* lists, hipStream_t s, hipEvent_t evStart, hipEvent_t evStop, bool* timed) {
    const int myItems = p.numWork;
    if (myItems <= 0) return false;
    int nwg = (myItems + 3) / 4;
    { static const int cap = getenv("SNB_DIRECT_WGS") ? atoi(getenv("SNB_DIRECT_WGS")) : 0; if (cap > 0 && nwg > cap) nwg = cap; }
    if (p.workCounter && p.gridCap > 0) nwg = p.cuLimit > 0 ? p.gridCap : std::min(nwg, p.gridCap);      // overlapped step: resident waves claim their items
    dim3 grid(nwg), block(256);
    if constexpr (std::is_same<Real, float>::value) {
        static const bool scalarEnergy = getenv("SNB_SCALAR_ENERGY_KERNEL") != nullptr;
        if (!wrap && !(energy && scalarEnergy) && !(p.useSwitch && MC == MC_NOCUTOFF)) {
            PairListParams<float> q;
            std::memset(&q, 0, sizeof(q));
            int nExclBlocks = 0, nListBlocks = 0;
            if (lists) { q = *lists; nExclBlocks = (q.nExclAtoms + 255) / 256; nListBlocks = nExclBlocks + (q.n + 255) / 256; }
            const size_t listLds = (lists && energy) ? sizeof(double) * 2 * q.nSlices : 0;      // the energy list bodies reduce per slice in LDS
            dim3 gridAll(nwg + nListBlocks);
            const bool poly = (MC == MC_EWALD || MC == MC_LJPME) && p.ewUsePoly;
#define SNB_PACKED(P, E, S) do { if (p.fixed) SNB_LAUNCH_LDS((k_directPacked<MC, P, E, S, true>), gridAll, listLds, p, q, nExclBlocks, nListBlocks); \
                                 else SNB_LAUNCH_LDS((k_directPacked<MC, P, E, S, false>), gridAll, listLds, p, q, nExclBlocks, nListBlocks); } while (0)
            if constexpr (MC == MC_NOCUTOFF) { if (energy) SNB_PACKED(false, true, false); else SNB_PACKED(false, false, false); }
            else if (p.useSwitch && MC != MC_LJPME) {      // (no switching function under LJPME, Q2)
                if (energy) { if (poly) SNB_PACKED(true, true, true); else SNB_PACKED(false, true, true); }
                else { if (poly) SNB_PACKED(true, false, true); else SNB_PACKED(false, false, true); }
            } else {
                if (energy) { if (poly) SNB_PACKED(true, true, false); else SNB_PACKED(false, true, false); }
                else { if (poly) SNB_PACKED(true, false, false); else SNB_PACKED(false, false, false); }
            }
#undef SNB_PACKED
            return lists != nullptr;
        }
    }
    // the scalar tile kernel (double precision, per-pair wrapping, the switches above) carries the pair lists the same way
    PairListParams<Real> q;
    std::memset(&q, 0, sizeof(q));
    int nExclBlocks = 0, nListBlocks = 0;
    static const bool noFusedLists = getenv("SNB_NO_FUSED_LISTS") != nullptr;
    const bool fuseLists = lists != nullptr && !noFusedLists;
    if (fuseLists) { q = *lists; nExclBlocks = (q.nExclAtoms + 255) / 256; nListBlocks = nExclBlocks + (q.n + 255) / 256; }
    const size_t listLds = (fuseLists && energy) ? sizeof(double) * 2 * q.nSlices : 0;
    dim3 gridAll(nwg + nListBlocks);
    if (wrap) {
        if (energy) SNB_LAUNCH_LDS((k_direct<Real, MC, true, true>), gridAll, listLds, p, q, nExclBlocks, nListBlocks);
        else SNB_LAUNCH_LDS((k_direct<Real, MC, true, false>), gridAll, listLds, p, q, nExclBlocks, nListBlocks);
    } else {
        if (energy) SNB_LAUNCH_LDS((k_direct<Real, MC, false, true>), gridAll, listLds, p, q, nExclBlocks, nListBlocks);
        else SNB_LAUNCH_LDS((k_direct<Real, MC, false, false>), gridAll, listLds, p, q, nExclBlocks, nListBlocks);
    }
    return fuseLists;
}

// Returns true when the launch also ran the pair lists passed in `lists` (single-precision forces-only tile kernel); otherwise the
// caller launches them itself.
template <typename Real> bool launchDirect(const DirectParams<Real>& p0, int mc, bool wrap, bool energy, const PairListParams<Real>* lists, hipStream_t s, hipEvent_t evStart, hipEvent_t evStop, bool* timed) {
    const DirectParams<Real>& p = p0;
    switch (mc) {
        case MC_NOCUTOFF: return launchDirectMC<Real, MC_NOCUTOFF>(p, wrap, energy, lists, s, evStart, evStop, timed);
        case MC_RF: return launchDirectMC<Real, MC_RF>(p, wrap, energy, lists, s, evStart, evStop, timed);
        case MC_EWALD: return launchDirectMC<Real, MC_EWALD>(p, wrap, energy, lists, s, evStart, evStop, timed);
        default: return launchDirectMC<Real, MC_LJPME>(p, wrap, energy, lists, s, evStart, evStop, timed);
    }
}
template bool launchDirect<float>(const DirectParams<float>&, int, bool, bool, const PairListParams<float>*, hipStream_t, hipEvent_t, hipEvent_t, bool*);
template bool launchDirect<double>(const DirectParams<double>&, int, bool, bool, const PairListParams<double>*, hipStream_t, hipEvent_t, hipEvent_t, bool*);

// ---- 1-4 exceptions: ReferenceSlicedLJCoulomb14.cpp:61-95 ----------------------------------------
template <typename Real, bool ENERGY> __device__ __forceinline__ void exceptionsBody(const PairListParams<Real>& p, const int blk) {
    extern __shared__ double s_sliceE[];   // [2*S]
    const int nS2 = 2 * p.nSlices;
    if (ENERGY) { for (int i = threadIdx.x; i < nS2; i += 256) s_sliceE[i] = 0.0; __syncthreads(); }
    const int k = blk * 256 + threadIdx.x;
    double e0 = 0, e1 = 0; int slice = 0;
    if (k < p.n) {
        int2 ij = p.pairs[k];
        ij.x = p.userToSorted[ij.x]; ij.y = p.userToSorted[ij.y];
        const auto par = p.params[k];
        const auto xi = p.posq[ij.x]; const auto xj = p.posq[ij.y];
        Real dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
        if (p.periodic) { Real inv[3] = {Real(1) / p.box[0], Real(1) / p.box[4], Real(1) / p.box[8]}; wrapDelta<Real>(dx, dy, dz, p.box, inv); }
        else unwrapDelta<Real>(dx, dy, dz, p.imageOffset, ij.x, ij.y);
        const Real invR = rsq(dx * dx + dy * dy + dz * dz);
        Real s2 = invR * par.x; s2 *= s2;
        const Real s6 = s2 * s2 * s2;
        slice = (int)par.w;
        const Real lamC = p.lambdas[2 * slice], lamL = p.lambdas[2 * slice + 1];
        Real dEdR = lamL * par.y * (Real(12) * s6 - Real(6)) * s6 + lamC * par.z * invR;
        dEdR *= invR * invR;
        fAdd(p, p.fx, ij.x, dEdR * dx); fAdd(p, p.fy, ij.x, dEdR * dy); fAdd(p, p.fz, ij.x, dEdR * dz);
        fAdd(p, p.fx, ij.y, -dEdR * dx); fAdd(p, p.fy, ij.y, -dEdR * dy); fAdd(p, p.fz, ij.y, -dEdR * dz);
        if (ENERGY && p.sliceNeed[slice]) { e0 = par.z * invR; e1 = par.y * (s6 - Real(1)) * s6; }
    }
    if (ENERGY) {
        if (k < p.n) {
            __hip_atomic_fetch_add(&s_sliceE[2 * slice], e0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&s_sliceE[2 * slice + 1], e1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nS2; i += 256) { const double v = s_sliceE[i]; if (v != 0.0) atomicAdd(&SNB_SLICE_E_PARTITION(p.sliceE, 2 * p.nSlices)[i], v); }
    }
}

// ---- Ewald exclusion corrections: ReferenceSlicedLJCoulombIxn.cpp:449-506 -------------------------
// One thread per ATOM walking its own exclusion list (CSR over sorted indices): every excluded pair is evaluated
// from both ends, so the force update is a plain read-modify-write of the atom's own accumulator -- no atomics
// (2 M scattered float atomics cost 70 us on MI355X; this costs a few us).  Energies: half a pair from each end,
// reduced per slice in LDS (ds_add_f64) and flushed with one global atomic per slice and work-group.
template <typename Real, bool ENERGY> __device__ __forceinline__ void exclusionAtomsBody(const PairListParams<Real>& p, const int blk) {
    extern __shared__ double s_sliceE[];   // [2*S]
    const int nS2 = 2 * p.nSlices;
    if (ENERGY) { for (int i = threadIdx.x; i < nS2; i += 256) s_sliceE[i] = 0.0; __syncthreads(); }
    const int a = blk * 256 + threadIdx.x;
    const int ua = a < p.n ? p.sortedToUser[a] : -1;
    if (ua >= 0) {
        const int e0 = p.exclStart[ua], e1 = p.exclStart[ua + 1];
        if (e1 > e0) {
            const auto xi = p.posq[a];
            const auto sei = p.sigeps[a];
            const int si = p.blockSubset[a >> 5];
            const Real qi = xi.w * Real(SNB_ONE_4PI_EPS0);
            const Real c6i = Real(8) * sei.x * sei.x * sei.x * sei.y;
            Real inv[3] = {Real(1) / p.box[0], Real(1) / p.box[4], Real(1) / p.box[8]};
            Real fx = 0, fy = 0, fz = 0;
            for (int e = e0; e < e1; e++) {
                const int b = p.userToSorted[p.exclList[e]];
                const auto xj = p.posq[b];
                Real dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
                if (p.periodic) wrapDelta<Real>(dx, dy, dz, p.box, inv); else unwrapDelta<Real>(dx, dy, dz, p.imageOffset, a, b);
                const Real r2 = dx * dx + dy * dy + dz * dz;
                const Real r = sqrt(r2);
                const Real invR = Real(1) / r;
                const Real ar = p.alpha * r;
                const int slice = sliceOf(si, p.blockSubset[b >> 5]);
                const Real lamC = p.lambdas[2 * slice], lamL = p.lambdas[2 * slice + 1];
                const Real qq = qi * xj.w;
                const Real ex = fexp(-ar * ar);
                // energy evaluations use libm erf in double even in the single-precision engine: the raw slice energy is a
                // small difference of large direct/reciprocal/exclusion sums and a 1e-7 bias in erf would show up in it.
                // The exclusion-correction ENERGY of a solvated box is a huge sum (2e7 kJ/mol for 300k atoms of water: every O-H and H-H pair of
                // every molecule) that the reciprocal-space energy cancels to a few thousand: it is formed in double from the stored
                // coordinates, charges and the double alpha -- products like k*qO*qH rounded to float are off by the SAME 1e-7 for all
                // 2e5 identical pairs, which alone was 0.8 kJ/mol (1.1e-3 of the water-water slice of the 96k-atom box).
                double erfv, rd = 0, qqd = 0;
                const bool wantE = ENERGY && p.sliceNeed[slice] != 0;
                if (wantE) {
                    rd = sqrt((double)dx * (double)dx + (double)dy * (double)dy + (double)dz * (double)dz);
                    qqd = (double)xi.w * (double)xj.w * SNB_ONE_4PI_EPS0;
                    erfv = erf(p.alpha64 * rd);
                } else erfv = erfOf(ar, ex);
                Real f = 0;
                if (erfv > 1e-6) {
                    f = -lamC * qq * invR * invR * invR * (Real(erfv) - ar * ex * Real(1.1283791670955126));
                    if (wantE) __hip_atomic_fetch_add(&s_sliceE[2 * slice], -0.5 * qqd * erfv / rd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else if (wantE)
                    __hip_atomic_fetch_add(&s_sliceE[2 * slice], -0.5 * p.alpha64 * 1.1283791670955126 * qqd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (p.ljpme) {
                    const auto sej = p.sigeps[b];
                    const Real c6 = c6i * (Real(8) * sej.x * sej.x * sej.x * sej.y);
                    const Real dar2 = p.alphaD * p.alphaD * r2, dar4 = dar2 * dar2, dar6 = dar4 * dar2;
                    const Real invR2 = invR * invR;
                    const Real expd = fexp(-dar2);
                    const Real coef = c6 * invR2 * invR2 * invR2;
                    if (wantE) __hip_atomic_fetch_add(&s_sliceE[2 * slice + 1], 0.5 * (double)(coef * (Real(1) - expd * (Real(1) + dar2 + Real(0.5) * dar4))), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    // reference: dEdR = -6 c6 r^-8 (...), forces[ii] -= lam*dEdR*delta  =>  +6 lam c6 r^-8 (...) * delta on ii
                    f += lamL * Real(6) * coef * invR2 * (Real(1) - expd * (Real(1) + dar2 + Real(0.5) * dar4 + dar6 * Real(1.0 / 6.0)));
                }
                fx += f * dx; fy += f * dy; fz += f * dz;
            }
            fAdd(p, p.fx, a, fx); fAdd(p, p.fy, a, fy); fAdd(p, p.fz, a, fz);   // atomics: the 1-4 blocks of the same launch add to the same atoms
        }
    }
    if (ENERGY) {
        __syncthreads();
        for (int i = threadIdx.x; i < nS2; i += 256) { const double v = s_sliceE[i]; if (v != 0.0) atomicAdd(&SNB_SLICE_E_PARTITION(p.sliceE, 2 * p.nSlices)[i], v); }
    }
}

// One launch for both O(N) pair lists: blocks [0, nExclBlocks) run the per-atom Ewald exclusion corrections (p.nExclAtoms atoms),
// the remaining blocks the 1-4 exceptions (p.n pairs).
template <typename Real, bool ENERGY> __global__ __launch_bounds__(256) void k_pairLists(const PairListParams<Real> p, const int nExclBlocks) {
    if ((int)blockIdx.x < nExclBlocks) { PairListParams<Real> q = p; q.n = p.nExclAtoms; exclusionAtomsBody<Real, ENERGY>(q, blockIdx.x); }
    else exceptionsBody<Real, ENERGY>(p, blockIdx.x - nExclBlocks);
}
template <typename Real> void launchPairLists(const PairListParams<Real>& p, bool energy, hipStream_t s) {
    const int nExclBlocks = (p.nExclAtoms + 255) / 256, nExcBlocks = (p.n + 255) / 256;
    if (nExclBlocks + nExcBlocks <= 0) return;
    dim3 grid(nExclBlocks + nExcBlocks), block(256);
    const size_t lds = sizeof(double) * 2 * p.nSlices;
    if (energy) hipLaunchKernelGGL((k_pairLists<Real, true>), grid, block, lds, s, p, nExclBlocks);
    else hipLaunchKernelGGL((k_pairLists<Real, false>), grid, block, 0, s, p, nExclBlocks);
}
template void launchPairLists<float>(const PairListParams<float>&, bool, hipStream_t);
template void launchPairLists<double>(const PairListParams<double>&, bool, hipStream_t);

}  // namespace snb
