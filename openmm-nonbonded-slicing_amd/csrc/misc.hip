// misc.hip -- per-step data movement kernels of the engine: positions user order -> sorted posq, forces sorted
// accumulators -> user order.  Pure HBM streaming (16 B/lane where the layout allows).
#include "snb_internal.h"
#include <algorithm>

namespace snb {

// posq[s].xyz = userPos[sortedToUser[s]] + imageOffset[s]; charge (.w) is kept.  Padding slots (sortedToUser < 0)
// keep their parked far-away coordinates.  The same pass clears the six force arrays of the atom (no separate memset node in the
// step graph) and, when a Coulomb mesh is given, writes the atom's packed mesh cell for the brick spreader (pme.hip, k_pmeCells).
template <typename Real, typename In>
__global__ void k_gatherPositions(const In* __restrict__ userPos, int stride, const int* __restrict__ sortedToUser,
                                  const Real* __restrict__ imageOffset, typename Vec<Real>::T4* __restrict__ posq, int nPadded,
                                  Real* __restrict__ forces, int nClear, const GatherCells<Real> gc) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    SNB_TRACE_START(gc.stepTrace, 0);
    if (gc.clearE) for (int i = s; i < gc.nClearE; i += gridDim.x * blockDim.x) gc.clearE[i] = 0.0;
    if (gc.zeroInts && s < gc.nZeroInts) gc.zeroInts[s] = 0;
    if (gc.zeroInts2) for (int i = s; i < gc.nZeroInts2; i += gridDim.x * blockDim.x) gc.zeroInts2[i] = 0;
    if (s >= nPadded) return;
    if (forces) {
        for (int k = 0; k < nClear; k++) forces[(size_t)k * nPadded + s] = Real(0);      // 7: 4 n direct-space (either layout) + 3 n reciprocal; 10 with 64-bit accumulators
    }
    const int u = sortedToUser[s];
    if (u < 0) { if (gc.cells) gc.cells[s] = -1; return; }
    auto v = posq[s];
    v.x = (Real)userPos[(size_t)u * stride] + imageOffset[3 * s];
    v.y = (Real)userPos[(size_t)u * stride + 1] + imageOffset[3 * s + 1];
    v.z = (Real)userPos[(size_t)u * stride + 2] + imageOffset[3 * s + 2];
    posq[s] = v;
    if (gc.posRef) {
        const auto r0 = gc.posRef[s];
        const Real ddx = v.x - r0.x, ddy = v.y - r0.y, ddz = v.z - r0.z;
        const Real d2 = ddx * ddx + ddy * ddy + ddz * ddz;
        if (d2 > gc.warn2) { gc.flags[0] = 1; if (d2 > gc.fail2) gc.flags[1] = 1; }      // plain stores of the same value: no atomics needed
    }
    if (gc.cells) {
        int cell = -1;
        if (gc.atomGrid[s] >= 0 && v.w != Real(0)) {
            int idx[3]; Real fr[3];
            gridCoord<Real>(gc.recip, gc.recipLo, v.x, v.y, v.z, gc.nx, gc.ny, gc.nz, idx, fr);
            cell = idx[0] | (idx[1] << 10) | (idx[2] << 20);
        }
        gc.cells[s] = cell;
    }
}

template <typename Real>
void launchGatherPositions(const void* userPos, int isDouble, int stride4, const int* sortedToUser, const Real* imageOffset,
                           typename Vec<Real>::T4* posq, int nPadded, Real* forces, int nClear, const GatherCells<Real>& gc, hipStream_t s) {
    if (nPadded <= 0) return;
    dim3 grid((nPadded + 255) / 256), block(256);
    const int stride = stride4 ? 4 : 3;
    if (isDouble) SNB_STAMPED_LAUNCH(0, (k_gatherPositions<Real, double>), grid, block, 0, s, (const double*)userPos, stride, sortedToUser, imageOffset, posq, nPadded, forces, nClear, gc);
    else SNB_STAMPED_LAUNCH(0, (k_gatherPositions<Real, float>), grid, block, 0, s, (const float*)userPos, stride, sortedToUser, imageOffset, posq, nPadded, forces, nClear, gc);
}

// In-place refresh of the sorted per-atom parameters from the user-order values (parameter offsets / updateParametersInContext)
template <typename Real>
__global__ void k_refreshParams(const int* __restrict__ sortedToUser, const Real* __restrict__ uCharge, const typename Vec<Real>::T2* __restrict__ uSigEps,
                                typename Vec<Real>::T4* __restrict__ posq, typename Vec<Real>::T2* __restrict__ sigeps, int nPadded) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nPadded) return;
    const int u = sortedToUser[s];
    if (u < 0) return;
    posq[s].w = uCharge[u];
    sigeps[s] = uSigEps[u];
}
template <typename Real>
void launchRefreshParams(const int* sortedToUser, const Real* uCharge, const typename Vec<Real>::T2* uSigEps, typename Vec<Real>::T4* posq,
                         typename Vec<Real>::T2* sigeps, int nPadded, hipStream_t s) {
    if (nPadded <= 0) return;
    hipLaunchKernelGGL((k_refreshParams<Real>), dim3((nPadded + 255) / 256), dim3(256), 0, s, sortedToUser, uCharge, uSigEps, posq, sigeps, nPadded);
}
template void launchRefreshParams<float>(const int*, const float*, const Vec<float>::T2*, Vec<float>::T4*, Vec<float>::T2*, int, hipStream_t);
template void launchRefreshParams<double>(const int*, const double*, const Vec<double>::T2*, Vec<double>::T4*, Vec<double>::T2*, int, hipStream_t);

template <typename Real, typename Out>
__global__ void k_finishForces(const Real* __restrict__ fx, const Real* __restrict__ fy, const Real* __restrict__ fz, int fs, int fixed,
                               const Real* __restrict__ fpx, const Real* __restrict__ fpy, const Real* __restrict__ fpz,
                               const int* __restrict__ userToSorted, int nAtoms, Out* __restrict__ out, int accumulate) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= nAtoms) return;
    const int s = userToSorted[u];
    Out x, y, z;
    if (fixed) {      // 64-bit fixed point, 2^32 per unit (SNB_MIXED)
        x = (Out)((double)reinterpret_cast<const long long*>(fx)[(size_t)s * fs] * (1.0 / 4294967296.0));
        y = (Out)((double)reinterpret_cast<const long long*>(fy)[(size_t)s * fs] * (1.0 / 4294967296.0));
        z = (Out)((double)reinterpret_cast<const long long*>(fz)[(size_t)s * fs] * (1.0 / 4294967296.0));
    } else { x = (Out)fx[(size_t)s * fs]; y = (Out)fy[(size_t)s * fs]; z = (Out)fz[(size_t)s * fs]; }
    if (fpx) { x += (Out)fpx[s]; y += (Out)fpy[s]; z += (Out)fpz[s]; }
    if (accumulate) { x += out[3 * (size_t)u]; y += out[3 * (size_t)u + 1]; z += out[3 * (size_t)u + 2]; }
    out[3 * (size_t)u] = x; out[3 * (size_t)u + 1] = y; out[3 * (size_t)u + 2] = z;
}

template <typename Real>
void launchFinishForces(const Real* fx, const Real* fy, const Real* fz, int fs, int fixed, const Real* fpx, const Real* fpy, const Real* fpz,
                        const int* userToSorted, int nAtoms, void* out, int isDouble, int accumulate, hipStream_t s) {
    if (nAtoms <= 0) return;
    dim3 grid((nAtoms + 255) / 256), block(256);
    if (isDouble) hipLaunchKernelGGL((k_finishForces<Real, double>), grid, block, 0, s, fx, fy, fz, fs, fixed, fpx, fpy, fpz, userToSorted, nAtoms, (double*)out, accumulate);
    else hipLaunchKernelGGL((k_finishForces<Real, float>), grid, block, 0, s, fx, fy, fz, fs, fixed, fpx, fpy, fpz, userToSorted, nAtoms, (float*)out, accumulate);
}

// ---- effective parameters on the device (the reference: nonbondedParameters.cc computeParameters :4-137, computePlasmaCorrection :139-179) ----
// One thread per particle: (q, sigma, eps) = base + sum over the particle's offsets of global[k] * delta, in double, then the engine's
// working form (q ; sigma/2, 2 sqrt(eps)) in user order.  Offsets are CSR by particle.
template <typename Real>
__global__ void k_particleParams(int n, const double* __restrict__ base, const int* __restrict__ offStart, const int* __restrict__ offGlobal,
                                 const double* __restrict__ offDelta, const double* __restrict__ globals, Real* __restrict__ uCharge,
                                 typename Vec<Real>::T2* __restrict__ uSigEps) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n) return;
    double q = base[3 * (size_t)u], sg = base[3 * (size_t)u + 1], ep = base[3 * (size_t)u + 2];
    if (offStart) for (int o = offStart[u]; o < offStart[u + 1]; o++) {
        const double v = globals[offGlobal[o]];
        q += v * offDelta[3 * (size_t)o]; sg += v * offDelta[3 * (size_t)o + 1]; ep += v * offDelta[3 * (size_t)o + 2];
    }
    uCharge[u] = (Real)q;
    typename Vec<Real>::T2 se; se.x = (Real)(0.5 * sg); se.y = (Real)(2.0 * sqrt(ep));
    uSigEps[u] = se;
}
// One thread per 1-4 pair: (qq, sigma, eps) = base + offsets -> (sigma, 4 eps, qq / (4 pi eps0), slice)
template <typename Real>
__global__ void k_exceptionParams(int n, const double* __restrict__ base, const int* __restrict__ offStart, const int* __restrict__ offGlobal,
                                  const double* __restrict__ offDelta, const double* __restrict__ globals, const int* __restrict__ slice,
                                  typename Vec<Real>::T4* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double qq = base[3 * (size_t)k], sg = base[3 * (size_t)k + 1], ep = base[3 * (size_t)k + 2];
    if (offStart) for (int o = offStart[k]; o < offStart[k + 1]; o++) {
        const double v = globals[offGlobal[o]];
        qq += v * offDelta[3 * (size_t)o]; sg += v * offDelta[3 * (size_t)o + 1]; ep += v * offDelta[3 * (size_t)o + 2];
    }
    typename Vec<Real>::T4 v; v.x = (Real)sg; v.y = (Real)(4.0 * ep); v.z = (Real)(SNB_ONE_4PI_EPS0 * qq); v.w = (Real)slice[k];
    out[k] = v;
}
// What the closed-form energy terms and the spreader need from the effective parameters AS STORED (rounded to Real): per subset the sums
// of q, q^2 and c6^2 (self energies, neutralising background: ReferenceSlicedLJCoulombIxn.cpp:203-222), and the largest |q| and |c6|
// (fixed-point scale of the brick spreader).  sums: [3 nsub] doubles, then two ints holding the maxima as float bit patterns.
template <typename Real>
__global__ void __launch_bounds__(1024) k_paramSums(int n, int nsub, const Real* __restrict__ uCharge, const typename Vec<Real>::T2* __restrict__ uSigEps, const int* __restrict__ uSubset,
                            double* __restrict__ partials) {
    extern __shared__ double s_sums[];      // [3 nsub]
    __shared__ int s_max[2];
    for (int i = threadIdx.x; i < 3 * nsub; i += blockDim.x) s_sums[i] = 0.0;
    if (threadIdx.x < 2) s_max[threadIdx.x] = 0;
    __syncthreads();
    float mq = 0.f, mc = 0.f;
    const int lane = threadIdx.x & 63;
    for (int u0 = blockIdx.x * blockDim.x; u0 < n; u0 += gridDim.x * blockDim.x) {      // (uniform trip count: the wave reductions below need every lane)
        const int u = u0 + threadIdx.x;
        const bool valid = u < n;
        double q = 0, c6 = 0; int sb = -1;
        if (valid) {
            q = (double)uCharge[u];
            const double hs = (double)uSigEps[u].x, se = (double)uSigEps[u].y;
            c6 = 8.0 * hs * hs * hs * se;
            sb = uSubset[u];
            mq = fmaxf(mq, fabsf((float)q)); mc = fmaxf(mc, fabsf((float)c6));
        }
        // one wave reduction per subset present in the wave (usually one), then three LDS adds by one lane -- 64 lanes adding to the
        // same three doubles were serialised (52 us for 300k atoms; a parameter change costs this pass)
        unsigned long long todo = __ballot(valid);
        while (todo) {
            const int sel = __shfl(sb, __builtin_ctzll(todo), 64);
            const bool mine = valid && sb == sel;
            double a = mine ? q : 0.0, b = mine ? q * q : 0.0, c = mine ? c6 * c6 : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); c += __shfl_xor(c, o, 64); }
            if (lane == 0) {
                __hip_atomic_fetch_add(&s_sums[3 * sel], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&s_sums[3 * sel + 1], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&s_sums[3 * sel + 2], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            todo &= ~__ballot(mine);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mq = fmaxf(mq, __shfl_xor(mq, o, 64)); mc = fmaxf(mc, __shfl_xor(mc, o, 64)); }
    if (lane == 0) { atomicMax(&s_max[0], __float_as_int(mq)); atomicMax(&s_max[1], __float_as_int(mc)); }      // non-negative floats order like their bit patterns
    __syncthreads();
    // partials, one row per work-group, summed by k_fixScale: atomics on ONE address are serialised in the memory-side unit (about 5 ns
    // each; 512 groups x 20 of them were the whole 50 us of the first version of this kernel)
    double* row = partials + (size_t)blockIdx.x * (3 * nsub + 1);
    for (int i = threadIdx.x; i < 3 * nsub; i += blockDim.x) row[i] = s_sums[i];
    if (threadIdx.x < 2) reinterpret_cast<int*>(row + 3 * nsub)[threadIdx.x] = s_max[threadIdx.x];
}
// fixed point: headroom x the largest per-atom value fits 31 bits.  A mesh point collects sum_a q_a w_a <= max|q| sum_a w_a, and the weights of
// all atoms at one point add up to the local number of atoms per mesh cell (partition of unity): the headroom is max(16, 8 x atoms per cell of
// THAT mesh) -- 16 for the usual 0.1 nm Coulomb mesh (0.17 atoms per cell), more for a coarse mesh, whose points collect many atoms
template <typename Real> __global__ void k_fixScale(const double* __restrict__ partials, int rows, int nsub, double* __restrict__ sums, Real* __restrict__ fix, double headQ, double headC) {
    const int w = 3 * nsub + 1, lane = threadIdx.x & 63;      // four waves share the columns; every wave finds the maxima, the first writes them
    for (int i = threadIdx.x >> 6; i < 3 * nsub; i += 4) {
        double a = 0;
        for (int r = lane; r < rows; r += 64) a += partials[(size_t)r * w + i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if (lane == 0) sums[i] = a;
    }
    int mq = 0, mc = 0;
    for (int r = lane; r < rows; r += 64) {
        const int* im = reinterpret_cast<const int*>(partials + (size_t)r * w + 3 * nsub);
        mq = max(mq, im[0]); mc = max(mc, im[1]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mq = max(mq, __shfl_xor(mq, o, 64)); mc = max(mc, __shfl_xor(mc, o, 64)); }
    if (threadIdx.x < 2) {
        const double m = fmax((double)__int_as_float(threadIdx.x ? mc : mq), 1e-30);
        const Real sc = (Real)(1073741824.0 / ((threadIdx.x ? headC : headQ) * m));
        fix[2 * threadIdx.x] = sc; fix[2 * threadIdx.x + 1] = (Real)(1.0 / (double)sc);
        reinterpret_cast<int*>(sums + 3 * nsub)[threadIdx.x] = threadIdx.x ? mc : mq;
    }
}
template <typename Real>
void launchParticleParams(int n, int nsub, const double* base, const int* offStart, const int* offGlobal, const double* offDelta, const double* globals,
                          const int* uSubset, Real* uCharge, typename Vec<Real>::T2* uSigEps, double* sums, Real* fix, double headQ, double headC, hipStream_t s) {
    // sums: [3 nsub + 1] totals, then SNB_PARAM_SUM_ROWS rows of [3 nsub + 1] partials
    double* partials = sums + 3 * (size_t)nsub + 1;
    const int rows = std::min((n + 1023) / 1024, SNB_PARAM_SUM_ROWS);
    if (n > 0) {
        hipLaunchKernelGGL((k_particleParams<Real>), dim3((n + 255) / 256), dim3(256), 0, s, n, base, offStart, offGlobal, offDelta, globals, uCharge, uSigEps);
        hipLaunchKernelGGL((k_paramSums<Real>), dim3(rows), dim3(1024), sizeof(double) * 3 * nsub, s, n, nsub, uCharge, uSigEps, uSubset, partials);
    }
    hipLaunchKernelGGL((k_fixScale<Real>), dim3(1), dim3(256), 0, s, partials, rows, nsub, sums, fix, headQ, headC);
}
template <typename Real>
void launchExceptionParams(int n, const double* base, const int* offStart, const int* offGlobal, const double* offDelta, const double* globals, const int* slice,
                           typename Vec<Real>::T4* out, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL((k_exceptionParams<Real>), dim3((n + 255) / 256), dim3(256), 0, s, n, base, offStart, offGlobal, offDelta, globals, slice, out);
}
template void launchParticleParams<float>(int, int, const double*, const int*, const int*, const double*, const double*, const int*, float*, Vec<float>::T2*, double*, float*, double, double, hipStream_t);
template void launchParticleParams<double>(int, int, const double*, const int*, const int*, const double*, const double*, const int*, double*, Vec<double>::T2*, double*, double*, double, double, hipStream_t);
template void launchExceptionParams<float>(int, const double*, const int*, const int*, const double*, const double*, const int*, Vec<float>::T4*, hipStream_t);
template void launchExceptionParams<double>(int, const double*, const int*, const int*, const double*, const double*, const int*, Vec<double>::T4*, hipStream_t);

// Raw slice energies: the kernels of an energy step add into SNB_SLICE_E_PARTS copies of the [S][2] table (chosen by work-group).
// This last kernel of the step sums the copies and adds the closed-form terms -- self energy and neutralising background
// (ReferenceSlicedLJCoulombIxn.cpp:203-222) from the per-subset parameter sums k_paramSums keeps on the device, and the long-range
// dispersion correction coef_s / V (ReferenceNonbondedSlicingKernels.cpp:244-249) -- so that an energy / derivative step needs no
// host arithmetic and no synchronisation: the host (or the caller's own kernel, snb_slice_energies_device) reads 2 S doubles when
// it wants them.
__global__ void k_finishSliceEnergies(const double* __restrict__ parts, double* __restrict__ out, int n, SliceFinish f) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double acc = 0;
    for (int part = 0; part < SNB_SLICE_E_PARTS; part++) acc += parts[(size_t)part * n + i];
    acc += sliceFinishClosedForm(f, i);
    out[i] = acc;
}
void launchFinishSliceEnergies(const double* parts, double* out, int n, const SliceFinish& f, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_finishSliceEnergies, dim3((n + 63) / 64), dim3(64), 0, s, parts, out, n, f);
}

// Zero fill as a KERNEL.  hipMemsetAsync captured into a hipGraph is a memset node, and a replayed memset node went wrong in round 4:
// the phase-A graph of an engine's in-line rebuilds, replayed after the two side-build graphs of the same engine (same topology, memset
// nodes of their own) had been instantiated, left `blockWideOut` full of non-zero words -- every atom then counts as a block of its own, 32 N
// padded slots -- although the node's destination, value and size were unchanged; with this kernel in its place the same replay is right
// (bench.py's displacement-triggered leg, docs/MEASUREMENT_LOG.md round 4 section 5).  Nothing that can be captured uses hipMemsetAsync now.
__global__ void k_zeroFill(int4* __restrict__ a, size_t n16, unsigned char* __restrict__ tail, int nTail) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) a[i] = make_int4(0, 0, 0, 0);
    if (blockIdx.x == 0 && (int)threadIdx.x < nTail) tail[threadIdx.x] = 0;
}
void launchZeroFill(void* ptr, size_t bytes, hipStream_t s) {      // ptr 16-byte aligned (hipMalloc)
    if (bytes == 0) return;
    const size_t n16 = bytes / 16; const int nTail = (int)(bytes - n16 * 16);
    const unsigned grid = (unsigned)std::max<size_t>(1, std::min<size_t>((n16 + 255) / 256, 2048));
    hipLaunchKernelGGL(k_zeroFill, dim3(grid), dim3(256), 0, s, (int4*)ptr, n16, (unsigned char*)ptr + n16 * 16, nTail);
}

// The displacement watch's flags (mapped host memory) cleared IN STREAM ORDER -- behind every step that used the old list, in front of the
// first one on the new list (a list exchanged without draining the queue: engine.hip finishSideBuild); an overrun seen by a step that was
// still queued when the host exchanged the lists is counted in flags[4] instead of being lost.
__global__ void k_dispFlagsReset(volatile int* flags) {
    if (threadIdx.x == 0) { if (flags[1]) flags[4] = flags[4] + 1; flags[0] = 0; flags[1] = 0; __threadfence_system(); }
}
void launchDispFlagsReset(int* flags, hipStream_t s) { hipLaunchKernelGGL(k_dispFlagsReset, dim3(1), dim3(64), 0, s, (volatile int*)flags); }

template void launchGatherPositions<float>(const void*, int, int, const int*, const float*, Vec<float>::T4*, int, float*, int, const GatherCells<float>&, hipStream_t);
template void launchGatherPositions<double>(const void*, int, int, const int*, const double*, Vec<double>::T4*, int, double*, int, const GatherCells<double>&, hipStream_t);
template void launchFinishForces<float>(const float*, const float*, const float*, int, int, const float*, const float*, const float*, const int*, int, void*, int, int, hipStream_t);
template void launchFinishForces<double>(const double*, const double*, const double*, int, int, const double*, const double*, const double*, const int*, int, void*, int, int, hipStream_t);

}  // namespace snb
