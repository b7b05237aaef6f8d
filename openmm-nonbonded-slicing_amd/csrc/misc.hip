// misc.hip -- per-step data movement kernels of the engine: positions user order -> sorted posq, forces sorted
// accumulators -> user order.  Pure HBM streaming (16 B/lane where the layout allows).
#include "snb_internal.h"

namespace snb {

// posq[s].xyz = userPos[sortedToUser[s]] + imageOffset[s]; charge (.w) is kept.  Padding slots (sortedToUser < 0)
// keep their parked far-away coordinates.  The same pass clears the six force arrays of the atom (no separate memset node in the
// step graph) and, when a Coulomb mesh is given, writes the atom's packed mesh cell for the brick spreader (pme.hip, k_pmeCells).
template <typename Real, typename In>
__global__ void k_gatherPositions(const In* __restrict__ userPos, int stride, const int* __restrict__ sortedToUser,
                                  const Real* __restrict__ imageOffset, typename Vec<Real>::T4* __restrict__ posq, int nPadded,
                                  Real* __restrict__ forces, const GatherCells<Real> gc) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nPadded) return;
    if (forces) {
#pragma unroll
        for (int k = 0; k < 7; k++) forces[(size_t)k * nPadded + s] = Real(0);      // 4 n direct-space (either layout) + 3 n reciprocal
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
            gridCoord<Real>(gc.recip, v.x, v.y, v.z, gc.nx, gc.ny, gc.nz, idx, fr);
            cell = idx[0] | (idx[1] << 10) | (idx[2] << 20);
        }
        gc.cells[s] = cell;
    }
}

template <typename Real>
void launchGatherPositions(const void* userPos, int isDouble, int stride4, const int* sortedToUser, const Real* imageOffset,
                           typename Vec<Real>::T4* posq, int nPadded, Real* forces, const GatherCells<Real>& gc, hipStream_t s) {
    if (nPadded <= 0) return;
    dim3 grid((nPadded + 255) / 256), block(256);
    const int stride = stride4 ? 4 : 3;
    if (isDouble) hipLaunchKernelGGL((k_gatherPositions<Real, double>), grid, block, 0, s, (const double*)userPos, stride, sortedToUser, imageOffset, posq, nPadded, forces, gc);
    else hipLaunchKernelGGL((k_gatherPositions<Real, float>), grid, block, 0, s, (const float*)userPos, stride, sortedToUser, imageOffset, posq, nPadded, forces, gc);
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
__global__ void k_finishForces(const Real* __restrict__ fx, const Real* __restrict__ fy, const Real* __restrict__ fz, int fs,
                               const Real* __restrict__ fpx, const Real* __restrict__ fpy, const Real* __restrict__ fpz,
                               const int* __restrict__ userToSorted, int nAtoms, Out* __restrict__ out, int accumulate) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= nAtoms) return;
    const int s = userToSorted[u];
    Out x = (Out)fx[(size_t)s * fs], y = (Out)fy[(size_t)s * fs], z = (Out)fz[(size_t)s * fs];
    if (fpx) { x += (Out)fpx[s]; y += (Out)fpy[s]; z += (Out)fpz[s]; }
    if (accumulate) { x += out[3 * (size_t)u]; y += out[3 * (size_t)u + 1]; z += out[3 * (size_t)u + 2]; }
    out[3 * (size_t)u] = x; out[3 * (size_t)u + 1] = y; out[3 * (size_t)u + 2] = z;
}

template <typename Real>
void launchFinishForces(const Real* fx, const Real* fy, const Real* fz, int fs, const Real* fpx, const Real* fpy, const Real* fpz,
                        const int* userToSorted, int nAtoms, void* out, int isDouble, int accumulate, hipStream_t s) {
    if (nAtoms <= 0) return;
    dim3 grid((nAtoms + 255) / 256), block(256);
    if (isDouble) hipLaunchKernelGGL((k_finishForces<Real, double>), grid, block, 0, s, fx, fy, fz, fs, fpx, fpy, fpz, userToSorted, nAtoms, (double*)out, accumulate);
    else hipLaunchKernelGGL((k_finishForces<Real, float>), grid, block, 0, s, fx, fy, fz, fs, fpx, fpy, fpz, userToSorted, nAtoms, (float*)out, accumulate);
}

// Raw slice energies: the kernels of an energy step add into SNB_SLICE_E_PARTS copies of the [S][2] table (chosen by work-group);
// this sums the copies on the device, as the last kernel of the step, so that an energy / derivative step needs no host
// synchronisation of its own and can be replayed from a graph -- the host reads the 2 S doubles when the caller asks for them.
__global__ void k_sumSliceParts(const double* __restrict__ parts, double* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double acc = 0;
    for (int part = 0; part < SNB_SLICE_E_PARTS; part++) acc += parts[(size_t)part * n + i];
    out[i] = acc;
}
void launchSumSliceParts(const double* parts, double* out, int n, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_sumSliceParts, dim3((n + 63) / 64), dim3(64), 0, s, parts, out, n);
}

template void launchGatherPositions<float>(const void*, int, int, const int*, const float*, Vec<float>::T4*, int, float*, const GatherCells<float>&, hipStream_t);
template void launchGatherPositions<double>(const void*, int, int, const int*, const double*, Vec<double>::T4*, int, double*, const GatherCells<double>&, hipStream_t);
template void launchFinishForces<float>(const float*, const float*, const float*, int, const float*, const float*, const float*, const int*, int, void*, int, int, hipStream_t);
template void launchFinishForces<double>(const double*, const double*, const double*, int, const double*, const double*, const double*, const int*, int, void*, int, int, hipStream_t);

}  // namespace snb
