// neighbor.hip -- GPU construction of the sorted atom order and the tile lists (rectangular periodic boxes).
//
// Stands where the reference relies on OpenMM utilities (third-party): ComputeSort for the PME atom order
// (CommonNonbondedSlicingKernels.cpp:515,1249) and NonbondedUtilities' block bounding boxes / neighbour list /
// exclusion tiles (registered at :721).  MI355X version: one radix sort by (subset, serpentine xy column, z), then one
// wavefront per 32-atom block gathers its j-atoms column by column (two-round 64-way search of the z-sorted run, AABB
// distance test, ballot compaction into LDS), builds the exclusion masks against the LDS-resident list and publishes its
// tiles and work items with one atomic allocation each.
#include "snb_internal.h"
#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>

namespace snb {

// ---- 1. sort keys -------------------------------------------------------------------------------------------------
template <typename Real, typename In>
__global__ void k_nbKeys(const NbParams<Real> p, const In* __restrict__ userPos, int stride) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= p.nAtoms) return;
    double x = (double)userPos[(size_t)u * stride], y = (double)userPos[(size_t)u * stride + 1], z = (double)userPos[(size_t)u * stride + 2];
    const double wx = x - floor(x / p.boxd[0]) * p.boxd[0], wy = y - floor(y / p.boxd[1]) * p.boxd[1], wz = z - floor(z / p.boxd[2]) * p.boxd[2];
    p.wrapped[3 * (size_t)u] = (Real)wx; p.wrapped[3 * (size_t)u + 1] = (Real)wy; p.wrapped[3 * (size_t)u + 2] = (Real)wz;
    p.offsetU[3 * (size_t)u] = (Real)(wx - x); p.offsetU[3 * (size_t)u + 1] = (Real)(wy - y); p.offsetU[3 * (size_t)u + 2] = (Real)(wz - z);
    int cx = (int)(wx / p.boxd[0] * p.ncx); cx = cx < 0 ? 0 : (cx >= p.ncx ? p.ncx - 1 : cx);
    int cy = (int)(wy / p.boxd[1] * p.ncy); cy = cy < 0 ? 0 : (cy >= p.ncy ? p.ncy - 1 : cy);
    const int serp = cx * p.ncy + ((cx & 1) ? (p.ncy - 1 - cy) : cy);
    double zf = wz / p.boxd[2]; zf = zf < 0 ? 0 : (zf > 1 ? 1 : zf);
    if (serp & 1) zf = 1.0 - zf;
    const unsigned long long zq = (unsigned long long)(zf * 1048575.0);
    p.keysIn[u] = ((unsigned long long)p.uSubset[u] << 44) | ((unsigned long long)serp << 20) | zq;
    p.valsIn[u] = u;
}

// ---- 2. scatter into the padded sorted order ---------------------------------------------------------------------
template <typename Real> __global__ void k_nbScatter(const NbParams<Real> p) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.nAtoms) return;
    const unsigned long long key = p.keysOut[t];
    const int u = p.valsOut[t];
    const int s = (int)(key >> 44);
    const int serp = (int)((key >> 20) & 0xFFFFFF);
    const int si = p.subsetPaddedStart[s] + (t - p.subsetStart[s]);
    p.sortedToUser[si] = u; p.userToSorted[u] = si;
    typename Vec<Real>::T4 v; v.x = p.wrapped[3 * (size_t)u]; v.y = p.wrapped[3 * (size_t)u + 1]; v.z = p.wrapped[3 * (size_t)u + 2]; v.w = p.uCharge[u];
    p.posq[si] = v;
    p.sigeps[si] = p.uSigEps[u];
    p.imageOffset[3 * (size_t)si] = p.offsetU[3 * (size_t)u]; p.imageOffset[3 * (size_t)si + 1] = p.offsetU[3 * (size_t)u + 1]; p.imageOffset[3 * (size_t)si + 2] = p.offsetU[3 * (size_t)u + 2];
    p.atomSubset[si] = s; p.atomGrid[si] = p.slotOfSubset[s];
    // column bookkeeping: linear (non-serpentine) column id, run boundaries of (subset, column)
    const int cx = serp / p.ncy, cyS = serp - cx * p.ncy;
    const int cy = (cx & 1) ? (p.ncy - 1 - cyS) : cyS;
    const int col = cx * p.ncy + cy;
    const bool first = (t == 0) || ((p.keysOut[t - 1] >> 20) != (key >> 20));
    const bool last = (t == p.nAtoms - 1) || ((p.keysOut[t + 1] >> 20) != (key >> 20));
    if (first) p.colRange[(size_t)s * p.ncx * p.ncy + col].x = si;
    if (last) p.colRange[(size_t)s * p.ncx * p.ncy + col].y = si + 1;
}

// padding slots: static far-away coordinates with zero parameters (they are also masked out of every tile)
template <typename Real> __global__ void k_nbPad(const NbParams<Real> p) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= p.nPadded) return;
    if (p.padFlag[s]) {
        typename Vec<Real>::T4 v; v.x = (Real)(1e9 + 1e6 * (s & 4095)); v.y = (Real)2e9; v.z = (Real)-3e9; v.w = 0;
        p.posq[s] = v;
        typename Vec<Real>::T2 z; z.x = 0; z.y = 0; p.sigeps[s] = z;
        p.sortedToUser[s] = -1; p.atomSubset[s] = -1; p.atomGrid[s] = -1;
        p.imageOffset[3 * (size_t)s] = 0; p.imageOffset[3 * (size_t)s + 1] = 0; p.imageOffset[3 * (size_t)s + 2] = 0;
    }
}

// ---- 3. block bounding boxes ---------------------------------------------------------------------------------------
template <typename Real> __global__ void k_nbBounds(const NbParams<Real> p) {
    const int b = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int k = threadIdx.x & 31;
    if (b >= p.nBlocks) return;
    const int s = b * 32 + k;
    const bool real = p.sortedToUser[s] >= 0;
    const auto v = p.posq[s];
    float mn[3] = {real ? (float)v.x : 3e38f, real ? (float)v.y : 3e38f, real ? (float)v.z : 3e38f};
    float mx[3] = {real ? (float)v.x : -3e38f, real ? (float)v.y : -3e38f, real ? (float)v.z : -3e38f};
#pragma unroll
    for (int o = 16; o > 0; o >>= 1)
#pragma unroll
        for (int d = 0; d < 3; d++) { mn[d] = fminf(mn[d], __shfl_xor(mn[d], o, 64)); mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], o, 64)); }
    if (k == 0) {
        bool tooWide = false;
        for (int d = 0; d < 3; d++) {
            p.blockCenter[3 * b + d] = 0.5f * (mn[d] + mx[d]); p.blockHalf[3 * b + d] = 0.5f * (mx[d] - mn[d]) + 1e-5f;
            if ((mx[d] - mn[d]) + 2.f * p.listCutoff >= (float)p.boxd[d]) tooWide = true;   // one image per j-atom needs extent + 2R < L
        }
        if (tooWide) atomicAdd(&p.counters[3], 1);
    }
}

// ---- 4. tiles ------------------------------------------------------------------------------------------------------
__device__ inline bool ownsPair(int I, int J) { return ((I + J) & 1) ? (I > J) : (I < J); }
__device__ inline int lanePrefix(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0)); }

constexpr int NB_CAP = 1024;        // j entries gathered per published chunk (32 tiles); larger neighbourhoods publish several chunks
constexpr int NB_MAXT = NB_CAP / 32;

// First index in the z-sorted run [a, b) whose z is not "before" zq (ascending runs: before = z < zq; descending runs:
// before = z > zq); b if every element is before.  64-way search: each round probes 64 evenly spaced elements.
template <typename Real> __device__ inline int runLowerBound(const NbParams<Real>& p, int a, int b, float zq, bool asc, int lane) {
    int lo = a, hi = b;
    while (hi > lo) {
        const int len = hi - lo;
        const int step = (len + 63) / 64;
        const int idx = lo + lane * step;
        bool before = false;
        if (idx < hi) { const float z = (float)p.posq[idx].z; before = asc ? (z < zq) : (z > zq); }
        const int nb = __popcll(__ballot(before));      // monotone run: exactly the first nb probes are "before"
        if (nb == 0) return lo;
        const int newLo = lo + (nb - 1) * step + 1;
        int newHi = lo + nb * step; if (newHi > hi) newHi = hi;
        lo = newLo; hi = newHi;
        if (step == 1) return lo;
    }
    return lo;
}

template <typename Real> __global__ __launch_bounds__(256) void k_nbBuildTiles(const NbParams<Real> p) {
    __shared__ int s_list[4][NB_CAP];
    __shared__ unsigned s_mask[4][NB_MAXT][32];
    __shared__ int s_tileSub[4][NB_MAXT];
    __shared__ int s_query[4][128];      // exclusion partners (sorted index) still to be located in the gathered list
    __shared__ int s_qrow[4][128];       // ... and the i-row each belongs to
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int I = blockIdx.x * 4 + wid;
    if (I >= p.nBlocks) return;
    int* list = s_list[wid];
    unsigned (*mask)[32] = s_mask[wid];
    int* tileSub = s_tileSub[wid];
    int* query = s_query[wid];
    int* qrow = s_qrow[wid];
    const float R = p.listCutoff, R2 = R * R;
    const float cxx = p.blockCenter[3 * I], cyy = p.blockCenter[3 * I + 1], czz = p.blockCenter[3 * I + 2];
    const float hx = p.blockHalf[3 * I], hy = p.blockHalf[3 * I + 1], hz = p.blockHalf[3 * I + 2];
    const float Lx = (float)p.boxd[0], Ly = (float)p.boxd[1], Lz = (float)p.boxd[2];
    const float colW = Lx / p.ncx, colH = Ly / p.ncy;
    const int il = lane & 31, half = lane >> 5;
    const int uI = p.sortedToUser[I * 32 + il];
    bool failed = false;

    // Masks (diagonal rule, padding, exclusions) for the `count` entries gathered so far, then publication of those
    // tiles and their work items.  A block whose neighbourhood exceeds the LDS list is published in several such chunks.
    auto flush = [&](int count, bool hasDiag) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int nT = count >> 5;
        for (int k = lane; k < nT * 32; k += 64) mask[k >> 5][k & 31] = 0u;
        __builtin_amdgcn_wave_barrier();
        if (hasDiag && half == 0) { unsigned m = 0; for (int j = 0; j <= il; j++) m |= 1u << j; mask[0][il] = m; }   // keep j > i only
        __builtin_amdgcn_wave_barrier();
        for (int t = half; t < nT; t += 2) {     // padded j slots are masked for every row, padded i rows entirely
            const int e = list[t * 32 + il];
            const unsigned long long bal = __ballot(e == -1);
            const unsigned jPad = half ? (unsigned)(bal >> 32) : (unsigned)bal;
            const unsigned row = (uI < 0) ? 0xFFFFFFFFu : jPad;
            if (row) atomicOr(&mask[t][il], row);
        }
        __builtin_amdgcn_wave_barrier();
        // exclusions: partners inside the block hit the diagonal tile directly; the others are looked up in the list
        auto resolve = [&](int nq) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int qi = 0; qi < nq; qi++) {
                const int target = query[qi], row = qrow[qi];
                for (int k2 = (hasDiag ? 32 : 0) + lane; k2 < count; k2 += 64)
                    if (list[k2] != -1 && (list[k2] & SNB_JIDX_MASK) == target) atomicOr(&mask[k2 >> 5][row], 1u << (k2 & 31));
            }
            __builtin_amdgcn_wave_barrier();
        };
        int nq = 0;
        const int e0 = uI >= 0 ? p.uExclStart[uI] : 0, e1 = uI >= 0 ? p.uExclStart[uI + 1] : 0;
        int maxLen = e1 - e0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(maxLen, o, 64); maxLen = maxLen > other ? maxLen : other; }
        for (int k = 0; k < maxLen; k++) {
            bool want = false; int sp = -1;
            if (half == 0 && e0 + k < e1) {
                sp = p.userToSorted[p.uExclList[e0 + k]];
                const int J = sp >> 5;
                if (J == I) { if (hasDiag) atomicOr(&mask[0][il], 1u << (sp & 31)); }
                else want = ownsPair(I, J);
            }
            const unsigned long long m = __ballot(want);
            const int nNew = __popcll(m);
            if (nq + nNew > 128) { resolve(nq); nq = 0; }
            if (want) { const int o = nq + lanePrefix(m); query[o] = sp; qrow[o] = il; }
            nq += nNew;
        }
        resolve(nq);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // publish
        // full (8-tile) work items and the at most one partial item go to separate queues: the engine appends the partial
        // ones behind the full ones so that the short items fill the tail of the launch
        int first = 0, w0 = 0, wp = 0;
        const int nFull = nT / 8, nPart = (nT & 7) ? 1 : 0;
        if (lane == 0) { first = atomicAdd(&p.counters[0], nT); w0 = atomicAdd(&p.counters[1], nFull); wp = nPart ? atomicAdd(&p.counters[4], 1) : 0; }
        first = __builtin_amdgcn_readfirstlane(first); w0 = __builtin_amdgcn_readfirstlane(w0); wp = __builtin_amdgcn_readfirstlane(wp);
        if (first + nT > p.tileCapacity || w0 + nFull > p.workCapacity || wp + nPart > p.workCapacity) { failed = true; return; }
        for (int k = lane; k < count; k += 64) p.tileJ[(size_t)first * 32 + k] = list[k];
        for (int t = 0; t < nT; t++) {
            const unsigned row = (lane < 32) ? mask[t][lane] : 0u;
            const bool any = __ballot(row != 0u) != 0ull;
            int mi = -1;
            if (any) {
                if (lane == 0) mi = atomicAdd(&p.counters[2], 1);
                mi = __builtin_amdgcn_readfirstlane(mi);
                if (mi >= p.maskCapacity) { failed = true; return; }
                if (lane < 32) p.masks[(size_t)mi * 32 + lane] = row;
            }
            if (lane == 0) p.tileInfo[first + t] = make_int4(tileSub[t], mi, 0, 0);
        }
        for (int k = lane; k < nFull; k += 64) p.workItems[w0 + k] = make_int4(I, first + 8 * k, 8, 0);
        if (nPart && lane == 0) p.workItemsPartial[wp] = make_int4(I, first + 8 * nFull, nT & 7, 0);
        __builtin_amdgcn_wave_barrier();
    };

    // diagonal tile
    if (lane < 32) list[lane] = (uI >= 0) ? ((I * 32 + lane) | (13 << SNB_JSHIFT_BITS)) : -1;
    if (lane == 0) tileSub[0] = p.blockSubset[I];
    int count = 32;
    bool hasDiag = true;

    const int cx0 = (int)floorf((cxx - hx - R) / colW), cx1 = (int)floorf((cxx + hx + R) / colW);
    const int cy0 = (int)floorf((cyy - hy - R) / colH), cy1 = (int)floorf((cyy + hy + R) / colH);
    const float zlo = czz - hz - R, zhi = czz + hz + R;
    for (int s = 0; s < p.nSubsets && !failed; s++) {
        int segStart = count;
        const int2* ranges = p.colRange + (size_t)s * p.ncx * p.ncy;
        for (int gx = cx0; gx <= cx1 && !failed; gx++) {
            const int kx = (gx < 0) ? -1 : (gx >= p.ncx ? 1 : 0);
            const int ccx = gx - kx * p.ncx;
            if (ccx < 0 || ccx >= p.ncx) continue;
            for (int gy = cy0; gy <= cy1 && !failed; gy++) {
                const int ky = (gy < 0) ? -1 : (gy >= p.ncy ? 1 : 0);
                const int ccy = gy - ky * p.ncy;
                if (ccy < 0 || ccy >= p.ncy) continue;
                const int2 rg = ranges[ccx * p.ncy + ccy];
                if (rg.y <= rg.x) continue;
                const int serp = ccx * p.ncy + ((ccx & 1) ? (p.ncy - 1 - ccy) : ccy);
                const bool asc = (serp & 1) == 0;
                for (int kz = -1; kz <= 1 && !failed; kz++) {
                    const float a = zlo - kz * Lz, b = zhi - kz * Lz;     // wanted z interval in the primary cell
                    if (b < 0.f || a >= Lz) continue;
                    const float eps = 2e-6f * Lz + 1e-6f;                 // quantised sort keys: widen by a hair
                    int i0, i1;
                    if (asc) { i0 = runLowerBound<Real>(p, rg.x, rg.y, a - eps, true, lane); i1 = runLowerBound<Real>(p, i0, rg.y, b + eps, true, lane); }
                    else { i0 = runLowerBound<Real>(p, rg.x, rg.y, b + eps, false, lane); i1 = runLowerBound<Real>(p, i0, rg.y, a - eps, false, lane); }
                    const float sx = kx * Lx, sy = ky * Ly, sz = kz * Lz;
                    const int code = (kx + 1) * 9 + (ky + 1) * 3 + (kz + 1);
                    for (int base = i0; base < i1 && !failed; base += 64) {
                        const int j = base + lane;
                        bool ok = j < i1;
                        if (ok) { const int J = j >> 5; ok = (J != I) && ownsPair(I, J); }
                        if (ok) {
                            const auto v = p.posq[j];
                            float dx = fabsf((float)v.x + sx - cxx) - hx, dy = fabsf((float)v.y + sy - cyy) - hy, dz = fabsf((float)v.z + sz - czz) - hz;
                            dx = dx > 0 ? dx : 0; dy = dy > 0 ? dy : 0; dz = dz > 0 ? dz : 0;
                            ok = dx * dx + dy * dy + dz * dz < R2;
                        }
                        const unsigned long long m = __ballot(ok);
                        const int nNew = __popcll(m);
                        if (count + nNew > NB_CAP - 32) {
                            // list full: close the current segment, publish this chunk and start a fresh list
                            const int padded = (count + 31) & ~31;
                            for (int k = count + lane; k < padded; k += 64) list[k] = -1;
                            for (int t = (segStart >> 5) + lane; t < (padded >> 5); t += 64) tileSub[t] = s;
                            flush(padded, hasDiag);
                            hasDiag = false; count = 0; segStart = 0;
                        }
                        if (ok) list[count + lanePrefix(m)] = j | (code << SNB_JSHIFT_BITS);
                        count += nNew;
                    }
                }
            }
        }
        // close the subset segment: pad to a whole tile, record the tiles' subset
        const int padded = (count + 31) & ~31;
        for (int k = count + lane; k < padded; k += 64) list[k] = -1;
        for (int t = (segStart >> 5) + lane; t < (padded >> 5); t += 64) tileSub[t] = s;
        count = padded;
    }
    if (!failed && count > 0) flush(count, hasDiag);
    if (failed && lane == 0) atomicAdd(&p.counters[3], 1);
}

// ---- driver ---------------------------------------------------------------------------------------------------------
template <typename Real> size_t nbSortTempBytes(int n) {
    size_t bytes = 0;
    unsigned long long* k = nullptr; int* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)n, 0, 64, (hipStream_t)0);
    return bytes;
}

template <typename Real> void launchNeighborBuild(const NbParams<Real>& p, const void* userPos, int isDouble, int stride4, void* sortTemp, size_t sortTempBytes, hipStream_t s) {
    const int n = p.nAtoms;
    const int stride = stride4 ? 4 : 3;
    dim3 block(256), gridN((n + 255) / 256);
    (void)hipMemsetAsync(p.counters, 0, sizeof(int) * 8, s);
    (void)hipMemsetAsync(p.colRange, 0, sizeof(int2) * (size_t)p.nSubsets * p.ncx * p.ncy, s);
    if (n > 0) {
        if (isDouble) hipLaunchKernelGGL((k_nbKeys<Real, double>), gridN, block, 0, s, p, (const double*)userPos, stride);
        else hipLaunchKernelGGL((k_nbKeys<Real, float>), gridN, block, 0, s, p, (const float*)userPos, stride);
        (void)rocprim::radix_sort_pairs(sortTemp, sortTempBytes, p.keysIn, p.keysOut, p.valsIn, p.valsOut, (size_t)n, 0, 44 + p.subsetBits, s);
        hipLaunchKernelGGL((k_nbPad<Real>), dim3((p.nPadded + 255) / 256), block, 0, s, p);
        hipLaunchKernelGGL((k_nbScatter<Real>), gridN, block, 0, s, p);
        hipLaunchKernelGGL((k_nbBounds<Real>), dim3((p.nBlocks + 7) / 8), block, 0, s, p);
        hipLaunchKernelGGL((k_nbBuildTiles<Real>), dim3((p.nBlocks + 3) / 4), block, 0, s, p);
    }
}

template size_t nbSortTempBytes<float>(int);
template size_t nbSortTempBytes<double>(int);
template void launchNeighborBuild<float>(const NbParams<float>&, const void*, int, int, void*, size_t, hipStream_t);
template void launchNeighborBuild<double>(const NbParams<double>&, const void*, int, int, void*, size_t, hipStream_t);

}  // namespace snb
