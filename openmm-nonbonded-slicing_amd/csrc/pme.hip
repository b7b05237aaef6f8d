// pme.hip -- sliced PME reciprocal pipeline for gfx950 (MI355X): B-spline charge spreading onto one grid per
// subset, hand-written batched 3D real FFT (mixed radix 2/3/4/5/7 Stockham in LDS), fused
// [x-FFT -> per-slice energy -> lambda-mixed convolution -> inverse x-FFT], and force interpolation from ONE
// pre-mixed potential grid per atom.
//
// Replaces platforms/common/src/kernels/pme.cc (findAtomGridIndex :1-22, gridSpreadCharge :24-122,
// reciprocalConvolution :138-189, gridEvaluateEnergy :191-274, gridInterpolateForce :276-391) and the cuFFT/VkFFT
// back-ends (platforms/cuda/src/CudaCuFFT3D.cpp, CudaVkFFT3D.cpp).  Arithmetic parity is with
// platforms/reference/src/ReferencePME.cpp:196-256 (index/fraction), :264-317 (B-splines), :320-396 (spread),
// :400-595 (convolution + sliced energy), :598-702 (interpolation).
//
// Differences from the reference GPU design, chosen for MI355X:
//  * the lambda mix  phi_I = sum_J lambda_IJ * eterm * S_J  is done in k-space inside the convolution kernel, so the
//    interpolation gathers 125 values per atom instead of 125*n (pme.cc:360-371);
//  * the last forward axis and the first inverse axis are the same axis, so both FFTs of that axis, the energy
//    evaluation and the convolution run in one kernel without leaving LDS: 5 grid passes instead of 8;
//  * the real axis (z) is transformed with the imaginary half implied, writing nz/2+1 complex outputs.
#include "snb_internal.h"
#include <cstdlib>
#include <type_traits>
#include <algorithm>
#include <cstdlib>

namespace snb {

thread_local KernelStamps* g_stamps = nullptr;
template <typename Real> static inline int stampSlot(const PmeParams<Real>& p, int k) { return k + (p.dispersion ? 8 : 0); }

// Wave priority of the reciprocal pipeline's front kernels (the ones an overlapped step runs beside the resident pair kernel, engine.hip
// overlapMode).  0 = none: the arbiter then serves the older pair-kernel waves first (measured best, docs/MEASUREMENT_LOG.md round 4).
#ifndef SNB_PME_PRIO_LEVEL
#define SNB_PME_PRIO_LEVEL 0
#endif
#if SNB_PME_PRIO_LEVEL > 0
#define SNB_PME_PRIO() __builtin_amdgcn_s_setprio(SNB_PME_PRIO_LEVEL)
#else
#define SNB_PME_PRIO() do {} while (0)
#endif

// x / d for 0 <= x < 2^22 without the ~35-instruction integer division: (x + 0.5) * (1/d) never lands within rounding error of an integer.
// Every dividend in this file is bounded by a work-group's LDS element count, a brick's line count or a mesh plane / slab of at most
// 1024 x 1024 points (PmePlan::init in engine.hip rejects larger meshes).
struct FastDiv {
    float inv; int d;
    __device__ explicit FastDiv(int d_) : inv(1.0f / (float)d_), d(d_) {}
    __device__ int div(int x) const { return (int)(((float)x + 0.5f) * inv); }
};

// LDS stride of a z line in k_spreadOwn's region: a stride that is a multiple of 64 dwords puts every line on the same banks (RZ = 64 for
// sz = 60: lanes = z-neighbours of one column then collide ~13-way: 47.7 us on c3 against 27.8); a few extra values per line spread them
template <bool FIXED> __host__ __device__ inline int ownLineStride(int RZ) {
    const int per = FIXED ? 4 : 2;                                  // values per 16 bytes: lines stay 16-byte aligned for the vector copy
    int st = (RZ + per - 1) / per * per;
    while (((st * (FIXED ? 1 : 2)) & 7) != 4) st += per;            // stride = 4 (mod 8) dwords: consecutive lines walk over all 64 banks
    return st;
}

template <typename Real> struct Cx { Real x, y; };
template <typename Real, int R1 = 0, int R2 = 0>
__device__ inline Cx<Real>* fftLines(Cx<Real>* a, Cx<Real>* b, int n, const int* factors, int nf, int sign, const Cx<Real>* tw, int nb, int BS, int tid, int nthreads);
template <typename Real> __device__ inline Cx<Real> cmul(Cx<Real> a, Cx<Real> b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

// ---------------------------------------------------------------------------------------------------
// B-splines of order 5 and their derivatives (ReferencePME.cpp:264-317), in registers.
// ---------------------------------------------------------------------------------------------------
template <typename Real> __device__ inline void bspline5(Real dr, Real* d, Real* dd) {
    d[4] = 0; d[1] = dr; d[0] = 1 - dr; d[2] = 0; d[3] = 0;
    // k = 3
    {
        const Real div = Real(0.5);
        d[2] = div * dr * d[1];
        d[1] = div * ((dr + 1) * d[0] + (2 - dr) * d[1]);
        d[0] = div * (1 - dr) * d[0];
    }
    // k = 4
    {
        const Real div = Real(1.0 / 3.0);
        d[3] = div * dr * d[2];
        d[2] = div * ((dr + 1) * d[1] + (3 - dr) * d[2]);
        d[1] = div * ((dr + 2) * d[0] + (2 - dr) * d[1]);
        d[0] = div * (1 - dr) * d[0];
    }
    dd[0] = -d[0];
    dd[1] = d[0] - d[1]; dd[2] = d[1] - d[2]; dd[3] = d[2] - d[3]; dd[4] = d[3] - d[4];
    {
        const Real div = Real(0.25);
        d[4] = div * dr * d[3];
        d[3] = div * ((dr + 1) * d[2] + (4 - dr) * d[3]);
        d[2] = div * ((dr + 2) * d[1] + (3 - dr) * d[2]);
        d[1] = div * ((dr + 3) * d[0] + (2 - dr) * d[1]);
        d[0] = div * (1 - dr) * d[0];
    }
}

// grid index + fraction (ReferencePME.cpp:245-254)
template <typename Real> __device__ inline Real pmeCharge(const PmeParams<Real>& p, int atom) {
    if (p.dispersion) { const auto se = p.sigeps[atom]; return Real(8) * se.x * se.x * se.x * se.y; }   // c6_i (ReferenceSlicedLJCoulombIxn.cpp:247)
    return p.posq[atom].w;
}

// ---------------------------------------------------------------------------------------------------
// Spreading: 32 lanes per atom, lane = one (x,y) stencil row, 5 float atomics along z.
// (ReferencePME.cpp:320-396).  The grids were cleared by a memset on the same stream.
// ---------------------------------------------------------------------------------------------------
template <typename Real> __global__ __launch_bounds__(256) void k_spread(const PmeParams<Real> p) {
    const int gid = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int r = threadIdx.x & 31;
    if (gid >= p.natoms || r >= 25) return;
    const int slot = p.atomGrid[gid];
    if (slot < 0) return;
    const Real q = pmeCharge(p, gid);
    if (q == Real(0)) return;
    const auto pos = p.posq[gid];
    int idx[3]; Real fr[3];
    gridCoord<Real>(p.recip, p.recipLo, pos.x, pos.y, pos.z, p.d.nx, p.d.ny, p.d.nz, idx, fr);
    Real tx[5], ty[5], tz[5], dtmp[5];
    bspline5<Real>(fr[0], tx, dtmp); bspline5<Real>(fr[1], ty, dtmp); bspline5<Real>(fr[2], tz, dtmp);
    const int ix = r / 5, iy = r - ix * 5;
    int xi = idx[0] + ix; if (xi >= p.d.nx) xi -= p.d.nx;
    int yi = idx[1] + iy; if (yi >= p.d.ny) yi -= p.d.ny;
    Real wxy = q;
#pragma unroll
    for (int k = 0; k < 5; k++) { if (k == ix) wxy *= tx[k]; }
#pragma unroll
    for (int k = 0; k < 5; k++) { if (k == iy) wxy *= ty[k]; }
    Real* row = p.gridReal + (((size_t)slot * p.d.nx + xi) * p.d.ny + yi) * p.d.nz;
#pragma unroll
    for (int iz = 0; iz < 5; iz++) {
        int zi = idx[2] + iz; if (zi >= p.d.nz) zi -= p.d.nz;
        __hip_atomic_fetch_add(&row[zi], wxy * tz[iz], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// packed mesh cell (10 bits per axis) of every sorted atom for the brick spreader's candidate scan, -1 for atoms that carry no charge
// on this mesh (padding included): one 4-byte load and three compares per candidate instead of a position load and gridCoord
template <typename Real> __global__ __launch_bounds__(256) void k_pmeCells(const PmeParams<Real> p) {
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= p.natoms) return;
    int cell = -1;
    if (p.atomGrid[a] >= 0 && pmeCharge(p, a) != Real(0)) {
        const auto pos = p.posq[a];
        int idx[3]; Real fr[3];
        gridCoord<Real>(p.recip, p.recipLo, pos.x, pos.y, pos.z, p.d.nx, p.d.ny, p.d.nz, idx, fr);
        cell = idx[0] | (idx[1] << 10) | (idx[2] << 20);
    }
    p.cells[a] = cell;
}

// ---------------------------------------------------------------------------------------------------
// Brick spreading (the production path for rectangular boxes): no global atomics, no grid memset.
// One work-group owns a (cx x cy x nz) column of one subset grid in LDS (as doubles: ds_add_f64 runs at ~9 cycles
// per wave-instruction on gfx950, ds_add_f32 at ~190 -- measured, tools/ubench_atomics.hip).  Atoms are already sorted
// by (subset, xy-column of exactly this size, z) for the pair kernel, so the candidates of a brick are the contiguous
// sorted ranges of its 3x3 column neighbourhood (one column of margin on each side covers the drift since the last
// re-sort: < skin/2 < one grid cell).  Every grid point is written exactly once, coalesced along z.
// ---------------------------------------------------------------------------------------------------
template <typename Real, bool FIXED, bool FUSEZ> __global__ __launch_bounds__(512) void k_spreadBrick(const PmeParams<Real> p) {
    extern __shared__ __align__(16) unsigned char s_brick_raw[];
    constexpr int NT = 512, LISTCAP = FIXED ? 8192 : 4096;      // (single precision: the list shares its LDS with the z-FFT buffers of the fused pass, which are larger)
    // a brick spans groupX x groupY sort columns (1 x 1 for the Coulomb mesh; more when a coarser mesh makes one column < 5 cells)
    const int ncx = p.sortNcx, ncy = p.sortNcy, nz = p.d.nz;
    const int cx = p.groupX * (p.d.nx / ncx), cy = p.groupY * (p.d.ny / ncy);       // brick size in cells of THIS mesh
    const int nbx = ncx / p.groupX, nby = ncy / p.groupY;
    const int ncol = ncx * ncy;
    // ... and one of zSlabs slabs along z (more, smaller work-groups: the bulk subset alone has only ~#columns busy bricks)
    const int nSlabs = FUSEZ ? 1 : p.zSlabs;      // (the fused-z instantiation is launched with one slab only)
    const int sz = nz / nSlabs;
    const int zs = blockIdx.x % nSlabs, brickId = blockIdx.x / nSlabs;
    const int slot = brickId / (nbx * nby), bcol = brickId - slot * (nbx * nby);
    const int Bx = bcol / nby, By = bcol - Bx * nby;
    const int x0 = Bx * cx, y0 = By * cy, z0 = zs * sz;
    const int npts = cx * cy * sz;
    // accumulation type in LDS: doubles with ds_add_f64 (double precision), or -- single precision -- 32-bit fixed point with two
    // z-adjacent points packed per ds_add_u64 (sign-extended low half, so the 64-bit sum is exact): 15 instead of 25 LDS atomics
    // per x-line (ds_add_f32 is ~20x slower than either on this chip, tools/ubench_atomics.hip).  sz and nz are even on this path.
    using Acc = typename std::conditional<FIXED, int, double>::type;
    Acc* brick = reinterpret_cast<Acc*>(s_brick_raw);
    int* list = reinterpret_cast<int*>(s_brick_raw + ((sizeof(Acc) * (size_t)npts + 15) & ~(size_t)15));       // [LISTCAP] (atom, x-line) entries
    __shared__ int s_count;
    __shared__ int s_rangeBegin[64], s_rangePrefix[65];
    const int tid = threadIdx.x;
    const Real fixScale = FIXED ? p.fixDev[0] : Real(1), fixInv = FIXED ? p.fixDev[1] : Real(1);
    for (int i = tid; i < npts; i += NT) brick[i] = Acc(0);
    if (tid == 0) s_count = 0;
    const int2* ranges = p.colRange + (size_t)p.gridSubset[slot] * ncol;
    // candidate columns: the brick's own columns, mLo columns below (stencil reach 4 cells + 1 cell of drift) and one above
    // (drift), each distinct column once; their atom ranges are concatenated into one virtual index space
    const int cpcx = p.d.nx / ncx, cpcy = p.d.ny / ncy;
    const int mLoX = (5 + cpcx - 1) / cpcx, mLoY = (5 + cpcy - 1) / cpcy;
    const int nvx = (p.groupX + mLoX + 1 < ncx) ? p.groupX + mLoX + 1 : ncx, nvy = (p.groupY + mLoY + 1 < ncy) ? p.groupY + mLoY + 1 : ncy;
    const int nr = nvx * nvy;                       // <= 7 * 7
    if (tid < nr) {
        const int tx = tid / nvy, ty = tid - tx * nvy;
        int ccx = (Bx * p.groupX - mLoX + tx) % ncx; if (ccx < 0) ccx += ncx;
        int ccy = (By * p.groupY - mLoY + ty) % ncy; if (ccy < 0) ccy += ncy;
        const int2 rg = ranges[ccx * ncy + ccy];
        s_rangeBegin[tid] = rg.x; s_rangePrefix[tid + 1] = rg.y - rg.x;
    }
    __syncthreads();
    if (tid == 0) { int acc = 0; s_rangePrefix[0] = 0; for (int r = 0; r < nr; r++) { acc += s_rangePrefix[r + 1]; s_rangePrefix[r + 1] = acc; } }
    __syncthreads();
    const int total = s_rangePrefix[nr];
    const int hx = p.d.nx / 2, hy = p.d.ny / 2, hz = nz / 2;
    if (total == 0) {      // no atom of this subset anywhere near: the brick is all zeros
        if constexpr (FUSEZ) {
            const int nzc = p.d.nzc;
            Cx<Real>* o0 = reinterpret_cast<Cx<Real>*>(p.gridCplx) + (size_t)slot * p.d.nx * p.d.ny * nzc;
            const FastDiv dzc0(nzc), dcy1(cy);
            for (int i = tid; i < cx * cy * nzc; i += NT) {
                const int l = dzc0.div(i), k = i - l * nzc;
                const int lx = dcy1.div(l), ly = l - lx * cy;
                o0[((size_t)(x0 + lx) * p.d.ny + (y0 + ly)) * nzc + k] = {Real(0), Real(0)};
            }
            return;
        }
        Real* g0 = p.gridReal + (size_t)slot * p.d.nx * p.d.ny * nz;
        const FastDiv dsz0(sz), dcy0(cy);
        for (int i = tid; i < npts; i += NT) {
            const int l = dsz0.div(i), z = i - l * sz;
            const int lx = dcy0.div(l), ly = l - lx * cy;
            g0[((size_t)(x0 + lx) * p.d.ny + (y0 + ly)) * nz + z0 + z] = Real(0);
        }
        return;
    }
    const bool trace = p.trace != nullptr;      // SNB_PME_TRACE: wall-clock split of the busy work-groups (scan / entries / z FFT + store)
    long long tT = trace ? (long long)wall_clock64() : 0, tScan = 0, tEnt = 0;
    // Phase 1 takes SCAN candidates per thread and round: a round is a chain of dependent latencies (range lookup in LDS, the cell load,
    // a six-step wave scan, an LDS atomic, two barriers: about 1 us whatever the work) and most candidates are rejected -- five of the nine
    // scanned columns are there only for atoms that drifted across a column border since the last re-sort.
    constexpr int SCAN = FIXED ? 2 : 1;
    for (int base = 0; base < total; base += SCAN * NT) {
        // phase 1: one thread per candidate atom; every x-line (atom, ix) of its 5x5 footprint that falls inside the brick becomes a list entry
        int nEnt[SCAN], ixLo[SCAN], aSel[SCAN], cellOf[SCAN];
#pragma unroll
        for (int c = 0; c < SCAN; c++) {      // all loads of the round in flight together
            const int v = base + c * NT + tid;
            aSel[c] = 0; cellOf[c] = -1;
            if (v < total) {
                int r = 0;
                while (v >= s_rangePrefix[r + 1]) r++;
                aSel[c] = s_rangeBegin[r] + (v - s_rangePrefix[r]);
                cellOf[c] = p.cells[aSel[c]];            // packed mesh cell of the atom (k_pmeCells), -1 = carries no charge on this mesh
            }
        }
        int mine = 0;
#pragma unroll
        for (int c = 0; c < SCAN; c++) {
            const int cell = cellOf[c];
            nEnt[c] = 0; ixLo[c] = 0;
            const int idx[3] = {cell & 1023, (cell >> 10) & 1023, (cell >> 20) & 1023};
            int rx = idx[0] - x0; if (rx > hx) rx -= p.d.nx; else if (rx < -hx) rx += p.d.nx;
            int ry = idx[1] - y0; if (ry > hy) ry -= p.d.ny; else if (ry < -hy) ry += p.d.ny;
            int rz = idx[2] - z0; if (rz > hz) rz -= nz; else if (rz < -hz) rz += nz;
            const bool zHit = nSlabs == 1 || (rz + 4 >= 0 && rz < sz);
            if (cell >= 0 && zHit && rx + 4 >= 0 && rx < cx && ry + 4 >= 0 && ry < cy) {
                ixLo[c] = rx < 0 ? -rx : 0;                                                   // lines with 0 <= rx + ix < cx
                nEnt[c] = ((cx - rx < 5) ? cx - rx : 5) - ixLo[c];
            }
            mine += nEnt[c];
        }
        {   // wave-aggregated append: one LDS atomic per wave instead of one same-address atomic per lane
            int incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += t; }
            int waveBase = 0;
            if ((tid & 63) == 63 && incl > 0) waveBase = atomicAdd(&s_count, incl);
            waveBase = __shfl(waveBase, 63, 64);
            int slot0 = waveBase + incl - mine;
#pragma unroll
            for (int c = 0; c < SCAN; c++) { for (int k = 0; k < nEnt[c]; k++) list[slot0 + k] = (aSel[c] << 3) | (ixLo[c] + k); slot0 += nEnt[c]; }
        }
        __syncthreads();
        const int count = s_count;
        __syncthreads();                                                    // everyone has read the count before anyone appends again
        if (count <= LISTCAP - 5 * SCAN * NT && base + SCAN * NT < total) continue;      // room for another round of candidates (uniform branch)
        if (trace) { const long long t = (long long)wall_clock64(); tScan += t - tT; tT = t; }
        // phase 2: one thread per (atom, x-line): weights, then the line's 5x5 (y,z) points that fall inside the brick into LDS
        int eN = 0; Real qN = Real(0); typename Vec<Real>::T4 posN = {};
        auto fetch = [&](int k) { if (k < count) { eN = list[k]; qN = pmeCharge(p, eN >> 3); posN = p.posq[eN >> 3]; } };
        fetch(tid);
        for (int k = tid; k < count; k += NT) {
            const int e = eN;
            const int ix = e & 7;
            const Real q = qN;
            const auto pos = posN;
            fetch(k + NT);
            int idx[3]; Real fr[3];
            gridCoord<Real>(p.recip, p.recipLo, pos.x, pos.y, pos.z, p.d.nx, p.d.ny, nz, idx, fr);
            int rx = idx[0] - x0; if (rx > hx) rx -= p.d.nx; else if (rx < -hx) rx += p.d.nx;
            int ry = idx[1] - y0; if (ry > hy) ry -= p.d.ny; else if (ry < -hy) ry += p.d.ny;
            Real tx[5], ty[5], tz[5], dtmp[5];
            bspline5<Real>(fr[0], tx, dtmp); bspline5<Real>(fr[1], ty, dtmp); bspline5<Real>(fr[2], tz, dtmp);
            if ((unsigned)(rx + ix) >= (unsigned)cx) continue;   // cannot happen while the scan and this pass agree on the cell; guards the LDS bounds
            const Real wx = q * (ix == 0 ? tx[0] : ix == 1 ? tx[1] : ix == 2 ? tx[2] : ix == 3 ? tx[3] : tx[4]);
            int zi[5];      // z index inside the slab, or -1 when the point belongs to another slab
#pragma unroll
            for (int iz = 0; iz < 5; iz++) {
                int z = idx[2] + iz - z0; if (z >= nz) z -= nz; else if (z < 0) z += nz;
                zi[iz] = z < sz ? z : -1;
            }
            Acc* plane = brick + (size_t)(rx + ix) * cy * sz;
            if constexpr (FIXED) {
                // pair slots in unwrapped z: (2P, 2P+1) for P = idx[2]>>1 + {0,1,2}; nz and the slab bounds are even, so a pair never straddles
                // the periodic wrap or a slab boundary
                const bool odd = idx[2] & 1;
                int zp[3];
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    int z = (idx[2] & ~1) + 2 * j - z0; if (z >= nz) z -= nz; else if (z < 0) z += nz;
                    zp[j] = (FUSEZ || z < sz) ? z : -1;      // (FUSEZ: one slab, the brick holds whole lines -- every point is inside)
                }
                const Real wq = wx * fixScale;
#pragma unroll
                for (int iy = 0; iy < 5; iy++) {
                    const int ly = ry + iy;
                    if (ly < 0 || ly >= cy) continue;
                    const Real wxy = wq * ty[iy];
                    int v[5];
#pragma unroll
                    for (int iz = 0; iz < 5; iz++) v[iz] = __float2int_rn(wxy * tz[iz]);
                    unsigned long long* line = reinterpret_cast<unsigned long long*>(plane + ly * sz);
                    const int lo0 = odd ? 0 : v[0], hi0 = odd ? v[0] : v[1];
                    const int lo1 = odd ? v[1] : v[2], hi1 = odd ? v[2] : v[3];
                    const int lo2 = odd ? v[3] : v[4], hi2 = odd ? v[4] : 0;
                    const int lo[3] = {lo0, lo1, lo2}, hi[3] = {hi0, hi1, hi2};
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        const unsigned long long packed = ((unsigned long long)(unsigned)(hi[j] + (lo[j] >> 31)) << 32) | (unsigned)lo[j];
                        if (FUSEZ || zp[j] >= 0) __hip_atomic_fetch_add(&line[zp[j] >> 1], packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            } else {
#pragma unroll
                for (int iy = 0; iy < 5; iy++) {
                    const int ly = ry + iy;
                    if (ly < 0 || ly >= cy) continue;
                    const Real wxy = wx * ty[iy];
                    Acc* line = plane + ly * sz;
#pragma unroll
                    for (int iz = 0; iz < 5; iz++)
                        if (zi[iz] >= 0) __hip_atomic_fetch_add(&line[zi[iz]], (double)(wxy * tz[iz]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        __syncthreads();
        if (tid == 0) s_count = 0;
        __syncthreads();
        if (trace) { const long long t = (long long)wall_clock64(); tEnt += t - tT; tT = t; }
    }
    auto traceOut = [&]() {
        if (trace && tid == 0) {
            const long long t = (long long)wall_clock64();
            atomicAdd((unsigned long long*)&p.trace[4], (unsigned long long)tScan); atomicAdd((unsigned long long*)&p.trace[5], (unsigned long long)tEnt);
            atomicAdd((unsigned long long*)&p.trace[6], (unsigned long long)(t - tT)); atomicAdd((unsigned long long*)&p.trace[7], 1ull);
        }
    };
    // Reading a point back.  Fixed point: the two halves of a 64-bit word were summed as ONE signed integer, hi * 2^32 + lo, so the upper
    // word holds hi - 1 whenever the lower half's sum is negative: the borrow is returned here.  (Without it every odd-z point beside a
    // negative even-z point was low by one unit, 1.2e-8 e: a uniform spurious charge of -1e-3 e per subset grid of c4, whose interaction
    // with the boundary double layers of the OTHER water shells put -0.03 kJ/mol into every cross slice -- 3.6e-3 of an 8 kJ/mol slice.)
    auto brickValue = [&](int i) -> Real {
        if constexpr (FIXED) { const int v = (int)brick[i]; return (Real)((i & 1) ? v + (int)((unsigned)brick[i - 1] >> 31) : v); }
        else return (Real)brick[i];
    };
    if constexpr (FUSEZ) {
        // forward z FFT of the brick's own lines straight out of LDS (zSlabs == 1: the brick holds whole lines), two real lines per
        // complex transform as in k_fftZ; the real grid is never written and the separate z pass is skipped
        const int nzc = p.d.nzc, nl = cx * cy, nb = (nl + 1) >> 1, BS = nb + 1;
        Cx<Real>* A = reinterpret_cast<Cx<Real>*>(list);                  // the entry list is dead by now
        Cx<Real>* B = A + (size_t)nz * BS;
        Cx<Real>* tw = B + (size_t)nz * BS;
        for (int k = tid; k < nz; k += NT) tw[k] = reinterpret_cast<const Cx<Real>*>(p.twz)[k];
        const FastDiv dz(nz), dzc(nzc), dcy2(cy);
        const Real inv = FIXED ? fixInv : Real(1);
        for (int it = tid; it < nb * nz; it += NT) {
            const int c = dz.div(it), k = it - c * nz;
            const Real a = brickValue((2 * c) * nz + k) * inv;
            const Real b = (2 * c + 1 < nl) ? brickValue((2 * c + 1) * nz + k) * inv : Real(0);
            A[k * BS + c] = {a, b};
        }
        Cx<Real>* R = fftLines<Real, 0, 0>(A, B, nz, p.d.fz, p.d.nfz, -1, tw, nb, BS, tid, NT);
        __syncthreads();
        Cx<Real>* out = reinterpret_cast<Cx<Real>*>(p.gridCplx) + (size_t)slot * p.d.nx * p.d.ny * nzc;
        for (int it = tid; it < nb * nzc; it += NT) {
            const int c = dzc.div(it), k = it - c * nzc;
            const Cx<Real> z = R[k * BS + c], m = R[(k == 0 ? 0 : nz - k) * BS + c];
            const int l0 = 2 * c, lx0 = dcy2.div(l0), ly0 = l0 - lx0 * cy;
            out[((size_t)(x0 + lx0) * p.d.ny + (y0 + ly0)) * nzc + k] = {Real(0.5) * (z.x + m.x), Real(0.5) * (z.y - m.y)};
            if (l0 + 1 < nl) {
                const int l1 = l0 + 1, lx1 = dcy2.div(l1), ly1 = l1 - lx1 * cy;
                out[((size_t)(x0 + lx1) * p.d.ny + (y0 + ly1)) * nzc + k] = {Real(0.5) * (z.y + m.y), Real(0.5) * (m.x - z.x)};
            }
        }
        traceOut();
        return;
    }
    Real* g = p.gridReal + (size_t)slot * p.d.nx * p.d.ny * nz;
    const FastDiv dsz(sz), dcy(cy);
    for (int i = tid; i < npts; i += NT) {
        const int l = dsz.div(i), z = i - l * sz;
        const int lx = dcy.div(l), ly = l - lx * cy;
        g[((size_t)(x0 + lx) * p.d.ny + (y0 + ly)) * nz + z0 + z] = FIXED ? brickValue(i) * fixInv : (Real)brick[i];
    }
}

template <typename Real> static int launchSpreadOwn(const PmeParams<Real>& p, hipStream_t s);      // -1: not applicable, 0: real mesh written, 1: forward z FFT done too, 2: ... and written plane-major
// Returns 1 when the spreader also did the forward z FFT (launchPmeForwardFFT must then skip its z pass), 2 when it left the spectrum
// plane-major for the plane path (launchPmePlanePath replaces forward FFT, convolution and inverse FFT), else 0.
template <typename Real> int launchPmeSpread(const PmeParams<Real>& p, hipStream_t s) {
    if (p.sortNcx > 0 && p.colRange != nullptr) {
        const int cx = p.groupX * (p.d.nx / p.sortNcx), cy = p.groupY * (p.d.ny / p.sortNcy);
        static const bool noFixed = getenv("SNB_NO_FIXED_SPREAD") != nullptr;      // test switch: f64 LDS accumulation in single precision too
        const bool fixed = std::is_same<Real, float>::value && !noFixed && p.d.nz % 2 == 0 && (p.d.nz / p.zSlabs) % 2 == 0;
        const size_t accBytes = fixed ? sizeof(int) : sizeof(double);
        const size_t brickBytes = (accBytes * (size_t)cx * cy * (p.d.nz / p.zSlabs) + 15) & ~(size_t)15;
        const size_t listBytes = sizeof(int) * (fixed ? 8192 : 4096);
        const int nbz = (cx * cy + 1) / 2;
        const size_t fftBytes = sizeof(Cx<Real>) * ((size_t)2 * p.d.nz * (nbz + 1) + p.d.nz);
        static const bool noFuse = getenv("SNB_NO_FUSED_Z") != nullptr;
        const bool fuse = !noFuse && p.zSlabs == 1 && brickBytes + std::max(listBytes, fftBytes) <= 64 * 1024;
        const size_t lds = brickBytes + (fuse ? std::max(listBytes, fftBytes) : listBytes);
        const int nblocks = p.nsub * (p.sortNcx / p.groupX) * (p.sortNcy / p.groupY) * p.zSlabs;
        if (!p.cellsReady) hipLaunchKernelGGL((k_pmeCells<Real>), dim3((p.natoms + 255) / 256), dim3(256), 0, s, p);
        if (p.ownSlabs > 0) { const int r = launchSpreadOwn<Real>(p, s); if (r >= 0) return r; }
#define SNB_SPREAD(FX, FZ) { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spreadBrick<Real, FX, FZ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                             SNB_STAMPED_LAUNCH(stampSlot(p, 1), (k_spreadBrick<Real, FX, FZ>), dim3(nblocks), dim3(512), lds, s, p); }
        if constexpr (std::is_same<Real, float>::value) {
            if (fixed) { if (fuse) SNB_SPREAD(true, true) else SNB_SPREAD(true, false) return fuse ? 1 : 0; }
        }
        if (fuse) SNB_SPREAD(false, true) else SNB_SPREAD(false, false)
#undef SNB_SPREAD
        return fuse ? 1 : 0;
    }
    // fallback (triclinic boxes, meshes without a usable column divisor): global float atomics
    launchZeroFill(p.gridReal, sizeof(Real) * (size_t)p.nsub * p.d.nx * p.d.ny * p.d.nz, s);      // (a kernel, not a memset node: misc.hip)
    if (p.natoms <= 0) return 0;
    SNB_STAMPED_LAUNCH(stampSlot(p, 1), (k_spread<Real>), dim3((p.natoms + 7) / 8), dim3(256), 0, s, p);
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Stockham FFT stages in LDS.  Data layout in LDS: a[k * BS + b]  (k = position on the line, b = line in the batch).
// One stage of radix P:  b[t + s(Pq + k)] = (sum_j a[t + s(q + m j)] W_P^{jk}) * W_n^{q k s}.
// ---------------------------------------------------------------------------------------------------
template <typename Real> __device__ inline void bf2(Cx<Real>& a, Cx<Real>& b) { Cx<Real> t = {a.x - b.x, a.y - b.y}; a = {a.x + b.x, a.y + b.y}; b = t; }

template <typename Real, int P> __device__ inline void butterflyGeneric(Cx<Real>* v, int sign) {
    // O(P^2) DFT with compile-time twiddles (P = 3, 5, 7); after unrolling every cos/sin below is a constant.
    Cx<Real> out[P];
#pragma unroll
    for (int k = 0; k < P; k++) {
        Cx<Real> acc = v[0];
#pragma unroll
        for (int j = 1; j < P; j++) {
            const int e = (j * k) % P;
            const double ang = -2.0 * SNB_PI * e / P;
            const Real c = (Real)__builtin_cos(ang);
            const Real sf = (Real)__builtin_sin(ang);       // forward: w = exp(-2 pi i e/P) = c + i*sf
            const Real wy = (sign < 0) ? sf : -sf;
            acc.x += v[j].x * c - v[j].y * wy;
            acc.y += v[j].x * wy + v[j].y * c;
        }
        out[k] = acc;
    }
#pragma unroll
    for (int k = 0; k < P; k++) v[k] = out[k];
}

template <typename Real, int P> __device__ inline void butterflyP(Cx<Real>* v, int sign) {
    if (P == 2) {
        bf2(v[0], v[1]);
    } else if (P == 4) {
        bf2(v[0], v[2]); bf2(v[1], v[3]);
        // v[3] *= -i (forward) or +i (inverse)
        Cx<Real> t = v[3];
        if (sign < 0) v[3] = {t.y, -t.x}; else v[3] = {-t.y, t.x};
        bf2(v[0], v[1]); bf2(v[2], v[3]);
        // outputs in order 0,2,1,3 -> reorder
        Cx<Real> o1 = v[2], o2 = v[1];
        v[1] = o1; v[2] = o2;
    } else if (P == 3) {
        // X1,2 = (v0 - (v1+v2)/2) -+ i (sqrt(3)/2)(v1 - v2)   (forward; signs swap for the inverse)
        const Cx<Real> t1 = {v[1].x + v[2].x, v[1].y + v[2].y};
        const Cx<Real> m1 = {v[0].x - Real(0.5) * t1.x, v[0].y - Real(0.5) * t1.y};
        const Real h = Real(0.86602540378443864676);
        Cx<Real> sx = {h * (v[1].x - v[2].x), h * (v[1].y - v[2].y)};
        if (sign > 0) { sx.x = -sx.x; sx.y = -sx.y; }
        v[0] = {v[0].x + t1.x, v[0].y + t1.y};
        v[1] = {m1.x + sx.y, m1.y - sx.x};            // m1 - i s
        v[2] = {m1.x - sx.y, m1.y + sx.x};            // m1 + i s
    } else if (P == 5) {
        const Real c1 = Real(0.30901699437494742410), c2 = Real(-0.80901699437494742410);
        const Real s1 = Real(0.95105651629515357212), s2 = Real(0.58778525229247312917);
        const Cx<Real> a1 = {v[1].x + v[4].x, v[1].y + v[4].y}, a2 = {v[2].x + v[3].x, v[2].y + v[3].y};
        const Cx<Real> b1 = {v[1].x - v[4].x, v[1].y - v[4].y}, b2 = {v[2].x - v[3].x, v[2].y - v[3].y};
        const Cx<Real> p1 = {v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y};
        const Cx<Real> p2 = {v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y};
        Cx<Real> q1 = {s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y};
        Cx<Real> q2 = {s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y};
        if (sign > 0) { q1.x = -q1.x; q1.y = -q1.y; q2.x = -q2.x; q2.y = -q2.y; }
        v[0] = {v[0].x + a1.x + a2.x, v[0].y + a1.y + a2.y};
        v[1] = {p1.x + q1.y, p1.y - q1.x};            // p1 - i q1
        v[4] = {p1.x - q1.y, p1.y + q1.x};            // p1 + i q1
        v[2] = {p2.x + q2.y, p2.y - q2.x};
        v[3] = {p2.x - q2.y, p2.y + q2.x};
    } else {
        butterflyGeneric<Real, P>(v, sign);
    }
}

template <typename Real, int P>
__device__ inline void fftStage(const Cx<Real>* a, Cx<Real>* b, int n, int len, int s, int sign, const Cx<Real>* tw, int nb, int BS, int tid, int nthreads) {
    const int m = len / P;
    const int perLine = n / P;
    const int items = perLine * nb;
    const float invNb = 1.0f / nb, invS = 1.0f / s;
    for (int it = tid; it < items; it += nthreads) {
        const int r = (int)((it + 0.5f) * invNb);
        const int bi = it - r * nb;
        const int q = (int)((r + 0.5f) * invS);
        const int t = r - q * s;
        Cx<Real> v[P];
#pragma unroll
        for (int j = 0; j < P; j++) v[j] = a[(t + s * (q + m * j)) * BS + bi];
        butterflyP<Real, P>(v, sign);
        const int base = t + s * P * q;
#pragma unroll
        for (int k = 0; k < P; k++) {
            Cx<Real> o = v[k];
            if (k > 0 && q > 0) {
                Cx<Real> w = tw[q * k * s];
                if (sign > 0) w.y = -w.y;
                o = cmul(o, w);
            }
            b[(base + s * k) * BS + bi] = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Two-pass line FFT, n = R1 * R2 with whole R1- and R2-point FFTs held in registers (R <= 16).  Per element it needs
// ~6x fewer instructions than the stage-by-stage Stockham above (one LDS round trip, indices computed once per
// R-point transform, compile-time inner twiddles), which matters because every FFT kernel here is VALU-issue-bound.
//   pass 1 (in place): for each n2 < R2:  y[k1][n2] = W_n^{n2 k1} * sum_{n1} x[n1*R2 + n2] W_R1^{n1 k1}
//   pass 2 (a -> b):   for each k1 < R1:  X[k1 + R1*k2] = sum_{n2} y[k1][n2] W_R2^{n2 k2}
// ---------------------------------------------------------------------------------------------------
template <int N> struct FirstFactor { static constexpr int value = (N % 4 == 0 && N > 4) ? 4 : (N % 2 == 0 && N > 2) ? 2 : (N % 3 == 0 && N > 3) ? 3 : (N % 5 == 0 && N > 5) ? 5 : N; };

template <typename Real, int N> __device__ __forceinline__ void fftReg(Cx<Real>* v, int sign) {
    constexpr int A = FirstFactor<N>::value;
    if constexpr (A == N) {
        butterflyP<Real, N>(v, sign);
    } else {
        constexpr int B = N / A;
        Cx<Real> out[N];
#pragma unroll
        for (int n2 = 0; n2 < B; n2++) {
            Cx<Real> t[A];
#pragma unroll
            for (int n1 = 0; n1 < A; n1++) t[n1] = v[n1 * B + n2];
            butterflyP<Real, A>(t, sign);
#pragma unroll
            for (int k1 = 0; k1 < A; k1++) {
                Cx<Real> o = t[k1];
                if (k1 > 0 && n2 > 0) {
                    const double ang = -2.0 * SNB_PI * (double)(n2 * k1) / N;
                    const Real c = (Real)__builtin_cos(ang), sf = (Real)__builtin_sin(ang);
                    const Cx<Real> w = {c, sign < 0 ? sf : -sf};
                    o = cmul(o, w);
                }
                v[k1 * B + n2] = o;
            }
        }
#pragma unroll
        for (int k1 = 0; k1 < A; k1++) {
            fftReg<Real, B>(&v[k1 * B], sign);
#pragma unroll
            for (int k2 = 0; k2 < B; k2++) out[k1 + A * k2] = v[k1 * B + k2];
        }
#pragma unroll
        for (int k = 0; k < N; k++) v[k] = out[k];
    }
}

template <typename Real, int R1>
__device__ __forceinline__ void fftPass1(Cx<Real>* a, int r2, int sign, const Cx<Real>* tw, int nb, int BS, int tid, int nthreads) {
    const int tasks = nb * r2;
    const float invNb = 1.0f / nb;
    for (int t = tid; t < tasks; t += nthreads) {
        const int n2 = (int)((t + 0.5f) * invNb);
        const int b = t - n2 * nb;
        Cx<Real> v[R1];
#pragma unroll
        for (int n1 = 0; n1 < R1; n1++) v[n1] = a[(n1 * r2 + n2) * BS + b];
        fftReg<Real, R1>(v, sign);
#pragma unroll
        for (int k1 = 0; k1 < R1; k1++) {
            Cx<Real> o = v[k1];
            if (k1 > 0 && n2 > 0) { Cx<Real> w = tw[n2 * k1]; if (sign > 0) w.y = -w.y; o = cmul(o, w); }
            a[(k1 * r2 + n2) * BS + b] = o;
        }
    }
}
template <typename Real, int R2>
__device__ __forceinline__ void fftPass2(const Cx<Real>* a, Cx<Real>* bOut, int r1, int sign, int nb, int BS, int tid, int nthreads) {
    const int tasks = nb * r1;
    const float invNb = 1.0f / nb;
    for (int t = tid; t < tasks; t += nthreads) {
        const int k1 = (int)((t + 0.5f) * invNb);
        const int b = t - k1 * nb;
        Cx<Real> v[R2];
#pragma unroll
        for (int n2 = 0; n2 < R2; n2++) v[n2] = a[(k1 * R2 + n2) * BS + b];
        fftReg<Real, R2>(v, sign);
#pragma unroll
        for (int k2 = 0; k2 < R2; k2++) bOut[(k1 + r1 * k2) * BS + b] = v[k2];
    }
}

#ifndef SNB_FFT_120
// the split of a 120-point line, measured on c3 (round 3, A/B through SNB_LIB_PATH): 8 x 15 (rounds 1-2) x kernel 53.2 us, inverse z 16.3,
// merge + forward z 29.8, y pass 21.3; 10 x 12: 49.7 / 14.8 / 28.2 / 21.6 (the 15-point pass has only nb x 8 tasks for a work-group's 512
// threads and the most registers); 12 x 10: y pass 28.1; 15 x 8: x kernel 85.4
#define SNB_FFT_120(X) X(10, 12)
#endif
// The kernels are instantiated per (R1, R2) pair (inlining every radix into one runtime switch made them allocate 248 VGPRs);
// sizes outside this list use the staged Stockham path (R1 = 0).
#define SNB_FFT_PAIRS(X) X(6, 7) X(6, 9) X(8, 8) X(8, 10) X(9, 10) X(8, 12) X(10, 10) X(9, 12) SNB_FFT_120(X) X(8, 16) X(12, 12) X(10, 16) X(12, 15) X(12, 16) X(15, 16) X(16, 16)

// Runs all stages; returns the buffer holding the result.  Caller must __syncthreads() before reading it.
template <typename Real, int R1, int R2>
__device__ inline Cx<Real>* fftLines(Cx<Real>* a, Cx<Real>* b, int n, const int* factors, int nf, int sign, const Cx<Real>* tw, int nb, int BS, int tid, int nthreads) {
    if constexpr (R1 > 0) {
        __syncthreads();
        fftPass1<Real, R1>(a, R2, sign, tw, nb, BS, tid, nthreads);
        __syncthreads();
        fftPass2<Real, R2>(a, b, R1, sign, nb, BS, tid, nthreads);
        return b;
    }
    int len = n, s = 1;
    for (int f = 0; f < nf; f++) {
        const int P = factors[f];
        __syncthreads();
        switch (P) {
            case 2: fftStage<Real, 2>(a, b, n, len, s, sign, tw, nb, BS, tid, nthreads); break;
            case 3: fftStage<Real, 3>(a, b, n, len, s, sign, tw, nb, BS, tid, nthreads); break;
            case 4: fftStage<Real, 4>(a, b, n, len, s, sign, tw, nb, BS, tid, nthreads); break;
            case 5: fftStage<Real, 5>(a, b, n, len, s, sign, tw, nb, BS, tid, nthreads); break;
            case 7: fftStage<Real, 7>(a, b, n, len, s, sign, tw, nb, BS, tid, nthreads); break;
            case 11: fftStage<Real, 11>(a, b, n, len, s, sign, tw, nb, BS, tid, nthreads); break;
            default: fftStage<Real, 13>(a, b, n, len, s, sign, tw, nb, BS, tid, nthreads); break;
        }
        Cx<Real>* tmp = a; a = b; b = tmp;
        len /= P; s *= P;
    }
    return a;
}

extern __shared__ __align__(16) unsigned char s_dyn[];

// Copy loops of the FFT kernels: `for (i = tid; i < n; i += NT) lds[f(i)] = global[g(i)]` compiles to ONE request in flight per thread
// (the LDS store of trip i sits between the loads of trips i and i+1), i.e. n/NT dependent memory round trips -- 15 of them in the y
// pass.  U requests are issued back to back into registers first, then stored.
template <int U, typename T, typename LoadF, typename StoreF>
__device__ __forceinline__ void batchedCopy(const int begin, const int end, const int stride, LoadF load, StoreF store) {
    for (int i0 = begin; i0 < end; i0 += U * stride) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const int i = i0 + u * stride; if (i < end) v[u] = load(i); }
#pragma unroll
        for (int u = 0; u < U; u++) { const int i = i0 + u * stride; if (i < end) store(i, v[u]); }
    }
}


// ---- z axis: real <-> half-complex.  One work-group transforms NL contiguous real lines, TWO PER COMPLEX FFT: lines 2c and 2c+1
// travel as the real and imaginary part of complex line c (z = a + i b, Z = A + i B with A, B Hermitian), so the z passes do half
// the butterflies of a zero-imaginary transform.  forward: A_k = (Z_k + conj Z_{n-k})/2, B_k = (Z_k - conj Z_{n-k})/(2i);
// inverse: Z_k = A_k + i B_k for k <= n/2 and conj(A_{n-k}) + i conj(B_{n-k}) above, then a = Re z, b = Im z.
template <typename Real, bool FORWARD, int R1, int R2> __global__ __launch_bounds__(512) void k_fftZ(const PmeParams<Real> p, int NL) {
    const int nz = p.d.nz, nzc = p.d.nzc;
    const int NC = NL >> 1;
    const int BS = NC + 1;   // padded batch stride (bank spread for the transposing LDS accesses)
    Cx<Real>* A = reinterpret_cast<Cx<Real>*>(s_dyn);
    Cx<Real>* B = A + (size_t)nz * BS;
    Cx<Real>* tw = B + (size_t)nz * BS;                 // roots of unity staged in LDS (the butterflies index them per item)
    const size_t nlines = (size_t)p.nsub * p.d.nx * p.d.ny;
    const size_t line0 = (size_t)blockIdx.x * NL;
    const int nl = (int)((nlines - line0) < (size_t)NL ? (nlines - line0) : (size_t)NL);   // real lines here
    const int nb = (nl + 1) >> 1;                                                           // complex lines here
    const int tid = threadIdx.x, NT = blockDim.x;
    const FastDiv dz(nz), dzc(nzc);
    for (int k = tid; k < nz; k += NT) tw[k] = reinterpret_cast<const Cx<Real>*>(p.twz)[k];
    if (FORWARD) {
        const Real* in = p.gridReal + line0 * nz;
        for (int it = tid; it < nb * nz; it += NT) {
            const int c = dz.div(it), k = it - c * nz;
            const Real a = in[(2 * c) * nz + k];
            const Real b = (2 * c + 1 < nl) ? in[(2 * c + 1) * nz + k] : Real(0);
            A[k * BS + c] = {a, b};
        }
        Cx<Real>* R = fftLines<Real, R1, R2>(A, B, nz, p.d.fz, p.d.nfz, -1, tw, nb, BS, tid, NT);
        __syncthreads();
        Cx<Real>* out = reinterpret_cast<Cx<Real>*>(p.gridCplx) + line0 * nzc;
        for (int it = tid; it < nb * nzc; it += NT) {
            const int c = dzc.div(it), k = it - c * nzc;
            const Cx<Real> z = R[k * BS + c], m = R[(k == 0 ? 0 : nz - k) * BS + c];
            out[(2 * c) * nzc + k] = {Real(0.5) * (z.x + m.x), Real(0.5) * (z.y - m.y)};
            if (2 * c + 1 < nl) out[(2 * c + 1) * nzc + k] = {Real(0.5) * (z.y + m.y), Real(0.5) * (m.x - z.x)};
        }
    } else {
        const Cx<Real>* in = reinterpret_cast<const Cx<Real>*>(p.gridCplx) + line0 * nzc;
        struct Two { Cx<Real> a, b; };
        batchedCopy<4, Two>(tid, nb * nzc, NT,
            [&](int it) {
                const int c = dzc.div(it), k = it - c * nzc;
                Two t; t.a = in[(2 * c) * nzc + k]; t.b = {Real(0), Real(0)};
                if (2 * c + 1 < nl) t.b = in[(2 * c + 1) * nzc + k];
                return t;
            },
            [&](int it, const Two& t) {
                const int c = dzc.div(it), k = it - c * nzc;
                const Cx<Real> a = t.a, b = t.b;
                A[k * BS + c] = {a.x - b.y, a.y + b.x};                                         // A_k + i B_k
                if (k > 0 && nz - k >= nzc) A[(nz - k) * BS + c] = {a.x + b.y, b.x - a.y};      // conj(A_k) + i conj(B_k)
            });
        Cx<Real>* R = fftLines<Real, R1, R2>(A, B, nz, p.d.fz, p.d.nfz, +1, tw, nb, BS, tid, NT);
        __syncthreads();
        Real* out = p.gridReal + line0 * nz;
        for (int it = tid; it < nb * nz; it += NT) {
            const int c = dz.div(it), k = it - c * nz;
            const Cx<Real> z = R[k * BS + c];
            out[(2 * c) * nz + k] = z.x;
            if (2 * c + 1 < nl) out[(2 * c + 1) * nz + k] = z.y;
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// Own-atoms spreader (round 3).  The scanning brick spreader above is VALU-bound (29 M VALU instructions per launch on c3, 85 % busy):
// every brick scans the atoms of nine columns to find the ones that reach it and every atom is processed by ~1.7 bricks per x-line.
// Here a work-group = (brick, z slab) handles ONLY the atoms sorted into its own columns whose mesh cell lies in the slab, one thread
// per atom (weights once, 25 lines, 75 packed LDS atomics), in an LDS region that covers the brick, the stencil's reach (+4) and a drift
// margin of M cells on either side: no scan, no candidate list, no atom seen twice.  The region goes to global memory (ownPartial) and
// k_spreadMerge sums, per z line, the regions that cover it -- integer sums in single precision, so the mesh does not depend on any
// order -- and runs the forward z FFT of the brick's lines.  An atom that has drifted out of its region (further than the margin: the
// neighbour list is overdue then) is recorded as a stray and added by the merge kernel, one by one.
// Replaces gridSpreadCharge (platforms/common/src/kernels/pme.cc:24-122) + the sort it relies on + the forward z pass.
// ---------------------------------------------------------------------------------------------------
template <typename Real, bool FIXED> __global__ __launch_bounds__(512) void k_spreadOwn(const PmeParams<Real> p) {
    SNB_PME_PRIO();
    SNB_TRACE_START(p.stepTrace, 6);
    extern __shared__ __align__(16) unsigned char s_brick_raw[];
    constexpr int NT = 512;
    using Acc = typename std::conditional<FIXED, int, double>::type;
    const int ncx = p.sortNcx, ncy = p.sortNcy, nz = p.d.nz;
    const int cx = p.groupX * (p.d.nx / ncx), cy = p.groupY * (p.d.ny / ncy);
    const int nbx = ncx / p.groupX, nby = ncy / p.groupY;
    const int nSlabs = p.ownSlabs, M = p.ownMargin, sz = nz / nSlabs;
    const int RX = cx + 4 + 2 * M, RY = cy + 4 + 2 * M, RZ = sz + 4;
    const int RZP = ownLineStride<FIXED>(RZ);                            // LDS stride of a z line (bank spread); the global copy is dense
    const int zs = blockIdx.x % nSlabs, brickId = blockIdx.x / nSlabs;
    const int slot = brickId / (nbx * nby), bcol = brickId - slot * (nbx * nby);
    const int Bx = bcol / nby, By = bcol - Bx * nby;
    const int xr0 = Bx * cx - M, yr0 = By * cy - M, z0 = zs * sz;      // region origin (x, y may be negative: compared modulo the mesh)
    const int nlines = RX * RY;
    Acc* region = reinterpret_cast<Acc*>(s_brick_raw);
    __shared__ int s_begin[16], s_pref[17], s_any;
    const int tid = threadIdx.x;
    const int nr = p.groupX * p.groupY;                                  // <= 16 (launcher)
    if (tid < nr) {
        const int gx = tid / p.groupY, gy = tid - gx * p.groupY;
        const int2 rg = p.colRange[(size_t)p.gridSubset[slot] * (ncx * ncy) + (Bx * p.groupX + gx) * ncy + By * p.groupY + gy];
        s_begin[tid] = rg.x; s_pref[tid + 1] = rg.y > rg.x ? rg.y - rg.x : 0;
    }
    if (tid == 0) s_any = 0;
    __syncthreads();
    if (tid == 0) { int acc = 0; s_pref[0] = 0; for (int r = 0; r < nr; r++) { acc += s_pref[r + 1]; s_pref[r + 1] = acc; } }
    __syncthreads();
    const int total = s_pref[nr];
    if (total == 0) { if (tid == 0) p.ownBusy[blockIdx.x] = 0; return; }
    {   // (RZP * sizeof(Acc) is a multiple of 16)
        int4* z4 = reinterpret_cast<int4*>(region);
        const int n4 = (int)((sizeof(Acc) * (size_t)nlines * RZP) >> 4);
        for (int i = tid; i < n4; i += NT) z4[i] = make_int4(0, 0, 0, 0);
    }
    __syncthreads();
    const Real fixScale = FIXED ? p.fixDev[0] : Real(1);
    bool mine = false;
    for (int v = tid; v < total; v += NT) {
        int r = 0;
        while (v >= s_pref[r + 1]) r++;
        const int a = s_begin[r] + (v - s_pref[r]);
        const int cell = p.cells[a];                                     // packed mesh cell (position-gather pass / k_pmeCells), -1: no charge on this mesh
        if (cell < 0) continue;
        const int rz = ((cell >> 20) & 1023) - z0;
        if (rz < 0 || rz >= sz) continue;                                // another slab's atom
        const auto pos = p.posq[a];
        const Real q = pmeCharge(p, a);
        int idx[3]; Real fr[3];
        gridCoord<Real>(p.recip, p.recipLo, pos.x, pos.y, pos.z, p.d.nx, p.d.ny, nz, idx, fr);
        int rx = idx[0] - xr0; if (rx < 0) rx += p.d.nx; else if (rx >= p.d.nx) rx -= p.d.nx;
        int ry = idx[1] - yr0; if (ry < 0) ry += p.d.ny; else if (ry >= p.d.ny) ry -= p.d.ny;
        if (rx + 4 >= RX || ry + 4 >= RY) {                              // drifted out of the region: the merge kernel adds it
            const int at = atomicAdd(p.strayCount, 1);
            p.strays[at] = make_int2(a, slot);
            continue;
        }
        mine = true;
        Real tx[5], ty[5], tz[5], dtmp[5];
        bspline5<Real>(fr[0], tx, dtmp); bspline5<Real>(fr[1], ty, dtmp); bspline5<Real>(fr[2], tz, dtmp);
        const Real wq = q * fixScale;
        if constexpr (FIXED) {
            // 32-bit fixed point, one ds_add_u32 per point: the scanning spreader packs two points per ds_add_u64 (15 instead of 25 LDS
            // atomics per x-line), which costs ~17 VALU instructions per line for the odd/even selects and the borrow; this kernel is
            // VALU-bound (one thread per atom, 25 lines), the LDS pipe has room (measured: 8.5 M VALU instructions per launch with pairs)
#pragma unroll
            for (int ix = 0; ix < 5; ix++) {
                const Real wx = wq * tx[ix];
                Acc* plane = region + (size_t)(rx + ix) * RY * RZP + rz;
#pragma unroll
                for (int iy = 0; iy < 5; iy++) {
                    const Real wxy = wx * ty[iy];
                    unsigned* line = reinterpret_cast<unsigned*>(plane + (ry + iy) * RZP);
#pragma unroll
                    for (int iz = 0; iz < 5; iz++) __hip_atomic_fetch_add(&line[iz], (unsigned)__float2int_rn(wxy * tz[iz]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        } else {
#pragma unroll
            for (int ix = 0; ix < 5; ix++) {
                const Real wx = wq * tx[ix];
                Acc* plane = region + (size_t)(rx + ix) * RY * RZP;
#pragma unroll
                for (int iy = 0; iy < 5; iy++) {
                    const Real wxy = wx * ty[iy];
                    Acc* line = plane + (ry + iy) * RZP + rz;
#pragma unroll
                    for (int iz = 0; iz < 5; iz++) __hip_atomic_fetch_add(&line[iz], (double)(wxy * tz[iz]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
    if (mine) s_any = 1;      // (same value from every writer)
    __syncthreads();
    const int busy = s_any;
    if (tid == 0) p.ownBusy[blockIdx.x] = busy;
    if (!busy) return;
    Acc* out = reinterpret_cast<Acc*>(p.ownPartial) + (size_t)blockIdx.x * nlines * RZ;
    // dense copy: a half-wave per line, 16 bytes per lane when a line is a whole number of them, else 8 (RZ is even: launcher)
    const int half = tid >> 5, hl = tid & 31;
    if ((RZ * sizeof(Acc)) % 16 == 0) {
        const int q = (int)(RZ * sizeof(Acc) / 16);
        for (int l = half; l < nlines; l += NT / 32)
            for (int k = hl; k < q; k += 32)
                reinterpret_cast<int4*>(out + (size_t)l * RZ)[k] = reinterpret_cast<const int4*>(region + (size_t)l * RZP)[k];
    } else {
        const int q = (int)(RZ * sizeof(Acc) / 8);
        for (int l = half; l < nlines; l += NT / 32)
            for (int k = hl; k < q; k += 32)
                reinterpret_cast<int2*>(out + (size_t)l * RZ)[k] = reinterpret_cast<const int2*>(region + (size_t)l * RZP)[k];
    }
}

// One work-group per brick: per z line of the brick, the sum of the regions of k_spreadOwn that cover it (the brick's own slabs, the
// reach of the bricks below it in x / y, the margin of the bricks above), then -- FUSEZ -- the forward z FFT of the brick's lines, two
// real lines per complex transform, written as the half-complex mesh; otherwise the real mesh.  One thread sums one 16-byte chunk of a
// line: at most 3 x 3 bricks and 2 slabs cover it; per y neighbour the 6 (predicated) loads are issued back to back -- the first version
// walked the candidates one dependent load at a time and took 101 us on c3.
template <typename Real, bool FIXED, bool FUSEZ, int R1, int R2, int NT> __global__ __launch_bounds__(NT) void k_spreadMerge(const PmeParams<Real> p, const int chunk, const int plane) {
    SNB_PME_PRIO();
    SNB_TRACE_START(p.stepTrace, 7);
    // NT = 256      // (a brick is ~18 complex lines: 512 threads left most of them idle in the FFT passes, and at the ~120 VGPRs of those passes 256-thread groups go four to a CU)
    using Acc = typename std::conditional<FIXED, int, double>::type;
    constexpr int CMAX = FIXED ? 4 : 2;                                    // values per 16-byte load
    const int ncx = p.sortNcx, ncy = p.sortNcy, nx = p.d.nx, ny = p.d.ny, nz = p.d.nz, nzc = p.d.nzc;
    const int cx = p.groupX * (nx / ncx), cy = p.groupY * (ny / ncy);
    const int nbx = ncx / p.groupX, nby = ncy / p.groupY;
    const int nSlabs = p.ownSlabs, M = p.ownMargin, sz = nz / nSlabs;
    const int RX = cx + 4 + 2 * M, RY = cy + 4 + 2 * M, RZ = sz + 4;
    const size_t npts = (size_t)RX * RY * RZ;
    const int slot = blockIdx.x / (nbx * nby), bcol = blockIdx.x - slot * (nbx * nby);
    const int Bx = bcol / nby, By = bcol - Bx * nby;
    const int x0 = Bx * cx, y0 = By * cy;
    const int nl = cx * cy, nb = (nl + 1) >> 1, BS = nb + 1;
    const int tid = threadIdx.x;
    const bool trace = p.trace != nullptr;      // SNB_PME_TRACE: wall-clock split of the busy work-groups (sum phase / strays + z FFT / store), 100 MHz ticks
    long long tr0 = 0, tr1 = 0, tr2 = 0;
    if (trace) tr0 = wall_clock64();
    Cx<Real>* A = reinterpret_cast<Cx<Real>*>(s_dyn);                     // FUSEZ: [nz][BS] (+ B, roots of unity)
    Cx<Real>* B = A + (size_t)nz * BS;
    Cx<Real>* tw = B + (size_t)nz * BS;
    // per neighbour brick j = jy * 3 + jx (uniform for the work-group): which of its slabs' regions were written this step (bit per slab)
    // and where its regions start -- worked out by nine threads, once (per line and candidate this was two integer modulos and a
    // flag loop: 5 of the busy work-groups' 25 us, SNB_PME_TRACE)
    __shared__ unsigned s_bmask[9];
    __shared__ int s_roff[9];                                              // (launcher: every element offset of ownPartial below 2^31)
    __shared__ int s_any;
    const int loX = (4 + M + cx - 1) / cx, hiX = (M + cx - 1) / cx, loY = (4 + M + cy - 1) / cy, hiY = (M + cy - 1) / cy;      // launcher: lo + hi + 1 <= 3
    if (tid == 0) s_any = 0;
    __syncthreads();
    if (tid < 9) {
        const int jy = tid / 3, jx = tid - jy * 3;
        const int dbx = jx - hiX, dby = jy - hiY;
        unsigned m = 0; int ro = 0;
        if (dbx <= loX && dby <= loY) {
            int bx2 = (Bx - dbx) % nbx; if (bx2 < 0) bx2 += nbx;
            int by2 = (By - dby) % nby; if (by2 < 0) by2 += nby;
            const int reg = ((slot * nbx + bx2) * nby + by2) * nSlabs;
            for (int sl = 0; sl < nSlabs; sl++) if (p.ownBusy[reg + sl]) m |= 1u << sl;      // (launcher: at most 32 slabs)
            ro = (int)((size_t)reg * npts) + (((jx - hiX) * cx + M) * RY + ((jy - hiY) * cy + M)) * RZ;      // + the line (lx, ly) of THIS brick: (lx * RY + ly) * RZ
        }
        s_bmask[tid] = m; s_roff[tid] = ro;
        if (m) s_any = 1;
    }
    if (FUSEZ) for (int k = tid; k < nz; k += NT) tw[k] = reinterpret_cast<const Cx<Real>*>(p.twz)[k];
    // (both requested BEFORE the barrier: behind it they were a dependent memory round trip of their own ahead of the sums -- the busy
    // work-groups spend 17 of their 25 us in dependent load rounds, SNB_PME_TRACE)
    const int nStray = *p.strayCount;      // strays (normally none): atoms whose footprint left the region of their own work-group; every brick looks at every stray
    const Real inv = FIXED ? p.fixDev[1] : Real(1);
    __syncthreads();
    long long trA = 0;
    if (trace) trA = wall_clock64();
    const Acc* partial = reinterpret_cast<const Acc*>(p.ownPartial);
    Real* greal = p.gridReal + (size_t)slot * nx * ny * nz;
    Cx<Real>* out = reinterpret_cast<Cx<Real>*>(p.gridCplx) + (size_t)slot * nx * ny * nzc;
    const FastDiv dzc(nzc), dcy2(cy);
    if (FUSEZ && !s_any && nStray == 0) {      // nothing of this subset anywhere near (uniform): the brick's spectrum is zero
        if (plane) {      // plane-major, brick-tiled spectrum [slot][kz][brick][line] (the plane path, k_planeXY)
            const FastDiv dnl(nl);
            for (int it = tid; it < nl * nzc; it += NT) {
                const int k = dnl.div(it), l = it - k * nl;
                out[((size_t)k * (nbx * nby) + bcol) * nl + l] = {Real(0), Real(0)};
            }
            return;
        }
        for (int it = tid; it < nl * nzc; it += NT) {
            const int l = dzc.div(it), k = it - l * nzc;
            const int lx = dcy2.div(l), ly = l - lx * cy;
            out[((size_t)(x0 + lx) * ny + (y0 + ly)) * nzc + k] = {Real(0), Real(0)};
        }
        return;
    }
    const int nch = nz / chunk;                                            // chunks per line (launcher: chunk divides sz, hence nz and RZ)
    // A thread stays on ONE line per pass (NT / nl threads share a line and stride over its chunks), so which regions cover the line -- the
    // 3 x 3 neighbour bricks' offsets and their busy slabs, ~20 instructions per candidate -- is worked out once per line, in registers,
    // instead of once per 16-byte chunk (7 k of the work-group's 11 k wave-instructions; a table in LDS instead of registers was slower:
    // 33.3 us against 28.3, one more LDS round trip per candidate).
    const int TPL = (NT / nl) > 0 ? (NT / nl) : 1, LPP = NT / TPL;        // threads per line, lines per pass
    const FastDiv dtpl(TPL);
    const int lineOfThread = dtpl.div(tid), sub = tid - lineOfThread * TPL;
    for (int l = lineOfThread; l < nl; l += LPP) {
        const int lx = dcy2.div(l), ly = l - lx * cy;
        // the neighbour tables are uniform: they travel in scalar registers; a thread keeps the offset of its line and one bit per
        // neighbour (does that brick's region reach this line?) -- 18 vector registers fewer than offsets and masks per candidate, which is
        // what keeps four work-groups on a CU (at 152 registers the 1728 empty bricks of c3's solute meshes queue behind three)
        const int lineOff = (lx * RY + ly) * RZ;
        unsigned okBits = 0;
#pragma unroll
        for (int jy = 0; jy < 3; jy++) {
            const int ry = ly + (jy - hiY) * cy + M;
#pragma unroll
            for (int jx = 0; jx < 3; jx++) {
                const int rx = lx + (jx - hiX) * cx + M;
                if (rx >= 0 && rx < RX && ry >= 0 && ry < RY) okBits |= 1u << (jy * 3 + jx);      // (bricks out of reach carry an empty mask)
            }
        }
        int cadd[9]; unsigned bmask[9];
#pragma unroll
        for (int j = 0; j < 9; j++) { cadd[j] = __builtin_amdgcn_readfirstlane(s_roff[j]); bmask[j] = (unsigned)__builtin_amdgcn_readfirstlane((int)s_bmask[j]); }
        for (int ch = sub; ch < nch; ch += TPL) {
        const int k0 = ch * chunk;
        // the two slabs whose regions hold plane k0: its own, and the one below when k0 is among that one's four extra planes
        const int s1 = k0 / sz, zl1 = k0 - s1 * sz;
        const int s0 = s1 == 0 ? nSlabs - 1 : s1 - 1, zl0 = zl1 + sz;
        const bool low = zl0 < RZ;
        Acc sum[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; c++) sum[c] = Acc(0);
        // all (<= 3 x 3 bricks) x (2 slabs) candidate loads of the chunk go out together: a thread has ~3 chunks, and with one round of
        // dependent loads per y neighbour the merge was 12 latencies long (28.5 us on c3)
        Acc v[18][CMAX];
        // per candidate: one bit test, one 64-bit add, one load (the slab offsets are per chunk, the load width is chosen once per chunk:
        // with ~2 waves per SIMD a wave's own instruction stream is latency, not throughput -- SNB_PME_TRACE: 5 us per chunk before this)
        const int sOff[2] = {(int)(s1 * (int)npts + zl1), (int)(s0 * (int)npts + zl0)};
        const unsigned sBit[2] = {1u << s1, low ? (1u << s0) : 0u};
        auto loadAll = [&](auto width) {
            constexpr int W = decltype(width)::value;      // values per load: CMAX (16 bytes), 2 (fixed point, 8 bytes) or 1
#pragma unroll
            for (int j = 0; j < 9; j++) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int u = j * 2 + h;
#pragma unroll
                    for (int c = 0; c < CMAX; c++) v[u][c] = Acc(0);
                    if (((okBits >> j) & 1u) && (bmask[j] & sBit[h])) {
                        const Acc* src = partial + (lineOff + cadd[j] + sOff[h]);
                        if constexpr (W == CMAX) {
                            if constexpr (FIXED) { const int4 t = *reinterpret_cast<const int4*>(src); v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w; }
                            else { const double2 t = *reinterpret_cast<const double2*>(src); v[u][0] = t.x; v[u][1] = t.y; }
                        } else if constexpr (FIXED && W == 2) { const int2 t = *reinterpret_cast<const int2*>(src); v[u][0] = t.x; v[u][1] = t.y; }
                        else v[u][0] = src[0];
                    }
                }
            }
        };
        if (chunk == CMAX) loadAll(std::integral_constant<int, CMAX>());
        else if (FIXED && chunk == 2) loadAll(std::integral_constant<int, 2>());
        else loadAll(std::integral_constant<int, 1>());
#pragma unroll
        for (int c = 0; c < CMAX; c++)
#pragma unroll
            for (int u = 0; u < 18; u++) sum[c] += v[u][c];
#pragma unroll
        for (int c = 0; c < CMAX; c++) if (c < chunk) {
            const Real val = (Real)sum[c] * inv;
            if (FUSEZ) reinterpret_cast<Real*>(&A[(k0 + c) * BS + (l >> 1)])[l & 1] = val;
            else greal[((size_t)(x0 + lx) * ny + (y0 + ly)) * nz + k0 + c] = val;
        }
        }
    }
    if (FUSEZ && (nl & 1)) for (int k = tid; k < nz; k += NT) A[k * BS + (nl >> 1)].y = Real(0);      // the odd line out has no partner
    __syncthreads();
    if (trace) tr1 = wall_clock64();
    for (int s = 0; s < nStray; s++) {
        const int2 e = p.strays[s];
        if (e.y != slot) continue;
        const auto pos = p.posq[e.x];
        int idx[3]; Real fr[3];
        gridCoord<Real>(p.recip, p.recipLo, pos.x, pos.y, pos.z, nx, ny, nz, idx, fr);
        int rx = idx[0] - x0; if (rx > nx / 2) rx -= nx; else if (rx < -(nx / 2)) rx += nx;
        int ry = idx[1] - y0; if (ry > ny / 2) ry -= ny; else if (ry < -(ny / 2)) ry += ny;
        if (rx + 4 < 0 || rx >= cx || ry + 4 < 0 || ry >= cy) continue;     // (uniform: every thread looks at the same stray)
        if (tid < 125) {
            const int ix = tid / 25, iy = (tid / 5) % 5, iz = tid % 5;
            const int lx = rx + ix, ly = ry + iy;
            if (lx >= 0 && lx < cx && ly >= 0 && ly < cy) {
                Real tx[5], ty[5], tz[5], dtmp[5];
                bspline5<Real>(fr[0], tx, dtmp); bspline5<Real>(fr[1], ty, dtmp); bspline5<Real>(fr[2], tz, dtmp);
                Real wxy = pmeCharge(p, e.x) * (FIXED ? p.fixDev[0] : Real(1)), wz = Real(0);
#pragma unroll
                for (int k = 0; k < 5; k++) { if (k == ix) wxy *= tx[k]; if (k == iz) wz = tz[k]; }
#pragma unroll
                for (int k = 0; k < 5; k++) if (k == iy) wxy *= ty[k];
                Real w = wxy * wz;
                if (FIXED) w = (Real)__float2int_rn(w) * inv;      // the value the fixed-point path would have added
                int z = idx[2] + iz; if (z >= nz) z -= nz;
                const int l = lx * cy + ly;
                if (FUSEZ) { Real* t = reinterpret_cast<Real*>(&A[z * BS + (l >> 1)]) + (l & 1); __hip_atomic_fetch_add(t, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
                else __hip_atomic_fetch_add(&greal[((size_t)(x0 + lx) * ny + (y0 + ly)) * nz + z], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if constexpr (FUSEZ) {
        Cx<Real>* R = fftLines<Real, R1, R2>(A, B, nz, p.d.fz, p.d.nfz, -1, tw, nb, BS, tid, NT);
        __syncthreads();
        if (trace) tr2 = wall_clock64();
        if (plane) {
            // plane-major and brick-tiled, [slot][kz][brick][line]: the brick's lines of one kz are ONE contiguous run (288 bytes on c3) and a
            // line pair is one 16-byte store.  (As [slot][kz][x][y] the same data went out in 48-byte runs a row apart: by SNB_PME_TRACE a busy
            // brick's store phase took 8.1 us against 3.2 for the line-major layout of the three-pass pipeline, and its loads 2.7 us more.)
            const FastDiv dnb(nb);
            Cx<Real>* tiled = out + (size_t)bcol * nl;
            const size_t kStride = (size_t)(nbx * nby) * nl;
            for (int it = tid; it < nb * nzc; it += NT) {
                const int k = dnb.div(it), c = it - k * nb;
                const Cx<Real> z = R[k * BS + c], m = R[(k == 0 ? 0 : nz - k) * BS + c];
                const Cx<Real> a = {Real(0.5) * (z.x + m.x), Real(0.5) * (z.y - m.y)}, b = {Real(0.5) * (z.y + m.y), Real(0.5) * (m.x - z.x)};
                Cx<Real>* dst = tiled + k * kStride + 2 * c;
                if constexpr (std::is_same<Real, float>::value) {
                    if (2 * c + 1 < nl && !(nl & 1)) { *reinterpret_cast<float4*>(dst) = make_float4(a.x, a.y, b.x, b.y); continue; }
                }
                dst[0] = a;
                if (2 * c + 1 < nl) dst[1] = b;
            }
        } else
        for (int it = tid; it < nb * nzc; it += NT) {
            const int c = dzc.div(it), k = it - c * nzc;
            const Cx<Real> z = R[k * BS + c], m = R[(k == 0 ? 0 : nz - k) * BS + c];
            const int l0 = 2 * c, lx0 = dcy2.div(l0), ly0 = l0 - lx0 * cy;
            out[((size_t)(x0 + lx0) * ny + (y0 + ly0)) * nzc + k] = {Real(0.5) * (z.x + m.x), Real(0.5) * (z.y - m.y)};
            if (l0 + 1 < nl) {
                const int l1 = l0 + 1, lx1 = dcy2.div(l1), ly1 = l1 - lx1 * cy;
                out[((size_t)(x0 + lx1) * ny + (y0 + ly1)) * nzc + k] = {Real(0.5) * (z.y + m.y), Real(0.5) * (m.x - z.x)};
            }
        }
        if (trace && tid == 0) {
            const long long t3 = wall_clock64();
            atomicAdd((unsigned long long*)&p.trace[3], (unsigned long long)(trA - tr0));      // (of which: the prologue up to its barrier)
            atomicAdd((unsigned long long*)&p.trace[4], (unsigned long long)(tr1 - tr0)); atomicAdd((unsigned long long*)&p.trace[5], (unsigned long long)(tr2 - tr1));
            atomicAdd((unsigned long long*)&p.trace[6], (unsigned long long)(t3 - tr2)); atomicAdd((unsigned long long*)&p.trace[7], 1ull);
        }
    }
}

// ---- strided axis (y): tiles of NB adjacent lines (adjacent = consecutive complex elements in memory) ----
// address(a, b, k) = a*strideA + b + k*strideK, b in [0, nbTotal)
template <typename Real, int R1, int R2> __global__ __launch_bounds__(512) void k_fftStrided(const PmeParams<Real> p, int n, size_t strideA, int nbTotal, size_t strideK,
                                                                          int NB, int tilesPerA, int sign, int axis) {
    const int a = blockIdx.x / tilesPerA, tile = blockIdx.x - a * tilesPerA;
    const int b0 = tile * NB;
    const int nb = (nbTotal - b0) < NB ? (nbTotal - b0) : NB;
    Cx<Real>* A = reinterpret_cast<Cx<Real>*>(s_dyn);
    Cx<Real>* B = A + (size_t)n * NB;
    Cx<Real>* tw = B + (size_t)n * NB;
    Cx<Real>* g = reinterpret_cast<Cx<Real>*>(p.gridCplx) + (size_t)a * strideA + b0;
    const int tid = threadIdx.x, NT = blockDim.x;      // 256 or 512 threads over the same LDS tile (launcher)
    for (int k = tid; k < n; k += NT) tw[k] = reinterpret_cast<const Cx<Real>*>(axis == 1 ? p.twy : p.twx)[k];
    const FastDiv dnb(nb);
    batchedCopy<8, Cx<Real>>(tid, n * nb, NT,
        [&](int it) { const int k = dnb.div(it), b = it - k * nb; return g[(size_t)k * strideK + b]; },
        [&](int it, const Cx<Real>& v) { const int k = dnb.div(it), b = it - k * nb; A[k * NB + b] = v; });
    Cx<Real>* R = fftLines<Real, R1, R2>(A, B, n, axis == 1 ? p.d.fy : p.d.fx, axis == 1 ? p.d.nfy : p.d.nfx, sign, tw, nb, NB, tid, NT);
    __syncthreads();
    for (int it = tid; it < n * nb; it += NT) {
        const int k = dnb.div(it), b = it - k * nb;
        g[(size_t)k * strideK + b] = R[k * NB + b];
    }
}

// ---- reciprocal-space kernel value (ReferencePME.cpp:425-471 Coulomb, :522-570 dispersion) -------------
template <typename Real> __device__ inline Real recipTermRaw(const PmeParams<Real>& p, int kx, int ky, int kz) {
    const int nx = p.d.nx, ny = p.d.ny, nz = p.d.nz;
    const Real mx = (Real)((kx < (nx + 1) / 2) ? kx : kx - nx);
    const Real my = (Real)((ky < (ny + 1) / 2) ? ky : ky - ny);
    const Real mz = (Real)((kz < (nz + 1) / 2) ? kz : kz - nz);
    const Real mhx = mx * p.recip[0];
    const Real mhy = mx * p.recip[3] + my * p.recip[4];
    const Real mhz = mx * p.recip[6] + my * p.recip[7] + mz * p.recip[8];
    const Real m2 = mhx * mhx + mhy * mhy + mhz * mhz;
    const Real bprod = p.modx[kx] * p.mody[ky] * p.modz[kz];
    if (!p.dispersion) {
        if (kx == 0 && ky == 0 && kz == 0) return Real(0);
        const Real factor = Real(SNB_PI * SNB_PI) / (p.alpha * p.alpha);
        const Real denom = m2 * Real(SNB_PI) * p.volume * bprod;
        return Real(SNB_ONE_4PI_EPS0) * exp(-factor * m2) / denom;
    } else {
        const Real boxfactor = Real(-2 * SNB_PI * 1.7724538509055159) / (Real(6) * p.volume);
        const Real denom = boxfactor / bprod;
        const Real bfac = Real(SNB_PI) / p.alpha;
        const Real fac1 = Real(2 * SNB_PI * SNB_PI * SNB_PI * 1.7724538509055159);
        const Real fac2 = p.alpha * p.alpha * p.alpha;
        const Real fac3 = Real(-2) * p.alpha * Real(SNB_PI * SNB_PI);
        const Real m = sqrt(m2), m3 = m * m2, b = bfac * m;
        return (fac1 * (Real)erfc((double)b) * m3 + exp(-b * b) * (fac2 + fac3 * m2)) * denom;
    }
}

// The reference transforms the FULL complex mesh and keeps the real part of the inverse (ReferencePME.cpp:598-606), which amounts to using
// the average of eterm(k) and eterm(-k).  The two are identical except where an index sits at the Nyquist frequency n/2 of an even
// mesh: there both k and -k carry m = -n/2 and, in a TRICLINIC cell, different |m_hat|^2.  The half-complex pipeline here stores one of
// the pair, so it applies the average explicitly on those planes (measured: 8e-5 of force error on a 24^3 dispersion mesh otherwise).
template <typename Real> __device__ inline Real recipTerm(const PmeParams<Real>& p, int kx, int ky, int kz) {
    const Real e = recipTermRaw<Real>(p, kx, ky, kz);
    if (2 * kx != p.d.nx && 2 * ky != p.d.ny && 2 * kz != p.d.nz) return e;
    const int jx = kx == 0 ? 0 : p.d.nx - kx, jy = ky == 0 ? 0 : p.d.ny - ky, jz = kz == 0 ? 0 : p.d.nz - kz;
    return Real(0.5) * (e + recipTermRaw<Real>(p, jx, jy, jz));
}

// ---- fused x-axis kernel: forward FFT_x, sliced energy, lambda-mixed convolution, inverse FFT_x ---------
// One work-group owns NB adjacent (ky,kz) columns for ALL held subsets: batch index = sub*NB + col.
// (NT = 256 for double precision with a 15- or 16-point register transform: at 512 threads the kernel is capped at 128 VGPRs and spilled
// -- 351 -> 488 us on c5's 180^3 mesh in round 3, which therefore stayed on the staged transform; at 256 threads it has 256)
template <typename Real, int R1, int R2, int NT = 512> __global__ __launch_bounds__(NT) void k_convolveX(const PmeParams<Real> p, int NB, int nCols) {
    const int nx = p.d.nx, nsub = p.nsub;
    const int BS = nsub * NB;
    const int c0 = blockIdx.x * NB;
    const int nbc = (nCols - c0) < NB ? (nCols - c0) : NB;
    Cx<Real>* A = reinterpret_cast<Cx<Real>*>(s_dyn);
    Cx<Real>* B = A + (size_t)nx * BS;
    Cx<Real>* tw = B + (size_t)nx * BS;                                // [nx] roots of unity
    Real* et = reinterpret_cast<Real*>(tw + nx);                       // [nx][NB]
    __shared__ double s_red[(NT / 64) * 4];
    const size_t strideK = (size_t)nCols;                              // ny*nzc
    const size_t strideSub = (size_t)nx * nCols;
    Cx<Real>* g = reinterpret_cast<Cx<Real>*>(p.gridCplx);
    const int tid = threadIdx.x;
    // load: when nbc < NB the unused batch slots are zero-filled so the FFT can run on the full batch shape
    const FastDiv dBS(BS), dNB(NB), dNzc(p.d.nzc);
    batchedCopy<8, Cx<Real>>(tid, nx * BS, NT,
        [&](int it) {
            const int k = dBS.div(it), bb = it - k * BS;
            const int sub = dNB.div(bb), col = bb - sub * NB;
            Cx<Real> v = {Real(0), Real(0)};
            if (col < nbc) v = g[sub * strideSub + (size_t)k * strideK + c0 + col];
            return v;
        },
        [&](int it, const Cx<Real>& v) { A[it] = v; });      // (k * BS + bb == it)
    for (int k = tid; k < nx; k += NT) tw[k] = reinterpret_cast<const Cx<Real>*>(p.twx)[k];
    Cx<Real>* S = fftLines<Real, R1, R2>(A, B, nx, p.d.fx, p.d.nfx, -1, tw, BS, BS, tid, NT);
    Cx<Real>* O = (S == A) ? B : A;
    __syncthreads();
    // eterm per (kx, col)
    for (int it = tid; it < nx * NB; it += NT) {
        const int kx = dNB.div(it), col = it - kx * NB;
        Real e = 0;
        if (col < nbc) {
            const int c = c0 + col;
            const int ky = dNzc.div(c), kz = c - ky * p.d.nzc;
            e = recipTerm<Real>(p, kx, ky, kz);
        }
        et[it] = e;
    }
    __syncthreads();
    // per-slice energies: E_II = 1/2 sum_k eterm |S_I|^2 ; E_IJ = sum_k eterm Re(S_I conj S_J) (ReferencePME.cpp:487-491),
    // over the full grid => Hermitian weight 2 for interior kz planes.
    if (p.wantEnergy && p.mix) {   // sharded engines get the energies by interpolation instead (k_interpolate)
        const int term = p.dispersion ? 1 : 0;
        // The wanted (I, J) pairs four at a time: one pass over the spectra and one reduction per group (a pass and two barriers per
        // slice cost the derivative step 12 us on c3; derivative-only steps ask for a few slices, the others are skipped altogether)
        constexpr int G = 4;
        int gI[G], gJ[G], ng = 0;
        auto flushGroup = [&]() {      // (uniform: every thread holds the same group)
            double acc[G] = {0, 0, 0, 0};
            for (int it = tid; it < nx * NB; it += NT) {
                const int kx = dNB.div(it), col = it - kx * NB;
                if (col >= nbc) continue;
                const int kz = (c0 + col) - dNzc.div(c0 + col) * p.d.nzc;
                const Real w = ((kz == 0 || (2 * kz == p.d.nz)) ? Real(1) : Real(2)) * et[it];
#pragma unroll
                for (int u = 0; u < G; u++) if (u < ng) {
                    const Cx<Real> a = S[kx * BS + gI[u] * NB + col], b = S[kx * BS + gJ[u] * NB + col];
                    acc[u] += (double)(w * (a.x * b.x + a.y * b.y));
                }
            }
#pragma unroll
            for (int u = 0; u < G; u++) {
                if (gI[u] == gJ[u]) acc[u] *= 0.5;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) acc[u] += __shfl_xor(acc[u], o, 64);
            }
            __syncthreads();
            if ((tid & 63) == 0) for (int u = 0; u < G; u++) s_red[(tid >> 6) * G + u] = acc[u];
            __syncthreads();
            if (tid < ng) {
                const int gi = p.gridSubset[gI[tid]], gj = p.gridSubset[gJ[tid]];
                const int slice = gi > gj ? gi * (gi + 1) / 2 + gj : gj * (gj + 1) / 2 + gi;
                double tot = 0; for (int w = 0; w < NT / 64; w++) tot += s_red[w * G + tid];
                atomicAdd(&SNB_SLICE_E_PARTITION(p.sliceE, p.nsubTotal * (p.nsubTotal + 1))[2 * slice + term], tot);
            }
            ng = 0;
        };
        for (int u = 0; u < G; u++) { gI[u] = 0; gJ[u] = 0; }
        for (int I = 0; I < nsub; I++)
            for (int J = 0; J <= I; J++) {
                const int gi = p.gridSubset[I], gj = p.gridSubset[J];
                if (!p.sliceNeed[gi > gj ? gi * (gi + 1) / 2 + gj : gj * (gj + 1) / 2 + gi]) continue;      // (uniform)
                gI[ng] = I; gJ[ng] = J; ng++;
                if (ng == G) flushGroup();
            }
        if (ng > 0) flushGroup();
    }
    // convolution with the lambda mix:  O_I = eterm * sum_J lambda[slice(I,J)][term] * S_J   (mix=0: O_I = eterm * S_I)
    bool mixedOnMatrixCores = false;
    if constexpr (std::is_same<Real, float>::value) {
        static_assert(sizeof(Cx<Real>) == 8 || !std::is_same<Real, float>::value, "");
        if (p.mix && nsub <= 4 && !p.mix16) {
            // Up to four subsets (round 3): v_mfma_f32_4x4x1_16b_f32 -- 16 blocks of a 4 x 1 by 1 x 4 product.  Lane l holds A[block l/4][row l%4]
            // and B[block l/4][col l%4] and receives D[block l/4][row r][col l%4] in register r (tools/ubench_mfma4x4.hip).  With A = one column
            // J of the lambda matrix (rows = output subsets I) and B = the spectra of subset J at the lanes' own 64 points (re / im are points),
            // four accumulating instructions leave every lane with the four mixed values of ITS point: one LDS read per J, one LDS write per I,
            // no shuffling.  The 16 x 16 x 4 form below spends ~60 VALU instructions per 16 points on operand indexing and uses 4 of its 16
            // output rows: 7200 wave-instructions per work-group, 46 % of this kernel's VALU work on c3; this form needs ~750.
            mixedOnMatrixCores = true;
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const int term = p.dispersion ? 1 : 0;
            const int lane = tid & 63, wave = tid >> 6, nWaves = NT / 64;
            float aReg[4];
#pragma unroll
            for (int J = 0; J < 4; J++) {
                const int I = lane & 3;
                float v = 0.f;
                if (I < nsub && J < nsub) {
                    const int gi = p.gridSubset[I], gj = p.gridSubset[J];
                    v = p.lambdas[2 * (gi > gj ? gi * (gi + 1) / 2 + gj : gj * (gj + 1) / 2 + gi) + term];
                }
                aReg[J] = v;
            }
            const int nPts = nx * NB * 2;
            const float* Sf = reinterpret_cast<const float*>(S);
            float* Of = reinterpret_cast<float*>(O);
            const FastDiv d2NB(2 * NB);
            for (int p0 = 64 * wave; p0 < nPts; p0 += 64 * nWaves) {
                const int pt = p0 + lane;
                const bool valid = pt < nPts;
                const int k = d2NB.div(valid ? pt : 0), rem = (valid ? pt : 0) - k * 2 * NB;
                const int base = 2 * k * BS + rem;                       // float index of (k, subset 0, col, re/im); subset J: + 2 NB J
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int J = 0; J < 4; J++) {
                    const float b = (valid && J < nsub) ? Sf[base + 2 * NB * J] : 0.f;
                    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(aReg[J], b, acc, 0, 0, 0);
                }
                if (valid) {
                    const float e = (float)et[k * NB + (rem >> 1)];
#pragma unroll
                    for (int I = 0; I < 4; I++) if (I < nsub) Of[base + 2 * NB * I] = acc[I] * e;
                }
            }
        } else if (p.mix && nsub <= 16) {
            // The mix is a dense [n x n] x [n x points] contraction: it runs on the matrix cores.  v_mfma_f32_16x16x4_f32:
            // A[i][k] (lane: i = l&15, k = l>>4) = lambda matrix (rows = output subset, K = input subset, 4 per instruction),
            // B[k][j] (lane: k = l>>4, j = l&15) = 16 spectral values (re/im are separate "points"), D rows (l>>4)*4+r, column l&15.
            mixedOnMatrixCores = true;
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const int term = p.dispersion ? 1 : 0;
            const int lane = tid & 63, wave = tid >> 6, nWaves = NT / 64;
            const int ai = lane & 15, ak = lane >> 4;
            float aReg[4];
#pragma unroll
            for (int kc = 0; kc < 4; kc++) {
                const int J = kc * 4 + ak;
                float v = 0.f;
                if (ai < nsub && J < nsub) {
                    const int gi = p.gridSubset[ai], gj = p.gridSubset[J];
                    v = p.lambdas[2 * (gi > gj ? gi * (gi + 1) / 2 + gj : gj * (gj + 1) / 2 + gi) + term];
                }
                aReg[kc] = v;
            }
            const int nPts = nx * NB * 2, nGroups = (nPts + 15) / 16, nK = (nsub + 3) / 4;
            const float* Sf = reinterpret_cast<const float*>(S);
            float* Of = reinterpret_cast<float*>(O);
            for (int g = wave; g < nGroups; g += nWaves) {
                const int pt = 16 * g + ai;
                const bool valid = pt < nPts;
                const int k = dNB.div(pt >> 1), rem = pt - k * 2 * NB, col = rem >> 1, c = rem & 1;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kc = 0; kc < 4; kc++) {
                    if (kc < nK) {
                        const int J = kc * 4 + ak;
                        const float b = (valid && J < nsub) ? Sf[2 * (k * BS + J * NB + col) + c] : 0.f;
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aReg[kc], b, acc, 0, 0, 0);
                    }
                }
                const float e = valid ? (float)et[k * NB + col] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int I = ak * 4 + r;
                    if (valid && I < nsub) Of[2 * (k * BS + I * NB + col) + c] = acc[r] * e;
                }
            }
        }
    }
    if (!mixedOnMatrixCores) {
        const int term = p.dispersion ? 1 : 0;
        for (int it = tid; it < nx * BS; it += NT) {
            const int k = dBS.div(it), bb = it - k * BS;
            const int I = dNB.div(bb), col = bb - I * NB;
            Cx<Real> acc = {Real(0), Real(0)};
            if (p.mix) {
                const int gi = p.gridSubset[I];
                for (int J = 0; J < nsub; J++) {
                    const int gj = p.gridSubset[J];
                    const int slice = gi > gj ? gi * (gi + 1) / 2 + gj : gj * (gj + 1) / 2 + gi;
                    const Real lam = p.lambdas[2 * slice + term];
                    const Cx<Real> v = S[k * BS + J * NB + col];
                    acc.x += lam * v.x; acc.y += lam * v.y;
                }
            } else
                acc = S[it];
            const Real e = et[k * NB + col];
            O[it] = {acc.x * e, acc.y * e};
        }
    }
    Cx<Real>* R = fftLines<Real, R1, R2>(O, S, nx, p.d.fx, p.d.nfx, +1, tw, BS, BS, tid, NT);
    __syncthreads();
    for (int it = tid; it < nx * BS; it += NT) {
        const int k = dBS.div(it), bb = it - k * BS;
        const int sub = dNB.div(bb), col = bb - sub * NB;
        if (col < nbc) g[sub * strideSub + (size_t)k * strideK + c0 + col] = R[it];
    }
}

// ---------------------------------------------------------------------------------------------------
// Plane path (round 3).  The y pass, the fused x kernel and the inverse y pass above are three trips of the complex meshes through HBM and
// ~80 us of latency-bound work-groups on c3 (21 + 38..44 + 21).  A (subset, kz) plane of a mesh up to ~128^2 is 115 KB in single
// precision: it fits the 160 KB of LDS of one CU, and a 120^3 mesh with 4 subsets has 244 such planes -- one round of 256 CUs.  So, when
// the merge kernel has written the half-complex spectrum plane-major and brick-tiled ([slot][kz][brick][line]), ONE kernel does, per plane and entirely in LDS:
// forward FFT_y, forward FFT_x, multiplication by the reciprocal-space kernel, inverse FFT_x, inverse FFT_y, and (energy steps) the
// slice energies.  The transforms are the two-pass register FFT run IN PLACE: the forward passes leave frequency k1 + R1 k2 at position
// k1 R2 + k2, the kernel value is looked up at that permuted index, and the inverse passes run in the opposite order and undo the
// permutation -- no reordering pass, no second LDS buffer.  The lambda mix moves behind it: it is linear, so mixing the convolved
// potentials psi_J of the subsets in (x, y, kz) space gives the same phi_I = sum_J lambda_IJ psi_J as mixing in k-space; the inverse z
// kernel below (k_fftZInvMix) does it on the matrix cores while it loads its lines.  Slice energies need no k-space either: by
// Parseval over the plane, sum_{kx,ky} S_I conj(eterm S_J) = sum_{x,y} Q~_I conj(psi~_J), with Q~ the plane as the merge kernel left it
// (kept in the first buffer: the convolved planes go to a second one) and psi~ the plane this work-group has just produced.
// Replaces, on this path, reciprocalConvolution + gridEvaluateEnergy (platforms/common/src/kernels/pme.cc:138-274) and four of the six
// 1D FFT passes of the vendor FFT the reference calls.
// ---------------------------------------------------------------------------------------------------
template <typename Real, int R, int SIGN, int TWMODE, bool STRIDED, typename PRE>
__device__ __forceinline__ void planePass(Cx<Real>* P, const int lines, const int lineStride, const int elemStride, const int other, const Cx<Real>* tw,
                                          const int tid, const int NT, PRE pre) {
    // One R-point transform per task, in place.  STRIDED: task j = n2, element k at position k * other + j (pass over n1 / k1);
    // else task j = k1, element k at position j * R + k (pass over n2 / k2).  TWMODE 1: W_n^{j k} after the transform (forward),
    // 2: conj W_n^{j k} before it (inverse).  Lanes run along `lines`.
    const int tasks = lines * other;
    const FastDiv dl(lines);
    for (int t = tid; t < tasks; t += NT) {
        const int j = dl.div(t), line = t - j * lines;
        Cx<Real>* base = P + line * lineStride;
        Cx<Real> v[R];
#pragma unroll
        for (int k = 0; k < R; k++) {
            const int pos = STRIDED ? (k * other + j) : (j * R + k);
            v[k] = base[pos * elemStride];
            const Real sc = pre(line, pos);
            v[k].x *= sc; v[k].y *= sc;
        }
        if (TWMODE == 2 && j > 0) {
#pragma unroll
            for (int k = 1; k < R; k++) { Cx<Real> w = tw[j * k]; w.y = -w.y; v[k] = cmul(v[k], w); }
        }
        fftReg<Real, R>(v, SIGN);
        if (TWMODE == 1 && j > 0) {
#pragma unroll
            for (int k = 1; k < R; k++) v[k] = cmul(v[k], tw[j * k]);
        }
#pragma unroll
        for (int k = 0; k < R; k++) {
            const int pos = STRIDED ? (k * other + j) : (j * R + k);
            base[pos * elemStride] = v[k];
        }
    }
}
struct PlaneNoScale { template <typename... A> __device__ float operator()(A...) const { return 1.0f; } };

// Rectangular planes and square ones without a kernel of their own (round 4): the two axes take splits of their own from the plan at run
// time (PmePlanDims px1..py2, any product of two of the radices below).  One kernel serves every combination: each pass is a switch over
// the pass bodies per radix -- inlined, it allocates the registers of its largest arm (120 VGPRs, no scratch at 1024 threads; as calls
// the bodies needed 160 B of stack per lane).  The bodies address LDS through offsets from s_dyn, so they stay ds_ instructions either way.
#define SNB_PLANE_RADICES(X) X(5) X(6) X(7) X(8) X(9) X(10) X(12) X(15) X(16)
template <int R, int SIGN, int TWMODE, bool STRIDED, typename PRE>
__device__ __forceinline__ void planePassCall(const int pOff, const int lines, const int lineStride, const int elemStride, const int other, const int twOff,
                                                        const int tid, const int NT, PRE pre) {
    Cx<float>* base = reinterpret_cast<Cx<float>*>(s_dyn);
    planePass<float, R, SIGN, TWMODE, STRIDED>(base + pOff, lines, lineStride, elemStride, other, base + twOff, tid, NT, pre);
}
template <int SIGN, int TWMODE, bool STRIDED, typename PRE>
__device__ __forceinline__ void planePassDyn(const int R, const int pOff, const int lines, const int lineStride, const int elemStride, const int other, const int twOff,
                                             const int tid, const int NT, PRE pre) {
    switch (R) {
#define X(A) case A: planePassCall<A, SIGN, TWMODE, STRIDED>(pOff, lines, lineStride, elemStride, other, twOff, tid, NT, pre); break;
        SNB_PLANE_RADICES(X)
#undef X
        default: break;
    }
}
// what k_planeXY needs from its two axes: static splits (square planes, R1 x R2 on both axes, one table of roots) or the plan's
template <int R1, int R2> struct PlaneSplitsStatic {
    static constexpr bool dynamic = false;
    __device__ PlaneSplitsStatic(const PmePlanDims&) {}
    __device__ int rx1() const { return R1; } __device__ int rx2() const { return R2; } __device__ int ry1() const { return R1; } __device__ int ry2() const { return R2; }
};
struct PlaneSplitsDynamic {
    static constexpr bool dynamic = true;
    int x1, x2, y1, y2;
    __device__ PlaneSplitsDynamic(const PmePlanDims& d) : x1(d.px1), x2(d.px2), y1(d.py1), y2(d.py2) {}
    __device__ int rx1() const { return x1; } __device__ int rx2() const { return x2; } __device__ int ry1() const { return y1; } __device__ int ry2() const { return y2; }
};

// The reciprocal-space kernel value of every plane position, in the permuted order the in-place forward passes leave (position q of an
// axis holds frequency q / R2 + R1 (q % R2)): [kz][px][py], filled at rebuild time (the box and alpha are fixed between rebuilds).
template <typename Real> __global__ __launch_bounds__(256) void k_planeEterm(const PmeParams<Real> p, Real* table) {
    const int nx = p.d.nx, ny = p.d.ny;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)p.d.nzc * nx * ny) return;
    const int py = (int)(i % ny), px = (int)((i / ny) % nx), kz = (int)(i / ((size_t)nx * ny));
    table[i] = recipTerm<Real>(p, px / p.d.px2 + p.d.px1 * (px % p.d.px2), py / p.d.py2 + p.d.py1 * (py % p.d.py2), kz);      // (each axis in the permuted order of its own in-place transform)
}

template <int R1, int R2, int NT> __global__ __launch_bounds__(NT) void k_planeXY(const PmeParams<float> p, const int NBY) {
    SNB_PME_PRIO();
    SNB_TRACE_START(p.stepTrace, 8);
    using Real = float;
    using Splits = typename std::conditional<R1 == 0, PlaneSplitsDynamic, PlaneSplitsStatic<R1, R2>>::type;      // R1 == 0: rectangular plane, splits from the plan
    const Splits sp(p.d);
    const int nx = p.d.nx, ny = p.d.ny, nzc = p.d.nzc;      // launcher: static splits: nx == ny == R1 * R2; dynamic: nx = rx1 rx2, ny = ry1 ry2
    const int PY = ny | 1;                                  // odd pitch: lanes along x (stride PY) and lanes along y (stride 1) are both conflict-free
    const int slot = blockIdx.x / nzc, kz = blockIdx.x - slot * nzc;
    Cx<Real>* P = reinterpret_cast<Cx<Real>*>(s_dyn);                  // [nx][PY]
    Cx<Real>* tw = P + (size_t)nx * PY;                                // [nx] roots of unity of the x axis (and of y on square planes), then [ny] for y on rectangular ones
    const int twXOff = nx * PY, twYOff = Splits::dynamic ? nx * PY + nx : nx * PY;
    __shared__ double s_red[NT / 64];
    const int tid = threadIdx.x;
    const size_t planeElems = (size_t)nx * ny;
    const Cx<Real>* in = reinterpret_cast<const Cx<Real>*>(p.gridCplx) + ((size_t)slot * nzc + kz) * planeElems;
    // convolved plane: written in the order the inverse z kernel reads, [x][y tile][slot][kz][NBY] (its lines of one work-group are contiguous)
    Cx<Real>* out = reinterpret_cast<Cx<Real>*>(p.planeB) + ((size_t)slot * nzc + kz) * NBY;
    const int tilesY = (ny + NBY - 1) / NBY;
    const size_t tileStride = (size_t)p.nsub * nzc * NBY;
    const FastDiv dNBY(NBY);
    for (int k = tid; k < nx; k += NT) tw[k] = reinterpret_cast<const Cx<Real>*>(p.twx)[k];
    if (Splits::dynamic) for (int k = tid; k < ny; k += NT) tw[nx + k] = reinterpret_cast<const Cx<Real>*>(p.twy)[k];
    const FastDiv dny(ny);
    const int nPairs = (int)(planeElems >> 1);                         // ny is even (launcher): 16-byte accesses never straddle a row
    // the merge kernel's planes are brick-tiled, [kz][brick][line of the brick]: element e sits at x = bx cx + lx, y = by cy + ly
    const int cx = p.groupX * (nx / p.sortNcx), cy = p.groupY * (ny / p.sortNcy), nbyT = p.sortNcy / p.groupY, nlT = cx * cy;
    const FastDiv dnlT(nlT), dnbyT(nbyT), dcyT(cy);
    auto tiledIndex = [&](int e) { const int b = dnlT.div(e), w = e - b * nlT, bxi = dnbyT.div(b), byi = b - bxi * nbyT, lx = dcyT.div(w), ly = w - lx * cy;
                                   return (bxi * cx + lx) * PY + byi * cy + ly; };
    const bool pairsOk = !(cy & 1);                                    // an even brick width keeps a 16-byte pair inside one line of the brick
    if (pairsOk)
        batchedCopy<8, float4>(tid, nPairs, NT,
            [&](int e) { return reinterpret_cast<const float4*>(in)[e]; },
            [&](int e, const float4& v) { const int i = tiledIndex(2 * e); P[i] = {v.x, v.y}; P[i + 1] = {v.z, v.w}; });
    else
        batchedCopy<8, float2>(tid, (int)planeElems, NT,
            [&](int e) { return reinterpret_cast<const float2*>(in)[e]; },
            [&](int e, const float2& v) { P[tiledIndex(e)] = {v.x, v.y}; });
    __syncthreads();
    // forward y (lines = x rows, elements along y), forward x (lines = y columns, elements along x, stride PY); then position q of an axis
    // holds frequency (q / r2) + r1 * (q % r2): the kernel value (tabulated in that order) is applied while the first inverse pass loads
    const Real* et = p.planeEterm + (size_t)kz * planeElems;
    auto eterm = [et, ny](int line, int pos) { return et[pos * ny + line]; };
    if constexpr (!Splits::dynamic) {
        planePass<Real, R1, -1, 1, true>(P, nx, PY, 1, R2, tw, tid, NT, PlaneNoScale());
        __syncthreads();
        planePass<Real, R2, -1, 0, false>(P, nx, PY, 1, R1, tw, tid, NT, PlaneNoScale());
        __syncthreads();
        planePass<Real, R1, -1, 1, true>(P, ny, 1, PY, R2, tw, tid, NT, PlaneNoScale());
        __syncthreads();
        planePass<Real, R2, -1, 0, false>(P, ny, 1, PY, R1, tw, tid, NT, PlaneNoScale());
        __syncthreads();
        planePass<Real, R2, +1, 0, false>(P, ny, 1, PY, R1, tw, tid, NT, eterm);
        __syncthreads();
        planePass<Real, R1, +1, 2, true>(P, ny, 1, PY, R2, tw, tid, NT, PlaneNoScale());
        __syncthreads();
        planePass<Real, R2, +1, 0, false>(P, nx, PY, 1, R1, tw, tid, NT, PlaneNoScale());
        __syncthreads();
        planePass<Real, R1, +1, 2, true>(P, nx, PY, 1, R2, tw, tid, NT, PlaneNoScale());
        __syncthreads();
    } else {
        planePassDyn<-1, 1, true>(sp.ry1(), 0, nx, PY, 1, sp.ry2(), twYOff, tid, NT, PlaneNoScale());
        __syncthreads();
        planePassDyn<-1, 0, false>(sp.ry2(), 0, nx, PY, 1, sp.ry1(), twYOff, tid, NT, PlaneNoScale());
        __syncthreads();
        planePassDyn<-1, 1, true>(sp.rx1(), 0, ny, 1, PY, sp.rx2(), twXOff, tid, NT, PlaneNoScale());
        __syncthreads();
        planePassDyn<-1, 0, false>(sp.rx2(), 0, ny, 1, PY, sp.rx1(), twXOff, tid, NT, PlaneNoScale());
        __syncthreads();
        planePassDyn<+1, 0, false>(sp.rx2(), 0, ny, 1, PY, sp.rx1(), twXOff, tid, NT, eterm);
        __syncthreads();
        planePassDyn<+1, 2, true>(sp.rx1(), 0, ny, 1, PY, sp.rx2(), twXOff, tid, NT, PlaneNoScale());
        __syncthreads();
        planePassDyn<+1, 0, false>(sp.ry2(), 0, nx, PY, 1, sp.ry1(), twYOff, tid, NT, PlaneNoScale());
        __syncthreads();
        planePassDyn<+1, 2, true>(sp.ry1(), 0, nx, PY, 1, sp.ry2(), twYOff, tid, NT, PlaneNoScale());
        __syncthreads();
    }
    batchedCopy<8, float4>(tid, nPairs, NT,
        [&](int e) { const int x = dny.div(2 * e), y = 2 * e - x * ny; const Cx<Real> a = P[x * PY + y], b = P[x * PY + y + 1]; return make_float4(a.x, a.y, b.x, b.y); },
        [&](int e, const float4& v) { const int x = dny.div(2 * e), y = 2 * e - x * ny, t = dNBY.div(y); *reinterpret_cast<float4*>(out + ((size_t)x * tilesY + t) * tileStride + (y - t * NBY)) = v; });
    // per-slice energies (ReferencePME.cpp:487-491): E_IJ = sum_k eterm Re(S_I conj S_J) over the full mesh (1/2 on the diagonal), here as
    // sum over the plane of Re(Q~_I conj psi~_J), Hermitian weight 2 for interior kz; this work-group holds psi~_J, J = its slot.  The sum is
    // symmetric in (I, J), so either of the two work-groups can take a pair: the one with the higher slot when I + J is odd, the lower one
    // otherwise (with "all pairs I >= J" the planes of slot 0 read up to nsub - 1 more planes than the others and the launch waited for them:
    // 38-41 us against 24 on c3's derivative steps)
    if (p.wantEnergy && p.mix) {
        const int term = p.dispersion ? 1 : 0;
        const int gj = p.gridSubset[slot];
        const double w = (kz == 0 || 2 * kz == p.d.nz) ? 1.0 : 2.0;
        for (int I = 0; I < p.nsub; I++) {
            if (I != slot) { const int hi = I > slot ? I : slot, lo = I > slot ? slot : I; if ((((hi + lo) & 1) ? hi : lo) != slot) continue; }      // (uniform) the other work-group's pair
            const int gi = p.gridSubset[I];
            const int slice = gi > gj ? gi * (gi + 1) / 2 + gj : gj * (gj + 1) / 2 + gi;
            if (!p.sliceNeed[slice]) continue;      // (uniform)
            const Cx<Real>* qc = reinterpret_cast<const Cx<Real>*>(p.gridCplx) + ((size_t)I * nzc + kz) * planeElems;      // (brick-tiled, like this work-group's input)
            const float4* q = reinterpret_cast<const float4*>(qc);
            double acc = 0;
            if (pairsOk) for (int e0 = tid; e0 < nPairs; e0 += 4 * NT) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const int e = e0 + u * NT; if (e < nPairs) v[u] = q[e]; }
                float part = 0.f;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int e = e0 + u * NT;
                    if (e < nPairs) {
                        const int i = tiledIndex(2 * e);
                        const Cx<Real> a = P[i], b = P[i + 1];
                        part += (v[u].x * a.x + v[u].y * a.y) + (v[u].z * b.x + v[u].w * b.y);
                    }
                }
                acc += (double)part;
            }
            else for (int e0 = tid; e0 < (int)planeElems; e0 += 4 * NT) {
                float part = 0.f;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int e = e0 + u * NT;
                    if (e < (int)planeElems) { const Cx<Real> v = qc[e], a = P[tiledIndex(e)]; part += v.x * a.x + v.y * a.y; }
                }
                acc += (double)part;
            }
            acc *= (I == slot) ? 0.5 * w : w;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
            __syncthreads();
            if ((tid & 63) == 0) s_red[tid >> 6] = acc;
            __syncthreads();
            if (tid == 0) {
                double tot = 0; for (int k = 0; k < NT / 64; k++) tot += s_red[k];
                atomicAdd(&SNB_SLICE_E_PARTITION(p.sliceE, p.nsubTotal * (p.nsubTotal + 1))[2 * slice + term], tot);
            }
        }
    }
}

// ---- inverse z pass of the plane path: lambda mix (matrix cores) + half-complex -> real -------------------------------------------
// One work-group takes NBY consecutive y at one x, all held subsets: per (kz, y) point it loads the convolved potentials psi~_J of the
// subsets from the plane-major buffer, mixes them (phi_I = sum_J lambda[slice(I,J)] psi~_J; v_mfma_f32_4x4x1_16b_f32 with A = a column of
// the lambda matrix and B = the lanes' own values, as in k_convolveX: every lane ends up with the mixed values of ITS point, real and
// imaginary part as two accumulation chains; up to 8 subsets, two groups of four output rows), packs subsets 2m and 2m+1 as the real and
// imaginary line of one complex transform (Z = A + iB with A, B Hermitian) and runs the inverse z FFT into the real mesh [slot][x][y][z]
// the interpolation reads.  mix == 0 (sharded engines): no mix, the subsets' own potentials.
template <int R1, int R2, int NT> __global__ __launch_bounds__(NT) void k_fftZInvMix(const PmeParams<float> p, const int NBY) {
    SNB_PME_PRIO();
    SNB_TRACE_START(p.stepTrace, 9);
    using Real = float;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int nx = p.d.nx, ny = p.d.ny, nz = p.d.nz, nzc = p.d.nzc, nsub = p.nsub;
    const int NP = (nsub + 1) >> 1;                                      // complex transforms per y: subsets (2m, 2m+1)
    const int tilesY = (ny + NBY - 1) / NBY;
    const int x = blockIdx.x / tilesY, y0 = (blockIdx.x - x * tilesY) * NBY;
    const int nby = (ny - y0) < NBY ? (ny - y0) : NBY;
    const int nb = nby * NP, BS = NBY * NP + 1;
    // With a two-pass split the inverse transform runs IN PLACE, as in k_planeXY: the half-complex lines are written at the permuted positions
    // the forward passes would have left (frequency k1 + R1 k2 at position k1 R2 + k2), the inverse passes run in the opposite order and
    // leave z in natural order -- one LDS buffer instead of two, 16 KB instead of 33 KB on c3, i.e. seven work-groups per CU instead of four:
    // the 1800 work-groups of a 120^3 mesh are resident at once instead of in 1.8 rounds.  (Staged Stockham fallback: two buffers.)
    constexpr bool INPLACE = R1 > 0;
    Cx<Real>* A = reinterpret_cast<Cx<Real>*>(s_dyn);
    Cx<Real>* B = A + (INPLACE ? (size_t)0 : (size_t)nz * BS);
    Cx<Real>* tw = B + (size_t)nz * BS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < nz; k += NT) tw[k] = reinterpret_cast<const Cx<Real>*>(p.twz)[k];
    const FastDiv dR1(INPLACE ? R1 : 1);
    auto zpos = [&](int k) { if constexpr (INPLACE) { const int k2 = dR1.div(k); return (k - k2 * R1) * R2 + k2; } else return k; };      // LDS position of frequency k
    const int term = p.dispersion ? 1 : 0;
    float aReg[2][8];
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
        for (int J = 0; J < 8; J++) {
            const int I = 4 * g + (lane & 3);
            float v = 0.f;
            if (I < nsub && J < nsub) {
                if (p.mix) {
                    const int gi = p.gridSubset[I], gj = p.gridSubset[J];
                    v = p.lambdas[2 * (gi > gj ? gi * (gi + 1) / 2 + gj : gj * (gj + 1) / 2 + gi) + term];
                } else v = (I == J) ? 1.f : 0.f;
            }
            aReg[g][J] = v;
        }
    const size_t subStride = (size_t)nzc * NBY;                          // [x][y tile][slot][kz][NBY]: this work-group's lines are one contiguous block
    const Cx<Real>* src = reinterpret_cast<const Cx<Real>*>(p.planeB) + (size_t)blockIdx.x * nsub * subStride;
    const int nPts = nzc * NBY;
    const FastDiv dNBY(NBY);
    const int nG = (nsub + 3) >> 2;
    for (int p0 = 64 * wave; p0 < nPts; p0 += NT) {      // (uniform trip count per wave: the matrix-core instructions run with every lane)
        const int pt = p0 + lane;
        const int kz = dNBY.div(pt < nPts ? pt : 0), yy = (pt < nPts ? pt : 0) - kz * NBY;
        const bool valid = pt < nPts && yy < nby;
        const Cx<Real>* s0 = src + (valid ? pt : 0);
        Cx<Real> v[8];
#pragma unroll
        for (int J = 0; J < 8; J++) { v[J] = {0.f, 0.f}; if (valid && J < nsub) v[J] = s0[J * subStride]; }
        f32x4 re[2], im[2];
#pragma unroll
        for (int g = 0; g < 2; g++) { re[g] = {0.f, 0.f, 0.f, 0.f}; im[g] = {0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int g = 0; g < 2; g++) if (g < nG) {
#pragma unroll
            for (int J = 0; J < 8; J++) if (J < nsub) {
                re[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(aReg[g][J], v[J].x, re[g], 0, 0, 0);
                im[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(aReg[g][J], v[J].y, im[g], 0, 0, 0);
            }
        }
        if (valid) {
#pragma unroll
            for (int m = 0; m < 4; m++) if (m < NP) {
                const int g = m >> 1, r = (2 * m) & 3;
                const Cx<Real> a = {re[g][r], im[g][r]};
                Cx<Real> b = {0.f, 0.f};
                if (2 * m + 1 < nsub) b = {re[g][r + 1], im[g][r + 1]};
                const int c = yy * NP + m;
                A[zpos(kz) * BS + c] = {a.x - b.y, a.y + b.x};                                          // A_k + i B_k
                if (kz > 0 && nz - kz >= nzc) A[zpos(nz - kz) * BS + c] = {a.x + b.y, b.x - a.y};    // conj(A_k) + i conj(B_k)
            }
        }
    }
    Cx<Real>* R = A;
    if constexpr (INPLACE) {
        __syncthreads();
        planePass<Real, R2, +1, 0, false>(A, nb, 1, BS, R1, tw, tid, NT, PlaneNoScale());
        __syncthreads();
        planePass<Real, R1, +1, 2, true>(A, nb, 1, BS, R2, tw, tid, NT, PlaneNoScale());
    } else R = fftLines<Real, R1, R2>(A, B, nz, p.d.fz, p.d.nfz, +1, tw, nb, BS, tid, NT);
    __syncthreads();
    const FastDiv dz(nz), dnp(NP);
    for (int it = tid; it < nb * nz; it += NT) {
        const int c = dz.div(it), k = it - c * nz;
        const int yy = dnp.div(c), m = c - yy * NP;
        const Cx<Real> z = R[k * BS + c];
        p.gridReal[(((size_t)(2 * m) * nx + x) * ny + (y0 + yy)) * nz + k] = z.x;
        if (2 * m + 1 < nsub) p.gridReal[(((size_t)(2 * m + 1) * nx + x) * ny + (y0 + yy)) * nz + k] = z.y;
    }
    SNB_TRACE_END(p.stepTrace, 10);
}

// ---- launch dispatch over the instantiated (R1, R2) pairs ------------------------------------------
// threads per work-group of the y FFT pass: 512 over the same LDS tile (two rounds of register sub-transforms become one and twice the
// waves hide the tile's load latency: 24.4 -> 22.4 us per pass on c3; SNB_FFT_THREADS=256 restores the narrower groups).  The z pass
// was measured the other way round (16.9 us with 256 threads, 20.9 with 512) and keeps 256.
static int fftyThreads() { static const int n = getenv("SNB_FFT_THREADS") ? atoi(getenv("SNB_FFT_THREADS")) : 512; return n == 256 ? 256 : 512; }
template <typename Real, bool FWD> static void launchFftZ(int r1, int r2, dim3 grid, size_t lds, hipStream_t s, const PmeParams<Real>& p, int NL) {
#define X(A, B) if (r1 == A && r2 == B) { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fftZ<Real, FWD, A, B>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); SNB_STAMPED_LAUNCH(stampSlot(p, FWD ? 2 : 6), (k_fftZ<Real, FWD, A, B>), grid, dim3(256), lds, s, p, NL); return; }
    SNB_FFT_PAIRS(X)
#undef X
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fftZ<Real, FWD, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SNB_STAMPED_LAUNCH(stampSlot(p, FWD ? 2 : 6), (k_fftZ<Real, FWD, 0, 0>), grid, dim3(256), lds, s, p, NL);
}
template <typename Real> static void launchFftStrided(int r1, int r2, dim3 grid, size_t lds, hipStream_t s, const PmeParams<Real>& p, int n, size_t strideA, int nbTotal, size_t strideK, int NB,
                                                      int tilesPerA, int sign, int axis) {
#define X(A, B) if (r1 == A && r2 == B) { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fftStrided<Real, A, B>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); SNB_STAMPED_LAUNCH(axis == 1 ? stampSlot(p, sign < 0 ? 3 : 5) : -1, (k_fftStrided<Real, A, B>), grid, dim3(fftyThreads()), lds, s, p, n, strideA, nbTotal, strideK, NB, tilesPerA, sign, axis); return; }
    SNB_FFT_PAIRS(X)
#undef X
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fftStrided<Real, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SNB_STAMPED_LAUNCH(axis == 1 ? stampSlot(p, sign < 0 ? 3 : 5) : -1, (k_fftStrided<Real, 0, 0>), grid, dim3(fftyThreads()), lds, s, p, n, strideA, nbTotal, strideK, NB, tilesPerA, sign, axis);
}
template <typename Real> static void launchConvolveX(int r1, int r2, dim3 grid, size_t lds, hipStream_t s, const PmeParams<Real>& p, int NB, int nCols) {
#define X(A, B) if (r1 == A && r2 == B) { \
        if constexpr (sizeof(Real) == 8 && (A > 12 || B > 12)) { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_convolveX<Real, A, B, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                                                                 SNB_STAMPED_LAUNCH(stampSlot(p, 4), (k_convolveX<Real, A, B, 256>), grid, dim3(256), lds, s, p, NB, nCols); } \
        else { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_convolveX<Real, A, B>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
               SNB_STAMPED_LAUNCH(stampSlot(p, 4), (k_convolveX<Real, A, B>), grid, dim3(512), lds, s, p, NB, nCols); } \
        return; }
    SNB_FFT_PAIRS(X)
#undef X
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_convolveX<Real, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SNB_STAMPED_LAUNCH(stampSlot(p, 4), (k_convolveX<Real, 0, 0>), grid, dim3(512), lds, s, p, NB, nCols);
}


// Plane path (k_planeXY + k_fftZInvMix): single precision, a square mesh plane that fits LDS with its row padding, an instantiated
// two-pass split, at most 8 held subsets, and the own-atoms spreader's merge kernel in front (it writes the plane-major spectrum).
#define SNB_PLANE_PAIRS(X) X(6, 7) X(6, 9) X(8, 8) X(8, 10) X(9, 10) X(8, 12) X(10, 10) X(9, 12) X(10, 12) X(8, 16)
// square plane with a split that has a kernel of its own (static splits: no calls, one table of roots); everything else takes the run-time kernel
static bool planeSquare(const PmePlanDims& d) {
    static const bool dyn = getenv("SNB_PLANE_DYNAMIC") != nullptr;      // measurement aid: the run-time-split kernel on square planes too
    if (dyn || d.nx != d.ny || d.px1 != d.py1 || d.px2 != d.py2) return false;
#define X(A, B) if (d.px1 == A && d.px2 == B) return true;
    SNB_PLANE_PAIRS(X)
#undef X
    return false;
}
template <typename Real> static size_t planeLds(const PmeParams<Real>& p) { return sizeof(Cx<Real>) * ((size_t)p.d.nx * (p.d.ny | 1) + p.d.nx + (planeSquare(p.d) ? 0 : p.d.ny)); }
template <typename Real> static bool planePathOK(const PmeParams<Real>& p) {
    static const bool off = getenv("SNB_NO_PLANE_FFT") != nullptr;      // test switch: the three-kernel y / x / y pipeline
    static const bool noRect = getenv("SNB_NO_RECT_PLANES") != nullptr; // test switch: rectangular planes on the three-pass pipeline, as before round 4
    if (off || !std::is_same<Real, float>::value || !p.planeB || !p.planeEterm) return false;
    if ((p.d.ny & 1) || p.d.px1 <= 0 || p.d.py1 <= 0 || p.d.px1 * p.d.px2 != p.d.nx || p.d.py1 * p.d.py2 != p.d.ny) return false;
    if (p.nsub > 8 || planeLds(p) > 156 * 1024) return false;
    return planeSquare(p.d) || !noRect;
}
// y lines per work-group of the inverse z kernel = tile of the convolved planes' layout (c4, 8 subsets: plane + z kernel 45.5 + 55.0 us with 8, 61.2 + 42.4 with 4: 32-byte runs in the plane kernel's store)
static int planeTileY(const PmeParams<float>& p) {
    static const int env = getenv("SNB_ZMIX_NBY") ? atoi(getenv("SNB_ZMIX_NBY")) : 0;
    return (env == 2 || env == 4 || env == 8) ? env : 8;
}
static void launchPlaneXY(const PmeParams<float>& p, hipStream_t s) {
    const size_t lds = planeLds(p);
    const int NBY = planeTileY(p);
    const dim3 grid((unsigned)(p.nsub * p.d.nzc));
    static const int nt = getenv("SNB_PLANE_NT") ? atoi(getenv("SNB_PLANE_NT")) : 1024;      // threads per plane: 1024 or 768
    if (!planeSquare(p.d)) {      // rectangular plane: the kernel with run-time splits
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_planeXY<0, 0, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        SNB_STAMPED_LAUNCH(stampSlot(p, 4), (k_planeXY<0, 0, 1024>), grid, dim3(1024), lds, s, p, NBY);
        return;
    }
#define X(A, B) if (p.d.px1 == A && p.d.px2 == B) { \
        if (nt == 768) { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_planeXY<A, B, 768>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                         SNB_STAMPED_LAUNCH(stampSlot(p, 4), (k_planeXY<A, B, 768>), grid, dim3(768), lds, s, p, NBY); } \
        else { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_planeXY<A, B, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
               SNB_STAMPED_LAUNCH(stampSlot(p, 4), (k_planeXY<A, B, 1024>), grid, dim3(1024), lds, s, p, NBY); } \
        return; }
    SNB_PLANE_PAIRS(X)
#undef X
}
static void launchFftZInvMix(const PmeParams<float>& p, hipStream_t s) {
    const int NP = (p.nsub + 1) / 2;
    const int NBY = planeTileY(p);
    bool twoPass = false;
#define X(A, B) if (p.d.rz1 == A && p.d.rz2 == B) twoPass = true;
    SNB_FFT_PAIRS(X)
#undef X
    const size_t lds = sizeof(Cx<float>) * ((size_t)(twoPass ? 1 : 2) * p.d.nz * (NBY * NP + 1) + p.d.nz);      // (in place with a two-pass split)
    const dim3 grid((unsigned)(p.d.nx * ((p.d.ny + NBY - 1) / NBY)));
    static const int ntEnv = getenv("SNB_ZMIX_NT") ? atoi(getenv("SNB_ZMIX_NT")) : 0;
    const bool wide = ntEnv ? ntEnv == 512 : NBY * NP >= 24;      // 24+ complex transforms per work-group (5-8 subsets): 512 threads
#define X(A, B) if (p.d.rz1 == A && p.d.rz2 == B) { \
        if (wide) { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fftZInvMix<A, B, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                    SNB_STAMPED_LAUNCH(stampSlot(p, 6), (k_fftZInvMix<A, B, 512>), grid, dim3(512), lds, s, p, NBY); } \
        else { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fftZInvMix<A, B, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
               SNB_STAMPED_LAUNCH(stampSlot(p, 6), (k_fftZInvMix<A, B, 256>), grid, dim3(256), lds, s, p, NBY); } \
        return; }
    SNB_FFT_PAIRS(X)
#undef X
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fftZInvMix<0, 0, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SNB_STAMPED_LAUNCH(stampSlot(p, 6), (k_fftZInvMix<0, 0, 256>), grid, dim3(256), lds, s, p, NBY);
}
// Rebuild time: the kernel-value table of the plane path (no-op when the mesh does not qualify).
template <typename Real> bool launchPlaneEterm(const PmeParams<Real>& p, Real* table, hipStream_t s) {      // true: the table was (re)filled
    PmeParams<Real> q = p; q.planeEterm = table;
    if (!table || !planePathOK<Real>(q)) return false;
    const size_t n = (size_t)p.d.nzc * p.d.nx * p.d.ny;
    hipLaunchKernelGGL((k_planeEterm<Real>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, table);
    return true;
}
template bool launchPlaneEterm<float>(const PmeParams<float>&, float*, hipStream_t);
template bool launchPlaneEterm<double>(const PmeParams<double>&, double*, hipStream_t);
// The middle of the pipeline on the plane path (after a spreader that returned 2): convolution + x / y transforms per plane, then mix + inverse z.
template <typename Real> void launchPmePlanePath(const PmeParams<Real>& p, hipStream_t s) {
    if constexpr (std::is_same<Real, float>::value) {
        launchPlaneXY(p, s);
        launchFftZInvMix(p, s);
    }
}
template void launchPmePlanePath<float>(const PmeParams<float>&, hipStream_t);
template void launchPmePlanePath<double>(const PmeParams<double>&, hipStream_t);

// own-atoms spreader: k_spreadOwn over (brick, slab) work-groups, then k_spreadMerge per brick (fused with the forward z FFT when its
// buffers fit LDS).  Geometry (slabs, margin) and the buffers come from the engine (sized at rebuild time).
template <typename Real> static int launchSpreadOwn(const PmeParams<Real>& p, hipStream_t s) {
    static const bool off = getenv("SNB_NO_OWN_SPREAD") != nullptr;      // test switch: the scanning brick spreader
    if (off || p.ownSlabs < 2 || !p.ownPartial || p.d.nz > 256) return -1;      // (one slab's region of nz + 4 planes would wrap onto itself)
    static const bool noFixed = getenv("SNB_NO_FIXED_SPREAD") != nullptr;
    const int cx = p.groupX * (p.d.nx / p.sortNcx), cy = p.groupY * (p.d.ny / p.sortNcy);
    const int sz = p.d.nz / p.ownSlabs;
    if (sz & 1) return -1;      // (the regions are copied 8 or 16 bytes at a time)
    const bool fixed = std::is_same<Real, float>::value && !noFixed;
    const int RX = cx + 4 + 2 * p.ownMargin, RY = cy + 4 + 2 * p.ownMargin, RZ = sz + 4;
    const size_t ldsOwn = (fixed ? sizeof(int) : sizeof(double)) * (size_t)RX * RY * (fixed ? ownLineStride<true>(RZ) : ownLineStride<false>(RZ));
    const int M = p.ownMargin;
    const int reachX = (4 + M + cx - 1) / cx + (M + cx - 1) / cx + 1, reachY = (4 + M + cy - 1) / cy + (M + cy - 1) / cy + 1;      // bricks whose regions cover a line
    if (ldsOwn > 64 * 1024 || RX > p.d.nx || RY > p.d.ny || p.groupX * p.groupY > 16 || reachX > 3 || reachY > 3 || p.ownSlabs > 32 || sz < 4) return -1;
    if ((size_t)p.nsub * (p.sortNcx / p.groupX) * (p.sortNcy / p.groupY) * p.ownSlabs * RX * RY * RZ >= ((size_t)1 << 31)) return -1;      // (the merge kernel keeps 32-bit element offsets into ownPartial)
    const int cmax = fixed ? 4 : 2, chunk = (sz % cmax == 0) ? cmax : ((fixed && sz % 2 == 0) ? 2 : 1);      // values per load of the merge kernel (16, 8 or 4 bytes)
    const int nbricks = p.nsub * (p.sortNcx / p.groupX) * (p.sortNcy / p.groupY);
    const int nb = (cx * cy + 1) / 2;
    const size_t ldsFft = sizeof(Cx<Real>) * ((size_t)2 * p.d.nz * (nb + 1) + p.d.nz);
    static const bool noFuse = getenv("SNB_NO_FUSED_Z") != nullptr;
    const bool fuse = !noFuse && ldsFft <= 120 * 1024;
    const int plane = (fuse && planePathOK<Real>(p)) ? 1 : 0;      // plane-major spectrum for k_planeXY
#define SNB_OWN(FX) { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spreadOwn<Real, FX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsOwn); \
                      SNB_STAMPED_LAUNCH(stampSlot(p, 1), (k_spreadOwn<Real, FX>), dim3(nbricks * p.ownSlabs), dim3(512), ldsOwn, s, p); }
    // a long mesh in double precision gives a 256-thread work-group a dozen 16-byte chunks per thread, each three rounds of dependent loads
    // (c5, 180^3: 309 us): 512 threads there
    static const int mergeNt = getenv("SNB_MERGE_NT") ? atoi(getenv("SNB_MERGE_NT")) : 0;      // test switch: 256 / 512 threads per brick
    const bool wideMerge = mergeNt ? mergeNt == 512 : (size_t)cx * cy * (p.d.nz / chunk) > 6 * 256;
#define SNB_MERGE(FX, FZ, A, B) { if (wideMerge) { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spreadMerge<Real, FX, FZ, A, B, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(FZ ? ldsFft : 0)); \
                                      SNB_STAMPED_LAUNCH(stampSlot(p, 2), (k_spreadMerge<Real, FX, FZ, A, B, 512>), dim3(nbricks), dim3(512), (FZ ? ldsFft : 0), s, p, chunk, plane); } \
                                  else { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spreadMerge<Real, FX, FZ, A, B, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(FZ ? ldsFft : 0)); \
                                      SNB_STAMPED_LAUNCH(stampSlot(p, 2), (k_spreadMerge<Real, FX, FZ, A, B, 256>), dim3(nbricks), dim3(256), (FZ ? ldsFft : 0), s, p, chunk, plane); } }
    bool done = false;
    if constexpr (std::is_same<Real, float>::value) {
        if (fixed) {
            SNB_OWN(true)
            if (fuse) {
#define X(A, B) if (!done && p.d.rz1 == A && p.d.rz2 == B) { SNB_MERGE(true, true, A, B) done = true; }
                SNB_FFT_PAIRS(X)
#undef X
                if (!done) { SNB_MERGE(true, true, 0, 0) done = true; }
            } else { SNB_MERGE(true, false, 0, 0) done = true; }
        }
    }
    if (!done) {
        SNB_OWN(false)
        if (fuse) {
#define X(A, B) if (!done && p.d.rz1 == A && p.d.rz2 == B) { SNB_MERGE(false, true, A, B) done = true; }
            SNB_FFT_PAIRS(X)
#undef X
            if (!done) { SNB_MERGE(false, true, 0, 0) done = true; }
        } else { SNB_MERGE(false, false, 0, 0) }
    }
#undef SNB_OWN
#undef SNB_MERGE
    return plane ? 2 : (fuse ? 1 : 0);
}

static size_t ldsBudget() { return 96 * 1024; }

template <typename Real> static int pickBatch(size_t bytesPerBatchElem, int maxB, int minB = 1) {
    int b = (int)(ldsBudget() / bytesPerBatchElem);
    if (b > maxB) b = maxB;
    if (b < minB) b = minB;
    return b;
}

// Columns per work-group of the y pass.  Single precision: as many as fit the LDS budget, at most 32 (c3: 32 + 29 columns 27.8 us, three
// tiles of 21 28.4).  Double precision (round 3): at least three tiles of even width -- 90^3 (46 columns) 16 + 16 + 14: 16.5 / 14.9 us per
// pass against 21.5 / 20.6 for 32 + 14; 180^3 six tiles of 16: 139 / 128 against 145 / 133 for 17 x 5 + 6; 120^3 keeps three tiles
// (four tiles of 16: 48.6 against 40.8)
template <typename Real> static int fftyBatch(int ny, int nzc, int forced) {
    const int cap = pickBatch<Real>((size_t)2 * ny * sizeof(Cx<Real>), forced > 0 ? forced : 32);
    if (forced > 0 || sizeof(Real) == 4) return cap;
    const int tiles = (nzc + cap - 1) / cap;
    if (tiles < 3 && nzc >= 24) return (nzc + 2) / 3;
    return cap > 2 ? (cap & ~1) : cap;      // (balanced tiles of 21 on the 120^3 mesh: 42.6 us against 40.8 for 25 + 25 + 11)
}
template <typename Real> void launchPmeForwardFFT(const PmeParams<Real>& p, hipStream_t s, bool zDone) {
    const int nx = p.d.nx, ny = p.d.ny, nz = p.d.nz, nzc = p.d.nzc;
    // z: real -> half complex (unless the brick spreader already did it)
    if (!zDone) {
        int NC = pickBatch<Real>((size_t)2 * nz * sizeof(Cx<Real>), 17) - 1;   // complex lines per work-group (two real lines each)
        NC &= ~1;
        if (NC < 2) NC = 2;
        const int NL = 2 * NC;
        const size_t lds = (size_t)2 * nz * (NC + 1) * sizeof(Cx<Real>) + (size_t)nz * sizeof(Cx<Real>);
        const size_t nlines = (size_t)p.nsub * nx * ny;
        launchFftZ<Real, true>(p.d.rz1, p.d.rz2, dim3((unsigned)((nlines + NL - 1) / NL)), lds, s, p, NL);
    }
    // y
    {
        static const int nbEnv = getenv("SNB_FFTY_NB") ? atoi(getenv("SNB_FFTY_NB")) : 0;   // measured on c3: 8: 40 us, 16: 31, 21: 28.4, 32: 27.8
        const int NB = fftyBatch<Real>(ny, nzc, nbEnv);
        const size_t lds = (size_t)2 * ny * NB * sizeof(Cx<Real>) + (size_t)ny * sizeof(Cx<Real>);
        const int tilesPerA = (nzc + NB - 1) / NB;
        launchFftStrided<Real>(p.d.ry1, p.d.ry2, dim3((unsigned)(p.nsub * nx * tilesPerA)), lds, s, p, ny, (size_t)ny * nzc, nzc, (size_t)nzc, NB, tilesPerA, -1, 1);
    }
}

template <typename Real> void launchPmeConvolution(const PmeParams<Real>& p, hipStream_t s) {
    const int nx = p.d.nx;
    const int nCols = p.d.ny * p.d.nzc;
    const size_t perCol = (size_t)2 * nx * p.nsub * sizeof(Cx<Real>) + (size_t)nx * sizeof(Real);
    const size_t twBytes = (size_t)nx * sizeof(Cx<Real>);
    // columns per work-group: 8 adjacent (ky,kz) columns are one 64-byte line per (subset, kx) row -- measured on c3 (4 subsets):
    // NB = 4: 68 us, 5: 64, 7: 69, 8: 48.5, 16: 60.5 -- so 8 when that fits ~72 KB of LDS, else 4, 2, 1
    static const int ldsKB = getenv("SNB_CONV_LDS_KB") ? atoi(getenv("SNB_CONV_LDS_KB")) : 72;
    int NB = 8;
    while (NB > 1 && perCol * NB > (size_t)ldsKB * 1024) NB >>= 1;
    const size_t lds = perCol * NB + twBytes;
    launchConvolveX<Real>(p.d.rx1, p.d.rx2, dim3((unsigned)((nCols + NB - 1) / NB)), lds, s, p, NB, nCols);
}

template <typename Real> void launchPmeInverseFFT(const PmeParams<Real>& p, hipStream_t s) {
    const int nx = p.d.nx, ny = p.d.ny, nz = p.d.nz, nzc = p.d.nzc;
    {
        static const int nbEnv = getenv("SNB_FFTY_NB") ? atoi(getenv("SNB_FFTY_NB")) : 0;   // measured on c3: 8: 40 us, 16: 31, 21: 28.4, 32: 27.8
        const int NB = fftyBatch<Real>(ny, nzc, nbEnv);
        const size_t lds = (size_t)2 * ny * NB * sizeof(Cx<Real>) + (size_t)ny * sizeof(Cx<Real>);
        const int tilesPerA = (nzc + NB - 1) / NB;
        launchFftStrided<Real>(p.d.ry1, p.d.ry2, dim3((unsigned)(p.nsub * nx * tilesPerA)), lds, s, p, ny, (size_t)ny * nzc, nzc, (size_t)nzc, NB, tilesPerA, +1, 1);
    }
    {
        int NC = pickBatch<Real>((size_t)2 * nz * sizeof(Cx<Real>), 17) - 1;   // complex lines per work-group (two real lines each)
        NC &= ~1;
        if (NC < 2) NC = 2;
        const int NL = 2 * NC;
        const size_t lds = (size_t)2 * nz * (NC + 1) * sizeof(Cx<Real>) + (size_t)nz * sizeof(Cx<Real>);
        const size_t nlines = (size_t)p.nsub * nx * ny;
        launchFftZ<Real, false>(p.d.rz1, p.d.rz2, dim3((unsigned)((nlines + NL - 1) / NL)), lds, s, p, NL);
    }
}

// x axis alone (used by the FFT unit-test hook; the pipeline itself uses the fused k_convolveX)
template <typename Real> void launchPmeFFTX(const PmeParams<Real>& p, int sign, hipStream_t s) {
    const int nx = p.d.nx, nCols = p.d.ny * p.d.nzc;
    const int NB = pickBatch<Real>((size_t)2 * nx * sizeof(Cx<Real>), 16);
    const size_t lds = (size_t)2 * nx * NB * sizeof(Cx<Real>) + (size_t)nx * sizeof(Cx<Real>);
    const int tilesPerA = (nCols + NB - 1) / NB;
    launchFftStrided<Real>(p.d.rx1, p.d.rx2, dim3((unsigned)(p.nsub * tilesPerA)), lds, s, p, nx, (size_t)nx * nCols, nCols, (size_t)nCols, NB, tilesPerA, sign, 0);
}
template void launchPmeFFTX<float>(const PmeParams<float>&, int, hipStream_t);
template void launchPmeFFTX<double>(const PmeParams<double>&, int, hipStream_t);

// ---------------------------------------------------------------------------------------------------
// Force interpolation (ReferencePME.cpp:598-702): 32 lanes per atom, lane = (x,y) stencil row.
// Unsharded: the atom reads the one pre-mixed grid of its own subset.  Sharded (mix=0): it loops over the
// grids this engine holds, scaling by lambda[slice(s_i, J)], and also accumulates E = 1/2 q psi_J(r_i).
// ---------------------------------------------------------------------------------------------------
template <typename Real> __global__ __launch_bounds__(256) void k_interpolate(const PmeParams<Real> p) {
    extern __shared__ double s_sliceE[];   // [2*S], sharded energy evaluation only
    const bool shardE = !p.mix && p.wantEnergy;
    const int nS2 = p.nsubTotal * (p.nsubTotal + 1);
    if (shardE) { for (int i = threadIdx.x; i < nS2; i += 256) s_sliceE[i] = 0.0; __syncthreads(); }
    const int gid = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int r = threadIdx.x & 31;
    if (gid < p.natoms) {
    const int si = p.atomSubset[gid];
    const Real q = si >= 0 ? pmeCharge(p, gid) : Real(0);
    const auto pos = p.posq[gid];
    int idx[3]; Real fr[3];
    gridCoord<Real>(p.recip, p.recipLo, pos.x, pos.y, pos.z, p.d.nx, p.d.ny, p.d.nz, idx, fr);
    Real tx[5], ty[5], tz[5], dx[5], dy[5], dz[5];
    bspline5<Real>(fr[0], tx, dx); bspline5<Real>(fr[1], ty, dy); bspline5<Real>(fr[2], tz, dz);
    Real fx = 0, fy = 0, fz = 0;
    const int term = p.dispersion ? 1 : 0;
    if (r < 25 && si >= 0 && q != Real(0)) {
        const int ix = r / 5, iy = r - ix * 5;
        int xi = idx[0] + ix; if (xi >= p.d.nx) xi -= p.d.nx;
        int yi = idx[1] + iy; if (yi >= p.d.ny) yi -= p.d.ny;
        Real txv = 0, dxv = 0, tyv = 0, dyv = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) { if (k == ix) { txv = tx[k]; dxv = dx[k]; } if (k == iy) { tyv = ty[k]; dyv = dy[k]; } }
        const int nG = p.mix ? 1 : p.nsub;
        for (int gI = 0; gI < nG; gI++) {
            int slot; Real lam = 1;
            if (p.mix) slot = p.atomGrid[gid];   // own subset's mixed grid
            else {
                slot = gI;
                const int gj = p.gridSubset[gI];
                const int slice = si > gj ? si * (si + 1) / 2 + gj : gj * (gj + 1) / 2 + si;
                lam = p.lambdas[2 * slice + term];
            }
            if (slot < 0) continue;
            const Real* row = p.gridReal + (((size_t)slot * p.d.nx + xi) * p.d.ny + yi) * p.d.nz;
            Real sz = 0, sdz = 0;
#pragma unroll
            for (int iz = 0; iz < 5; iz++) {
                int zi = idx[2] + iz; if (zi >= p.d.nz) zi -= p.d.nz;
                const Real gv = row[zi];
                sz += tz[iz] * gv; sdz += dz[iz] * gv;
            }
            fx += lam * dxv * tyv * sz; fy += lam * txv * dyv * sz; fz += lam * txv * tyv * sdz;
            if (shardE) {
                // E[slice(s_i, J)] += 1/2 q_i psi_J(r_i): equal to the k-space Gram sum because spreading and interpolation
                // use the same B-splines; the off-diagonal slice receives its two halves from the two owner ranks
                const int gj = p.gridSubset[gI];
                const int slice = si > gj ? si * (si + 1) / 2 + gj : gj * (gj + 1) / 2 + si;
                __hip_atomic_fetch_add(&s_sliceE[2 * slice + term], 0.5 * (double)q * (double)(txv * tyv * sz), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { fx += __shfl_xor(fx, o, 64); fy += __shfl_xor(fy, o, 64); fz += __shfl_xor(fz, o, 64); }
    if (r == 0) {
        const Real nx = p.d.nx, ny = p.d.ny, nz = p.d.nz;
        const Real gx = -q * (fx * nx * p.recip[0]);
        const Real gy = -q * (fx * nx * p.recip[3] + fy * ny * p.recip[4]);
        const Real gz = -q * (fx * nx * p.recip[6] + fy * ny * p.recip[7] + fz * nz * p.recip[8]);
        if (p.dispersion) { p.fpx[gid] += gx; p.fpy[gid] += gy; p.fpz[gid] += gz; }   // second pipeline of LJPME adds
        else { p.fpx[gid] = gx; p.fpy[gid] = gy; p.fpz[gid] = gz; }
    }
    }
    if (shardE) {
        __syncthreads();
        for (int i = threadIdx.x; i < nS2; i += 256) { const double v = s_sliceE[i]; if (v != 0.0) atomicAdd(&SNB_SLICE_E_PARTITION(p.sliceE, p.nsubTotal * (p.nsubTotal + 1))[i], v); }
    }
}

// ---------------------------------------------------------------------------------------------------
// Brick interpolation: the mirror image of k_spreadBrick.  One work-group stages the (c_x+6) x (c_y+6) x nz neighbourhood of one
// column brick of a potential grid in LDS with coalesced loads (halo: 4 stencil cells + 1 cell of drift on either side), then ONE
// THREAD PER ATOM evaluates its 125-point stencil out of LDS (~14 wave-instructions per atom instead of ~180 for the 32-lanes-per-atom
// gather kernel k_interpolate, which recomputes the B-splines 32 times and reduces by shuffles).
//
// Sharded engines (mix == 0): a rank holds the UNMIXED potential grids of its own subsets and every atom needs
// sum_J lambda[slice(s_i, J)] * (gradient of grid J).  One work-group per column brick loops over the held grids: stage the brick of
// grid J, then one thread per atom of ANY subset in the brick's columns adds lambda * gradient into the atom's reciprocal force
// (plain read-modify-write: the work-group is the only writer of its columns' atoms; the gather pass cleared the arrays), and --
// on energy steps -- E[slice(s_i, J)] += 1/2 q_i psi_J(r_i).  Replaces the 32-lanes-per-atom gather kernel, which
// cost ~80 us per held grid at 300k atoms.
// (occupancy note: ~100 VGPRs => one 1024-thread work-group per CU, 400 bricks = two rounds of ~20 us on c3; forcing 64 VGPRs spills
// and measures 71 us, 512-thread groups 56 us, z slabs 70 us -- this shape, 52 us, is the best of those)
template <typename Real, int NT> __global__ __launch_bounds__(NT) void k_interpolateBricks(const PmeParams<Real> p, const int zSlabs) {
    SNB_TRACE_START(p.stepTrace, 11);
    extern __shared__ __align__(16) unsigned char s_brick_raw[];
    constexpr int HALO_LO = 1, EXTRA = 6;
    const int ncx = p.sortNcx, ncy = p.sortNcy, nz = p.d.nz;
    const int cx = p.groupX * (p.d.nx / ncx), cy = p.groupY * (p.d.ny / ncy);
    const int nby = ncy / p.groupY;
    const int ncol = ncx * ncy;
    // the brick is cut into zSlabs slabs along z (more, smaller work-groups): a slab owns the atoms whose base cell lies in it and
    // stages sz + 4 planes (the stencil reaches 4 cells up, with periodic wrap)
    const int zs = blockIdx.x % zSlabs, brickId = blockIdx.x / zSlabs;
    const int sz = nz / zSlabs, z0 = zs * sz, bz = sz + 4;      // four wrap-around planes on top even for a single slab: the five z points of a stencil line are then always consecutive in LDS
    const int Bx = brickId / nby, By = brickId - Bx * nby;
    const int x0 = Bx * cx, y0 = By * cy;
    const int bx = cx + EXTRA, by = cy + EXTRA;
    const int tid = threadIdx.x;
    // energy / derivative steps whose last kernel this is: the last work-group also sums the 64 partitions of the raw slice energies and adds
    // the closed-form terms (k_finishSliceEnergies, ~4 us + a launch gap as a kernel of its own).  Every contribution is complete by now: the
    // tile kernel's launches and the pair lists precede this kernel, the reciprocal sums were made by the plane / x kernels, and an
    // unsharded interpolation adds none.  One wave per entry, one lane per partition.
    if (p.finOut != nullptr && blockIdx.x == gridDim.x - 1) {
        const int lane = tid & 63;
        for (int i = tid >> 6; i < p.finN; i += NT / 64) {
            double acc = 0;
            for (int part = lane; part < SNB_SLICE_E_PARTS; part += 64) acc += p.finParts[(size_t)part * p.finN + i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
            if (lane == 0) p.finOut[i] = acc + sliceFinishClosedForm(p.fin, i);
        }
    }
    Real* brick = reinterpret_cast<Real*>(s_brick_raw);
    double* sE = reinterpret_cast<double*>(s_brick_raw + ((sizeof(Real) * (size_t)bx * by * bz + 15) & ~(size_t)15));   // [2*S] on energy steps
    const bool wantE = p.wantEnergy != 0 && !p.mix;            // unsharded energies come from the k-space Gram sums (k_convolveX)
    const int nS2 = p.nsubTotal * (p.nsubTotal + 1);
    const int term = p.dispersion ? 1 : 0;
    if (wantE) for (int i = tid; i < nS2; i += NT) sE[i] = 0.0;
    __shared__ int s_begin[256], s_pref[257];
    const int nr = p.nsubTotal * p.groupX * p.groupY;          // launcher guarantees nr <= 256
    if (tid < nr) {
        const int si = tid / (p.groupX * p.groupY), gxy = tid - si * (p.groupX * p.groupY);
        const int gx = gxy / p.groupY, gy = gxy - gx * p.groupY;
        const int2 rg = p.colRange[(size_t)si * ncol + (Bx * p.groupX + gx) * ncy + By * p.groupY + gy];
        s_begin[tid] = rg.x; s_pref[tid + 1] = rg.y > rg.x ? rg.y - rg.x : 0;
    }
    __syncthreads();
    if (tid == 0) { int acc = 0; s_pref[0] = 0; for (int r = 0; r < nr; r++) { acc += s_pref[r + 1]; s_pref[r + 1] = acc; } }
    __syncthreads();
    const int total = s_pref[nr];
    if (total == 0) return;
    long long tA = 0, tLoad = 0, tComp = 0;
    const bool trace = p.trace != nullptr;
    const FastDiv dnz(bz), dby(by);
    const int g2 = p.groupX * p.groupY;
    for (int slot = 0; slot < p.nsub; slot++) {
        const int gj = p.gridSubset[slot];
        // unsharded (mix == 1): grid `slot` is the lambda-mixed potential felt by subset gj, so only that subset's atoms read it (and a
        // brick without such atoms is not even staged); sharded: every subset's atoms read every held grid, scaled by lambda
        const int vBegin = p.mix ? s_pref[gj * g2] : 0, vEnd = p.mix ? s_pref[(gj + 1) * g2] : total;
        if (vBegin == vEnd) continue;                          // uniform over the work-group
        const Real* g = p.gridReal + (size_t)slot * p.d.nx * p.d.ny * nz;
        __syncthreads();                                       // previous grid's readers are done with the brick
        if (trace) tA = (long long)wall_clock64();
        bool staged = false;
        if constexpr (std::is_same<Real, float>::value) {
            if (zSlabs == 1 && (nz & 3) == 0) {
                // whole z lines, 16 bytes per lane: a half-wave (31 of its 32 lanes for nz = 120) copies one line of nz + 4 floats, the last
                // chunk being the wrap-around planes; no division per element, and every request of a thread is in flight at once
                const int nLines = bx * by, chunks = bz >> 2;              // bz = nz + 4
                const int half = tid >> 5, hl = tid & 31;
                for (int c0 = 0; c0 < chunks; c0 += 32) {
                    const int ch = c0 + hl;
#pragma unroll 8
                    for (int line = half; line < nLines; line += NT / 32) {
                        if (ch < chunks) {
                            const int lx = dby.div(line), ly = line - lx * by;
                            int x = x0 + lx - HALO_LO; if (x < 0) x += p.d.nx; else if (x >= p.d.nx) x -= p.d.nx;
                            int y = y0 + ly - HALO_LO; if (y < 0) y += p.d.ny; else if (y >= p.d.ny) y -= p.d.ny;
                            int zg = 4 * ch; if (zg >= nz) zg -= nz;
                            *reinterpret_cast<float4*>(brick + (size_t)line * bz + 4 * ch) = *reinterpret_cast<const float4*>(g + ((size_t)x * p.d.ny + y) * nz + zg);
                        }
                    }
                }
                staged = true;
            }
        }
        // general shape: eight loads in flight per thread (one per trip left the staging latency-bound: 27 round trips, 14.6 us of a 29 us work-group)
        if (!staged) for (int i0 = tid; i0 < bx * by * bz; i0 += 8 * NT) {
            Real v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + u * NT;
                v[u] = Real(0);
                if (i < bx * by * bz) {
                    const int l = dnz.div(i), z = i - l * bz;
                    const int lx = dby.div(l), ly = l - lx * by;
                    int x = x0 + lx - HALO_LO; if (x < 0) x += p.d.nx; else if (x >= p.d.nx) x -= p.d.nx;
                    int y = y0 + ly - HALO_LO; if (y < 0) y += p.d.ny; else if (y >= p.d.ny) y -= p.d.ny;
                    int zg = z0 + z; if (zg >= nz) zg -= nz;
                    v[u] = g[((size_t)x * p.d.ny + y) * nz + zg];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; u++) { const int i = i0 + u * NT; if (i < bx * by * bz) brick[i] = v[u]; }
        }
        __syncthreads();
        if (trace) { const long long t = (long long)wall_clock64(); tLoad += t - tA; tA = t; }
        // the atoms of ALL subsets in the brick's columns, as one concatenated index space (their runs are short when there are many
        // subsets: one pass per subset would leave most of the 1024 threads idle)
        {
            for (int v0 = vBegin; v0 < vEnd; v0 += NT) {
                const int v = v0 + tid;
                if (v < vEnd) {
                    int r = 0;
#pragma unroll
                    for (int st = 128; st > 0; st >>= 1) if (r + st < nr && s_pref[r + st] <= v) r += st;
                    const int a = s_begin[r] + (v - s_pref[r]);
                    const int si = p.atomSubset[a];
                    const Real q = si >= 0 ? pmeCharge(p, a) : Real(0);
                    // the step's user-order force of this atom (fused k_finishForces): direct-space accumulator + reciprocal force
                    auto deliver = [&](Real rx_, Real ry_, Real rz_) {
                        const int u = p.sortedToUser[a];
                        if (u < 0) return;
                        if (p.dfixed) {      // SNB_MIXED: 64-bit fixed-point accumulators, summed with the reciprocal part in double
                            const double k = 1.0 / 4294967296.0;
                            const double X = (double)reinterpret_cast<const long long*>(p.dfx)[(size_t)a * p.dfs] * k + (double)rx_, Y = (double)reinterpret_cast<const long long*>(p.dfy)[(size_t)a * p.dfs] * k + (double)ry_,
                                         Z = (double)reinterpret_cast<const long long*>(p.dfz)[(size_t)a * p.dfs] * k + (double)rz_;
                            if (p.outIsDouble) {
                                double* o = reinterpret_cast<double*>(p.outForces) + 3 * (size_t)u;
                                if (p.outAccumulate) { o[0] += X; o[1] += Y; o[2] += Z; } else { o[0] = X; o[1] = Y; o[2] = Z; }
                            } else {
                                float* o = reinterpret_cast<float*>(p.outForces) + 3 * (size_t)u;
                                if (p.outAccumulate) { o[0] += (float)X; o[1] += (float)Y; o[2] += (float)Z; } else { o[0] = (float)X; o[1] = (float)Y; o[2] = (float)Z; }
                            }
                            return;
                        }
                        const Real X = p.dfx[(size_t)a * p.dfs] + rx_, Y = p.dfy[(size_t)a * p.dfs] + ry_, Z = p.dfz[(size_t)a * p.dfs] + rz_;
                        if (p.outIsDouble) {
                            double* o = reinterpret_cast<double*>(p.outForces) + 3 * (size_t)u;
                            if (p.outAccumulate) { o[0] += (double)X; o[1] += (double)Y; o[2] += (double)Z; } else { o[0] = (double)X; o[1] = (double)Y; o[2] = (double)Z; }
                        } else {
                            float* o = reinterpret_cast<float*>(p.outForces) + 3 * (size_t)u;
                            if (p.outAccumulate) { o[0] += (float)X; o[1] += (float)Y; o[2] += (float)Z; } else { o[0] = (float)X; o[1] = (float)Y; o[2] = (float)Z; }
                        }
                    };
                    if (q == Real(0)) {                        // padding slots inside a run, uncharged atoms
                        if (p.outForces && si >= 0 && (zSlabs == 1 || zs == 0)) deliver(p.fpx[a], p.fpy[a], p.fpz[a]);
                        continue;
                    }
                    const int slice = si > gj ? si * (si + 1) / 2 + gj : gj * (gj + 1) / 2 + si;
                    const Real lam = p.mix ? Real(1) : p.lambdas[2 * slice + term];
                    const auto pos = p.posq[a];
                    int idx[3]; Real fr[3];
                    gridCoord<Real>(p.recip, p.recipLo, pos.x, pos.y, pos.z, p.d.nx, p.d.ny, nz, idx, fr);
                    if (idx[2] < z0 || idx[2] >= z0 + sz) continue;       // another slab's atom
                    int rx = idx[0] - x0; if (rx > p.d.nx / 2) rx -= p.d.nx; else if (rx < -(p.d.nx / 2)) rx += p.d.nx;
                    int ry = idx[1] - y0; if (ry > p.d.ny / 2) ry -= p.d.ny; else if (ry < -(p.d.ny / 2)) ry += p.d.ny;
                    Real tx[5], ty[5], tz[5], dx[5], dy[5], dz[5];
                    bspline5<Real>(fr[0], tx, dx); bspline5<Real>(fr[1], ty, dy); bspline5<Real>(fr[2], tz, dz);
                    int zi[5];                                           // global (wrapped) z of the five stencil planes (slow path only)
#pragma unroll
                    for (int iz = 0; iz < 5; iz++) { int z = idx[2] + iz; zi[iz] = z >= nz ? z - nz : z; }
                    const int zb = idx[2] - z0;                          // slab-local z of the first plane; the other four follow without a wrap
                    Real fx = 0, fy = 0, fz = 0, psi = 0;
                    const bool inBrick = rx >= -HALO_LO && rx + 4 < bx - HALO_LO && ry >= -HALO_LO && ry + 4 < by - HALO_LO;
                    // separable accumulation: per x plane the y-sums of (value, y-derivative, z-derivative), then one x step -- 13 FMAs per
                    // stencil line instead of 18; the brick / global-memory choice is made once per atom, outside the 25 lines
                    auto plane = [&](auto&& lineOf, int ix) {
                        Real sV = 0, sDy = 0, sDz = 0;
#pragma unroll
                        for (int iy = 0; iy < 5; iy++) {
                            const Real* line = lineOf(ix, iy);
                            Real sv = 0, sdz = 0;
#pragma unroll
                            for (int iz = 0; iz < 5; iz++) { const Real gv = line[iz]; sv += tz[iz] * gv; sdz += dz[iz] * gv; }
                            sV += ty[iy] * sv; sDy += dy[iy] * sv; sDz += ty[iy] * sdz;
                        }
                        fx += dx[ix] * sV; fy += tx[ix] * sDy; fz += tx[ix] * sDz; psi += tx[ix] * sV;
                    };
                    if (inBrick) {
                        const Real* base = brick + (size_t)((rx + HALO_LO) * by + (ry + HALO_LO)) * bz + zb;
#pragma unroll
                        for (int ix = 0; ix < 5; ix++) plane([&](int jx, int jy) { return base + (size_t)(jx * by + jy) * bz; }, ix);
                    } else {   // drifted further than the halo since the last re-sort: correct but slow path through global memory
                        for (int ix = 0; ix < 5; ix++) {
                            int x = idx[0] + ix; if (x >= p.d.nx) x -= p.d.nx;
                            for (int iy = 0; iy < 5; iy++) {
                                int y = idx[1] + iy; if (y >= p.d.ny) y -= p.d.ny;
                                const Real* line = g + ((size_t)x * p.d.ny + y) * nz;
                                Real sv = 0, sdz = 0;
#pragma unroll
                                for (int iz = 0; iz < 5; iz++) { const Real gv = line[zi[iz]]; sv += tz[iz] * gv; sdz += dz[iz] * gv; }
                                fx += dx[ix] * ty[iy] * sv; fy += tx[ix] * dy[iy] * sv; fz += tx[ix] * ty[iy] * sdz; psi += tx[ix] * ty[iy] * sv;
                            }
                        }
                    }
                    const Real nx = p.d.nx, ny = p.d.ny, nzr = p.d.nz;
                    const Real ql = -q * lam;
                    const Real gx = p.fpx[a] + ql * (fx * nx * p.recip[0]);
                    const Real gy = p.fpy[a] + ql * (fx * nx * p.recip[3] + fy * ny * p.recip[4]);
                    const Real gz = p.fpz[a] + ql * (fx * nx * p.recip[6] + fy * ny * p.recip[7] + fz * nzr * p.recip[8]);
                    // (the reciprocal force is stored even when this kernel also delivers the step's user-order force: snb_get_forces into
                    // ANOTHER buffer rebuilds the force from the direct accumulators + fpx..)
                    p.fpx[a] = gx; p.fpy[a] = gy; p.fpz[a] = gz;
                    if (p.outForces) deliver(gx, gy, gz);
                    if (wantE) __hip_atomic_fetch_add(&sE[2 * slice + term], 0.5 * (double)q * (double)psi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        if (trace) { __syncthreads(); tComp += (long long)wall_clock64() - tA; }
    }
    if (trace && tid == 0) { atomicAdd((unsigned long long*)&p.trace[0], (unsigned long long)tLoad); atomicAdd((unsigned long long*)&p.trace[1], (unsigned long long)tComp); atomicAdd((unsigned long long*)&p.trace[2], 1ull); }
    if (wantE) {
        __syncthreads();
        for (int i = tid; i < nS2; i += NT) { const double v = sE[i]; if (v != 0.0) atomicAdd(&SNB_SLICE_E_PARTITION(p.sliceE, p.nsubTotal * (p.nsubTotal + 1))[i], v); }
    }
    SNB_TRACE_END(p.stepTrace, 12);
}

template <typename Real> static bool launchInterpolateBricks(const PmeParams<Real>& p0, hipStream_t s) {
    PmeParams<Real> p = p0;
    {
        const int cx = p.groupX * (p.d.nx / p.sortNcx), cy = p.groupY * (p.d.ny / p.sortNcy);
        static const int zsEnv = getenv("SNB_INTERP_ZSLABS") ? atoi(getenv("SNB_INTERP_ZSLABS")) : 0;
        int zSlabs = 1;      // measured on c3: 1 slab 52 us, 2 slabs 70, 4 slabs 72 (every slab rescans the columns' atoms)
        if (zsEnv > 0 && p.d.nz % zsEnv == 0 && p.d.nz / zsEnv >= 8) zSlabs = zsEnv;
        auto ldsFor = [&](int slabs) { return ((sizeof(Real) * (size_t)(cx + 6) * (cy + 6) * (p.d.nz / slabs + 4) + 15) & ~(size_t)15) + sizeof(double) * p.nsubTotal * (p.nsubTotal + 1); };
        // a brick that does not fit LDS in one piece (double precision on a large mesh: 12 x 12 x 184 doubles = 212 KB for the 180^3 mesh of
        // c5) is cut into the fewest z slabs that do, rather than falling back to the 32-lanes-per-atom gather kernel (424 us there)
        if (zsEnv <= 0) for (int k = 2; k <= 8 && ldsFor(zSlabs) > 150 * 1024; k++) if (p.d.nz % k == 0 && p.d.nz / k >= 8) zSlabs = k;
        // few, wide bricks (a coarse mesh whose bricks span 2 x 2 sort columns: the 60^3 dispersion mesh of c3l has 100 of them, 3000 atoms
        // each, on 256 CUs: 58.8 us against 26.6 for the 120^3 Coulomb mesh): slabs buy work-groups; every slab rescans the bricks' atoms,
        // which is why the fine mesh (400 bricks) does not take them
        if (zsEnv <= 0 && zSlabs == 1) {
            const int nb1 = (p.sortNcx / p.groupX) * (p.sortNcy / p.groupY);
            for (int k = 2; k <= 6 && nb1 * zSlabs < 192; k++) if (p.d.nz % k == 0 && p.d.nz / k >= 10) zSlabs = k;
        }
        const size_t lds = ldsFor(zSlabs);
        static const bool noBrick = getenv("SNB_NO_INTERP_BRICKS") != nullptr;   // testing aid: force the 32-lanes-per-atom kernel
        if (lds <= 150 * 1024 && !noBrick && p.nsubTotal * p.groupX * p.groupY <= 256) {
            const int nblocks = (p.sortNcx / p.groupX) * (p.sortNcy / p.groupY) * zSlabs;
            if (!p.mix) p.outForces = nullptr;      // (sharded engines visit an atom once per held grid: the separate finish pass stays)
            // 512-thread work-groups go two to a CU when the brick fits LDS twice (the kernel's ~100 VGPRs allow 16 waves per CU either way):
            // with more bricks than CUs the second round of 1024-thread groups runs half empty
            static const int ntEnv = getenv("SNB_INTERP_THREADS") ? atoi(getenv("SNB_INTERP_THREADS")) : 0;
            const bool narrow = ntEnv ? ntEnv == 512 : (nblocks > 256 && lds <= 76 * 1024);
            if (narrow) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(&k_interpolateBricks<Real, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                SNB_STAMPED_LAUNCH(stampSlot(p, 7), (k_interpolateBricks<Real, 512>), dim3(nblocks), dim3(512), lds, s, p, zSlabs);
            } else {
                hipFuncSetAttribute(reinterpret_cast<const void*>(&k_interpolateBricks<Real, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                SNB_STAMPED_LAUNCH(stampSlot(p, 7), (k_interpolateBricks<Real, 1024>), dim3(nblocks), dim3(1024), lds, s, p, zSlabs);
            }
            return p.outForces != nullptr;
        }
    }
    const size_t lds = (!p.mix && p.wantEnergy) ? sizeof(double) * p.nsubTotal * (p.nsubTotal + 1) : 0;
    SNB_STAMPED_LAUNCH(stampSlot(p, 7), (k_interpolate<Real>), dim3((p.natoms + 7) / 8), dim3(256), lds, s, p);
    return false;
}

template <typename Real> bool launchPmeInterpolate(const PmeParams<Real>& p, hipStream_t s) {
    if (p.natoms <= 0) return false;
    if (p.sortNcx > 0 && p.colRange != nullptr) {
        // The kernel needs ~100 VGPRs, so one 1024-thread work-group occupies a CU: with more bricks than CUs the launch runs in rounds.
        // Wider bricks (2 x 1, 2 x 2 columns) cut the count below the CU count and the halo overhead with it, as long as LDS allows.
        PmeParams<Real> q = p;
        static const int gEnv = getenv("SNB_INTERP_GROUP") ? atoi(getenv("SNB_INTERP_GROUP")) : -1;
        // (measured on c3, 400 single-column bricks: 1024 threads on 2 x 1-column bricks 35.2 us, 1024 threads on single columns 39.3,
        // 512 threads on single columns -- two work-groups per CU -- 29.1: when the single-column brick fits LDS twice, do not widen)
        const size_t single = sizeof(Real) * (size_t)(q.groupX * (q.d.nx / q.sortNcx) + 6) * (q.groupY * (q.d.ny / q.sortNcy) + 6) * (q.d.nz + 4) + 1024;
        for (int step = 0; step < 2; step++) {
            const int nb = (q.sortNcx / q.groupX) * (q.sortNcy / q.groupY);
            if (gEnv >= 0 ? step >= gEnv : (nb <= 256 || single <= 76 * 1024)) break;
            PmeParams<Real> t = q;
            if (step == 0 && t.sortNcx % (2 * t.groupX) == 0) t.groupX *= 2; else if (t.sortNcy % (2 * t.groupY) == 0) t.groupY *= 2; else break;
            const size_t need = sizeof(Real) * (size_t)(t.groupX * (t.d.nx / t.sortNcx) + 6) * (t.groupY * (t.d.ny / t.sortNcy) + 6) * (t.d.nz + 4) + 1024;
            if (need > 150 * 1024 || t.nsubTotal * t.groupX * t.groupY > 256) break;
            q = t;
        }
        return launchInterpolateBricks<Real>(q, s);
    }
    const size_t lds = (!p.mix && p.wantEnergy) ? sizeof(double) * p.nsubTotal * (p.nsubTotal + 1) : 0;
    SNB_STAMPED_LAUNCH(stampSlot(p, 7), (k_interpolate<Real>), dim3((p.natoms + 7) / 8), dim3(256), lds, s, p);
    return false;
}

template int launchPmeSpread<float>(const PmeParams<float>&, hipStream_t);
template int launchPmeSpread<double>(const PmeParams<double>&, hipStream_t);
template void launchPmeForwardFFT<float>(const PmeParams<float>&, hipStream_t, bool);
template void launchPmeForwardFFT<double>(const PmeParams<double>&, hipStream_t, bool);
template void launchPmeConvolution<float>(const PmeParams<float>&, hipStream_t);
template void launchPmeConvolution<double>(const PmeParams<double>&, hipStream_t);
template void launchPmeInverseFFT<float>(const PmeParams<float>&, hipStream_t);
template void launchPmeInverseFFT<double>(const PmeParams<double>&, hipStream_t);
template bool launchPmeInterpolate<float>(const PmeParams<float>&, hipStream_t);
template bool launchPmeInterpolate<double>(const PmeParams<double>&, hipStream_t);

// ---- host helpers ---------------------------------------------------------------------------------
bool factorize(int n, int* factors, int* nf) {
    int k = 0;
    while (n % 4 == 0) { factors[k++] = 4; n /= 4; }
    while (n % 2 == 0) { factors[k++] = 2; n /= 2; }
    while (n % 3 == 0) { factors[k++] = 3; n /= 3; }
    while (n % 5 == 0) { factors[k++] = 5; n /= 5; }
    while (n % 7 == 0) { factors[k++] = 7; n /= 7; }
    while (n % 11 == 0) { factors[k++] = 11; n /= 11; }      // 11 and 13: the reference's GPU FFT is legal up to 13-smooth sizes
    while (n % 13 == 0) { factors[k++] = 13; n /= 13; }      // (platforms/common/include/FFT3DFactory.h:45-47)
    *nf = k;
    return n == 1 && k <= 16;
}

// n = r1 * r2 for one of the instantiated (R1, R2) pairs; false (and 0, 0) if n is not in the list
// plane path: n = r1 * r2 with both factors among the radices k_planeXY has pass bodies for, r1 <= r2, the most balanced pair; the
// pairs with a kernel of their own (SNB_PLANE_PAIRS) first
bool splitPlane(int n, int* r1, int* r2) {
#define X(A, B) if (n == (A) * (B)) { *r1 = A; *r2 = B; return true; }
    SNB_PLANE_PAIRS(X)
#undef X
    const int radices[] = {
#define X(A) A,
        SNB_PLANE_RADICES(X)
#undef X
    };
    int best = 0;
    for (int a : radices) for (int b : radices) if (a <= b && a * b == n && a > best) { best = a; *r1 = a; *r2 = b; }
    if (best) return true;
    *r1 = *r2 = 0;
    return false;
}

bool splitTwoPass(int n, int* r1, int* r2) {
#define X(A, B) if (n == (A) * (B)) { *r1 = A; *r2 = B; return true; }
    SNB_FFT_PAIRS(X)
#undef X
    *r1 = *r2 = 0;
    return false;
}

int legalGridSize(int n) {
    if (n < 6) n = 6;
    for (;; n++) {
        int f[32], nf, m = n;
        for (int p : {2, 3, 5, 7, 11, 13}) while (m % p == 0) m /= p;
        if (m == 1 && factorize(n, f, &nf)) return n;
    }
}

}  // namespace snb
