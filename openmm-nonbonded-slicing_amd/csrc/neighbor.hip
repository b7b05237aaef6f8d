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
#include <algorithm>
#include <string.h>
#include <rocprim/rocprim.hpp>

namespace snb {

// Periodic cell in OpenMM's reduced form: a = (ax, 0, 0), b = (bx, by, 0), c = (cx, cy, cz), |bx| <= ax/2, |cx| <= ax/2, |cy| <= by/2.
struct Lattice { float ax, bx, by, cx, cy, cz; };
template <typename P> __device__ inline bool getenvBoxWalk(const P& p) { return p.boxWalk != 0; }      // SNB_NB_BOX_WALK=1: the round-2 candidate walk (test switch)
template <typename Real> __device__ inline Lattice latticeOf(const NbParams<Real>& p) {
    return Lattice{(float)p.boxm[0], (float)p.boxm[3], (float)p.boxm[4], (float)p.boxm[6], (float)p.boxm[7], (float)p.boxm[8]};
}
// nearest periodic image of a displacement (ReferenceForce::getDeltaRPeriodic's triclinic order: c, then b, then a)
__device__ inline void minImage(float& dx, float& dy, float& dz, const Lattice& L) {
    float s = rintf(dz / L.cz); dx -= s * L.cx; dy -= s * L.cy; dz -= s * L.cz;
    s = rintf(dy / L.by); dx -= s * L.bx; dy -= s * L.by;
    s = rintf(dx / L.ax); dx -= s * L.ax;
}
// ---- 0. bounding box of the user positions (non-periodic methods: the builder runs in an enclosing cell with no image in reach) ------
__device__ inline int orderedInt(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }
template <typename In> __global__ void k_nbExtent(const In* __restrict__ userPos, int stride, int n, int* __restrict__ ext) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    if (u < n) for (int d = 0; d < 3; d++) { const float v = (float)userPos[(size_t)u * stride + d]; lo[d] = v; hi[d] = v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int d = 0; d < 3; d++) { lo[d] = fminf(lo[d], __shfl_xor(lo[d], o, 64)); hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o, 64)); }
    if ((threadIdx.x & 63) == 0) for (int d = 0; d < 3; d++) { atomicMin(&ext[d], orderedInt(lo[d])); atomicMax(&ext[3 + d], orderedInt(hi[d])); }
}
void launchExtent(const void* userPos, int isDouble, int stride4, int n, int* ext, hipStream_t s) {
    const int init[6] = {0x7F7FFFFF, 0x7F7FFFFF, 0x7F7FFFFF, (int)0x80800000, (int)0x80800000, (int)0x80800000};      // ordered +max / -max
    (void)hipMemcpyAsync(ext, init, sizeof(init), hipMemcpyHostToDevice, s);
    const int stride = stride4 ? 4 : 3;
    if (isDouble) hipLaunchKernelGGL((k_nbExtent<double>), dim3((n + 255) / 256), dim3(256), 0, s, (const double*)userPos, stride, n, ext);
    else hipLaunchKernelGGL((k_nbExtent<float>), dim3((n + 255) / 256), dim3(256), 0, s, (const float*)userPos, stride, n, ext);
}

// ---- 1. sort keys -------------------------------------------------------------------------------------------------
template <typename Real, typename In>
__global__ void k_nbKeys(const NbParams<Real> p, const In* __restrict__ userPos, int stride) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= p.nAtoms) return;
    const double x0 = (double)userPos[(size_t)u * stride], y0 = (double)userPos[(size_t)u * stride + 1], z0 = (double)userPos[(size_t)u * stride + 2];
    const double x = x0 - p.origin[0], y = y0 - p.origin[1], z = z0 - p.origin[2];      // (origin != 0 only for the enclosing cell of a non-periodic system)
    // wrap into the primary cell IN FRACTIONAL COORDINATES (the parallelepiped 0 <= f < 1, the same fundamental domain as the PME
    // mesh); columns and z keys are fractional too, so a triclinic cell sorts exactly like a rectangular one
    const double* m = p.boxm;
    double fz = z / m[8], fy = (y - fz * m[7]) / m[4], fx = (x - fy * m[3] - fz * m[6]) / m[0];
    fx -= floor(fx); fy -= floor(fy); fz -= floor(fz);
    if (fx >= 1.0) fx = 0.0; if (fy >= 1.0) fy = 0.0; if (fz >= 1.0) fz = 0.0;      // -1e-17 - floor(-1e-17) == 1.0
    const double wx = fx * m[0] + fy * m[3] + fz * m[6], wy = fy * m[4] + fz * m[7], wz = fz * m[8];
    p.wrapped[3 * (size_t)u] = (Real)wx; p.wrapped[3 * (size_t)u + 1] = (Real)wy; p.wrapped[3 * (size_t)u + 2] = (Real)wz;
    p.offsetU[3 * (size_t)u] = (Real)(wx - x0); p.offsetU[3 * (size_t)u + 1] = (Real)(wy - y0); p.offsetU[3 * (size_t)u + 2] = (Real)(wz - z0);
    int cx = (int)(fx * p.ncx); cx = cx < 0 ? 0 : (cx >= p.ncx ? p.ncx - 1 : cx);
    int cy = (int)(fy * p.ncy); cy = cy < 0 ? 0 : (cy >= p.ncy ? p.ncy - 1 : cy);
    const int serp = cx * p.ncy + ((cx & 1) ? (p.ncy - 1 - cy) : cy);
    double zf = fz; zf = zf < 0 ? 0 : (zf > 1 ? 1 : zf);
    if (serp & 1) zf = 1.0 - zf;
    const unsigned long long zq = (unsigned long long)(zf * 1048575.0);
    p.keysIn[u] = ((unsigned long long)p.uSubset[u] << (20 + p.colBits)) | ((unsigned long long)serp << 20) | zq;      // only the bits in use are sorted
    p.valsIn[u] = u;
}

// ---- 1b. block segmentation ---------------------------------------------------------------------------------------
// Blocks are 32 consecutive atoms of the sorted order, but never across a subset boundary and never across a JUMP: two consecutive
// atoms further apart (minimum image) than `jumpDist` -- the hollow of a shell-shaped subset, scattered ions, a solute straddling the
// periodic boundary.  Every such segment is padded to a multiple of 32, so a block's bounding box stays small however sparse its
// subset is.  segKey[t] = t at segment starts (0 elsewhere) -> inclusive max scan = start of t's segment -> padding owed at segment
// ends -> exclusive sum scan = padding slots before t; padded index of t = t + padBefore[t].
template <typename Real> __global__ void k_nbJumpFlags(const NbParams<Real> p) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.nAtoms) return;
    bool start = t == 0;
    if (!start) {
        start = (p.keysOut[t] >> (20 + p.colBits)) != (p.keysOut[t - 1] >> (20 + p.colBits));
        if (!start) {
            const int u = p.valsOut[t], v = p.valsOut[t - 1];
            float dx = (float)p.wrapped[3 * (size_t)u] - (float)p.wrapped[3 * (size_t)v], dy = (float)p.wrapped[3 * (size_t)u + 1] - (float)p.wrapped[3 * (size_t)v + 1],
                  dz = (float)p.wrapped[3 * (size_t)u + 2] - (float)p.wrapped[3 * (size_t)v + 2];
            minImage(dx, dy, dz, latticeOf(p));
            start = dx * dx + dy * dy + dz * dz > p.jumpDist * p.jumpDist;
        }
        // second pass: every atom of a block found too wide by k_nbWideDetect becomes a segment (= block) of its own
        if (!start && p.blockWide) start = p.blockWide[(t + p.padBefore[t]) >> 5] || p.blockWide[(t - 1 + p.padBefore[t - 1]) >> 5];
    }
    p.segKey[t] = start ? t : 0;
}
// A chain of moderately spaced atoms (scattered ions, say) passes the jump test and can still make a 32-atom block too extended for
// the one-image-per-j-atom tile scheme (extent + 2R < L).  Blocks with an atom further than maxHalfExtent from their first atom
// (minimum image, any axis) are flagged; the second segmentation pass dissolves them into single-atom blocks.
template <typename Real> __global__ void k_nbWideDetect(const NbParams<Real> p) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.nAtoms) return;
    const int si = t + p.padBefore[t];
    const int u = p.valsOut[t], u0 = p.valsOut[t - (si & 31)];
    float dx = (float)p.wrapped[3 * (size_t)u] - (float)p.wrapped[3 * (size_t)u0], dy = (float)p.wrapped[3 * (size_t)u + 1] - (float)p.wrapped[3 * (size_t)u0 + 1],
          dz = (float)p.wrapped[3 * (size_t)u + 2] - (float)p.wrapped[3 * (size_t)u0 + 2];
    minImage(dx, dy, dz, latticeOf(p));
    if (fabsf(dx) > p.maxHalfExtent[0] || fabsf(dy) > p.maxHalfExtent[1] || fabsf(dz) > p.maxHalfExtent[2]) p.blockWideOut[si >> 5] = 1;
}
template <typename Real> __global__ void k_nbPadExtra(const NbParams<Real> p) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.nAtoms) return;
    const bool isEnd = (t == p.nAtoms - 1) || (p.segKey[t + 1] == t + 1);
    const int len = t - p.segStart[t] + 1;
    p.padExtra[t] = isEnd ? ((32 - (len & 31)) & 31) : 0;
}
template <typename Real> __global__ void k_nbPadTotal(const NbParams<Real> p) {
    if (blockIdx.x == 0 && threadIdx.x == 0) p.counters[7] = p.nAtoms + p.padBefore[p.nAtoms - 1] + p.padExtra[p.nAtoms - 1];
}

// ---- 2. scatter into the padded sorted order ---------------------------------------------------------------------
template <typename Real> __global__ void k_nbScatter(const NbParams<Real> p) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.nAtoms) return;
    const unsigned long long key = p.keysOut[t];
    const int u = p.valsOut[t];
    const int s = (int)(key >> (20 + p.colBits));
    const int serp = (int)((key >> 20) & ((1u << p.colBits) - 1u));
    const int si0 = t + p.padBefore[t];           // padded index in sorted order
    const int si = si0;
    // (round 4: the engine sizes the padded arrays from the PREVIOUS rebuild's count plus a margin instead of waiting for this one's;
    // an atom that would land beyond them raises the overflow count, the engine then repeats the rebuild with the exact count)
    if (si0 >= p.nPadded) { atomicAdd(&p.counters[3], 1); return; }
    if ((si0 & 31) == 0) p.blockSubset[si0 >> 5] = s;
    p.sortedToUser[si] = u; p.userToSorted[u] = si;
    typename Vec<Real>::T4 v; v.x = p.wrapped[3 * (size_t)u]; v.y = p.wrapped[3 * (size_t)u + 1]; v.z = p.wrapped[3 * (size_t)u + 2]; v.w = p.uCharge[u];
    // A 32-atom block of a sparse subset can straddle the periodic boundary (consecutive occupied columns 1 and ncy-2, say): its atoms
    // are stored in the image nearest to the block's first atom, so that the block's bounding box stays compact.  Stored positions
    // may therefore lie up to one box length outside [0, L); imageOffset absorbs the shift and tile image codes reach +-2.
    Real sh[3] = {0, 0, 0};
    {
        const int u0 = p.valsOut[t - (si0 & 31)];          // the block's first atom: blocks never cross segments, so ranks t-(si0&31)..t are its atoms
        const float d0x = (float)v.x - (float)p.wrapped[3 * (size_t)u0], d0y = (float)v.y - (float)p.wrapped[3 * (size_t)u0 + 1], d0z = (float)v.z - (float)p.wrapped[3 * (size_t)u0 + 2];
        float dx = d0x, dy = d0y, dz = d0z;
        const Lattice L = latticeOf(p);
        minImage(dx, dy, dz, L);
        // the difference is an exact lattice vector: rebuild it from integer coefficients so that no rounding noise enters the coordinates
        const int kz = (int)rintf((dz - d0z) / L.cz);
        const int ky = (int)rintf(((dy - d0y) - kz * L.cy) / L.by);
        const int kx = (int)rintf(((dx - d0x) - ky * L.bx - kz * L.cx) / L.ax);
        sh[0] = (Real)(kx * p.boxm[0] + ky * p.boxm[3] + kz * p.boxm[6]); sh[1] = (Real)(ky * p.boxm[4] + kz * p.boxm[7]); sh[2] = (Real)(kz * p.boxm[8]);
        p.atomCell[si] = (kx + 1) | ((ky + 1) << 2) | ((kz + 1) << 4);      // lattice cell of the stored position (exact: never re-derived from rounded coordinates)
    }
    v.x += sh[0]; v.y += sh[1]; v.z += sh[2];
    p.posq[si] = v;
    p.sigeps[si] = p.uSigEps[u];
    p.imageOffset[3 * (size_t)si] = p.offsetU[3 * (size_t)u] + sh[0]; p.imageOffset[3 * (size_t)si + 1] = p.offsetU[3 * (size_t)u + 1] + sh[1]; p.imageOffset[3 * (size_t)si + 2] = p.offsetU[3 * (size_t)u + 2] + sh[2];
    p.atomSubset[si] = s; p.atomGrid[si] = p.slotOfSubset[s];
    // column bookkeeping: linear (non-serpentine) column id, run boundaries of (subset, column)
    const int cx = serp / p.ncy, cyS = serp - cx * p.ncy;
    const int cy = (cx & 1) ? (p.ncy - 1 - cyS) : cyS;
    const int col = cx * p.ncy + cy;
    // z-bucket index of the (subset, column) run: 64 buckets of the key's 20-bit z (already direction-flipped for odd columns, so every
    // run ascends in it).  zIndex[b] = first padded index (IN SORTED ORDER, si0) of the run with bucket >= b (k_nbZPrefix fills the empty
    // buckets); positions, not counts, because segment padding may sit inside a run.
    atomicMin(&p.zIndex[((size_t)s * p.ncx * p.ncy + col) * 65 + (int)((key & 0xFFFFF) >> 14)], si0);
    // the run's interval of the padded order
    // sorted order: the run's first and last atom
    if ((t == 0) || ((p.keysOut[t - 1] >> 20) != (key >> 20))) p.colRange[(size_t)s * p.ncx * p.ncy + col].x = si;
    if ((t == p.nAtoms - 1) || ((p.keysOut[t + 1] >> 20) != (key >> 20))) p.colRange[(size_t)s * p.ncx * p.ncy + col].y = si + 1;
}

template <typename Real> __global__ void k_nbZPrefix(const NbParams<Real> p) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.nSubsets * p.ncx * p.ncy) return;
    int* h = p.zIndex + (size_t)c * 65;
    int2 rg = p.colRange[c];
    if (rg.y <= rg.x) { rg = make_int2(0, 0); p.colRange[c] = rg; }      // empty run: (0, 0), never looked up (k_nbClear left (INT_MAX, 0))
    int nxt = rg.y;
    h[64] = nxt;
    for (int b = 63; b >= 0; b--) { const int v = h[b]; nxt = v < nxt ? v : nxt; h[b] = nxt; }
}

// padding slots: static far-away coordinates with zero parameters (they are also masked out of every tile)
template <typename Real> __global__ void k_nbPad(const NbParams<Real> p) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= p.nPadded) return;
    if (p.sortedToUser[s] < 0) {
        typename Vec<Real>::T4 v; v.x = (Real)(1e9 + 1e6 * (s & 4095)); v.y = (Real)2e9; v.z = (Real)-3e9; v.w = 0;
        p.posq[s] = v;
        typename Vec<Real>::T2 z; z.x = 0; z.y = 0; p.sigeps[s] = z;
        p.atomSubset[s] = -1; p.atomGrid[s] = -1; p.atomCell[s] = 1 | (1 << 2) | (1 << 4);
        p.imageOffset[3 * (size_t)s] = 0; p.imageOffset[3 * (size_t)s + 1] = 0; p.imageOffset[3 * (size_t)s + 2] = 0;
        if ((s & 31) == 0) p.blockSubset[s >> 5] = 0;      // a block that STARTS with a padding slot is all padding (the spare blocks behind the last segment): any valid subset
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
            if ((mx[d] - mn[d]) + 2.f * p.listCutoff >= (float)p.boxm[4 * d]) tooWide = true;   // one image per j-atom needs extent + 2R < cell height (ax, by, cz)
        }
        if (tooWide) atomicAdd(&p.counters[3], 1);
    }
}

// ---- 4. tiles ------------------------------------------------------------------------------------------------------
__device__ inline bool ownsPair(int I, int J) { return ((I + J) & 1) ? (I > J) : (I < J); }
__device__ inline int nbSliceOf(int a, int b) { return a > b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a; }
__device__ inline int lanePrefix(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0)); }

constexpr int NB_PARTS = 64;         // allocation counters are partitioned (block I -> partition I % 64, one 128-byte line each): 9 k wavefronts bumping the
                                     // same three counters were serialised in L2 and cost ~250 us of a 600 us build
constexpr int NB_CAP = 768;         // j entries gathered per published chunk (24 tiles); larger neighbourhoods publish several chunks.  768: four
                                    // work-groups per CU instead of three with 1024 (LDS), builder 0.52 -> 0.43 ms on c3; 640 adds 2 % tiles
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

template <typename Real> __global__ __launch_bounds__(256, 4) void k_nbBuildTiles(const NbParams<Real> p) {
    __shared__ int s_list[4][NB_CAP];
    __shared__ unsigned s_mask[4][NB_MAXT][32];
    __shared__ int s_tileSub[4][NB_MAXT];
    __shared__ int s_query[4][128];      // exclusion partners (sorted index) still to be located in the gathered list
    __shared__ int s_qrow[4][128];       // ... and the i-row each belongs to
    __shared__ int s_imgI[4][27][5]; __shared__ float s_imgF[4][27][2];      // surviving lattice images of the block (see below)
    __shared__ int s_cmb[4][3][64];      // candidate runs of the current batch of (column, z image) combinations: start, exclusive prefix, image code
    __shared__ float4 s_ipos[4][16]; __shared__ float2 s_iposZ[4][16];     // the block's own atoms, two per entry: (x0, x1, y0, y1) and (z0, z1) (exact pruning of the gathered list)
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // sharded engines build only the i-blocks they own, I % shardPeriod in [shardBegin, shardBegin + shardWidth) (a deterministic
    // rule, the same on every rank); the grid is sized for the owned blocks so that every wave of a work-group has work
    // (in DESCENDING order of the owned blocks: the sorted order ends with the small subsets, whose blocks see several j-subsets and take
    // up to twice the median time -- started last they were the tail of the launch, SNB_NB_TRACE: span 451 us with the last start at 285)
    const int kOwned = p.nOwned - 1 - (blockIdx.x * 4 + wid);
    if (kOwned < 0) return;
    const int I = (kOwned / p.shardWidth) * p.shardPeriod + p.shardBegin + kOwned % p.shardWidth;
    if (I >= p.nBlocks) return;
    const long long tStart = p.dbgOut ? (long long)wall_clock64() : 0;
    int* list = s_list[wid];
    unsigned (*mask)[32] = s_mask[wid];
    int* tileSub = s_tileSub[wid];
    int* query = s_query[wid];
    int* qrow = s_qrow[wid];
    int* cmbStart = s_cmb[wid][0]; int* cmbPrefix = s_cmb[wid][1]; int* cmbCode = s_cmb[wid][2];
    const float R = p.listCutoff, R2 = R * R;
    const float cxx = p.blockCenter[3 * I], cyy = p.blockCenter[3 * I + 1], czz = p.blockCenter[3 * I + 2];
    const float hx = p.blockHalf[3 * I], hy = p.blockHalf[3 * I + 1], hz = p.blockHalf[3 * I + 2];
    const int il = lane & 31, half = lane >> 5;
    const int uI = p.sortedToUser[I * 32 + il];
    if (__shfl(uI, 0, 64) < 0) return;      // a block that starts with a padding slot is all padding (the spare blocks behind the last segment): no tiles
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
        // every lane keeps its share of the gathered list in registers (entry k = lane + 64 r) and compares it with each queued partner:
        // one LDS broadcast per query instead of a pass over the LDS list per query
        int mine[NB_CAP / 64];
#pragma unroll
        for (int r = 0; r < NB_CAP / 64; r++) { const int k2 = lane + 64 * r; const int e = (k2 < count && !(hasDiag && k2 < 32)) ? list[k2] : -1; mine[r] = (e == -1) ? -1 : (e & SNB_JIDX_MASK); }
        auto resolve = [&](int nq) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int qi = 0; qi < nq; qi++) {
                const int target = query[qi], row = qrow[qi];
#pragma unroll
                for (int r = 0; r < NB_CAP / 64; r++)
                    if (mine[r] == target) { const int k2 = lane + 64 * r; atomicOr(&mask[k2 >> 5][row], 1u << (k2 & 31)); }
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
        const int CH = p.itemTiles;                       // tiles per full work item
        const int nFull = nT / CH, nPart = (nT % CH) ? 1 : 0;
        const int part = I & (NB_PARTS - 1);
        int* cnt = p.counters + 32 * (1 + part);
        const int tileRegion = p.tileCapacity / NB_PARTS, workRegion = p.workCapacity / NB_PARTS;
        if (lane == 0) { first = atomicAdd(&cnt[0], nT); w0 = atomicAdd(&cnt[1], nFull); wp = nPart ? atomicAdd(&cnt[4], 1) : 0; }
        first = __builtin_amdgcn_readfirstlane(first); w0 = __builtin_amdgcn_readfirstlane(w0); wp = __builtin_amdgcn_readfirstlane(wp);
        if (first + nT > tileRegion || w0 + nFull > workRegion || wp + nPart > workRegion) { failed = true; return; }   // the host grows the regions and retries
        first += part * tileRegion; w0 += part * workRegion; wp += part * workRegion;
        for (int k = lane; k < count; k += 64) p.tileJ[(size_t)first * 32 + k] = list[k];
        // mask words: lanes 0..31 handle the rows of even tiles, lanes 32..63 of odd tiles; one allocation for all masked tiles of the chunk
        unsigned long long anyBits = 0ull;      // bit t = tile t has a non-zero mask
        for (int t2 = 0; t2 < nT; t2 += 2) {
            const int t = t2 + half;
            const unsigned row = (t < nT) ? mask[t][il] : 0u;
            const unsigned long long bal = __ballot(row != 0u);
            if ((unsigned)bal) anyBits |= 1ull << t2;
            if ((unsigned)(bal >> 32)) anyBits |= 1ull << (t2 + 1);
        }
        const int nMasked = __popcll(anyBits);
        int mi0 = 0;
        if (nMasked > 0) {
            if (lane == 0) mi0 = atomicAdd(&cnt[2], nMasked);
            mi0 = __builtin_amdgcn_readfirstlane(mi0);
            if (mi0 + nMasked > p.maskCapacity / NB_PARTS) { failed = true; return; }
            mi0 += part * (p.maskCapacity / NB_PARTS);
        }
        for (int t2 = 0; t2 < nT; t2 += 2) {
            const int t = t2 + half;
            if (t < nT && ((anyBits >> t) & 1ull)) p.masks[(size_t)(mi0 + __popcll(anyBits & ((1ull << t) - 1ull))) * 32 + il] = mask[t][il];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int subIb = p.blockSubset[I];
        for (int t = lane; t < nT; t += 64) p.tileInfo[first + t] = make_int4(nbSliceOf(subIb, tileSub[t]), ((anyBits >> t) & 1ull) ? mi0 + __popcll(anyBits & ((1ull << t) - 1ull)) : -1, tileSub[t], 0);
        const int subI = p.blockSubset[I];      // carried in the work item: the pair kernel needs it before the block's atoms arrive
        for (int k = lane; k < nFull; k += 64) p.workItemsStage[w0 + k] = make_int4(I, first + CH * k, CH, subI);
        if (nPart && lane == 0) p.workItemsPartial[wp] = make_int4(I, first + CH * nFull, nT % CH, subI);
        __builtin_amdgcn_wave_barrier();
    };

    const long long tProlog = p.dbgOut ? (long long)wall_clock64() : 0;
    // diagonal tile
    if (lane < 32) { list[lane] = (uI >= 0) ? ((I * 32 + lane) | (SNB_JCODE_CENTER << SNB_JSHIFT_BITS)) : -1; }
    if (lane == 0) tileSub[0] = p.blockSubset[I];
    int count = 32;
    bool hasDiag = true;

    // the block's box grown by the list radius: a j-atom is wanted if one of its periodic images lies inside
    const Lattice Lt = latticeOf(p);
    const float X0 = cxx - hx - R, X1 = cxx + hx + R, Y0 = cyy - hy - R, Y1 = cyy + hy + R, Z0 = czz - hz - R, Z1 = czz + hz + R;
    const float feps = 1e-5f;                             // fractional slack for float-vs-double rounding of column / bucket borders
    const bool rectCell = Lt.bx == 0.f && Lt.cx == 0.f && Lt.cy == 0.f && !getenvBoxWalk(p);
    // The lattice images that can reach the grown box, with their column rectangles and fractional z ranges: geometry only, so it is
    // worked out once per block and reused for every j-subset (typically 1-4 of the 27 images survive).
    int (*imgI)[5] = s_imgI[wid]; float (*imgF)[2] = s_imgF[wid];
    int nImg = 0;
    for (int img = 0; img < 27; img++) {
        const int kx = img % 3 - 1, ky = (img / 3) % 3 - 1, kz = img / 9 - 1;
        // where the WRAPPED atoms of this image must lie: the grown box moved back by kx a + ky b + kz c, in fractional coordinates
        const float tz = kz * Lt.cz, ty = ky * Lt.by + kz * Lt.cy, tx = kx * Lt.ax + ky * Lt.bx + kz * Lt.cx;
        float fz0 = (Z0 - tz) / Lt.cz - feps, fz1 = (Z1 - tz) / Lt.cz + feps;
        if (fz1 < 0.f || fz0 >= 1.f) continue;
        fz0 = fz0 < 0.f ? 0.f : fz0; fz1 = fz1 > 1.f ? 1.f : fz1;
        const float sy0 = fz0 * Lt.cy, sy1 = fz1 * Lt.cy;                       // shear of y with z
        float fy0 = (Y0 - ty - (sy0 > sy1 ? sy0 : sy1)) / Lt.by - feps, fy1 = (Y1 - ty - (sy0 < sy1 ? sy0 : sy1)) / Lt.by + feps;
        if (fy1 < 0.f || fy0 >= 1.f) continue;
        fy0 = fy0 < 0.f ? 0.f : fy0; fy1 = fy1 > 1.f ? 1.f : fy1;
        const float sx0 = fz0 * Lt.cx, sx1 = fz1 * Lt.cx, ux0 = fy0 * Lt.bx, ux1 = fy1 * Lt.bx;     // shear of x with z and y
        float fx0 = (X0 - tx - (sx0 > sx1 ? sx0 : sx1) - (ux0 > ux1 ? ux0 : ux1)) / Lt.ax - feps, fx1 = (X1 - tx - (sx0 < sx1 ? sx0 : sx1) - (ux0 < ux1 ? ux0 : ux1)) / Lt.ax + feps;
        if (fx1 < 0.f || fx0 >= 1.f) continue;
        fx0 = fx0 < 0.f ? 0.f : fx0; fx1 = fx1 > 1.f ? 1.f : fx1;
        const int gx0 = (int)(fx0 * p.ncx), gy0 = (int)(fy0 * p.ncy);
        int gx1 = (int)(fx1 * p.ncx), gy1 = (int)(fy1 * p.ncy);
        gx1 = gx1 > p.ncx - 1 ? p.ncx - 1 : gx1; gy1 = gy1 > p.ncy - 1 ? p.ncy - 1 : gy1;
        const int nyc = gy1 - gy0 + 1, ncols = (gx1 - gx0 + 1) * nyc;
        const int code = (kz + 1) * 9 + (ky + 1) * 3 + (kx + 1);
        if (lane == 0) { imgI[nImg][0] = gx0; imgI[nImg][1] = gy0; imgI[nImg][2] = nyc; imgI[nImg][3] = ncols; imgI[nImg][4] = code; imgF[nImg][0] = fz0; imgF[nImg][1] = fz1; }
        nImg++;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // Exact pruning.  The box test admits every atom within R of the block's bounding box; an atom only belongs in the list if it is
    // within R of one of the block's ATOMS, which removes about a sixth of them (and with them a sixth of the pair kernel's tiles).
    // The entries gathered since the last call are re-tested densely (64 survivors of the box test per pass) and compacted in place.
    float4* ipos = s_ipos[wid]; float2* iposZ = s_iposZ[wid];
    const bool exact = p.exactPrune != 0;
    if (lane < 16) {
        const auto q0 = p.posq[I * 32 + 2 * lane], q1 = p.posq[I * 32 + 2 * lane + 1];
        ipos[lane] = make_float4((float)q0.x, (float)q1.x, (float)q0.y, (float)q1.y); iposZ[lane] = make_float2((float)q0.z, (float)q1.z);
    }
    int filtered = 32;                                    // list[0 .. filtered) has been through the exact test (the diagonal tile needs none)
    auto exactFilter = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int out = filtered;
        for (int b0 = filtered; b0 < count; b0 += 64) {
            const int k = b0 + lane;
            bool keep = k < count;
            const int e = keep ? list[k] : 0;
            float px = 0.f, py = 0.f, pz = 0.f;
            if (keep) {
                const int sc = (e >> SNB_JSHIFT_BITS) & 127;
                const int kx = sc / 25 - 2, ky = (sc % 25) / 5 - 2, kz = sc % 5 - 2;
                const auto q = p.posq[e & SNB_JIDX_MASK];
                px = (float)q.x + kx * Lt.ax + ky * Lt.bx + kz * Lt.cx; py = (float)q.y + ky * Lt.by + kz * Lt.cy; pz = (float)q.z + kz * Lt.cz;
            }
            typedef float v2f __attribute__((ext_vector_type(2)));
            float best = 3e38f;
#pragma unroll 8
            for (int a = 0; a < 16; a++) {                // two i-atoms per pass (packed fp32), LDS broadcast reads
                const float4 xy = ipos[a]; const float2 zz = iposZ[a];
                const v2f ddx = v2f{xy.x, xy.y} - px, ddy = v2f{xy.z, xy.w} - py, ddz = v2f{zz.x, zz.y} - pz;
                const v2f d2 = ddx * ddx + ddy * ddy + ddz * ddz;
                best = fminf(best, fminf(d2.x, d2.y));
            }
            keep = keep && best < R2;
            __builtin_amdgcn_wave_barrier();             // every lane has read its entry before the slots below it are rewritten
            const unsigned long long m = __ballot(keep);
            if (keep) { const int o = out + lanePrefix(m); list[o] = e; }
            out += __popcll(m);
        }
        count = out; filtered = out;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    for (int s = 0; s < p.nSubsets && !failed; s++) {
        int segStart = count;
        const int2* ranges = p.colRange + (size_t)s * p.ncx * p.ncy;
        const int* zIndex = p.zIndex + (size_t)s * p.ncx * p.ncy * 65;
        int nCmb = 0;                                     // candidate runs queued in cmbStart / cmbPrefix (lengths) / cmbCode

        // Walks the queued runs, 64 candidates at a time (full lanes, independent loads): exact box test of every candidate in the
        // image its run was queued for, ballot compaction into the tile list.
        auto runCandidates = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int len = lane < nCmb ? cmbPrefix[lane] : 0;
            int incl = len;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
            const int total = __shfl(incl, 63, 64);
            __builtin_amdgcn_wave_barrier();
            cmbPrefix[lane] = incl - len;                 // exclusive prefix; lanes >= nCmb hold `total`
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int v0 = 0; v0 < total && !failed; v0 += 64) {
                const int v = v0 + lane;
                bool ok = v < total;
                int j = 0, code = SNB_JCODE_CENTER;
                if (ok) {
                    int k = 0;     // last run whose exclusive prefix is <= v (empty runs are never queued)
#pragma unroll
                    for (int st = 32; st > 0; st >>= 1) if (k + st < 64 && cmbPrefix[k + st] <= v) k += st;
                    // the run is stored in the COMPACT index space of the blocks this i-block owns (every other block on either side of it,
                    // ownsPair): position g of that space is atom (g & 31) of the (g >> 5)-th block of parity pi
                    const int g = cmbStart[k] + (v - cmbPrefix[k]), cc = cmbCode[k];
                    code = cc & 255;
                    j = ((((g >> 5) << 1) | (cc >> 8)) << 5) | (g & 31);
                }
                if (ok) {
                    // `code` names the lattice image (kx, ky, kz in -1..1) of the WRAPPED j position; the stored position may already sit in
                    // a neighbouring cell (compact blocks), so the shift applied to it -- and recorded in the tile -- is the difference
                    int kx = code % 3 - 1, ky = (code / 3) % 3 - 1, kz = code / 9 - 1;
                    const auto q = p.posq[j];
                    const int sc = p.atomCell[j];
                    kx -= (sc & 3) - 1; ky -= ((sc >> 2) & 3) - 1; kz -= ((sc >> 4) & 3) - 1;
                    const float px = (float)q.x + kx * Lt.ax + ky * Lt.bx + kz * Lt.cx, py = (float)q.y + ky * Lt.by + kz * Lt.cy, pz = (float)q.z + kz * Lt.cz;
                    float dx = fabsf(px - cxx) - hx, dy = fabsf(py - cyy) - hy, dz = fabsf(pz - czz) - hz;
                    dx = dx > 0 ? dx : 0; dy = dy > 0 ? dy : 0; dz = dz > 0 ? dz : 0;
                    ok = (dx * dx + dy * dy + dz * dz < R2) && ((float)q.y < 1e8f) && kx >= -2 && kx <= 2 && ky >= -2 && ky <= 2 && kz >= -2 && kz <= 2;   // (padding slots are parked at y = 2e9)
                    code = (kx + 2) * 25 + (ky + 2) * 5 + (kz + 2);
                }
                const unsigned long long m = __ballot(ok);
                const int nNew = __popcll(m);
                if (count + nNew > NB_CAP - 32 && exact) exactFilter();      // make room first
                if (count + nNew > NB_CAP - 32) {
                    // list full: close the current segment, publish this chunk and start a fresh list
                    const int padded = (count + 31) & ~31;
                    for (int k = count + lane; k < padded; k += 64) list[k] = -1;
                    for (int t = (segStart >> 5) + lane; t < (padded >> 5); t += 64) tileSub[t] = s;
                    flush(padded, hasDiag);
                    hasDiag = false; count = 0; segStart = 0; filtered = 0;
                }
                if (ok) { const int o = count + lanePrefix(m); list[o] = j | (code << SNB_JSHIFT_BITS); }
                count += nNew;
            }
            __builtin_amdgcn_wave_barrier();
            nCmb = 0;
        };

        for (int im = 0; im < nImg && !failed; im++) {
            const int gx0 = imgI[im][0], gy0 = imgI[im][1], nyc = imgI[im][2], ncols = imgI[im][3], code = imgI[im][4];
            const float fz0 = imgF[im][0], fz1 = imgF[im][1];
            for (int c0 = 0; c0 < ncols && !failed; c0 += 64) {
                const int c = c0 + lane;
                int cStart = 0, cLen = 0;
                if (c < ncols) {
                    const int ccx = gx0 + c / nyc, ccy = gy0 + c % nyc;
                    const int col = ccx * p.ncy + ccy;
                    const int2 rg = ranges[col];
                    if (rg.y > rg.x) {
                        const int serp = ccx * p.ncy + ((ccx & 1) ? (p.ncy - 1 - ccy) : ccy);
                        float fa = fz0, fb = fz1;
                        if (rectCell) {
                            // (round 3) the grown BOX reaches a column rectangle at xy distance d only over |z - box| < sqrt(R^2 - d^2): the columns at
                            // the corners of the rectangle range get a short z interval or none (the box walk tested a third more candidates)
                            const int kx = code % 3 - 1, ky = (code / 3) % 3 - 1, kz = code / 9 - 1;
                            const float wx = Lt.ax / p.ncx, wy = Lt.by / p.ncy, slack = 1e-5f * (Lt.ax + Lt.by);
                            const float xlo = ccx * wx + kx * Lt.ax - slack, xhi = (ccx + 1) * wx + kx * Lt.ax + slack;
                            const float ylo = ccy * wy + ky * Lt.by - slack, yhi = (ccy + 1) * wy + ky * Lt.by + slack;
                            float ddx = fmaxf(0.f, fmaxf(xlo - (cxx + hx), (cxx - hx) - xhi)), ddy = fmaxf(0.f, fmaxf(ylo - (cyy + hy), (cyy - hy) - yhi));
                            const float rz2 = R2 - ddx * ddx - ddy * ddy;
                            if (rz2 <= 0.f) { fa = 1.f; fb = 0.f; }
                            else {
                                const float rz = sqrtf(rz2) * 1.0001f;
                                const float za = (czz - hz - rz - kz * Lt.cz) / Lt.cz - feps, zb = (czz + hz + rz - kz * Lt.cz) / Lt.cz + feps;
                                fa = fmaxf(fa, za); fb = fminf(fb, zb);
                            }
                        }
                        if (fb > fa) {
                        if (serp & 1) { const float t = 1.f - fb; fb = 1.f - fa; fa = t; }
                        int blo = ((int)(fa * 1048575.0f) >> 14) - 1, bhi = ((int)(fb * 1048575.0f) >> 14) + 1;   // one bucket of slack for float rounding
                        blo = blo < 0 ? 0 : blo; bhi = bhi > 63 ? 63 : bhi;
                        const int* zi = zIndex + (size_t)col * 65;
                        cStart = zi[blo]; cLen = zi[bhi + 1] - zi[blo];      // an interval of the sorted padded order
                        }
                    }
                }
                // Only the blocks this i-block owns are enumerated (round 4: half of every run used to be walked for nothing): the part of the
                // run below the block's own atoms holds owned blocks of parity (I & 1) ^ 1, the part above it of parity I & 1.  Each part is
                // queued as an interval of that parity's compact index space: f(x) = 32 * (owned blocks below block x >> 5) + (x & 31 if block
                // x >> 5 is owned), so that the walk below visits owned atoms only, in the order it always has.
#pragma unroll
                for (int part = 0; part < 2; part++) {
                    const int pi = part == 0 ? ((I & 1) ^ 1) : (I & 1);
                    int a = cStart, b = cStart + cLen;
                    if (part == 0) b = b < I * 32 ? b : I * 32; else a = a > (I + 1) * 32 ? a : (I + 1) * 32;
                    auto compact = [pi](int x) { const int B = x >> 5; return (((B + 1 - pi) >> 1) << 5) + (((B & 1) == pi) ? (x & 31) : 0); };
                    const int g0 = b > a ? compact(a) : 0, gLen = b > a ? compact(b) - g0 : 0;
                    const unsigned long long mq = __ballot(gLen > 0);
                    const int nQ = __popcll(mq);
                    if (nQ == 0) continue;      // (uniform)
                    if (nCmb + nQ > 64) runCandidates();
                    if (gLen > 0) { const int o = nCmb + lanePrefix(mq); cmbStart[o] = g0; cmbPrefix[o] = gLen; cmbCode[o] = code | (pi << 8); }
                    nCmb += nQ;
                }
            }
        }
        if (nCmb > 0 && !failed) runCandidates();
        if (exact && !failed) exactFilter();
        // close the subset segment: pad to a whole tile, record the tiles' subset
        const int padded = (count + 31) & ~31;
        for (int k = count + lane; k < padded; k += 64) list[k] = -1;
        for (int t = (segStart >> 5) + lane; t < (padded >> 5); t += 64) tileSub[t] = s;
        count = padded; filtered = padded;
    }
    const long long tGather = p.dbgOut ? (long long)wall_clock64() : 0;
    if (!failed && count > 0) flush(count, hasDiag);
    if (failed && lane == 0) atomicAdd(&p.counters[3], 1);
    if (p.dbgOut && lane == 0) { p.dbgOut[2 * I] = tStart; p.dbgOut[2 * I + 1] = (long long)wall_clock64(); p.dbgOut[2 * p.nBlocks + 2 * I] = tProlog; p.dbgOut[2 * p.nBlocks + 2 * I + 1] = tGather; }
}

// ---- clears of phase B: builder counters (not [7], the padded count of phase A), column ranges, z index (large = unset), user map ----
template <typename Real> __global__ void k_nbClear(const NbParams<Real> p) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nCols = (size_t)p.nSubsets * p.ncx * p.ncy;
    if (i < 7 || (i >= 32 && i < (size_t)32 * (1 + NB_PARTS))) p.counters[i] = 0;
    if (i < nCols) p.colRange[i] = make_int2(0x7FFFFFFF, 0);      // atomicMin / atomicMax targets of k_nbScatter; k_nbZPrefix turns empty runs into (0, 0)
    if (i < nCols * 65) p.zIndex[i] = 0x7F7F7F7F;
    if (i < (size_t)p.nPadded) p.sortedToUser[i] = -1;
}

// ---- 5. work items of the 64 partitions -> one contiguous array, full (8-tile) items first; totals into counters[0..7] ------------
template <typename Real> __global__ __launch_bounds__(256) void k_nbCompactWork(const NbParams<Real> p) {
    __shared__ int s_off[2][NB_PARTS + 1];
    if (threadIdx.x == 0) {
        int accF = 0, accP = 0, tiles = 0, masksN = 0, maxTiles = 0, maxWork = 0;
        for (int q = 0; q < NB_PARTS; q++) {
            const int* c = p.counters + 32 * (1 + q);
            s_off[0][q] = accF; s_off[1][q] = accP;
            accF += c[1]; accP += c[4]; tiles += c[0]; masksN += c[2];
            maxTiles = maxTiles > c[0] ? maxTiles : c[0]; maxTiles = maxTiles > c[2] ? maxTiles : c[2];
            maxWork = maxWork > c[1] ? maxWork : c[1]; maxWork = maxWork > c[4] ? maxWork : c[4];
        }
        s_off[0][NB_PARTS] = accF; s_off[1][NB_PARTS] = accP;
        if (blockIdx.x == 0) { p.counters[0] = tiles; p.counters[1] = accF; p.counters[2] = masksN; p.counters[4] = accP; p.counters[5] = maxTiles; p.counters[6] = maxWork; }
    }
    __syncthreads();
    const int q = blockIdx.x % NB_PARTS, which = blockIdx.x / NB_PARTS;      // which = 0: full items, 1: partial items
    const int workRegion = p.workCapacity / NB_PARTS;
    const int n = s_off[which][q + 1] - s_off[which][q];
    const int4* src = (which ? p.workItemsPartial : p.workItemsStage) + (size_t)q * workRegion;
    int4* dst = p.workItems + (which ? s_off[0][NB_PARTS] : 0) + s_off[which][q];
    if (n > workRegion) return;      // overflowed partition: the host retries with larger regions
    if ((size_t)(dst - p.workItems) + (size_t)n > (size_t)2 * p.workCapacity) return;      // (behind an overflowed partition the offsets are meaningless: nothing may be written past the array)
    for (int i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
}

// ---- driver ---------------------------------------------------------------------------------------------------------
template <typename Real> size_t nbSortTempBytes(int n) {
    size_t bytes = 0, b2 = 0, b3 = 0;
    unsigned long long* k = nullptr; int* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)n, 0, 64, (hipStream_t)0);
    (void)rocprim::inclusive_scan(nullptr, b2, v, v, (size_t)n, rocprim::maximum<int>(), (hipStream_t)0);
    (void)rocprim::exclusive_scan(nullptr, b3, v, v, 0, (size_t)n, rocprim::plus<int>(), (hipStream_t)0);
    return std::max(bytes, std::max(b2, b3));
}

// Phase A: wrapped coordinates, sort, block segmentation.  Afterwards counters[7] holds the padded atom count (read it back, size the
// per-slot arrays, then run phase B).
template <typename Real> void launchNeighborSort(const NbParams<Real>& p, const void* userPos, int isDouble, int stride4, void* sortTemp, size_t sortTempBytes, hipStream_t s) {
    const int n = p.nAtoms;
    const int stride = stride4 ? 4 : 3;
    dim3 block(256), gridN((n + 255) / 256);
    launchZeroFill(p.counters, sizeof(int) * 32 * (1 + NB_PARTS), s);      // (a kernel, not a memset node: misc.hip)
    if (n <= 0) return;
    if (isDouble) hipLaunchKernelGGL((k_nbKeys<Real, double>), gridN, block, 0, s, p, (const double*)userPos, stride);
    else hipLaunchKernelGGL((k_nbKeys<Real, float>), gridN, block, 0, s, p, (const float*)userPos, stride);
    (void)rocprim::radix_sort_pairs(sortTemp, sortTempBytes, p.keysIn, p.keysOut, p.valsIn, p.valsOut, (size_t)n, 0, 20 + p.colBits + p.subsetBits, s);
    launchZeroFill(p.blockWideOut, sizeof(int) * (size_t)n, s);
    NbParams<Real> p0 = p;
    p0.blockWide = nullptr;                                 // pass 1: subset boundaries and jumps
    for (int pass = 0; pass < 2; pass++) {
        hipLaunchKernelGGL((k_nbJumpFlags<Real>), gridN, block, 0, s, pass == 0 ? p0 : p);
        (void)rocprim::inclusive_scan(sortTemp, sortTempBytes, p.segKey, p.segStart, (size_t)n, rocprim::maximum<int>(), s);
        hipLaunchKernelGGL((k_nbPadExtra<Real>), gridN, block, 0, s, p);
        (void)rocprim::exclusive_scan(sortTemp, sortTempBytes, p.padExtra, p.padBefore, 0, (size_t)n, rocprim::plus<int>(), s);
        if (pass == 0) hipLaunchKernelGGL((k_nbWideDetect<Real>), gridN, block, 0, s, p);   // pass 2 dissolves the blocks it flags
    }
    hipLaunchKernelGGL((k_nbPadTotal<Real>), dim3(1), dim3(64), 0, s, p);
}

// Phase B: padded sorted arrays, column ranges and z index, block bounds, tiles, work items.
template <typename Real> void launchNeighborBuild(const NbParams<Real>& p, hipStream_t s) {
    const int n = p.nAtoms;
    dim3 block(256), gridN((n + 255) / 256);
    {   // one launch instead of five memset nodes (each costs ~10 us of launch gap at this point of the rebuild)
        const size_t nCols = (size_t)p.nSubsets * p.ncx * p.ncy;
        const size_t most = std::max(std::max(nCols * 65, (size_t)p.nPadded), (size_t)32 * (1 + NB_PARTS));
        hipLaunchKernelGGL((k_nbClear<Real>), dim3((unsigned)((most + 255) / 256)), block, 0, s, p);
    }
    if (n > 0) {
        hipLaunchKernelGGL((k_nbScatter<Real>), gridN, block, 0, s, p);
        hipLaunchKernelGGL((k_nbPad<Real>), dim3((p.nPadded + 255) / 256), block, 0, s, p);
        hipLaunchKernelGGL((k_nbZPrefix<Real>), dim3((p.nSubsets * p.ncx * p.ncy + 255) / 256), block, 0, s, p);
        hipLaunchKernelGGL((k_nbBounds<Real>), dim3((p.nBlocks + 7) / 8), block, 0, s, p);
        const int nOwned = (p.nBlocks / p.shardPeriod) * p.shardWidth + std::min(std::max(p.nBlocks % p.shardPeriod - p.shardBegin, 0), p.shardWidth);
        if (nOwned > 0) {
            NbParams<Real> pb = p; pb.nOwned = nOwned;
            hipLaunchKernelGGL((k_nbBuildTiles<Real>), dim3((nOwned + 3) / 4), block, 0, s, pb);
        }
        hipLaunchKernelGGL((k_nbCompactWork<Real>), dim3(2 * NB_PARTS), block, 0, s, p);
    }
}

// The build's eight totals into mapped host memory, then a sequence number behind a system-scope fence: the host spins on the number
// instead of sleeping in hipStreamSynchronize (whose wake-up costs 30-45 us of idle GPU per rebuild).
__global__ void k_nbPublish(const int* __restrict__ counters, volatile int* __restrict__ hostOut, int seq) {
    if (threadIdx.x < 8) hostOut[threadIdx.x] = counters[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) hostOut[8] = seq;
}
void launchNeighborPublish(const int* counters, int* hostMapped, int seq, hipStream_t s) { hipLaunchKernelGGL(k_nbPublish, dim3(1), dim3(64), 0, s, counters, (volatile int*)hostMapped, seq); }

template size_t nbSortTempBytes<float>(int);
template size_t nbSortTempBytes<double>(int);
template void launchNeighborSort<float>(const NbParams<float>&, const void*, int, int, void*, size_t, hipStream_t);
template void launchNeighborSort<double>(const NbParams<double>&, const void*, int, int, void*, size_t, hipStream_t);
template void launchNeighborBuild<float>(const NbParams<float>&, hipStream_t);
template void launchNeighborBuild<double>(const NbParams<double>&, hipStream_t);

}  // namespace snb
