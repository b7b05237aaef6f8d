// snb_internal.h -- shared declarations of the MI355X SlicedNonbondedForce engine (not part of the C ABI).
//
// Data layout in HBM (see DESIGN.md "Data layout"):
//   atoms are kept in a SORTED order: by subset, then along a serpentine path of xy-columns, by z inside a
//   column; each subset is padded to a multiple of 32 so that every 32-atom BLOCK has one subset.  All hot
//   arrays are struct-of-arrays over the sorted index:
//     posq   : Real4[Npad]   x,y,z (box-wrapped at build time, continuous afterwards), q
//     sigeps : Real2[Npad]   sigma/2, 2*sqrt(epsilon)      (reference convention, ReferenceNonbondedSlicingKernels.cpp:364-368)
//     fx,fy,fz : Real[Npad]  direct-space force accumulators (atomics)
//   tiles (32 i-atoms of one block x 32 individually gathered j-atoms of ONE subset):
//     tileJ    : int32[T][32]  sorted j index | (periodic-image code << 27)
//     tileInfo : int4[T]       x = slice index of (i-block subset, j subset), y = exclusion-mask index or -1, z = j subset
//     masks    : uint32[M][32] bit j of word i set = pair (i,j) excluded
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <string>
#include <vector>

#define SNB_ONE_4PI_EPS0 138.93545764438198
#define SNB_PI 3.14159265358979323846
#define SNB_EPSILON0 (1.0 / (4.0 * SNB_PI * SNB_ONE_4PI_EPS0))
// tile j entry = sorted atom index (25 bits) | periodic image code << 25; code = (kx+2)*25 + (ky+2)*5 + (kz+2), k in -2..2: the j atom is taken
// at its stored position + k * box diagonal (stored positions of a block are re-imaged to be compact, so k reaches +-2); -1 = padding slot
#define SNB_JIDX_MASK 0x01FFFFFF
#define SNB_JSHIFT_BITS 25
#define SNB_JCODE_CENTER 62
// Per-slice energies are accumulated into one of 64 copies of the [S][2] table, chosen by work-group index, and summed by the host
// after the copy back: tens of thousands of wavefronts adding into the same ~20 doubles were serialised in L2 (0.7 ms of a 0.95 ms
// energy-step pair kernel).
#define SNB_SLICE_E_PARTS 64
#define SNB_SLICE_E_PARTITION(base, stride) ((base) + (size_t)(blockIdx.x & (SNB_SLICE_E_PARTS - 1)) * (size_t)(stride))
#define SNB_PME_ORDER 5
// SNB_STEP_TRACE slots (100 MHz wall clock): kernel start = the stamp of its work-group 0, end = the latest work-group's exit
#define SNB_TRACE_START(ptr, slot) do { if ((ptr) && blockIdx.x == 0 && threadIdx.x == 0) (ptr)[slot] = (long long)wall_clock64(); } while (0)
// (a plain store: the last work-group to leave writes last, near enough; an atomic maximum from every work-group serialises at ~15 ns each on one address)
#define SNB_TRACE_END(ptr, slot) do { if ((ptr) && threadIdx.x == 0) (ptr)[slot] = (long long)wall_clock64(); } while (0)

// Degrees of the double-precision pair kernel's polynomials (forces-only steps): Chebyshev truncation error of the Ewald factor Bt on
// [0, (cutoff + skin + 0.02)^2] at alpha = 2.6283/nm: degree 14 3.9e-10 of Bt(0), 15 4.7e-11, 16 5.6e-12, 17 6.6e-13; of the LJPME dispersion
// factor Gd: 16 1.3e-11, 17 1.4e-12, 18 1.5e-13.  Round 2 ran both at degree 20 (1e-13 / 4e-15): five orders beyond the 1e-5 parity bar is
// plenty, and each degree is one f64 FMA per pair (c5: 2.32 -> ms for the pair kernel).
#define SNB_EW_DEG_F64 16
#define SNB_DISP_DEG_F64 17

namespace snb {

template <typename Real> struct Vec;
template <> struct Vec<float> { using T2 = float2; using T4 = float4; };
template <> struct Vec<double> { using T2 = double2; using T4 = double4; };

// Method classes the pair kernel is compiled for.
enum { MC_NOCUTOFF = 0, MC_RF = 1, MC_EWALD = 2, MC_LJPME = 3 };

template <typename Real> struct DirectParams {
    const typename Vec<Real>::T4* posq;
    const typename Vec<Real>::T2* sigeps;
    const int* blockSubset;   // [numBlocks]
    const int4* workItems;    // [numItems] (block, first tile, tile count <= 8, -), longest first
    const int* tileJ;         // [T*32]
    const int4* tileInfo;     // [T]
    const unsigned* masks;    // [M*32]
    Real* fx; Real* fy; Real* fz; int fs;   // direct-space force accumulators: component bases and the index stride of an atom (1: three arrays; 4: one (x,y,z,-) record per atom)
    int fixed;                              // SNB_MIXED: the accumulators are 64-bit fixed point (2^32 per kJ/mol/nm) behind the same bases
    double* sliceE;           // [S*2] raw energies
    const Real* lambdas;      // [S*2]
    const int* sliceNeed;     // [S] energy steps: non-zero = this slice's raw energies are wanted (derivative-only steps ask for a few slices)
    int numWork, workStart, workStride;   // sharding: items workStart, workStart+workStride, ...
    int nsub;
    Real cutoff2, krf, crf, alpha, alphaD, k4pe;          // k4pe = ONE_4PI_EPS0
    Real alpha2l2e;                                        // alpha^2 * log2(e)
    Real dispPoly[21];                                     // LJPME, double, forces only: alpha_d^8 e^{-x} sum x^k/(k+4)!, x = (alpha_d r)^2, degree 20 in the same t
    Real ewPolyE[14];                                      // energy steps of the packed kernel: erf(alpha r)/r ~ sum_k ewPolyE[k] t^k (degree 13, same t)
    Real ewPoly[21]; Real ewScale; int ewUsePoly;         // forces-only paths: Bt(r^2) ~ sum_k ewPoly[k] t^k, t = r^2 * ewScale - 1; degree 11 (float) / 20 (double), engine.hip buildEwaldPoly
    Real invCut6, multShift6;                              // LJPME potential shifts
    int useSwitch; Real switchDist, invSwitchWidth;
    Real box[9]; Real invBoxDiag[3];                       // for the per-pair wrap variant
    Real boxDiag[3];                                       // rectangular box edge lengths (image codes are decoded against them)
    // Overlapped steps (engine.hip enqueueStep, "overlap"): the work items are handed out through a device counter to two launches of
    // the tile kernel -- a resident first launch beside the reciprocal pipeline, held to cuLimit work-groups per CU so that the PME
    // work-groups always find room, and a second launch that fills the chip once the pipeline is done.  Null: the static item loop.
    int* workCounter;         // [SNB_WORK_SHARDS][32] claim counters, one per 128-byte line (zeroed by the position-gather pass); direct.hip WorkClaim
    int* cuSlots;             // [SNB_CU_SLOTS] resident work-groups of the limited launch per physical CU (zeroed by the same pass), or null
    int cuLimit;              // work-groups of this launch allowed to stay on one CU (0: no limit)
    int cuBaseMax;            // ... chosen by position: highest VGPR_BASE (units of 8 registers) a staying wave may have; < 0: by order of arrival (direct.hip cuResident)
    int* cuTrace;             // SNB_OVERLAP_DEBUG: [SNB_CU_SLOTS][4][2] raw GPR_ALLOC / LDS_ALLOC registers of the first four arrivals per CU, or null
    int gridCap;              // work-groups to launch in the dynamic mode
    long long* stepTrace; int traceSlot;      // SNB_STEP_TRACE: wall-clock stamps of a replayed step (start by work-group 0, latest end), see engine.hip
    int listsLast;            // the launch's pair-list work-groups (exclusion corrections, 1-4) come after its tile work-groups instead of before them
};
// physical CU key of a work-group: XCC_ID (3 bits) | SE_ID, SH_ID, CU_ID of HW_ID (bits 15:8)
#define SNB_CU_SLOTS 2048
#define SNB_WORK_SHARDS 16
#define SNB_OVERLAP_INTS (SNB_WORK_SHARDS * 32 + SNB_CU_SLOTS)      // the step's overlap scratch: claim counters, then the CU table

template <typename Real> struct PairListParams {  // 1-4 exceptions: one thread per pair; exclusion corrections: one thread per atom
    const typename Vec<Real>::T4* posq;
    const typename Vec<Real>::T2* sigeps;
    const int* blockSubset;
    const int* exclStart;     // [N+1] exclusion CSR in USER order (static across re-sorts)
    const int* exclList;      // user partner indices
    const int* sortedToUser; const int* userToSorted;
    int nSlices;
    const int2* pairs;        // 1-4 pairs, USER indices (mapped through userToSorted in the kernel)
    const typename Vec<Real>::T4* params;   // 1-4: (sigma, 4 eps, k*qq, slice bits) ; exclusion: (k*qi*qj, c6i*c6j, -, slice bits)
    int n;                    // number of 1-4 pairs
    int nExclAtoms;           // atoms visited by the exclusion-correction part (0: none)
    Real* fx; Real* fy; Real* fz; int fs, fixed;
    double* sliceE;
    const Real* lambdas;
    const int* sliceNeed;     // [S] as in DirectParams
    int periodic;
    const Real* imageOffset;  // [Npad*3] wrapped - user coordinates (to undo the wrap for non-periodic exceptions)
    Real box[9];
    Real alpha, alphaD;
    double alpha64;           // the Ewald alpha unrounded (energy evaluations of the exclusion corrections run in double)
    int ljpme;
};

struct PmePlanDims {
    int nx, ny, nz, nzc;      // nzc = nz/2+1
    int nfx, nfy, nfz;        // number of radix factors per axis
    int fx[16], fy[16], fz[16];
    int rx1, rx2, ry1, ry2, rz1, rz2;   // two-pass register-FFT split n = r1*r2 per axis (0 = use the staged Stockham path)
    int px1, px2, py1, py2;             // plane path (pme.hip k_planeXY): splits of the x and y axes into radices it has pass bodies for (0 = none)
};

// Per-kernel begin/end stamps of a timed (eager) step: the engine points g_stamps at a set of event pairs before it enqueues the step;
// the launchers below it stamp their kernel with hipExtLaunchKernelGGL (the duration rocprofv3 reports, no marker-packet overhead) when
// a slot is offered.  Host-side only; null on graph-captured steps.
struct KernelStamps { hipEvent_t start[16] = {}, stop[16] = {}; bool used[16] = {}; };
extern thread_local KernelStamps* g_stamps;
#define SNB_STAMPED_LAUNCH(SLOT, KERNEL, GRID, BLOCK, LDS, STREAM, ...) do { \
        snb::KernelStamps* ks_ = snb::g_stamps; const int slot_ = (SLOT); \
        if (ks_ && slot_ >= 0 && slot_ < 16 && ks_->start[slot_] && !ks_->used[slot_]) { ks_->used[slot_] = true; hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, ks_->start[slot_], ks_->stop[slot_], 0, __VA_ARGS__); } \
        else hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, __VA_ARGS__); } while (0)

struct SliceFinish {
    const double* sums;       // [3 nsub] per-subset sum q, sum q^2, sum c6^2 (k_paramSums), or null
    const double* dispCoef;   // [S] dispersion-correction coefficients, or null
    double selfCoulomb;       // -ONE_4PI_EPS0 alpha / sqrt(pi)          (x sum q^2, diagonal slices)
    double selfDispersion;    // alpha_d^6 / 12                           (x sum c6^2, LJPME)
    double background;        // -1 / (4 alpha^2) / (2 eps0 V)            (x Q_a Q_b, x 2 off the diagonal)
    double invVolume;         // 1 / V                                    (x dispersion coefficient)
};
// closed-form terms of raw slice-energy entry i = 2 * slice + term (term 0: Coulomb, 1: dispersion), added to the sum of its partitions
__device__ inline double sliceFinishClosedForm(const SliceFinish& f, int i) {
    const int slice = i >> 1, term = i & 1;
    int a = 0;
    while ((a + 1) * (a + 2) / 2 <= slice) a++;      // slice = a (a + 1) / 2 + b, b <= a
    const int b = slice - a * (a + 1) / 2;
    double acc = 0;
    if (f.sums) {
        if (term == 0) {
            if (a == b) acc += f.selfCoulomb * f.sums[3 * a + 1];
            acc += (a == b ? 1.0 : 2.0) * f.sums[3 * a] * f.sums[3 * b] * f.background;
        } else if (a == b) acc += f.selfDispersion * f.sums[3 * a + 2];
    }
    if (term == 1 && f.dispCoef) acc += f.dispCoef[slice] * f.invVolume;
    return acc;
}
template <typename Real> struct PmeParams {
    PmePlanDims d;
    int nsub;                 // grids held (all subsets, or this shard's)
    int natoms;               // padded sorted atoms
    const typename Vec<Real>::T4* posq;
    const typename Vec<Real>::T2* sigeps;
    const int* atomSubset;    // [Npad] sorted
    const int* atomGrid;      // [Npad] grid slot of the atom's subset, or -1 (not owned / padding)
    const Real* fixDev;       // single-precision brick spreader: LDS accumulation in 32-bit fixed point: [0] scale, [1] its inverse (device: follows the parameters)
    long long* stepTrace;     // SNB_STEP_TRACE: [16] wall-clock stamps of the step's kernels (engine.hip printStepTrace)
    long long* trace;         // SNB_PME_TRACE: [0] summed load ticks, [1] summed compute ticks, [2] work-groups (interpolation bricks; 100 MHz ticks)
    int cellsReady;           // cells[] already hold this mesh's cells (written by the position-gather pass)
    int* cells;               // [Npad] scratch: packed mesh cell per atom (k_pmeCells), brick spreader only
    Real* gridReal;           // [nsub][nx][ny][nz]
    typename Vec<Real>::T2* gridCplx;   // [nsub][nx][ny][nzc]  (plane path: [nsub][nzc][brick][line of the brick], the z-transformed charges)
    const Real* planeEterm;             // plane path: reciprocal-space kernel values [nzc][nx][ny] in the permuted order of the in-place transforms (filled at rebuild time)
    typename Vec<Real>::T2* planeB;     // plane path (pme.hip k_planeXY): convolved potentials before the lambda mix, in the inverse z kernel's order [nx][y tile][nsub][nzc][8]; null: path off
    const typename Vec<Real>::T2* twx; const typename Vec<Real>::T2* twy; const typename Vec<Real>::T2* twz;   // roots of unity exp(-2 pi i k/n)
    const Real* modx; const Real* mody; const Real* modz;       // B-spline moduli
    Real recip[9];            // reciprocal box (ReferencePME.cpp:186-194)
    Real recipLo[9];          // single precision: the rounding remainder of recip (double-float pair), see gridCoord
    Real alpha, volume;
    int dispersion;           // 0: Coulomb charges & kernel, 1: LJPME dispersion
    const Real* lambdas;      // [S*2]
    const int* sliceNeed;     // [S] energy steps: slices whose reciprocal energy is wanted
    const int* gridSubset;    // [nsub] subset id of each held grid
    int nsubTotal;
    int mix;                  // 1: lambda-mix the potentials in k-space (unsharded); 0: plain convolution (sharded)
    int mix16;                // test switch (SNB_MIX_16X16): the 16 x 16 x 4 matrix-core form of the mix also for <= 4 subsets
    double* sliceE;
    Real* fpx; Real* fpy; Real* fpz;   // reciprocal force accumulators (plain stores when unsharded)
    int wantEnergy;
    int sortNcx, sortNcy;      // brick kernels: number of sort columns (0 = use the atomic / gather fallbacks)
    int groupX, groupY;        // sort columns per brick (brick = group * nx/sortNcx cells, at least 5)
    int zSlabs;                // bricks are also cut into this many slabs along z (nz % zSlabs == 0)
    const int2* colRange;      // [nsubTotal][ncx*ncy] sorted-atom range of every (subset, xy column)
    // own-atoms spreader (round 3, pme.hip k_spreadOwn / k_spreadMerge): every (brick, z slab) work-group accumulates ONLY the atoms sorted
    // into its columns, in an LDS region that includes the stencil's reach and a drift margin, and leaves it in ownPartial; the merge kernel
    // sums the overlapping regions per z line (exact integer sums in single precision) and runs the forward z FFT.  ownSlabs == 0: off.
    int ownSlabs, ownMargin;   // z slabs per brick; drift margin in mesh cells on either side of the brick (x and y)
    void* ownPartial;          // [nsub * bricks * ownSlabs][RX * RY * RZ] int (fixed point) or double
    int* ownBusy;              // [nsub * bricks * ownSlabs] 1 when the region was written this step
    int2* strays;              // (atom, grid slot) of atoms whose footprint left their work-group's region; merged one by one (normally none)
    int* strayCount;           // zeroed by the position-gather pass
    // brick interpolation of the step's LAST mesh, unsharded: the atom's thread also writes the step's user-order force,
    // direct-space accumulator + reciprocal force, into the caller's buffer (what k_finishForces does as a launch of its own)
    void* outForces; int outIsDouble, outAccumulate;      // [N][3] in the caller's type, or null
    const double* finParts; double* finOut; int finN; SliceFinish fin;      // finOut != null: one work-group of the interpolation also sums the slice-energy partitions (the fused k_finishSliceEnergies)
    const Real* dfx; const Real* dfy; const Real* dfz; int dfs, dfixed;   // direct-space accumulators (component bases, atom stride, 64-bit fixed point)
    const int* sortedToUser;
};

// classic Ewald reciprocal sum (ewald.hip)
template <typename Real> struct EwaldParams {
    int natoms, nsub, nk;
    const typename Vec<Real>::T4* posq; const int* atomSubset;
    const int3* kvec;          // [nk] half-space k-vectors in the reference's enumeration order (ReferenceSlicedLJCoulombIxn.cpp:288-355)
    Real* cosSin;              // [nk][2*nsub] per-subset structure factors
    Real recipBox[3];          // 2 pi / L
    double factorEwald, recipCoeff;
    const Real* lambdas; double* sliceE; int wantEnergy;
    Real* fpx; Real* fpy; Real* fpz;
};
template <typename Real> void launchEwald(const EwaldParams<Real>& p, hipStream_t s);

// GPU neighbour build (neighbor.hip)
template <typename Real> struct NbParams {
    int nAtoms, nPadded, nBlocks, nSubsets, subsetBits, colBits, ncx, ncy;      // sort key = subset | serpentine column (colBits) | z (20 bits)
    double boxm[9];  // periodic cell, rows a, b, c in OpenMM's reduced (lower-triangular) form
    double origin[3];  // subtracted from the user positions before wrapping (the enclosing cell of a non-periodic system; else 0)
    float listCutoff;
    int boxWalk;     // test switch (SNB_NB_BOX_WALK): candidate runs from the grown box alone, without the per-column z ranges of round 3
    float jumpDist;  // consecutive sorted atoms further apart than this start a new (padded) block segment
    // static, user order
    const int* uSubset; const Real* uCharge; const typename Vec<Real>::T2* uSigEps;
    const int* uExclStart; const int* uExclList;
    const int* slotOfSubset;
    int* blockSubset;
    int* segKey; int* segStart; int* padExtra; int* padBefore;   // block segmentation scratch (segStart and padBefore may alias)
    int nOwned;                                                  // i-blocks this engine builds (set by the launcher)
    const int* blockWide; int* blockWideOut;                     // [nAtoms] flags of over-extended blocks of the first segmentation pass
    float maxHalfExtent[3];                                      // a block is over-extended when an atom is further than this from its first atom
    // scratch
    Real* wrapped; Real* offsetU; unsigned long long* keysIn; unsigned long long* keysOut; int* valsIn; int* valsOut;
    float* blockCenter; float* blockHalf;
    // outputs
    int* sortedToUser; int* userToSorted; typename Vec<Real>::T4* posq; typename Vec<Real>::T2* sigeps; Real* imageOffset;
    int* atomSubset; int* atomGrid; int2* colRange;
    int* atomCell;   // [Npad] lattice cell (kx+1 | ky+1 << 2 | kz+1 << 4) of the stored position relative to the wrapped one (compact blocks)
    int* zIndex;     // [nSubsets * ncx * ncy][65] atoms of the (subset, column) run below each of 64 z buckets (scratch of the builder)
    int* tileJ; int4* tileInfo; unsigned* masks; int4* workItems; int4* workItemsStage; int4* workItemsPartial;
    int* counters;   // [32 * 65]: line 0 = totals ([0] tiles, [1] full work items, [2] masks, [3] overflow events, [4] partial items, [5] max tiles|masks and
                     // [6] max work items of a partition), lines 1..64 = the partitions' allocation counters (same slots)
    int tileCapacity, workCapacity, maskCapacity;
    int itemTiles;               // tiles per full work item of the pair kernel (8; SNB_ITEM_TILES)
    int exactPrune;              // k_nbBuildTiles: re-test the gathered atoms against the block's atoms (SNB_BOX_PRUNE=1 switches it off)
    int shardBegin, shardWidth, shardPeriod;   // tiles and work items are built only for i-blocks with block % shardPeriod in [shardBegin, shardBegin + shardWidth)
    long long* dbgOut;   // SNB_NB_TRACE: per-block start/end wall_clock64 stamps (100 MHz) of the tile builder
};
void launchExtent(const void* userPos, int isDouble, int stride4, int n, int* ext, hipStream_t s);   // ext[6]: ordered-int min xyz, max xyz
template <typename Real> size_t nbSortTempBytes(int n);
template <typename Real> void launchNeighborSort(const NbParams<Real>& p, const void* userPos, int isDouble, int stride4, void* sortTemp, size_t sortTempBytes, hipStream_t s);
template <typename Real> void launchNeighborBuild(const NbParams<Real>& p, hipStream_t s);
void launchDispFlagsReset(int* flagsMapped, hipStream_t s);      // misc.hip
void launchZeroFill(void* ptr, size_t bytes, hipStream_t s);      // misc.hip: zero fill as a kernel (never a graph memset node)
void launchNeighborPublish(const int* counters, int* hostMapped, int seq, hipStream_t s);   // counters[0..7] + sequence number into mapped host memory (the host spins on it)

// ---- launchers implemented in the .hip translation units -------------------------------------
template <typename Real> bool launchDirect(const DirectParams<Real>& p, int methodClass, bool wrap, bool energy, const PairListParams<Real>* lists, hipStream_t s,
                                          hipEvent_t evStart = nullptr, hipEvent_t evStop = nullptr, bool* timed = nullptr);   // true: lists ran inside the launch
template <typename Real> void launchPairLists(const PairListParams<Real>& p, bool energy, hipStream_t s);
template <typename Real> int launchPmeSpread(const PmeParams<Real>& p, hipStream_t s);   // 1: forward z FFT already done; 2: ... and the spectrum is plane-major (plane path)
template <typename Real> bool launchPlaneEterm(const PmeParams<Real>& p, Real* table, hipStream_t s);   // rebuild time: fills the plane path's kernel-value table
template <typename Real> void launchPmePlanePath(const PmeParams<Real>& p, hipStream_t s);   // after a spreader that returned 2: k_planeXY + k_fftZInvMix instead of forward FFT, convolution, inverse FFT
template <typename Real> void launchPmeForwardFFT(const PmeParams<Real>& p, hipStream_t s, bool zDone);
template <typename Real> void launchPmeConvolution(const PmeParams<Real>& p, hipStream_t s);   // fused x-FFT, energy, convolution, inverse x-FFT
template <typename Real> void launchPmeInverseFFT(const PmeParams<Real>& p, hipStream_t s);
template <typename Real> void launchPmeFFTX(const PmeParams<Real>& p, int sign, hipStream_t s);   // x axis alone (test hook)
template <typename Real> bool launchPmeInterpolate(const PmeParams<Real>& p, hipStream_t s);   // true: the kernel also delivered the user-order forces (p.outForces)
// fractional mesh coordinate of a position: cell index and offset inside the cell (ReferencePME.cpp:268-305); shared by the PME
// kernels and the position-gather pass so that both always agree on an atom's cell.
// Single precision carries the fraction as a double-float pair.  The plain form t = frac(x r) n rounds x r at magnitude 1 (6e-8) and
// r itself to 6e-8 relative, i.e. the offset inside the cell to 1e-5 cells on a 180-point mesh -- 2e-6 nm, times the gradient of the
// reciprocal force (1e4 kJ/mol/nm^2 on a water atom) = 0.02 kJ/mol/nm per atom: measured on c5 (1M atoms, 180^3 + 90^3) as
// 0.011-0.03 on atoms whose total force is 5, 2e-3 relative; the exact products below leave 1e-7 cells (tools/dbg_tail.py).
__device__ inline void twoProdAdd(float a, float bh, float bl, float& hi, float& lo) {      // (hi, lo) += a * (bh + bl), hi + lo exact up to second order
    const float ph = __fmul_rn(a, bh), pl = fmaf(a, bh, -ph) + a * bl;      // (__fmul_rn / __fadd_rn: never contracted into an fma)
    const float s = __fadd_rn(hi, ph), bb = s - hi;
    lo += ((hi - (s - bb)) + (ph - bb)) + pl;
    hi = s;
}
template <typename Real> __device__ inline void gridCoord(const Real* recip, const Real* recipLo, Real x, Real y, Real z, int nx, int ny, int nz, int* idx, Real* frac) {
    const int n[3] = {nx, ny, nz};
#pragma unroll
    for (int d = 0; d < 3; d++) {
        if constexpr (sizeof(Real) == 4) {
            float hi = __fmul_rn(x, recip[d]), lo = fmaf(x, recip[d], -hi) + x * recipLo[d];
            if (recip[3 + d] != 0.f) twoProdAdd(y, recip[3 + d], recipLo[3 + d], hi, lo);      // (lower-triangular reciprocal box: uniform branches,
            if (recip[6 + d] != 0.f) twoProdAdd(z, recip[6 + d], recipLo[6 + d], hi, lo);      //  a rectangular box takes neither)
            const float u = hi - floorf(hi);                      // exact
            const float fn = (float)n[d];
            const float ph = __fmul_rn(u, fn), pl = fmaf(u, fn, -ph) + lo * fn;      // (u + lo) n as a pair
            float tf = floorf(ph);
            float fr = (ph - tf) + pl;                            // ph - tf exact
            if (fr < 0.f) { fr += 1.f; tf -= 1.f; } else if (fr >= 1.f) { fr -= 1.f; tf += 1.f; }
            if (fr < 0.f) fr = 0.f;
            int ti = (int)tf;
            ti = ti >= n[d] ? ti - n[d] : (ti < 0 ? ti + n[d] : ti);
            frac[d] = fr; idx[d] = ti;
        } else {
            Real t = x * recip[d] + y * recip[3 + d] + z * recip[6 + d];
            t = (t - floor(t)) * n[d];
            int ti = (int)t;
            frac[d] = t - ti;
            idx[d] = ti >= n[d] ? ti - n[d] : ti;
        }
    }
}


// Coulomb-mesh geometry handed to the position-gather pass (cells == nullptr: no mesh / brick spreader not in use)
template <typename Real> struct GatherCells {
    Real recip[9], recipLo[9]; int nx, ny, nz; int* cells; const int* atomGrid;
    // displacement watch (posRef == nullptr: off): positions at the last rebuild; flags[0] |= 1 when an atom has moved further than
    // sqrt(warn2) (time to rebuild), flags[1] |= 1 beyond sqrt(fail2) = skin/2 (the list may already have missed a pair)
    const typename Vec<Real>::T4* posRef; int* flags; Real warn2, fail2;
    double* clearE; int nClearE;      // energy steps: the slice-energy partitions, zeroed by this pass (a memset of their own was a 5 us launch)
    int* zeroInts; int nZeroInts;     // small per-step counters reset by this pass (the spreader's stray-atom counts)
    long long* stepTrace;             // SNB_STEP_TRACE
    int* zeroInts2; int nZeroInts2;   // ... and the work counter + CU table of an overlapped step's pair kernel
};

template <typename Real> void launchGatherPositions(const void* userPos, int isDouble, int stride4, const int* sortedToUser, const Real* imageOffset,
                                                    typename Vec<Real>::T4* posq, int nPadded, Real* forces, int nClear, const GatherCells<Real>& gc, hipStream_t s);      // forces: 7 * nPadded values cleared
template <typename Real> void launchRefreshParams(const int* sortedToUser, const Real* uCharge, const typename Vec<Real>::T2* uSigEps, typename Vec<Real>::T4* posq,
                                                  typename Vec<Real>::T2* sigeps, int nPadded, hipStream_t s);
template <typename Real> void launchFinishForces(const Real* fx, const Real* fy, const Real* fz, int fs, int fixed, const Real* fpx, const Real* fpy, const Real* fpz,
                                                 const int* userToSorted, int nAtoms, void* out, int isDouble, int accumulate, hipStream_t s);

// closed-form terms of the raw slice energies (k_finishSliceEnergies); a null pointer / zero factor switches a term off
void launchFinishSliceEnergies(const double* parts, double* out, int n, const SliceFinish& f, hipStream_t s);
#define SNB_PARAM_SUM_ROWS 256      // work-groups of k_paramSums, each leaving one row of partial sums
template <typename Real>
void launchParticleParams(int n, int nsub, const double* base, const int* offStart, const int* offGlobal, const double* offDelta, const double* globals,
                          const int* uSubset, Real* uCharge, typename Vec<Real>::T2* uSigEps, double* sums, Real* fix, double headQ, double headC, hipStream_t s);
template <typename Real>
void launchExceptionParams(int n, const double* base, const int* offStart, const int* offGlobal, const double* offDelta, const double* globals, const int* slice,
                           typename Vec<Real>::T4* out, hipStream_t s);

int legalGridSize(int n);
bool factorize(int n, int* factors, int* nfactors);
bool splitTwoPass(int n, int* r1, int* r2);
bool splitPlane(int n, int* r1, int* r2);

}  // namespace snb
