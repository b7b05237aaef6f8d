// engine.hip -- host side of the MI355X SlicedNonbondedForce engine and its C ABI (include/snb.h).
//
// Functionally it stands where the reference has CommonCalcSlicedNonbondedForceKernel
// (platforms/common/src/CommonNonbondedSlicingKernels.cpp: commonInitialize :256-844, execute :846-1402,
// copyParametersToContext :1404-1568) plus the OpenMM utilities that class leans on (NonbondedUtilities'
// neighbour list and tile driver, BondedUtilities, ComputeSort) -- but none of its structure: there is no JIT, no
// ComputeContext; a handle owns plain HIP buffers and enqueues precompiled gfx950 kernels on one stream.
#include "../../include/snb.h"
#include "snb_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <numeric>
#include <set>
#include <tuple>
#include <type_traits>

namespace snb {

static thread_local std::string g_createError;

struct HipError { std::string msg; };
#define HIPCHECK(expr)                                                                                         \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) throw HipError{std::string(#expr) + ": " + hipGetErrorString(e_)};               \
    } while (0)

template <typename T> struct DevBuf {
    T* p = nullptr; size_t n = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    void resize(size_t m) {
        if (m <= n && p) return;
        if (p) { (void)hipFree(p); p = nullptr; }
        n = m > 0 ? m : 1;
        HIPCHECK(hipMalloc((void**)&p, sizeof(T) * n));
    }
    void upload(const std::vector<T>& h, hipStream_t s) {
        resize(h.size());
        if (!h.empty()) HIPCHECK(hipMemcpyAsync(p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice, s));
    }
};

// Small asynchronous host-to-device updates (lambdas, energy-slice mask, dispersion coefficients, global parameter values) go through a ring
// of pinned slots, one per in-flight update, each guarded by an event: the setter's own vector may be overwritten by the next call while an
// earlier copy is still queued, and a copy from pageable memory is either staged synchronously by the runtime or reads the source late
// (ADVICE r02).  A slot is reused only after its copy has completed; with 8 slots that wait never happens in practice.
struct PinnedRing {
    static constexpr int SLOTS = 8;
    struct Slot { void* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool pending = false; };
    Slot slots[SLOTS]; int next = 0;
    ~PinnedRing() { for (auto& s : slots) { if (s.ev) (void)hipEventDestroy(s.ev); if (s.p) (void)hipHostFree(s.p); } }
    void copy(void* dst, const void* src, size_t bytes, hipStream_t st) {
        if (bytes == 0) return;
        static const bool direct = getenv("SNB_NO_PINNED_RING") != nullptr;      // test switch: the copy straight from the caller's array (synchronised)
        if (direct) { HIPCHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st)); HIPCHECK(hipStreamSynchronize(st)); return; }
        Slot& s = slots[next]; next = (next + 1) % SLOTS;
        if (s.pending) { HIPCHECK(hipEventSynchronize(s.ev)); s.pending = false; }
        if (s.cap < bytes) { if (s.p) (void)hipHostFree(s.p); s.p = nullptr; s.cap = std::max<size_t>(bytes, 4096); HIPCHECK(hipHostMalloc(&s.p, s.cap, hipHostMallocDefault)); }
        if (!s.ev) HIPCHECK(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
        std::memcpy(s.p, src, bytes);
        HIPCHECK(hipMemcpyAsync(dst, s.p, bytes, hipMemcpyHostToDevice, st));
        HIPCHECK(hipEventRecord(s.ev, st)); s.pending = true;
    }
    template <typename T> void upload(DevBuf<T>& d, const std::vector<T>& h, hipStream_t st) { d.resize(h.size()); copy(d.p, h.data(), sizeof(T) * h.size(), st); }
};

struct EngineBase {
    snb_config cfg;
    std::string err;
    virtual ~EngineBase() {}
    virtual void setParticles(const double*, const double*, const double*, const int32_t*) = 0;
    virtual void setExceptions(int32_t, const int32_t*, const double*, const double*, const double*, const int32_t*) = 0;
    virtual void setLambdas(const double*) = 0;
    virtual void setParameterOffsets(int, int, const int32_t*, const int32_t*, const double*, int, const int32_t*, const int32_t*, const double*) = 0;
    virtual void setGlobalParameters(int, const double*) = 0;
    virtual void setEnergySlices(const int32_t*) = 0;
    virtual void setDispersion(const double*) = 0;
    virtual void setBox(const double*) = 0;
    virtual void setPositions(const void*, int, int, int) = 0;
    virtual void requestRebuild() = 0;
    virtual void execute(int, int, int, int, double*) = 0;
    virtual void getForces(void*, int, int, int) = 0;
    virtual void setForceOutput(void*, int, int) = 0;
    virtual void setShardBlocks(int, int, int) = 0;
    virtual void getSliceEnergies(double*) = 0;
    virtual const double* sliceEnergiesDevice() = 0;
    virtual void sync() = 0;
    virtual void getStats(snb_stats*) = 0;
    virtual void resetTimers() = 0;
    virtual void setTimingInterval(int) = 0;
    virtual void getPme(double*, int32_t*, bool dispersion) = 0;
};

// ---- dispersion coefficients (SlicedNonbondedForceImpl.cpp:150-185, 263-354), doubles throughout ----
static double evalIntegral(double r, double rs, double rc, double sigma) {
    double A = 1 / (rc - rs), A2 = A * A, A3 = A2 * A;
    double sig2 = sigma * sigma, sig6 = sig2 * sig2 * sig2;
    double rs2 = rs * rs, rs3 = rs * rs2;
    double r2 = r * r, r3 = r * r2, r4 = r * r3, r5 = r * r4, r6 = r * r5, r9 = r3 * r6;
    return sig6 * A3 *
           ((sig6 * (+rs3 * 28 * (6 * rs2 * A2 + 15 * rs * A + 10) - r * rs2 * 945 * (rs2 * A2 + 2 * rs * A + 1) +
                     r2 * rs * 1080 * (2 * rs2 * A2 + 3 * rs * A + 1) - r3 * 420 * (6 * rs2 * A2 + 6 * rs * A + 1) + r4 * 756 * (2 * rs * A2 + A) -
                     r5 * 378 * A2) -
             r6 * (+rs3 * 84 * (6 * rs2 * A2 + 15 * rs * A + 10) - r * rs2 * 3780 * (rs2 * A2 + 2 * rs * A + 1) +
                   r2 * rs * 7560 * (2 * rs2 * A2 + 3 * rs * A + 1))) /
                (252 * r9) -
            std::log(r) * 10 * (6 * rs2 * A2 + 6 * rs * A + 1) + r * 15 * (2 * rs * A2 + A) - r2 * 3 * A2);
}

static void dispersionCoefficients(int n, int nsub, const double* sigma, const double* epsilon, const int32_t* subset, double cutoff,
                                   int useSwitch, double switchDist, double* out) {
    const int S = nsub * (nsub + 1) / 2;
    std::map<std::tuple<double, double, int>, double> classes;
    for (int i = 0; i < n; i++) classes[std::make_tuple(sigma[i], epsilon[i], (int)subset[i])] += 1.0;
    std::vector<double> s1(S, 0.0), s2(S, 0.0), s3(S, 0.0);
    auto sl = [](int a, int b) { return a > b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a; };
    for (auto& c : classes) {
        double sg = std::get<0>(c.first), ep = std::get<1>(c.first); int s = std::get<2>(c.first);
        double count = c.second * (c.second + 1) / 2;
        double q = sg * sg, s6 = q * q * q;
        int slice = s * (s + 3) / 2;
        s1[slice] += count * ep * s6 * s6; s2[slice] += count * ep * s6;
        if (useSwitch) s3[slice] += count * ep * (evalIntegral(cutoff, switchDist, cutoff, sg) - evalIntegral(switchDist, switchDist, cutoff, sg));
    }
    for (auto a = classes.begin(); a != classes.end(); ++a)
        for (auto b = classes.begin(); b != a; ++b) {
            double sg = 0.5 * (std::get<0>(a->first) + std::get<0>(b->first));
            double ep = std::sqrt(std::get<1>(a->first) * std::get<1>(b->first));
            int slice = sl(std::get<2>(a->first), std::get<2>(b->first));
            double count = a->second * b->second;
            double q = sg * sg, s6 = q * q * q;
            s1[slice] += count * ep * s6 * s6; s2[slice] += count * ep * s6;
            if (useSwitch) s3[slice] += count * ep * (evalIntegral(cutoff, switchDist, cutoff, sg) - evalIntegral(switchDist, switchDist, cutoff, sg));
        }
    double N = n, numInteractions = (N * (N + 1)) / 2;
    for (int s = 0; s < S; s++)
        out[s] = 8 * N * N * SNB_PI * ((s1[s] / numInteractions) / (9 * std::pow(cutoff, 9)) - (s2[s] / numInteractions) / (3 * std::pow(cutoff, 3)) + s3[s] / numInteractions);
}

// ---- B-spline moduli (ReferencePME.cpp:88-183) ----
static void bsplineModuli(int n, int order, std::vector<double>& out) {
    std::vector<double> data(order, 0.0), bsp(std::max(n, order + 1), 0.0);
    data[order - 1] = 0; data[1] = 0; data[0] = 1;
    for (int k = 3; k < order; k++) {
        double div = 1.0 / (k - 1.0);
        data[k - 1] = 0;
        for (int l = 1; l < (k - 1); l++) data[k - l - 1] = div * (l * data[k - l - 2] + (k - l) * data[k - l - 1]);
        data[0] = div * data[0];
    }
    double div = 1.0 / (order - 1);
    data[order - 1] = 0;
    for (int l = 1; l < (order - 1); l++) data[order - l - 1] = div * (l * data[order - l - 2] + (order - l) * data[order - l - 1]);
    data[0] = div * data[0];
    for (int i = 1; i <= order; i++) bsp[i] = data[i - 1];
    out.assign(n, 0.0);
    for (int i = 0; i < n; i++) {
        double sc = 0, ss = 0;
        for (int j = 0; j < n && j <= order; j++) { double arg = (2.0 * SNB_PI * i * j) / n; sc += bsp[j] * std::cos(arg); ss += bsp[j] * std::sin(arg); }
        out[i] = sc * sc + ss * ss;
    }
    for (int i = 0; i < n; i++)
        if (out[i] < 1.0e-7) out[i] = (out[(i - 1 + n) % n] + out[(i + 1) % n]) / 2;
}

template <typename Real> struct PmePlan {
    PmePlanDims d;
    double alpha = 0;
    bool dispersion = false;
    DevBuf<Real> gridReal;
    DevBuf<typename Vec<Real>::T2> gridCplx, gridCplxB, twx, twy, twz;      // (gridCplxB: second complex mesh of the plane path, single precision)
    DevBuf<Real> modx, mody, modz;
    DevBuf<Real> planeEterm; bool planeEtermReady = false, planeEtermFilled = false; double planeEtermKey[10] = {0};      // plane path: kernel-value table, refilled at a rebuild when the box or alpha changed
    // own-atoms spreader (pme.hip k_spreadOwn / k_spreadMerge): geometry and buffers, sized at rebuild time
    int ownSlabs = 0, ownMargin = 1; DevBuf<unsigned char> ownPartial; DevBuf<int> ownBusy; DevBuf<int2> strays;
    void init(const int g[3], int nGrids, hipStream_t s) {
        d.nx = g[0]; d.ny = g[1]; d.nz = g[2]; d.nzc = g[2] / 2 + 1;
        if (!factorize(d.nx, d.fx, &d.nfx) || !factorize(d.ny, d.fy, &d.nfy) || !factorize(d.nz, d.fz, &d.nfz)) throw HipError{"PME mesh size is not FFT-legal"};
        // pme.hip divides index values by reciprocal multiplication (FastDiv: exact for dividends below 2^22).  Its dividends are LDS element
        // indices of one work-group (< 40 960), line / chunk counts of one brick, and indices into one mesh plane or one (y, kz) slab:
        // at most 1024 x 1024 = 2^20 under this cap.
        if (d.nx > 1024 || d.ny > 1024 || d.nz > 1024) throw HipError{"PME mesh dimensions above 1024 are not supported"};
        splitTwoPass(d.nx, &d.rx1, &d.rx2); splitTwoPass(d.ny, &d.ry1, &d.ry2); splitTwoPass(d.nz, &d.rz1, &d.rz2);
        splitPlane(d.nx, &d.px1, &d.px2); splitPlane(d.ny, &d.py1, &d.py2);
        // measured on MI355X (120^3 = 8 x 15, 4 grids, single precision, Winograd radix-3/5 butterflies): two-pass register FFT vs
        // staged Stockham: inverse z 18.2 vs 23.5 us, y 30.6 vs 32.4, fused x/convolution 68.6 vs 71.0.  SNB_FFT_TWOPASS=0/1 overrides.
        // Round 3, double precision (c5, 180^3 = 12 x 15 and 90^3 = 9 x 10): the register passes win on the y and z axes (y 183 -> 150 us,
        // inverse z 185 -> 134, 90^3: 32 -> 22 and 21 -> 14) and in the fused x kernel of the 90^3 mesh (63 -> 42), but the x kernel of the
        // 180^3 mesh, which also holds the spectra of all subsets, spills with 15-point transforms in double (351 -> 488): staged there.
        bool twoPass = true;
        if (const char* e = getenv("SNB_FFT_TWOPASS")) twoPass = atoi(e) != 0;
        if (!twoPass) d.rx1 = d.ry1 = d.rz1 = d.rx2 = d.ry2 = d.rz2 = 0;
        // (round 4: the fused x kernel runs 15- / 16-point transforms in double with 256 threads and 256 registers; SNB_CONVX_STAGED_F64=1 restores the staged form)
        if (sizeof(Real) == 8 && std::max(d.rx1, d.rx2) > 12 && getenv("SNB_CONVX_STAGED_F64")) d.rx1 = d.rx2 = 0;
        gridReal.resize((size_t)nGrids * d.nx * d.ny * d.nz);
        gridCplx.resize((size_t)nGrids * d.nx * d.ny * d.nzc);
        if (sizeof(Real) == 4 && (size_t)d.nx * (d.ny | 1) * 8 <= 156 * 1024) { gridCplxB.resize((size_t)nGrids * d.nx * (d.ny + 8) * d.nzc);      // (plane path, planes that fit LDS; y padded to whole tiles of the inverse z kernel)
             planeEterm.resize((size_t)d.nx * d.ny * d.nzc); planeEtermReady = false; }
        auto tw = [&](int n, DevBuf<typename Vec<Real>::T2>& buf) {
            std::vector<typename Vec<Real>::T2> h(n);
            for (int k = 0; k < n; k++) { double a = -2.0 * SNB_PI * k / n; h[k].x = (Real)std::cos(a); h[k].y = (Real)std::sin(a); }
            buf.upload(h, s);
        };
        tw(d.nx, twx); tw(d.ny, twy); tw(d.nz, twz);
        auto md = [&](int n, DevBuf<Real>& buf) {
            std::vector<double> m; bsplineModuli(n, SNB_PME_ORDER, m);
            std::vector<Real> h(m.begin(), m.end());
            buf.upload(h, s);
        };
        md(d.nx, modx); md(d.ny, mody); md(d.nz, modz);
        HIPCHECK(hipStreamSynchronize(s));
    }
};

template <typename Real> class Engine : public EngineBase {
    using T4 = typename Vec<Real>::T4;
    using T2 = typename Vec<Real>::T2;
public:
    int N, nsub, S;
    hipStream_t stream = nullptr; bool ownStream = false;
    // ring of per-execute event sets, harvested lazily into cumulative kernel times (no per-step host sync)
    static constexpr int RING = 32;
    hipEvent_t evRebuild[3] = {nullptr, nullptr, nullptr};
    hipEvent_t evStepDone[2] = {nullptr, nullptr}; long long stepCounter = 0;      // displacement-triggered rebuilds: end-of-execute events
    hipStream_t stream2 = nullptr; hipEvent_t evFork = nullptr, evJoin = nullptr, evPairA = nullptr;
    // measured on c3: serial 0.80 ms/step, forked 0.87 (default priority) / 1.32 (high or low priority): the graph's cross-stream
    // dependencies cost more than the overlap returns, so the fork is opt-in
    bool concurrentPme = getenv("SNB_CONCURRENT_PME") && atoi(getenv("SNB_CONCURRENT_PME"));
    // Overlapped steps (round 4; the reference runs its reciprocal pipeline on a queue of its own beside the pair kernel,
    // CommonNonbondedSlicingKernels.cpp:520-530, 1176-1179, 1377-1380).  Graph steps run the PME chain on stream2 while a first launch of
    // the tile kernel, held to overlapCuLimit work-groups per CU (its work-groups count themselves per physical CU and leave when the
    // CU is full), runs beside it; a second launch behind the chain fills the chip.  Both launches claim their work items from one
    // device counter, so the split follows the chain's actual duration.  Eager (stamped) steps stay serial: the per-kernel timers keep
    // measuring every kernel alone.  dOverlap: [0] the counter, [16 ...] the SNB_CU_SLOTS residency counts; zeroed by the gather pass.
    int overlapMode = getenv("SNB_OVERLAP") ? atoi(getenv("SNB_OVERLAP")) : 1;      // default on (round 4: c3 0.414 -> 0.377 ms per step with derivatives)
    int overlapCuLimit = getenv("SNB_OVERLAP_CU_LIMIT") ? atoi(getenv("SNB_OVERLAP_CU_LIMIT")) : 2;
    int overlapGridA = getenv("SNB_OVERLAP_GRID_A") ? atoi(getenv("SNB_OVERLAP_GRID_A")) : 0;      // 0: six work-groups per CU
    int overlapGridB = getenv("SNB_OVERLAP_GRID_B") ? atoi(getenv("SNB_OVERLAP_GRID_B")) : 0;      // 0: four work-groups per CU
    long long overlapMinTiles = getenv("SNB_OVERLAP_MIN_TILES") ? atoll(getenv("SNB_OVERLAP_MIN_TILES")) : 100000;      // below this the pair kernel is shorter than the PME chain and the fork only costs (c2, 65k tiles: +2 %)
    int numCUs = 256; DevBuf<int> dOverlap, dOverlapTrace;
    struct EvSet { hipEvent_t e[5]; bool pending = false; KernelStamps ks; };   // start, direct0, direct1(=recip0 after pair lists), recip1, end; per-kernel stamps (snb_stats.sum_kernel_ms)
    std::vector<EvSet> ring; int ringPos = 0;
    // host-side definition
    std::vector<double> charge, sigma, epsilon; std::vector<int32_t> subset;
    std::vector<int32_t> excPairs; std::vector<double> excQQ, excSigma, excEps; std::vector<int32_t> excForce14;
    std::vector<double> lambdas, dispCoef; std::vector<Real> hLambdas; PinnedRing pinned;
    double box[9] = {0}; bool haveBox = false, haveParticles = false;
    // positions
    const void* devUserPos = nullptr; int posIsDouble = 1, posStride4 = 0; bool havePositions = false;
    DevBuf<unsigned char> ownedPos;
    // sorted state
    int* hNbPub = nullptr; int* dNbPub = nullptr; int nbPubSeq = 0;      // the rebuild's totals in mapped host memory + sequence number (gpuRebuild)
    // Rebuild BESIDE the steps (fixed rebuild interval; startSideBuild / finishSideBuild): `sideLead` steps before a rebuild falls due the
    // positions are copied aside and the whole GPU build runs on a stream of its own into the second set of list buffers (`shadow`), while the
    // steps go on with the list in use; when the rebuild falls due the two sets change places.  The build's small latency-bound kernels fill
    // gaps of the steps instead of standing between them (measured with a second engine as the builder, tools/async_rebuild_probe.py: 24 us
    // per step for a rebuild every 20 steps, against 38-40 in line).
    struct ListShadow {
        DevBuf<int> dUserToSorted, dSortedToUser, atomSubset, atomGrid, blockSubset, tileJ; DevBuf<int2> colRange; DevBuf<T4> posq, posRef; DevBuf<T2> sigeps;
        DevBuf<Real> imageOffset; DevBuf<int4> tileInfo, workItems; DevBuf<unsigned> masks;
    } shadow;
    hipStream_t streamBuild = nullptr; hipEvent_t evSnap = nullptr, evBuilt = nullptr, evFlagsReset = nullptr; bool flagsResetPending = false; DevBuf<unsigned char> posSnap;
    bool sortGraphSuspect = false;      // see gpuRebuild
    bool sideMode = true, sideBuilding = false, sidePending = false; int sideLead = 3, sideSeq = 0; long long sideBuilds = 0, sideDiscarded = 0;
    int npadPredict = 0; long long padMispredictions = 0;      // > 0: size of the padded arrays the next GPU rebuild assumes (gpuRebuild); how often that was too small
    int Npad = 0, numBlocks = 0; int64_t numTiles = 0, numMaskTiles = 0, shardTiles = 0; bool wrapMode = false;
    std::vector<int> sortedToUser, userToSorted;
    DevBuf<T4> posq; DevBuf<T2> sigeps; DevBuf<Real> forceBuf, imageOffset, dLambdas;
    struct FView { Real* p = nullptr; } fx, fy, fz, fpx, fpy, fpz;   // views of forceBuf (7 Npad values, cleared by the position-gather pass)
    int fstride = 1;      // index stride of an atom in the direct-space accumulators (three arrays fx | fy | fz: 1)
    // SNB_MIXED: single-precision arithmetic, direct-space forces accumulated in 64-bit fixed point (direct.hip, fAdd) -- three arrays of
    // Npad 64-bit words (6 Npad floats, one spare), then the three reciprocal arrays: 10 Npad floats, all cleared by the gather pass
    bool fixedForces() const { return sizeof(Real) == 4 && cfg.precision == SNB_MIXED; }
    int forceArrays() const { return fixedForces() ? 10 : 7; }
    void layoutForces() {
        forceBuf.resize((size_t)forceArrays() * Npad);
        if (fixedForces()) {
            fstride = 1;
            fx.p = forceBuf.p; fy.p = fx.p + 2 * (size_t)Npad; fz.p = fx.p + 4 * (size_t)Npad;      // bases of 64-bit arrays
            fpx.p = forceBuf.p + 7 * (size_t)Npad; fpy.p = fpx.p + Npad; fpz.p = fpy.p + Npad;
            return;
        }
        fstride = 1;
        fx.p = forceBuf.p; fy.p = fx.p + Npad; fz.p = fx.p + 2 * (size_t)Npad;
        fpx.p = forceBuf.p + 4 * (size_t)Npad; fpy.p = fpx.p + Npad; fpz.p = fpy.p + Npad;
    }
    DevBuf<int> pmeCells, dZIndex, dScanA, dScanB, dScanC, dAtomCell, dExtent;
    double tileCell[9] = {0};      // the cell the tile image codes refer to (the box; an enclosing cell for CutoffNonPeriodic)
    DevBuf<long long> dNbTrace, dPmeTrace, dStepTrace;      // dStepTrace: SNB_STEP_TRACE, device wall-clock stamps of the last replayed step
    bool cellsFromGather = false;   // this step's gather pass already wrote the Coulomb-mesh cells
    bool traceThisStep = false;
    // Parameter offsets on the device (the reference: platforms/common/src/kernels/nonbondedParameters.cc:4-179).  charge/sigma/epsilon
    // and the exception arrays above hold the BASE values; effective = base + sum_k global[k] * delta is formed by k_particleParams /
    // k_exceptionParams whenever a base value, an offset or a global parameter changes -- no re-sort, no tile rebuild, no graph re-capture,
    // no host synchronisation.  The same pass reduces what the closed-form energy terms and the spreader's fixed-point scale need.
    struct Offset { int target, global; double d[3]; };
    std::vector<Offset> offP, offE; std::vector<double> gValues; int nGlobals = 0;
    bool basePDirty = true, baseEDirty = true, offsetsDirty = true, globalsDirty = true;
    DevBuf<double> dBaseP, dOffPDelta, dBase14, dOff14Delta, dGlobals, dParamSums;      // dParamSums: [3 nsub] (sum q, sum q^2, sum c6^2) + 2 ordered-int maxima
    DevBuf<int> dOffPStart, dOffPGlobal, dOff14Start, dOff14Global, dSlice14;
    DevBuf<Real> dFixScale;      // [4]: fixed-point scale and its inverse of the charge mesh, then of the dispersion mesh
    DevBuf<int> dSortedToUser, dUserToSorted, blockSubset, tileJ, atomSubset, atomGrid, gridSubset, exclStart, exclList;
    // GPU neighbour build: static user-order data and scratch
    DevBuf<int> dUSubset, dSubsetStart, dSubsetPaddedStart, dSlotOfSubset, dValsIn, dValsOut, dCounters; DevBuf<Real> dUCharge, dWrapped, dOffsetU; DevBuf<T2> dUSigEps;
    DevBuf<unsigned char> dPadFlag, dSortTemp; DevBuf<unsigned long long> dKeysIn, dKeysOut; DevBuf<float> dBlockCenter, dBlockHalf;
    std::vector<int> hSubsetStart, hSubsetPaddedStart, staticBlkSubset; int staticNpad = 0; size_t tileCap = 0; bool staticDirty = true, gpuBuilt = false;
    DevBuf<int2> pairs14, pairsExcl, colRange; DevBuf<int4> tileInfo, workItems, workItemsStage, workItemsPartial; int numWorkItems = 0; int colCells[2] = {0, 0}; DevBuf<unsigned> masks;
    DevBuf<T4> params14, paramsExcl; int n14 = 0, nExcl = 0;
    DevBuf<int> dSliceNeedAll, dSliceNeedSel; std::vector<int> sliceNeedSel;      // energy steps: every slice / the slices a derivative-only step (include_energy == 2) must produce
    DevBuf<double> dDispCoef, sliceE, sliceTotal;      // 64 partitioned copies of the raw [S][2] energies, and their sum (last kernel of an energy step)
    bool energyPending = false, energySelective = false;      // the last energy step's sums are still on the device
    std::vector<double> hostSliceE;   // raw energies of the last energy evaluation (device part + host terms)
    PmePlan<Real> pme, dpme; int nGrids = 0; std::vector<int> ownedSubsets; DevBuf<int> dStrayCount;      // [2] stray atoms of the own-atoms spreader (Coulomb mesh, dispersion mesh)
    bool needRebuild = true, paramsDirty = true; int stepsSinceRebuild = 0;
    // direct-space ownership: i-block I belongs to this engine when I % shardPeriod lies in [shardBegin, shardEnd); the default is
    // (shard_rank, shard_rank + 1, shard_count); snb_set_shard_blocks lets the host rebalance direct-space work between ranks
    int shardBegin = 0, shardEnd = 1, shardPeriod = 1;
    bool ownsBlock(int b) const { const int r = b % shardPeriod; return r >= shardBegin && r < shardEnd; }
    void setShardBlocks(int begin, int end, int period) override {
        if (period < 1 || begin < 0 || end < begin || end > period) throw std::runtime_error("snb_set_shard_blocks: need 0 <= begin <= end <= period");
        if (begin == shardBegin && end == shardEnd && period == shardPeriod) return;
        shardBegin = begin; shardEnd = end; shardPeriod = period; needRebuild = true;
    }
    bool valuesDirty = false, excValuesDirty = false, haveExceptions = false;   // parameter values changed, structure did not
    // displacement watch: reference positions of the last rebuild and two flags in mapped host memory (read without synchronising)
    DevBuf<T4> posRef; int* hDispFlags = nullptr; int* dDispFlags = nullptr; int64_t listOverruns = 0;
    std::vector<int3> hKvec; DevBuf<int3> dKvec; DevBuf<Real> dCosSin;
    struct GraphKey {
        const void* pos; int isDouble, stride4; bool direct, recip; int energy; void* out; int outDouble, outAcc;      // energy: 0 forces only, 1 all slice energies, 2 selected slices
        bool operator==(const GraphKey& o) const { return pos == o.pos && isDouble == o.isDouble && stride4 == o.stride4 && direct == o.direct && recip == o.recip && energy == o.energy && out == o.out && outDouble == o.outDouble && outAcc == o.outAcc; }
    };
    void* outPtr = nullptr; int outIsDouble = 0, outAccumulate = 0; bool outputWritten = false;   // snb_set_force_output
    // captured step graphs, a few at a time: a caller that alternates between position (or output) buffers keeps one graph per buffer
    struct CachedGraph { GraphKey key; hipGraphExec_t exec; bool stale; };      // stale: the arguments changed (rebuild, parameters): re-captured and UPDATED in place at the next use
    std::vector<CachedGraph> graphs; size_t graphVictim = 0; long long execCount = 0;
    static constexpr size_t MAX_GRAPHS = 4;
    int timingInterval = 32; long long stampCounter = 0;
    void setTimingInterval(int n) override { timingInterval = n; execCount = 0; }
    struct SortGraph { std::vector<unsigned char> key; hipGraphExec_t exec; };
    std::vector<SortGraph> sortGraphs; size_t sortGraphVictim = 0; bool sortGraphBroken = false;      // phase A of the neighbour rebuild (one per buffer set / position source: up to 4)
    // A rebuild (or a changed 1-4 list, box, dispersion table) changes kernel arguments, not the step's kernels: the executable graphs are kept
    // and refreshed with hipGraphExecUpdate from a new capture.  Instantiating the forked graph of an overlapped step anew costs the host
    // ~0.7 ms (a linear one 10 us), i.e. 0.3 ms of idle GPU after every rebuild; an update that fails falls back to a new instantiation.
    void dropGraph() { for (auto& g : graphs) g.stale = true; }
    void destroyGraphs() { for (auto& g : graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec); graphs.clear(); graphVictim = 0; }
    bool lastRecip = false;
    snb_stats stats;

    Engine(const snb_config& c) {
        cfg = c; N = c.n_atoms; nsub = c.n_subsets; S = nsub * (nsub + 1) / 2;
        std::memset(&stats, 0, sizeof(stats));
        HIPCHECK(hipSetDevice(c.device));
        if (c.stream) stream = (hipStream_t)c.stream; else { HIPCHECK(hipStreamCreate(&stream)); ownStream = true; }
        ring.resize(RING);
        { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, c.device) == hipSuccess && prop.multiProcessorCount > 0) numCUs = prop.multiProcessorCount; }
        if (const char* e = getenv("SNB_SIDE_REBUILD")) sideMode = atoi(e) != 0;
        if (const char* e = getenv("SNB_SIDE_LEAD")) sideLead = std::max(1, atoi(e));
        if (overlapMode) { dOverlap.resize(SNB_OVERLAP_INTS); HIPCHECK(hipMemsetAsync(dOverlap.p, 0, sizeof(int) * SNB_OVERLAP_INTS, stream)); }
        if (getenv("SNB_STEP_TRACE")) { dStepTrace.resize(16); HIPCHECK(hipMemsetAsync(dStepTrace.p, 0, sizeof(long long) * 16, stream)); }
        if (overlapMode && getenv("SNB_OVERLAP_DEBUG")) { dOverlapTrace.resize(SNB_CU_SLOTS * 8); HIPCHECK(hipMemsetAsync(dOverlapTrace.p, 0xff, sizeof(int) * SNB_CU_SLOTS * 8, stream)); }
        for (auto& r : ring) { for (int k = 0; k < 5; k++) HIPCHECK(hipEventCreate(&r.e[k])); for (int k = 0; k < 16; k++) { HIPCHECK(hipEventCreate(&r.ks.start[k])); HIPCHECK(hipEventCreate(&r.ks.stop[k])); } }
        charge.assign(N, 0.0); sigma.assign(N, 1.0); epsilon.assign(N, 0.0); subset.assign(N, 0);
        lambdas.assign((size_t)S * 2, 1.0); dispCoef.assign(S, 0.0); hostSliceE.assign((size_t)S * 2, 0.0);
        sliceE.resize((size_t)S * 2 * SNB_SLICE_E_PARTS); sliceTotal.resize((size_t)S * 2);
        dDispCoef.upload(dispCoef, stream);
        sliceNeedSel.assign(S, 1); dSliceNeedAll.upload(sliceNeedSel, stream); dSliceNeedSel.upload(sliceNeedSel, stream);
        HIPCHECK(hipStreamSynchronize(stream));
        if (cfg.shard_count < 1) cfg.shard_count = 1;
        shardBegin = cfg.shard_count > 1 ? cfg.shard_rank : 0; shardEnd = shardBegin + 1; shardPeriod = cfg.shard_count;
        // tabulated Ewald force factor: opt-in.  Measured on MI355X it only trades 4 % of the VALU instructions for LDS gathers (the packed
        // analytic erfc is already cheap) and leaves the kernel time unchanged, so the analytic form stays the default.
        if (cfg.method >= SNB_Ewald) buildEwaldPoly();
        HIPCHECK(hipHostMalloc((void**)&hDispFlags, 64, hipHostMallocMapped));
        std::memset(hDispFlags, 0, 64);      // [0] warn, [1] overrun, [4] overruns counted on the device (k_dispFlagsReset)
        HIPCHECK(hipHostGetDevicePointer((void**)&dDispFlags, hDispFlags, 0));
        HIPCHECK(hipHostMalloc((void**)&hNbPub, 64, hipHostMallocMapped));
        std::memset(hNbPub, 0, 64);
        HIPCHECK(hipHostGetDevicePointer((void**)&dNbPub, hNbPub, 0));
        for (int s = 0; s < nsub; s++) if (s % cfg.shard_count == cfg.shard_rank) ownedSubsets.push_back(s);
        nGrids = cfg.shard_count == 1 ? nsub : (int)ownedSubsets.size();
        if (isPme()) {
            for (int d = 0; d < 3; d++) cfg.grid[d] = legalGridSize(cfg.grid[d]);
            pme.alpha = cfg.alpha; pme.dispersion = false;
            if (nGrids > 0) pme.init(cfg.grid, nGrids, stream); else { pme.d.nx = cfg.grid[0]; pme.d.ny = cfg.grid[1]; pme.d.nz = cfg.grid[2]; }
            if (cfg.method == SNB_LJPME) {
                for (int d = 0; d < 3; d++) cfg.dgrid[d] = legalGridSize(cfg.dgrid[d]);
                dpme.alpha = cfg.alpha_d; dpme.dispersion = true;
                if (nGrids > 0) dpme.init(cfg.dgrid, nGrids, stream); else { dpme.d.nx = cfg.dgrid[0]; dpme.d.ny = cfg.dgrid[1]; dpme.d.nz = cfg.dgrid[2]; }
            }
            std::vector<int> gs = cfg.shard_count == 1 ? std::vector<int>() : ownedSubsets;
            if (cfg.shard_count == 1) { gs.resize(nsub); std::iota(gs.begin(), gs.end(), 0); }
            gridSubset.upload(gs, stream);
        }
    }
    ~Engine() override {
        (void)hipStreamSynchronize(stream);
        if (getenv("SNB_VERBOSE") && stats.n_rebuilds > 0) fprintf(stderr, "[snb] rebuilds: %lld, of them %lld built beside the steps; %lld side builds discarded\n", (long long)stats.n_rebuilds, sideBuilds, sideDiscarded);
        if (dStepTrace.p && getenv("SNB_STEP_TRACE")) {      // device wall clock (100 MHz) of the last replayed step, relative to the start of its gather pass
            long long t[16] = {0}; (void)hipMemcpy(t, dStepTrace.p, sizeof(t), hipMemcpyDeviceToHost);
            auto us = [&](int k) { return t[k] ? (t[k] - t[0]) / 100.0 : -1.0; };
            fprintf(stderr, "[snb] step trace (us after the gather pass started; -1: not run): pair A %.1f .. %.1f | pair B %.1f .. %.1f | own %.1f | merge %.1f | plane %.1f | mix+z %.1f .. %.1f | interpolation %.1f .. %.1f\n",
                    us(2), us(3), us(4), us(5), us(6), us(7), us(8), us(9), us(10), us(11), us(12));
        }
        if (dPmeTrace.p) {
            long long g[8] = {0}; (void)hipMemcpy(g, dPmeTrace.p, 64, hipMemcpyDeviceToHost);
            if (g[7] > 0 && g[3] > 0) fprintf(stderr, "[snb] merge kernel prologue (busy flags, roots of unity, barrier): %.2f us per busy work-group\n", g[3] / 100.0 / g[7]);
            if (g[7] > 0) fprintf(stderr, "[snb] spreading (busy work-groups; scanning spreader: scan / entries / z FFT + store; merge kernel: sums / strays + z FFT / store): %.2f us, %.2f us, %.2f us per work-group (%lld work-groups)\n", g[4] / 100.0 / g[7], g[5] / 100.0 / g[7], g[6] / 100.0 / g[7], g[7]);
        }
        if (dPmeTrace.p) { long long h[4] = {0, 0, 0, 0}; (void)hipMemcpy(h, dPmeTrace.p, 32, hipMemcpyDeviceToHost); if (h[2] > 0) fprintf(stderr, "[snb] interpolation bricks: mean load %.2f us, mean compute %.2f us per work-group (%lld work-groups)\n", h[0] / 100.0 / h[2], h[1] / 100.0 / h[2], h[2]); }
        destroyGraphs();
        if (streamBuild) { (void)hipStreamSynchronize(streamBuild); (void)hipStreamDestroy(streamBuild); if (evSnap) (void)hipEventDestroy(evSnap); if (evBuilt) (void)hipEventDestroy(evBuilt); if (evFlagsReset) (void)hipEventDestroy(evFlagsReset); }
        for (auto& g : sortGraphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
        sortGraphs.clear();
        for (auto& r : ring) { for (int k = 0; k < 5; k++) (void)hipEventDestroy(r.e[k]); for (int k = 0; k < 16; k++) { (void)hipEventDestroy(r.ks.start[k]); (void)hipEventDestroy(r.ks.stop[k]); } }
        for (int k = 0; k < 3; k++) if (evRebuild[k]) (void)hipEventDestroy(evRebuild[k]);
        for (int k = 0; k < 2; k++) if (evStepDone[k]) (void)hipEventDestroy(evStepDone[k]);
        if (hDispFlags) (void)hipHostFree(hDispFlags);
        if (hNbPub) (void)hipHostFree(hNbPub);
        if (stream2) { (void)hipEventDestroy(evFork); (void)hipEventDestroy(evJoin); (void)hipEventDestroy(evPairA); (void)hipStreamDestroy(stream2); }
        if (ownStream) (void)hipStreamDestroy(stream);
    }
    // Real-space Ewald force factor of the single-precision forces-only pair kernel:
    //   [erfc(ar)/r + 2a/sqrt(pi) exp(-(ar)^2)] / r^2 = 1/r^3 - Bt(r^2),   Bt(r^2) = [erf(ar) - 2ar/sqrt(pi) exp(-(ar)^2)] / r^3,
    // Bt is an entire function of r^2 (Bt(0) = 4a^3/(3 sqrt(pi))), so a polynomial in t = 2 r^2/r2max - 1 (Chebyshev fit over
    // [0, (cutoff+skin)^2], converted to monomials in t, |t| <= 1) reproduces it: degree 11 to 1e-7 of Bt(0) in single precision
    // (12 packed FMAs replace v_exp + v_rcp + the A&S erfc polynomial), degree 16 to ~6e-12 in double (replaces libm erfc + exp).  Absolute force error per pair stays below that of the A&S path at short range and
    // below 2e-6 * qq near the cutoff (tools/ewald_poly_check.py).
    static constexpr int EW_DEG = sizeof(Real) == 4 ? 11 : SNB_EW_DEG_F64;      // 1e-7 resp. ~6e-12 of Bt(0)
    double ewPoly[EW_DEG + 1] = {0}; double ewPolyE[14] = {0}; double dispPoly[21] = {0}; double ewR2Max = 1;
    void buildEwaldPoly() {
        const double rmax = cfg.cutoff + std::max(cfg.neighbor_padding, 0.0) + 0.02, a = cfg.alpha;
        ewR2Max = rmax * rmax;
        auto bt = [&](double r2) {
            const double r = std::sqrt(r2), z = a * r;
            if (z < 2e-2) return a * a * a * (4.0 / (3.0 * std::sqrt(SNB_PI))) * (1.0 - 0.6 * z * z + (3.0 / 14.0) * z * z * z * z - (1.0 / 18.0) * z * z * z * z * z * z);
            return (std::erf(z) - 2.0 * z / std::sqrt(SNB_PI) * std::exp(-z * z)) / (r2 * r);
        };
        chebyshevToMonomial(bt, EW_DEG, ewPoly);
        // pair energies of the packed kernel: Et(r^2) = erf(a r)/r, entire in r^2 as well, Et(0) = 2a/sqrt(pi)
        auto et = [&](double r2) {
            const double r = std::sqrt(r2), z = a * r;
            if (z < 1e-3) return a * (2.0 / std::sqrt(SNB_PI)) * (1.0 - z * z / 3.0 + z * z * z * z / 10.0);
            return std::erf(z) / r;
        };
        chebyshevToMonomial(et, 13, ewPolyE);
        if (cfg.method == SNB_LJPME) {      // dispersion force factor of the double-precision pair kernel
            const double ad = cfg.alpha_d;
            auto gd = [&](double r2) {
                const double x = ad * ad * r2;
                double sum = 0, term = 1.0 / 24.0;      // sum_k x^k / (k+4)!
                for (int k = 0; k < 60; k++) { sum += term; term *= x / (k + 5); if (term < 1e-18 * sum) break; }
                return std::pow(ad, 8.0) * std::exp(-x) * sum;
            };
            chebyshevToMonomial(gd, SNB_DISP_DEG_F64, dispPoly);
        }
    }
    // Chebyshev interpolant of f(r^2) over [0, ewR2Max] at 96 nodes, truncated at `deg`, as monomial coefficients in t = 2 r^2/ewR2Max - 1
    template <typename Fn> void chebyshevToMonomial(Fn f, int deg, double* out) {
        const int M = 96;
        std::vector<double> c(deg + 1);
        for (int k = 0; k <= deg; k++) {
            double acc = 0;
            for (int j = 0; j < M; j++) { const double x = std::cos(SNB_PI * (j + 0.5) / M); acc += f(0.5 * (x + 1.0) * ewR2Max) * std::cos(SNB_PI * k * (j + 0.5) / M); }
            c[k] = acc * 2.0 / M;
        }
        c[0] *= 0.5;
        // T_0 = 1, T_1 = x, T_{k+1} = 2 x T_k - T_{k-1}
        std::vector<double> Tm(deg + 1, 0.0), Tc(deg + 1, 0.0), Tn(deg + 1, 0.0);
        Tm[0] = 1; if (deg >= 1) Tc[1] = 1;
        for (int i = 0; i <= deg; i++) out[i] = 0;
        out[0] += c[0];
        if (deg >= 1) for (int i = 0; i <= deg; i++) out[i] += c[1] * Tc[i];
        for (int k = 2; k <= deg; k++) {
            for (int i = 0; i <= deg; i++) Tn[i] = (i > 0 ? 2.0 * Tc[i - 1] : 0.0) - Tm[i];
            for (int i = 0; i <= deg; i++) { out[i] += c[k] * Tn[i]; Tm[i] = Tc[i]; Tc[i] = Tn[i]; }
        }
    }
    bool isPme() const { return cfg.method == SNB_PME || cfg.method == SNB_LJPME; }
    bool isPeriodic() const { return cfg.method >= SNB_CutoffPeriodic; }

    void setParticles(const double* q, const double* sg, const double* ep, const int32_t* sub) override {
        bool subsetsChanged = !haveParticles;
        for (int i = 0; i < N; i++) {
            if (sub[i] < 0 || sub[i] >= nsub) throw HipError{"subset out of range"};
            if (subset[i] != sub[i]) subsetsChanged = true;      // subsets decide the sorted order and the block layout
        }
        charge.assign(q, q + N); sigma.assign(sg, sg + N); epsilon.assign(ep, ep + N); subset.assign(sub, sub + N);
        basePDirty = true;
        haveParticles = true;
        // new charges / sigmas / epsilons alone (parameter offsets, updateParametersInContext) do not touch the neighbour structure:
        // they are refreshed in place (refreshValues) instead of going through a rebuild
        if (subsetsChanged) { needRebuild = true; paramsDirty = true; staticDirty = true; } else valuesDirty = true;
    }
    void setExceptions(int32_t m, const int32_t* pairs, const double* qq, const double* sg, const double* ep, const int32_t* f14) override {
        for (int k = 0; k < m; k++)
            if (pairs[2 * k] < 0 || pairs[2 * k] >= N || pairs[2 * k + 1] < 0 || pairs[2 * k + 1] >= N || pairs[2 * k] == pairs[2 * k + 1]) throw HipError{"exception particle index out of range"};
        for (auto& o : offE) if (o.target >= m) throw HipError{"snb_set_exceptions: a parameter offset refers to an exception past the new count (clear or re-send the offsets first)"};
        std::vector<int32_t> np(pairs, pairs + 2 * (size_t)m);
        const bool pairsChanged = !haveExceptions || np != excPairs;      // the exclusion masks live in the tiles
        excPairs.swap(np); excQQ.assign(qq, qq + m); excSigma.assign(sg, sg + m); excEps.assign(ep, ep + m);
        if (f14) excForce14.assign(f14, f14 + m); else excForce14.assign(m, 0);
        haveExceptions = true;
        baseEDirty = true;
        if (pairsChanged) { needRebuild = true; paramsDirty = true; staticDirty = true; } else excValuesDirty = true;
    }
    void setParameterOffsets(int nGlob, int nP, const int32_t* pIdx, const int32_t* pGlob, const double* pDelta, int nE, const int32_t* eIdx, const int32_t* eGlob, const double* eDelta) override {
        if (nGlob < 0 || nP < 0 || nE < 0) throw HipError{"snb_set_parameter_offsets: negative count"};
        const int m = (int)(excPairs.size() / 2);
        for (int k = 0; k < nP; k++) if (pIdx[k] < 0 || pIdx[k] >= N || pGlob[k] < 0 || pGlob[k] >= nGlob) throw HipError{"snb_set_parameter_offsets: particle offset out of range"};
        for (int k = 0; k < nE; k++) if (eIdx[k] < 0 || eIdx[k] >= m || eGlob[k] < 0 || eGlob[k] >= nGlob) throw HipError{"snb_set_parameter_offsets: exception offset out of range (set the exceptions first)"};
        std::vector<Offset> nP_(nP), nE_(nE);
        for (int k = 0; k < nP; k++) nP_[k] = Offset{pIdx[k], pGlob[k], {pDelta[3 * k], pDelta[3 * k + 1], pDelta[3 * k + 2]}};
        for (int k = 0; k < nE; k++) nE_[k] = Offset{eIdx[k], eGlob[k], {eDelta[3 * k], eDelta[3 * k + 1], eDelta[3 * k + 2]}};
        auto same = [](const std::vector<Offset>& a, const std::vector<Offset>& b, bool structureOnly) {
            if (a.size() != b.size()) return false;
            for (size_t k = 0; k < a.size(); k++) {
                if (a[k].target != b[k].target || a[k].global != b[k].global) return false;
                if (!structureOnly && (a[k].d[0] != b[k].d[0] || a[k].d[1] != b[k].d[1] || a[k].d[2] != b[k].d[2])) return false;
            }
            return true;
        };
        if (nGlob == nGlobals && same(offP, nP_, false) && same(offE, nE_, false)) return;      // unchanged (copyParametersToContext re-sends everything)
        // an exception that carries an offset is a 1-4 interaction whatever its base values (Q6): a different set of them changes the 1-4 list
        const bool structure = nGlob != nGlobals || !same(offE, nE_, true) || !same(offP, nP_, true);
        offP.swap(nP_); offE.swap(nE_);
        if (nGlob != nGlobals) { nGlobals = nGlob; gValues.assign(nGlob, 0.0); globalsDirty = true; }
        offsetsDirty = true;
        if (structure) { needRebuild = true; paramsDirty = true; staticDirty = true; } else { valuesDirty = true; excValuesDirty = true; }
    }
    void setGlobalParameters(int n, const double* v) override {
        if (n != nGlobals) throw HipError{"snb_set_global_parameters: count differs from snb_set_parameter_offsets"};
        bool changed = false;
        for (int k = 0; k < n; k++) if (gValues[k] != v[k]) { gValues[k] = v[k]; changed = true; }
        if (!changed) return;
        globalsDirty = true;
        if (!offP.empty()) valuesDirty = true;
        if (!offE.empty()) excValuesDirty = true;
    }
    void setLambdas(const double* l) override {
        lambdas.assign(l, l + (size_t)S * 2);
        hLambdas.assign(lambdas.begin(), lambdas.end());
        pinned.upload(dLambdas, hLambdas, stream);      // (pinned slot per in-flight update: no synchronisation per lambda change, and the next call may overwrite hLambdas)
    }
    void setEnergySlices(const int32_t* m) override {
        for (int i = 0; i < S; i++) sliceNeedSel[i] = m[i] != 0;
        pinned.upload(dSliceNeedSel, sliceNeedSel, stream);      // (same device buffer, so captured graphs stay valid)
    }
    void setDispersion(const double* c) override {
        if (c) dispCoef.assign(c, c + S); else dispCoef.assign(S, 0.0);
        pinned.upload(dDispCoef, dispCoef, stream);      // read by the last kernel of an energy step
        dropGraph();
    }
    void setBox(const double* b) override {
        if (b[1] != 0 || b[2] != 0 || b[5] != 0) throw HipError{"box vectors must be in reduced (lower triangular) form"};
        bool changed = !haveBox;
        for (int i = 0; i < 9; i++) { if (box[i] != b[i]) changed = true; box[i] = b[i]; }
        haveBox = true;
        if (changed) needRebuild = true;
    }
    void setPositions(const void* pos, int isDevice, int isDouble, int stride4) override {
        posIsDouble = isDouble; posStride4 = stride4;
        const size_t bytes = (size_t)N * (stride4 ? 4 : 3) * (isDouble ? 8 : 4);
        if (isDevice) devUserPos = pos;
        else {
            ownedPos.resize(bytes);
            HIPCHECK(hipMemcpyAsync(ownedPos.p, pos, bytes, hipMemcpyHostToDevice, stream));
            HIPCHECK(hipStreamSynchronize(stream));   // the caller's array may be freed on return
            devUserPos = ownedPos.p;
        }
        havePositions = true;
    }
    void requestRebuild() override { needRebuild = true; }
    void sync() override { HIPCHECK(hipStreamSynchronize(stream)); }

    // ------------------------------------------------------------------------------------------
    // Neighbour structure: sort, blocks, per-atom gathered j-tiles, exclusion masks (host, v1).
    // ------------------------------------------------------------------------------------------
    static inline bool owns(int I, int J) { return ((I + J) & 1) ? (I > J) : (I < J); }

    void rebuild() {
        static const bool verbose = getenv("SNB_VERBOSE") != nullptr;
        const auto tr0 = std::chrono::steady_clock::now();
        if (verbose) { HIPCHECK(hipStreamSynchronize(stream)); }      // (diagnostic only: separates the wait for the queued steps from the rebuild's own time)
        const auto tr1 = std::chrono::steady_clock::now();
        dropGraph();   // buffers may move and every kernel argument block changes
        if (staticDirty) uploadStatic();
        gpuBuilt = false;
        if (cfg.host_neighbor_build || !gpuRebuild()) hostRebuild();
        if (verbose) fprintf(stderr, "[snb] rebuild: waited %.0f us for the queued steps, then %.0f us (host) for the build itself; device time %.0f us\n",
                             std::chrono::duration<double, std::micro>(tr1 - tr0).count(), std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tr1).count(), stats.last_rebuild_ms * 1e3);
        pmeCells.resize(Npad);   // per-slot scratch is sized here: nothing may allocate while a step is being captured into a graph
        if (isPme()) { if (!dStrayCount.p) { dStrayCount.resize(2); HIPCHECK(hipMemsetAsync(dStrayCount.p, 0, 2 * sizeof(int), stream)); } planOwnSpread(pme); if (cfg.method == SNB_LJPME) planOwnSpread(dpme); planPlaneTable(pme); if (cfg.method == SNB_LJPME) planPlaneTable(dpme); }
        posRef.resize(Npad);
        HIPCHECK(hipMemcpyAsync(posRef.p, posq.p, sizeof(T4) * (size_t)Npad, hipMemcpyDeviceToDevice, stream));
        if (hDispFlags[1]) listOverruns++;      // an atom had moved more than skin/2 before this rebuild came
        hDispFlags[0] = hDispFlags[1] = 0;
        flagsResetPending = false;      // (the queue has been drained: a reset enqueued at an earlier exchange has run)
    }

    void hostRebuild() {
        auto t0 = std::chrono::steady_clock::now();
        // 1. host copy of the user positions
        std::vector<double> hp((size_t)N * 3);
        {
            const int st = posStride4 ? 4 : 3;
            if (posIsDouble) {
                std::vector<double> tmp((size_t)N * st);
                HIPCHECK(hipMemcpy(tmp.data(), devUserPos, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost));
                for (int i = 0; i < N; i++) for (int d = 0; d < 3; d++) hp[3 * (size_t)i + d] = tmp[(size_t)i * st + d];
            } else {
                std::vector<float> tmp((size_t)N * st);
                HIPCHECK(hipMemcpy(tmp.data(), devUserPos, sizeof(float) * tmp.size(), hipMemcpyDeviceToHost));
                for (int i = 0; i < N; i++) for (int d = 0; d < 3; d++) hp[3 * (size_t)i + d] = tmp[(size_t)i * st + d];
            }
        }
        const bool periodic = isPeriodic();
        const double R = (cfg.method == SNB_NoCutoff) ? 1e300 : cfg.cutoff + cfg.neighbor_padding;
        const bool rect = box[3] == 0 && box[6] == 0 && box[7] == 0;
        // 2. wrap into the primary cell (fractional coordinates; lower-triangular box)
        std::vector<double> wp(hp), off((size_t)N * 3, 0.0);
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        if (periodic) {
            for (int i = 0; i < N; i++) {
                double* x = &wp[3 * (size_t)i];
                double s2 = std::floor(x[2] / box[8]); x[0] -= s2 * box[6]; x[1] -= s2 * box[7]; x[2] -= s2 * box[8];
                double s1 = std::floor(x[1] / box[4]); x[0] -= s1 * box[3]; x[1] -= s1 * box[4];
                double s0 = std::floor(x[0] / box[0]); x[0] -= s0 * box[0];
                for (int d = 0; d < 3; d++) off[3 * (size_t)i + d] = x[d] - hp[3 * (size_t)i + d];
            }
            lo[0] = lo[1] = lo[2] = 0; hi[0] = box[0]; hi[1] = box[4]; hi[2] = box[8];
            if (!rect) { lo[0] = std::min(0.0, box[3]) + std::min(0.0, box[6]); hi[0] = box[0] + std::max(0.0, box[3]) + std::max(0.0, box[6]); lo[1] = std::min(0.0, box[7]); hi[1] = box[4] + std::max(0.0, box[7]); }
        } else {
            for (int i = 0; i < N; i++) for (int d = 0; d < 3; d++) { lo[d] = std::min(lo[d], wp[3 * (size_t)i + d]); hi[d] = std::max(hi[d], wp[3 * (size_t)i + d]); }
            if (N == 0) { lo[0] = lo[1] = lo[2] = 0; hi[0] = hi[1] = hi[2] = 1; }
            for (int d = 0; d < 3; d++) { hi[d] += 1e-6 + 1e-9 * std::fabs(hi[d]); }
        }
        double ext[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        for (int d = 0; d < 3; d++) if (!(ext[d] > 1e-9)) ext[d] = 1e-9;
        // 3. sort: (subset, serpentine xy column, z up/down)
        const double volume = ext[0] * ext[1] * ext[2];
        const double aTarget = std::cbrt(32.0 * volume / std::max(N, 1));
        int ncx = std::max(1, std::min(2048, (int)std::lround(ext[0] / aTarget)));
        int ncy = std::max(1, std::min(2048, (int)std::lround(ext[1] / aTarget)));
        // With PME on a rectangular box the sort columns are made commensurate with the mesh (whole grid cells, >= 5 wide),
        // so the same sorted order feeds the brick-spreading kernel (pme.hip k_spreadBrick).
        colCells[0] = colCells[1] = 0;
        if (isPme() && rect && nGrids > 0) {
            auto pick = [&](int n, double L) {
                int best = 0; double bestErr = 1e300;
                for (int d = 5; d <= 16 && d <= n; d++) if (n % d == 0) { double e = std::fabs(d * L / n - aTarget); if (e < bestErr) { bestErr = e; best = d; } }
                return best;
            };
            const int px = pick(pme.d.nx, box[0]), py = pick(pme.d.ny, box[4]);
            const size_t brickBytes = sizeof(double) * (size_t)px * py * pme.d.nz;
            if (px > 0 && py > 0 && brickBytes <= 60 * 1024 && std::max(pme.d.nx, std::max(pme.d.ny, pme.d.nz)) < 1024) {      // (the packed mesh cell of an atom holds 10 bits per axis)
            colCells[0] = px; colCells[1] = py; ncx = pme.d.nx / px; ncy = pme.d.ny / py; }
        }
        std::vector<uint64_t> key(N);
        std::vector<int> colOfAtom(N);
        for (int i = 0; i < N; i++) {
            const double* x = &wp[3 * (size_t)i];
            int cx = std::min(ncx - 1, std::max(0, (int)((x[0] - lo[0]) / ext[0] * ncx)));
            int cy = std::min(ncy - 1, std::max(0, (int)((x[1] - lo[1]) / ext[1] * ncy)));
            colOfAtom[i] = cx * ncy + cy;
            int col = cx * ncy + ((cx & 1) ? (ncy - 1 - cy) : cy);
            double zf = std::min(1.0, std::max(0.0, (x[2] - lo[2]) / ext[2]));
            if (col & 1) zf = 1.0 - zf;
            uint64_t zq = (uint64_t)(zf * 1048575.0);
            key[i] = ((uint64_t)subset[i] << 44) | ((uint64_t)col << 20) | zq;
        }
        std::vector<int> order(N);
        std::iota(order.begin(), order.end(), 0);
        std::sort(order.begin(), order.end(), [&](int a, int b) { return key[a] != key[b] ? key[a] < key[b] : a < b; });
        sortedToUser.clear(); userToSorted.assign(N, -1);
        std::vector<int> blkSubset;
        {
            size_t k = 0;
            for (int s = 0; s < nsub; s++) {
                size_t start = sortedToUser.size();
                while (k < (size_t)N && subset[order[k]] == s) { userToSorted[order[k]] = (int)sortedToUser.size(); sortedToUser.push_back(order[k]); k++; }
                while ((sortedToUser.size() - start) % 32) sortedToUser.push_back(-1);
                for (size_t b = start / 32; b < sortedToUser.size() / 32; b++) blkSubset.push_back(s);
            }
        }
        Npad = (int)sortedToUser.size(); numBlocks = Npad / 32;
        if (colCells[0] > 0) {   // sorted range of every (subset, column): atoms of one column are contiguous in the sorted order
            const int ncol = ncx * ncy;
            std::vector<int2> hRange((size_t)nsub * ncol, make_int2(0, 0));
            for (int s = 0; s < Npad; s++) {
                const int u = sortedToUser[s]; if (u < 0) continue;
                int2& rg = hRange[(size_t)subset[u] * ncol + colOfAtom[u]];
                if (rg.y == 0) rg.x = s;
                rg.y = s + 1;
            }
            colRange.upload(hRange, stream);
        }
        if (Npad >= (1 << SNB_JSHIFT_BITS) - 1) throw HipError{"too many atoms for the 25-bit tile index"};
        // 4. sorted parameter arrays (effective parameter values: formed on the device, fetched for this host-side path)
        std::vector<Real> effQ(N); std::vector<T2> effSE(N);
        if (N > 0) { HIPCHECK(hipMemcpy(effQ.data(), dUCharge.p, sizeof(Real) * N, hipMemcpyDeviceToHost)); HIPCHECK(hipMemcpy(effSE.data(), dUSigEps.p, sizeof(T2) * N, hipMemcpyDeviceToHost)); }
        std::vector<T4> hPosq(Npad); std::vector<T2> hSigeps(Npad); std::vector<Real> hOff((size_t)Npad * 3, Real(0));
        std::vector<int> hAtomSubset(Npad, -1), hAtomGrid(Npad, -1);
        std::vector<int> slotOfSubset(nsub, -1);
        if (cfg.shard_count == 1) std::iota(slotOfSubset.begin(), slotOfSubset.end(), 0);
        else for (size_t g = 0; g < ownedSubsets.size(); g++) slotOfSubset[ownedSubsets[g]] = (int)g;
        for (int s = 0; s < Npad; s++) {
            int u = sortedToUser[s];
            if (u >= 0) {
                hPosq[s].x = (Real)wp[3 * (size_t)u]; hPosq[s].y = (Real)wp[3 * (size_t)u + 1]; hPosq[s].z = (Real)wp[3 * (size_t)u + 2]; hPosq[s].w = effQ[u];
                hSigeps[s] = effSE[u];
                for (int d = 0; d < 3; d++) hOff[3 * (size_t)s + d] = (Real)off[3 * (size_t)u + d];
                hAtomSubset[s] = subset[u]; hAtomGrid[s] = slotOfSubset[subset[u]];
            } else {   // parked padding atom: zero parameters, far away, distinct
                hPosq[s].x = (Real)(1e9 + 1e6 * (s & 4095)); hPosq[s].y = (Real)2e9; hPosq[s].z = (Real)-3e9; hPosq[s].w = 0;
                hSigeps[s].x = 0; hSigeps[s].y = 0;
            }
        }
        // 5. block bounding boxes (real atoms only)
        std::vector<double> bc((size_t)numBlocks * 3, 0.0), bh((size_t)numBlocks * 3, 0.0);
        double maxFullExt[3] = {0, 0, 0};
        for (int b = 0; b < numBlocks; b++) {
            double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
            for (int k = 0; k < 32; k++) {
                int u = sortedToUser[b * 32 + k]; if (u < 0) continue;
                for (int d = 0; d < 3; d++) { mn[d] = std::min(mn[d], wp[3 * (size_t)u + d]); mx[d] = std::max(mx[d], wp[3 * (size_t)u + d]); }
            }
            for (int d = 0; d < 3; d++) { bc[3 * (size_t)b + d] = 0.5 * (mn[d] + mx[d]); bh[3 * (size_t)b + d] = 0.5 * (mx[d] - mn[d]); maxFullExt[d] = std::max(maxFullExt[d], mx[d] - mn[d]); }
        }
        // 6. tiles
        wrapMode = false;
        bool allPairs = (cfg.method == SNB_NoCutoff);
        if (periodic) {
            if (!rect) { wrapMode = true; allPairs = true; }
            else for (int d = 0; d < 3; d++) if (!(maxFullExt[d] + 2 * R < box[4 * d])) { wrapMode = true; allPairs = true; }
        }
        std::vector<int> hTileJ; std::vector<int4> hTileInfo; std::vector<int2> hBlockTiles(numBlocks);
        std::vector<unsigned> hMasks;
        // exclusion CSR over user indices
        std::vector<int> exStart(N + 1, 0), exList;
        {
            const size_t m = excPairs.size() / 2;
            for (size_t k = 0; k < m; k++) { exStart[excPairs[2 * k] + 1]++; exStart[excPairs[2 * k + 1] + 1]++; }
            for (int i = 0; i < N; i++) exStart[i + 1] += exStart[i];
            exList.resize(exStart[N]);
            std::vector<int> fill(N, 0);
            for (size_t k = 0; k < m; k++) { int a = excPairs[2 * k], b = excPairs[2 * k + 1]; exList[exStart[a] + fill[a]++] = b; exList[exStart[b] + fill[b]++] = a; }
        }
        // cell grid over sorted real atoms (rectangular domains only; allPairs mode does not need it)
        int nc[3] = {1, 1, 1}; std::vector<int> cellStart, cellAtoms;
        double csz[3] = {ext[0], ext[1], ext[2]};
        if (!allPairs) {
            const double target = std::max(aTarget, R / 3.0);
            for (int d = 0; d < 3; d++) { nc[d] = std::max(1, std::min(512, (int)(ext[d] / target))); csz[d] = ext[d] / nc[d]; }
            const size_t ncell = (size_t)nc[0] * nc[1] * nc[2];
            cellStart.assign(ncell + 1, 0);
            std::vector<int> cellOf(Npad, -1);
            for (int s = 0; s < Npad; s++) {
                int u = sortedToUser[s]; if (u < 0) continue;
                int c[3];
                for (int d = 0; d < 3; d++) c[d] = std::min(nc[d] - 1, std::max(0, (int)((wp[3 * (size_t)u + d] - lo[d]) / csz[d])));
                cellOf[s] = (c[0] * nc[1] + c[1]) * nc[2] + c[2];
                cellStart[cellOf[s] + 1]++;
            }
            for (size_t c = 0; c < ncell; c++) cellStart[c + 1] += cellStart[c];
            cellAtoms.resize(cellStart[ncell]);
            std::vector<int> fill(ncell, 0);
            for (int s = 0; s < Npad; s++) if (cellOf[s] >= 0) cellAtoms[cellStart[cellOf[s]] + fill[cellOf[s]]++] = s;
        }
        std::vector<int> slotOf(Npad, -1);            // sorted j index -> position in this block's candidate list
        std::vector<std::pair<int, int>> cand;        // (key = sj<<27 | index, code)
        std::vector<int> tileMask;                    // per tile of the current block: mask index or -1
        numMaskTiles = 0;
        for (int I = 0; I < numBlocks; I++) {
            cand.clear();
            const double* c = &bc[3 * (size_t)I]; const double* h = &bh[3 * (size_t)I];
            bool emptyBlock = true;
            for (int k = 0; k < 32; k++) if (sortedToUser[I * 32 + k] >= 0) emptyBlock = false;
            if (!emptyBlock) {
                if (allPairs) {
                    for (int J = 0; J < numBlocks; J++) {
                        if (J == I || !owns(I, J)) continue;
                        for (int k = 0; k < 32; k++) if (sortedToUser[J * 32 + k] >= 0) cand.push_back({J * 32 + k, SNB_JCODE_CENTER});
                    }
                } else {
                    int cmin[3], cmax[3];
                    for (int d = 0; d < 3; d++) {
                        cmin[d] = (int)std::floor((c[d] - h[d] - R - lo[d]) / csz[d]);
                        cmax[d] = (int)std::floor((c[d] + h[d] + R - lo[d]) / csz[d]);
                        if (!periodic) { cmin[d] = std::max(cmin[d], 0); cmax[d] = std::min(cmax[d], nc[d] - 1); }
                    }
                    for (int ix = cmin[0]; ix <= cmax[0]; ix++) for (int iy = cmin[1]; iy <= cmax[1]; iy++) for (int iz = cmin[2]; iz <= cmax[2]; iz++) {
                        int cc[3] = {ix, iy, iz}, img[3] = {0, 0, 0};
                        for (int d = 0; d < 3; d++) { img[d] = (int)std::floor((double)cc[d] / nc[d]); cc[d] -= img[d] * nc[d]; }
                        if (std::abs(img[0]) > 1 || std::abs(img[1]) > 1 || std::abs(img[2]) > 1) continue;
                        const double sh[3] = {img[0] * box[0], img[1] * box[4], img[2] * box[8]};
                        const int code = (img[0] + 2) * 25 + (img[1] + 2) * 5 + (img[2] + 2);
                        const int cell = (cc[0] * nc[1] + cc[1]) * nc[2] + cc[2];
                        for (int a = cellStart[cell]; a < cellStart[cell + 1]; a++) {
                            const int sj = cellAtoms[a]; const int J = sj >> 5;
                            if (J == I || !owns(I, J)) continue;
                            const int u = sortedToUser[sj];
                            double d2 = 0;
                            for (int d = 0; d < 3; d++) { double dd = std::fabs(wp[3 * (size_t)u + d] + sh[d] - c[d]) - h[d]; if (dd > 0) d2 += dd * dd; }
                            if (d2 < R * R) cand.push_back({sj, code});
                        }
                    }
                }
            }
            // group by j subset, ascending index inside a group
            std::sort(cand.begin(), cand.end(), [&](const std::pair<int, int>& a, const std::pair<int, int>& b) {
                int sa = blkSubset[a.first >> 5], sb = blkSubset[b.first >> 5];
                return sa != sb ? sa < sb : a.first < b.first;
            });
            const int firstTile = (int)hTileInfo.size();
            // diagonal tile first
            tileMask.clear();
            {
                for (int k = 0; k < 32; k++) hTileJ.push_back((sortedToUser[I * 32 + k] >= 0) ? ((I * 32 + k) | (SNB_JCODE_CENTER << SNB_JSHIFT_BITS)) : -1);
                int mi = (int)(hMasks.size() / 32);
                hMasks.resize(hMasks.size() + 32, 0u);
                for (int i = 0; i < 32; i++) { unsigned m = 0; for (int j = 0; j <= i; j++) m |= 1u << j; hMasks[(size_t)mi * 32 + i] = m; }   // keep j > i only
                hTileInfo.push_back(make_int4(blkSubset[I] * (blkSubset[I] + 3) / 2, mi, blkSubset[I], 0));      // (slice, mask, j subset)
                tileMask.push_back(mi);
                for (int k = 0; k < 32; k++) slotOf[I * 32 + k] = k;    // slots 0..31 = diagonal tile
            }
            size_t pos = 0;
            while (pos < cand.size()) {
                const int sjSub = blkSubset[cand[pos].first >> 5];
                int cnt = 0;
                const int tIndex = (int)hTileInfo.size() - firstTile;
                while (pos < cand.size() && cnt < 32 && blkSubset[cand[pos].first >> 5] == sjSub) {
                    hTileJ.push_back(cand[pos].first | (cand[pos].second << SNB_JSHIFT_BITS));
                    slotOf[cand[pos].first] = tIndex * 32 + cnt;
                    cnt++; pos++;
                }
                for (; cnt < 32; cnt++) hTileJ.push_back(-1);
                { const int a = std::max(blkSubset[I], sjSub), b = std::min(blkSubset[I], sjSub); hTileInfo.push_back(make_int4(a * (a + 1) / 2 + b, -1, sjSub, 0)); }
                tileMask.push_back(-1);
            }
            // exclusion masks
            for (int k = 0; k < 32; k++) {
                int u = sortedToUser[I * 32 + k]; if (u < 0) continue;
                for (int e = exStart[u]; e < exStart[u + 1]; e++) {
                    const int sj = userToSorted[exList[e]];
                    const int sl = slotOf[sj];
                    if (sl < 0) continue;
                    const int t = sl >> 5, bit = sl & 31;
                    if (tileMask[t] < 0) { tileMask[t] = (int)(hMasks.size() / 32); hMasks.resize(hMasks.size() + 32, 0u); hTileInfo[firstTile + t].y = tileMask[t]; }
                    hMasks[(size_t)tileMask[t] * 32 + k] |= 1u << bit;
                }
            }
            // padding slots (partially filled tiles, padded i rows) are masked out as well, so that parked padding
            // coordinates can never contribute (in the per-pair-wrap variant they would be folded back into the box)
            {
                unsigned iPadRows = 0;
                for (int k = 0; k < 32; k++) if (sortedToUser[I * 32 + k] < 0) iPadRows |= 1u << k;
                for (int t = 0; t < (int)tileMask.size(); t++) {
                    unsigned jPad = 0;
                    for (int k = 0; k < 32; k++) if (hTileJ[(size_t)(firstTile + t) * 32 + k] == -1) jPad |= 1u << k;
                    if (!jPad && !iPadRows) continue;
                    if (tileMask[t] < 0) { tileMask[t] = (int)(hMasks.size() / 32); hMasks.resize(hMasks.size() + 32, 0u); hTileInfo[firstTile + t].y = tileMask[t]; }
                    for (int k = 0; k < 32; k++) hMasks[(size_t)tileMask[t] * 32 + k] |= ((iPadRows >> k) & 1u) ? 0xFFFFFFFFu : jPad;
                }
            }
            for (int t = 0; t < (int)tileMask.size(); t++) if (tileMask[t] >= 0) numMaskTiles++;
            // reset the scratch map
            for (int k = 0; k < 32; k++) slotOf[I * 32 + k] = -1;
            for (auto& cd : cand) slotOf[cd.first] = -1;
            hBlockTiles[I] = make_int2(firstTile, (int)hTileInfo.size() - firstTile);
            if (emptyBlock) { hBlockTiles[I].y = 0; }
        }
        numTiles = (int64_t)hTileInfo.size();
        // work items: runs of <= 8 tiles of one i-block (fine grain => several rounds of waves per CU, small tail)
        std::vector<int4> hWork;
        const int CH = 8;
        // sharded engines keep the work items of the i-blocks they own (block index % shard_count == shard_rank): a rule every rank
        // evaluates identically, whatever order its own builder emitted the items in
        shardTiles = 0;
        for (int b = 0; b < numBlocks; b++) {
            if (!ownsBlock(b)) continue;
            shardTiles += hBlockTiles[b].y;
            for (int o = 0; o < hBlockTiles[b].y; o += CH) hWork.push_back(make_int4(b, hBlockTiles[b].x + o, std::min(CH, hBlockTiles[b].y - o), blkSubset[b]));
        }
        std::stable_sort(hWork.begin(), hWork.end(), [&](const int4& a, const int4& b) { return a.z > b.z; });
        numWorkItems = (int)hWork.size();
        // 7. upload
        posq.upload(hPosq, stream); sigeps.upload(hSigeps, stream); imageOffset.upload(hOff, stream);
        dSortedToUser.upload(sortedToUser, stream); dUserToSorted.upload(userToSorted, stream);
        blockSubset.upload(blkSubset, stream); workItems.upload(hWork, stream);
        tileJ.upload(hTileJ, stream); tileInfo.upload(hTileInfo, stream); masks.upload(hMasks, stream);
        atomSubset.upload(hAtomSubset, stream); atomGrid.upload(hAtomGrid, stream);
        layoutForces();
        HIPCHECK(hipStreamSynchronize(stream));
        needRebuild = false; paramsDirty = false; stepsSinceRebuild = 0;
        stats.n_rebuilds++; stats.n_host_rebuilds++;
        stats.last_rebuild_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }

    // Static (sort-independent) device data: 1-4 list and exclusion CSR in USER indices (Q6:
    // ReferenceNonbondedSlicingKernels.cpp:99-112, 129-131), per-atom parameters in user order, the padded subset layout.
    // 1-4 list in user order: the exceptions with non-zero base parameters, or flagged by the caller, or carrying a parameter offset (Q6).
    // Structure only (pairs, slices, base values, offsets in CSR per 1-4 entry); the values are formed on the device (syncParameters).
    void upload14() {
        const size_t m = excPairs.size() / 2;
        // captured step graphs bake in the 1-4 count, the list pointers and the grid of the list blocks: same pairs with a different set of
        // non-zero / flagged exceptions changes all three, so the graphs go whenever any of them moved
        const int oldN14 = n14; const void* const oldPairs = pairs14.p; const void* const oldParams = params14.p;
        std::vector<char> hasOffset(m, 0);
        for (auto& o : offE) hasOffset[o.target] = 1;
        std::vector<int2> p14; std::vector<double> b14; std::vector<int> sl14, where(m, -1);
        auto sl = [](int a, int b) { return a > b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a; };
        for (size_t k = 0; k < m; k++) {
            const int a = excPairs[2 * k], b = excPairs[2 * k + 1];
            if (excQQ[k] != 0.0 || excEps[k] != 0.0 || excForce14[k] || hasOffset[k]) {
                where[k] = (int)p14.size();
                p14.push_back(make_int2(a, b));
                b14.push_back(excQQ[k]); b14.push_back(excSigma[k]); b14.push_back(excEps[k]);
                sl14.push_back(sl(subset[a], subset[b]));
            }
        }
        n14 = (int)p14.size(); nExcl = (int)m;
        std::vector<int> start(n14 + 1, 0), glob(offE.size()); std::vector<double> delta(3 * offE.size());
        for (auto& o : offE) start[where[o.target] + 1]++;
        for (int k = 0; k < n14; k++) start[k + 1] += start[k];
        { std::vector<int> fill(n14, 0); for (auto& o : offE) { const int e = where[o.target], at = start[e] + fill[e]++; glob[at] = o.global; for (int d = 0; d < 3; d++) delta[3 * (size_t)at + d] = o.d[d]; } }
        pairs14.upload(p14, stream); dBase14.upload(b14, stream); dSlice14.upload(sl14, stream);
        dOff14Start.upload(start, stream); dOff14Global.upload(glob, stream); dOff14Delta.upload(delta, stream);
        params14.resize(n14);
        stats.n_14 = n14; stats.n_exclusions = nExcl;
        HIPCHECK(hipStreamSynchronize(stream));      // (host staging vectors go out of scope)
        baseEDirty = false;
        if (n14 != oldN14 || pairs14.p != oldPairs || params14.p != oldParams) dropGraph();
    }
    void uploadParticleBase() {
        std::vector<double> hb(3 * (size_t)N);
        for (int i = 0; i < N; i++) { hb[3 * (size_t)i] = charge[i]; hb[3 * (size_t)i + 1] = sigma[i]; hb[3 * (size_t)i + 2] = epsilon[i]; }
        std::vector<int> start((size_t)N + 1, 0), glob(offP.size()); std::vector<double> delta(3 * offP.size());
        for (auto& o : offP) start[o.target + 1]++;
        for (int i = 0; i < N; i++) start[i + 1] += start[i];
        { std::vector<int> fill(N, 0); for (auto& o : offP) { const int at = start[o.target] + fill[o.target]++; glob[at] = o.global; for (int d = 0; d < 3; d++) delta[3 * (size_t)at + d] = o.d[d]; } }
        dBaseP.upload(hb, stream); dOffPStart.upload(start, stream); dOffPGlobal.upload(glob, stream); dOffPDelta.upload(delta, stream);
        dUCharge.resize(N); dUSigEps.resize(N); dParamSums.resize((3 * (size_t)nsub + 1) * (1 + SNB_PARAM_SUM_ROWS)); dFixScale.resize(4);
        HIPCHECK(hipStreamSynchronize(stream));
        basePDirty = false;
    }
    // Effective parameters on the device, from whatever changed: base values / offsets (re-uploaded), global parameter values (a few
    // doubles).  Enqueued on the engine's stream ahead of the step; nothing here waits for the GPU unless a base array was re-uploaded.
    void syncParameters(bool particles, bool exceptions) {
        if (offsetsDirty) { basePDirty = true; baseEDirty = true; offsetsDirty = false; }
        if (basePDirty) { uploadParticleBase(); particles = true; }
        if (baseEDirty) { upload14(); exceptions = true; }
        if (globalsDirty) {
            dGlobals.resize(std::max(nGlobals, 1));
            if (nGlobals > 0) pinned.copy(dGlobals.p, gValues.data(), sizeof(double) * nGlobals, stream);
            globalsDirty = false;
        }
        if (particles) {
            // headroom of the spreader's 32-bit fixed point, per mesh: max(16, 8 x atoms per mesh cell) (misc.hip, k_fixScale)
            static const double forced = getenv("SNB_FIX_HEADROOM") ? atof(getenv("SNB_FIX_HEADROOM")) : 0.0;      // test switch: a fixed headroom (16 = the rule before the mesh-dependent one)
            auto headroom = [&](const PmePlanDims& d) { const double cells = (double)std::max(d.nx, 1) * std::max(d.ny, 1) * std::max(d.nz, 1); return forced > 0 ? forced : std::max(16.0, 8.0 * (double)N / cells); };
            launchParticleParams<Real>(N, nsub, dBaseP.p, offP.empty() ? nullptr : dOffPStart.p, dOffPGlobal.p, dOffPDelta.p, dGlobals.p, dUSubset.p, dUCharge.p, dUSigEps.p,
                                       dParamSums.p, dFixScale.p, headroom(pme.d), headroom(dpme.d), stream);
        }
        if (exceptions) launchExceptionParams<Real>(n14, dBase14.p, offE.empty() ? nullptr : dOff14Start.p, dOff14Global.p, dOff14Delta.p, dGlobals.p, dSlice14.p, params14.p, stream);
    }
    // New parameter VALUES with the same subsets / exception pairs (copyParametersToContext, parameter offsets: the reference recomputes
    // them on the device per changed global parameter, nonbondedParameters.cc:4-179): effective values on the device, then the sorted
    // per-atom arrays are rewritten in place -- no re-sort, no tile rebuild, no host synchronisation, and the captured step graphs stay
    // valid (nothing derived from the values is baked into them: the spreader reads its fixed-point scale from device memory).
    void refreshValues() {
        const bool particles = valuesDirty;
        syncParameters(valuesDirty, excValuesDirty);
        if (particles) launchRefreshParams<Real>(dSortedToUser.p, dUCharge.p, dUSigEps.p, posq.p, sigeps.p, Npad, stream);
        valuesDirty = excValuesDirty = false;
    }

    void uploadStatic() {
        const size_t m = excPairs.size() / 2;
        std::vector<int> hStart((size_t)N + 1, 0), hList(2 * m);
        for (size_t k = 0; k < m; k++) {
            const int a = excPairs[2 * k], b = excPairs[2 * k + 1];
            hStart[a + 1]++; hStart[b + 1]++;
        }
        for (int i = 0; i < N; i++) hStart[i + 1] += hStart[i];
        { std::vector<int> fill(N, 0); for (size_t k = 0; k < m; k++) { const int a = excPairs[2 * k], b = excPairs[2 * k + 1]; hList[hStart[a] + fill[a]++] = b; hList[hStart[b] + fill[b]++] = a; } }
        exclStart.upload(hStart, stream); exclList.upload(hList, stream);
        // user-order parameters: formed on the device from base values, offsets and global parameters
        dUSubset.upload(std::vector<int>(subset.begin(), subset.end()), stream);
        HIPCHECK(hipStreamSynchronize(stream));
        basePDirty = baseEDirty = true;
        syncParameters(true, true);
        // padded subset layout (depends on subset populations only)
        std::vector<int> cnt(nsub, 0);
        for (int i = 0; i < N; i++) cnt[subset[i]]++;
        hSubsetStart.assign(nsub + 1, 0); hSubsetPaddedStart.assign(nsub + 1, 0);
        for (int k = 0; k < nsub; k++) { hSubsetStart[k + 1] = hSubsetStart[k] + cnt[k]; hSubsetPaddedStart[k + 1] = hSubsetPaddedStart[k] + ((cnt[k] + 31) / 32) * 32; }
        staticNpad = hSubsetPaddedStart[nsub];
        std::vector<unsigned char> pad(std::max(staticNpad, 1), 0);
        staticBlkSubset.clear();
        for (int k = 0; k < nsub; k++) {
            for (int x = hSubsetPaddedStart[k] + cnt[k]; x < hSubsetPaddedStart[k + 1]; x++) pad[x] = 1;
            for (int bb = hSubsetPaddedStart[k] / 32; bb < hSubsetPaddedStart[k + 1] / 32; bb++) staticBlkSubset.push_back(k);
        }
        dSubsetStart.upload(hSubsetStart, stream); dSubsetPaddedStart.upload(hSubsetPaddedStart, stream); dPadFlag.upload(pad, stream);
        std::vector<int> slot(nsub, -1);
        if (cfg.shard_count == 1) std::iota(slot.begin(), slot.end(), 0); else for (size_t g = 0; g < ownedSubsets.size(); g++) slot[ownedSubsets[g]] = (int)g;
        dSlotOfSubset.upload(slot, stream);
        HIPCHECK(hipStreamSynchronize(stream));
        staticDirty = false;
        npadPredict = 0;      // (subsets or exclusions may have changed: the next rebuild waits for its padded count again)
    }

    // ------------------------------------------------------------------------------------------
    // GPU neighbour build (neighbor.hip): rectangular periodic boxes with cutoff.  Returns false when the
    // host builder must take over (per-pair-wrap regime, gather-capacity overflow).
    // ------------------------------------------------------------------------------------------
    // the totals of a build through mapped host memory: spins on the sequence number k_nbPublish writes last (falls back to a copy after 2 s)
    void waitForTotals(int seq, int* h) {
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(2);
        volatile int* pub = hNbPub;
        while (pub[8] != seq && std::chrono::steady_clock::now() < deadline) { __builtin_ia32_pause(); }
        if (pub[8] == seq) { for (int k = 0; k < 8; k++) h[k] = pub[k]; }
        else { HIPCHECK(hipMemcpyAsync(h, dCounters.p, sizeof(int) * 8, hipMemcpyDeviceToHost, stream)); HIPCHECK(hipStreamSynchronize(stream)); }
    }
    // a build whose totals are in: host-side counts, the next prediction, statistics
    void acceptBuild(const int* h, float sortMs) {
        numTiles = h[0]; numWorkItems = h[1] + h[4]; numMaskTiles = h[2]; wrapMode = false;
        if (getenv("SNB_DEBUG_WORK")) {      // consistency of the work list: the items must cover every tile exactly once
            std::vector<int4> hw(numWorkItems);
            HIPCHECK(hipMemcpy(hw.data(), workItems.p, sizeof(int4) * numWorkItems, hipMemcpyDeviceToHost));
            long long sumZ = 0; int hist[9] = {0};
            for (auto& w : hw) { sumZ += w.z; hist[std::min(std::max(w.z, 0), 8)]++; }
            fprintf(stderr, "[snb] work list: %d items (%d full + %d partial), tiles %d, sum of item sizes %lld, sizes 1..8:", numWorkItems, h[1], h[4], (int)numTiles, sumZ);
            for (int k = 1; k <= 8; k++) fprintf(stderr, " %d", hist[k]);
            fprintf(stderr, "\n");
        }
        shardTiles = numTiles;      // the builder only emitted the blocks this engine owns
        // next rebuild's array size: this count + 0.4 % + 8 blocks (c3: 300 k atoms move its count by a few blocks between rebuilds); never shrinking,
        // so that the buffers -- and the step graph's arguments -- stay where they are
        npadPredict = std::max(npadPredict, ((int)(h[7] * 1.004) + 256 + 31) / 32 * 32);
        { static const int shortBy = getenv("SNB_NB_PREDICT_SHORT") ? atoi(getenv("SNB_NB_PREDICT_SHORT")) : 0;      // test switch: predict this many blocks too FEW (exercises the repeat path)
          if (shortBy > 0) npadPredict = std::max(32, h[7] - 32 * shortBy); }
        stats.n_rebuilds++;
        float gpuMs = 0;   // device time of the build (the host clock would also count the queued steps this call waited for)
        HIPCHECK(hipEventElapsedTime(&gpuMs, evRebuild[0], evRebuild[1]));
        stats.last_rebuild_ms = gpuMs + sortMs;
    }

    // ---- the rebuild beside the steps ----
    void swapListSets() {
        auto sw = [](auto& a, auto& b) { std::swap(a.p, b.p); std::swap(a.n, b.n); };
        sw(dUserToSorted, shadow.dUserToSorted); sw(dSortedToUser, shadow.dSortedToUser); sw(atomSubset, shadow.atomSubset); sw(atomGrid, shadow.atomGrid);
        sw(blockSubset, shadow.blockSubset); sw(tileJ, shadow.tileJ); sw(colRange, shadow.colRange); sw(posq, shadow.posq); sw(posRef, shadow.posRef);
        sw(sigeps, shadow.sigeps); sw(imageOffset, shadow.imageOffset); sw(tileInfo, shadow.tileInfo); sw(workItems, shadow.workItems); sw(masks, shadow.masks);
    }
    // whether the rebuild that falls due `sideLead` executes from now may be built beside the steps: a list built on the GPU with a predicted
    // padded count is in use, nothing but the positions has changed since, fixed interval.  (Displacement-triggered rebuilds were given side
    // builds too, timed by a guess of the watch's next interval -- built, parity green, and slower than rebuilding in line at the watch's own
    // pace: 0.3634 against 0.3539 ms per step on c3, the guess brings rebuilds forward and a third of them still came in line.  Removed.)
    bool sideBuildPossible() const {
        return sideMode && !sidePending && gpuBuilt && cfg.rebuild_interval > sideLead + 1 && cfg.neighbor_padding > 0 && isPeriodic() && !cfg.host_neighbor_build && !cfg.disable_graph
               && !needRebuild && !paramsDirty && !staticDirty && !valuesDirty && !excValuesDirty && npadPredict > 0 && npadPredict == Npad && hNbPub && dNbPub && devUserPos;
    }
    // Copies the positions aside (in stream order: the positions of the step just enqueued) and enqueues the whole build on streamBuild, into
    // the shadow buffers.  Nothing the steps use is touched; the host does not wait.
    void startSideBuild() {
        if (!streamBuild) {
            int lo = 0, hi = 0;
            HIPCHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            // (normal priority: at the lowest the build crawls and the steps end up waiting for it, at the highest its kernels push the tile kernel aside)
            HIPCHECK(hipStreamCreateWithPriority(&streamBuild, hipStreamNonBlocking, getenv("SNB_SIDE_PRIO_HIGH") ? hi : (getenv("SNB_SIDE_PRIO_LOW") ? lo : (lo + hi) / 2)));
            HIPCHECK(hipEventCreateWithFlags(&evSnap, hipEventDisableTiming)); HIPCHECK(hipEventCreateWithFlags(&evBuilt, hipEventDisableTiming)); HIPCHECK(hipEventCreateWithFlags(&evFlagsReset, hipEventDisableTiming));
        }
        const size_t bytes = (size_t)N * (posStride4 ? 4 : 3) * (posIsDouble ? 8 : 4);
        posSnap.resize(bytes);
        HIPCHECK(hipMemcpyAsync(posSnap.p, devUserPos, bytes, hipMemcpyDeviceToDevice, stream));
        HIPCHECK(hipEventRecord(evSnap, stream));
        HIPCHECK(hipStreamWaitEvent(streamBuild, evSnap, 0));
        const hipStream_t liveStream = stream; const void* livePos = devUserPos;
        const int liveCells[2] = {colCells[0], colCells[1]};
        swapListSets(); stream = streamBuild; devUserPos = posSnap.p; sideBuilding = true;
        bool ok = false;
        try {
            ok = gpuRebuild();
            if (ok) { posRef.resize(Npad); HIPCHECK(hipMemcpyAsync(posRef.p, posq.p, sizeof(T4) * (size_t)Npad, hipMemcpyDeviceToDevice, stream)); }
            HIPCHECK(hipEventRecord(evBuilt, stream));
        } catch (...) { sideBuilding = false; stream = liveStream; devUserPos = livePos; swapListSets(); colCells[0] = liveCells[0]; colCells[1] = liveCells[1]; throw; }
        sideBuilding = false; stream = liveStream; devUserPos = livePos; swapListSets();
        if (!ok) { colCells[0] = liveCells[0]; colCells[1] = liveCells[1]; HIPCHECK(hipEventSynchronize(evBuilt)); }      // (whatever was enqueued has finished with the scratch arrays)
        sidePending = ok;
    }
    // drops a pending side build (its result will not be used): waits until it has finished with the scratch arrays
    void cancelSideBuild() {
        if (!sidePending) return;
        HIPCHECK(hipEventSynchronize(evBuilt));
        sidePending = false; sideDiscarded++;
    }
    // The rebuild has fallen due and a side build is pending: true = its list is now the one in use (nothing else to do); false = it failed
    // (padded count or a partition over its capacity) and the caller rebuilds in line.
    bool finishSideBuild() {
        sidePending = false;
        int h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        waitForTotals(sideSeq, h);
        HIPCHECK(hipEventSynchronize(evBuilt));      // (returns at once: the totals are the build's last kernel but one)
        static const bool verbose = getenv("SNB_VERBOSE") != nullptr;
        static const bool reject = getenv("SNB_SIDE_REJECT") != nullptr;      // test switch: every side build is discarded and the rebuild repeated in line
        if (h[7] < N || (h[7] & 31) || h[7] > Npad || h[3] != 0 || reject) {
            if (verbose) fprintf(stderr, "[snb] side rebuild discarded (padded count %d of %d, overflow flag %d): rebuilding in line\n", h[7], Npad, h[3]);
            if (h[7] > Npad) { npadPredict = 0; padMispredictions++; }
            sideDiscarded++;
            return false;
        }
        HIPCHECK(hipStreamWaitEvent(stream, evBuilt, 0));      // stream order for the steps that follow
        dropGraph();      // every kernel argument block changes
        swapListSets();
        acceptBuild(h, 0.f);
        wrapMode = false; gpuBuilt = true; needRebuild = false; stepsSinceRebuild = 0;
        launchDispFlagsReset(dDispFlags, stream);      // (in stream order: steps on the old list may still be queued, and their flags belong to it)
        HIPCHECK(hipEventRecord(evFlagsReset, stream)); flagsResetPending = true;      // (until then the host copy still shows the old list's flags)
        sideBuilds++;
        return true;
    }

    bool gpuRebuild() {
        if (cfg.method == SNB_NoCutoff || N < 64) return false;
        if (sideBuilding && !(npadPredict > 0 && npadPredict == Npad && getenv("SNB_NB_SYNC_PADDED") == nullptr)) return false;
        static const bool hostTriclinic = getenv("SNB_HOST_TRICLINIC") != nullptr;      // testing aid: old behaviour
        if (hostTriclinic && (!isPeriodic() || !(box[3] == 0 && box[6] == 0 && box[7] == 0))) return false;
        const double R = cfg.cutoff + cfg.neighbor_padding;
        // The cell the builder works in: the periodic box, or -- CutoffNonPeriodic -- a rectangular cell around the atoms' bounding box with
        // more than a list radius of empty margin on every side, so that no periodic image of anything is ever in reach
        double cell[9], origin[3] = {0, 0, 0};
        for (int i = 0; i < 9; i++) cell[i] = box[i];
        if (!isPeriodic()) {
            dExtent.resize(6);
            launchExtent(devUserPos, posIsDouble, posStride4, N, dExtent.p, stream);
            int h[6];
            HIPCHECK(hipMemcpyAsync(h, dExtent.p, sizeof(h), hipMemcpyDeviceToHost, stream));
            HIPCHECK(hipStreamSynchronize(stream));
            auto dec = [](int i) { i = i >= 0 ? i : i ^ 0x7FFFFFFF; float f; std::memcpy(&f, &i, 4); return (double)f; };
            const double margin = R + 0.5 * cfg.neighbor_padding + 0.05;
            for (int d = 0; d < 3; d++) {
                const double lo = dec(h[d]), hi = dec(h[3 + d]);
                if (!(hi >= lo) || !std::isfinite(lo) || !std::isfinite(hi)) return false;
                origin[d] = lo - margin;
                cell[4 * d] = (hi - lo) + 2.0 * margin;
            }
            cell[1] = cell[2] = cell[3] = cell[5] = cell[6] = cell[7] = 0;
        }
        const double* box = cell;      // (shadows the member for the rest of the build)
        const double volume = box[0] * box[4] * box[8];
        const double aTarget = std::cbrt(32.0 * volume / std::max(N, 1));
        for (int d = 0; d < 3; d++) if (!(4.0 * aTarget + 2 * R < box[4 * d])) return false;   // tile-image scheme needs extent + 2R < L (checked exactly on the GPU too)
        auto t0 = std::chrono::steady_clock::now();
        int ncx = std::max(1, std::min(2048, (int)std::lround(box[0] / aTarget)));
        int ncy = std::max(1, std::min(2048, (int)std::lround(box[4] / aTarget)));
        colCells[0] = colCells[1] = 0;
        if (isPme() && nGrids > 0) {
            auto pick = [&](int n, double L) {
                int best = 0; double bestErr = 1e300;
                for (int d = 5; d <= 16 && d <= n; d++) if (n % d == 0) { double er = std::fabs(d * L / n - aTarget); if (er < bestErr) { bestErr = er; best = d; } }
                return best;
            };
            const int px = pick(pme.d.nx, box[0]), py = pick(pme.d.ny, box[4]);
            if (px > 0 && py > 0 && sizeof(double) * (size_t)px * py * pme.d.nz <= 60 * 1024 && std::max(pme.d.nx, std::max(pme.d.ny, pme.d.nz)) < 1024) {      // (the packed mesh cell of an atom holds 10 bits per axis)
            colCells[0] = px; colCells[1] = py; ncx = pme.d.nx / px; ncy = pme.d.ny / py; }
        }
        // phase A: sort and block segmentation (the padded atom count depends on where the sorted order jumps)
        dUserToSorted.resize(N);
        colRange.resize((size_t)nsub * ncx * ncy); dZIndex.resize((size_t)nsub * ncx * ncy * 65);
        dWrapped.resize((size_t)3 * N); dOffsetU.resize((size_t)3 * N); dKeysIn.resize(N); dKeysOut.resize(N); dValsIn.resize(N); dValsOut.resize(N);
        dScanA.resize(N); dScanB.resize(N); dCounters.resize(32 * 65);
        const size_t tempBytes = nbSortTempBytes<Real>(N);
        dSortTemp.resize(tempBytes);
        NbParams<Real> p;
        std::memset(&p, 0, sizeof(p));
        p.nAtoms = N; p.nSubsets = nsub; p.ncx = ncx; p.ncy = ncy;
        p.subsetBits = 1; while ((1 << p.subsetBits) < nsub) p.subsetBits++;
        p.colBits = 1; while ((1ll << p.colBits) < (long long)ncx * ncy) p.colBits++;
        for (int i = 0; i < 9; i++) p.boxm[i] = box[i];
        for (int d = 0; d < 3; d++) p.origin[d] = origin[d];
        for (int i = 0; i < 9; i++) tileCell[i] = box[i];
        p.listCutoff = (float)R;
        { static const bool bw = getenv("SNB_NB_BOX_WALK") != nullptr; p.boxWalk = bw ? 1 : 0; }
        p.jumpDist = (float)(2.0 * std::sqrt(2.0) * std::max(box[0] / ncx, box[4] / ncy));   // neighbours along the sort path of a dense region are closer than this
        p.uSubset = dUSubset.p; p.uCharge = dUCharge.p; p.uSigEps = dUSigEps.p; p.uExclStart = exclStart.p; p.uExclList = exclList.p;
        p.slotOfSubset = dSlotOfSubset.p;
        p.wrapped = dWrapped.p; p.offsetU = dOffsetU.p; p.keysIn = dKeysIn.p; p.keysOut = dKeysOut.p; p.valsIn = dValsIn.p; p.valsOut = dValsOut.p;
        p.segKey = dValsIn.p; p.segStart = dScanA.p; p.padExtra = dScanB.p; p.padBefore = dScanA.p;   // valsIn is dead once the sort has run
        dScanC.resize(N); p.blockWide = dScanC.p; p.blockWideOut = dScanC.p;
        for (int d = 0; d < 3; d++) p.maxHalfExtent[d] = (float)(0.45 * (box[4 * d] - 2.0 * R));   // extent <= 0.9 (L - 2R)
        p.userToSorted = dUserToSorted.p; p.colRange = colRange.p; p.zIndex = dZIndex.p; p.counters = dCounters.p;
        if (!evRebuild[0]) { HIPCHECK(hipEventCreate(&evRebuild[0])); HIPCHECK(hipEventCreate(&evRebuild[1])); HIPCHECK(hipEventCreate(&evRebuild[2])); }
        HIPCHECK(hipEventRecord(evRebuild[0], stream));
        {
            // Phase A is ~30 small launches (key pass, radix sort, segmentation scans) with nothing but launch latency between them:
            // captured once per (parameters, buffers, position pointer) and replayed, the way the step itself is.  Any failure to
            // capture (a library call that is not capturable) falls back to plain launches for good.
            struct SortKey { NbParams<Real> p; const void* pos; int isDouble, stride4; void* temp; size_t tempBytes; };
            static_assert(std::is_trivially_copyable<SortKey>::value, "SortKey is compared bytewise");
            SortKey key; std::memset(&key, 0, sizeof(key));
            key.p = p; key.pos = devUserPos; key.isDouble = posIsDouble; key.stride4 = posStride4; key.temp = dSortTemp.p; key.tempBytes = tempBytes;
            static const bool noSortGraph = getenv("SNB_NO_SORT_GRAPH") != nullptr;
            bool replayed = false;
            // (round 4: a replayed phase-A graph once left 32 N padded slots -- its memset node for blockWideOut had stopped zeroing after other
            // graphs with memset nodes had been instantiated; phase A now zero-fills with a kernel, misc.hip launchZeroFill.  `sortGraphSuspect`
            // is the insurance that caught nothing since: a padded count that doubles out of a replayed graph is built once more without it.)
            // (above 2^20 keys rocprim's radix sort takes its onesweep path, which clears its histogram with hipMemsetAsync -- a memset node if
            // captured: such systems issue phase A as plain launches)
            if (!noSortGraph && !sortGraphBroken && !cfg.disable_graph && !sortGraphSuspect && N <= (1 << 20)) {
                hipGraphExec_t sortGraphExec = nullptr;
                for (auto& g : sortGraphs) if (g.key.size() == sizeof(key) && std::memcmp(g.key.data(), &key, sizeof(key)) == 0) sortGraphExec = g.exec;
                if (!sortGraphExec) {
                    hipGraph_t graph = nullptr;
                    if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                        bool threw = false;
                        try { launchNeighborSort<Real>(p, devUserPos, posIsDouble, posStride4, dSortTemp.p, tempBytes, stream); }
                        catch (...) { threw = true; }      // never leave the caller's stream in capture mode
                        hipError_t endErr = hipStreamEndCapture(stream, &graph);
                        if (threw) { endErr = hipErrorUnknown; (void)hipGetLastError(); }
                        if (endErr == hipSuccess && graph && hipGraphInstantiate(&sortGraphExec, graph, nullptr, nullptr, 0) == hipSuccess) {
                            SortGraph g; g.key.assign(reinterpret_cast<const unsigned char*>(&key), reinterpret_cast<const unsigned char*>(&key) + sizeof(key)); g.exec = sortGraphExec;
                            if (sortGraphs.size() < 4) sortGraphs.push_back(g);
                            else { (void)hipGraphExecDestroy(sortGraphs[sortGraphVictim].exec); sortGraphs[sortGraphVictim] = g; sortGraphVictim = (sortGraphVictim + 1) % 4; }
                        } else { sortGraphExec = nullptr; sortGraphBroken = true; (void)hipGetLastError(); }
                        if (graph) (void)hipGraphDestroy(graph);
                    } else { sortGraphBroken = true; (void)hipGetLastError(); }
                }
                if (sortGraphExec) { HIPCHECK(hipGraphLaunch(sortGraphExec, stream)); replayed = true; }
            }
            if (!replayed) launchNeighborSort<Real>(p, devUserPos, posIsDouble, posStride4, dSortTemp.p, tempBytes, stream);
        }
        HIPCHECK(hipEventRecord(evRebuild[2], stream));
        // The padded atom count depends on where the sorted order jumps, i.e. it is known on the device only.  Round 4: from the second
        // rebuild on the arrays are sized by the previous count plus a margin (a few spare all-padding blocks at the end) and the build
        // goes on without waiting; the exact count comes back with the tile counters below, and an overflow (k_nbScatter counts the
        // atoms that did not fit) repeats the rebuild with the exact count.  Saves one host round trip (~75 us) per rebuild.
        static const bool noPredict = getenv("SNB_NB_SYNC_PADDED") != nullptr;      // test switch: wait for the count, as before round 4
        const bool predicted = npadPredict > 0 && !noPredict;
        if (predicted) Npad = npadPredict;
        else {
            int npadDev = 0;
            HIPCHECK(hipMemcpyAsync(&npadDev, dCounters.p + 7, sizeof(int), hipMemcpyDeviceToHost, stream));
            HIPCHECK(hipStreamSynchronize(stream));
            if (npadDev < N || (npadDev & 31)) throw HipError{"neighbour build: inconsistent padded atom count"};
            Npad = npadDev;
        }
        numBlocks = Npad / 32;
        if (Npad >= (1 << SNB_JSHIFT_BITS) - 1) throw HipError{"too many atoms for the 25-bit tile index"};
        // outputs / scratch sized by the padded count
        posq.resize(Npad); sigeps.resize(Npad); imageOffset.resize((size_t)3 * Npad);
        dSortedToUser.resize(Npad); atomSubset.resize(Npad); atomGrid.resize(Npad); blockSubset.resize(numBlocks);
        layoutForces();
        dBlockCenter.resize((size_t)3 * numBlocks); dBlockHalf.resize((size_t)3 * numBlocks);
        if (tileCap < (size_t)numBlocks * 40) tileCap = (size_t)numBlocks * 40;
        if (tileCap < 64 * 128) tileCap = 64 * 128;      // 64 allocation partitions, each with room for a few blocks' worth of tiles
        float sortMs = 0;
        if (!predicted) HIPCHECK(hipEventElapsedTime(&sortMs, evRebuild[0], evRebuild[2]));      // (predicted: the sort is part of the span measured below)
        for (int attempt = 0; attempt < 3; attempt++) {
            tileJ.resize(tileCap * 32); tileInfo.resize(tileCap); masks.resize(tileCap * 32); workItems.resize(2 * (tileCap / 4 + 2 * numBlocks + 64)); workItemsStage.resize(tileCap / 4 + 2 * numBlocks + 64); workItemsPartial.resize(tileCap / 4 + 2 * numBlocks + 64);
            p.nPadded = Npad; p.nBlocks = numBlocks; p.blockSubset = blockSubset.p;
            p.shardBegin = shardBegin; p.shardWidth = shardEnd - shardBegin; p.shardPeriod = shardPeriod;
            { static const bool boxOnly = getenv("SNB_BOX_PRUNE") != nullptr; p.exactPrune = boxOnly ? 0 : 1; }
            { static const int it = getenv("SNB_ITEM_TILES") ? std::max(1, std::min(32, atoi(getenv("SNB_ITEM_TILES")))) : 8; p.itemTiles = it; }
            p.blockCenter = dBlockCenter.p; p.blockHalf = dBlockHalf.p;
            p.sortedToUser = dSortedToUser.p; p.userToSorted = dUserToSorted.p; p.posq = posq.p; p.sigeps = sigeps.p; p.imageOffset = imageOffset.p;
            p.atomSubset = atomSubset.p; p.atomGrid = atomGrid.p; p.colRange = colRange.p; p.zIndex = dZIndex.p;
            dAtomCell.resize(Npad); p.atomCell = dAtomCell.p;
            static const bool nbTrace = getenv("SNB_NB_TRACE") != nullptr;
            if (nbTrace) { dNbTrace.resize((size_t)4 * numBlocks); p.dbgOut = dNbTrace.p; }
            p.tileJ = tileJ.p; p.tileInfo = tileInfo.p; p.masks = masks.p; p.workItems = workItems.p; p.workItemsStage = workItemsStage.p; p.workItemsPartial = workItemsPartial.p; p.counters = dCounters.p;
            p.tileCapacity = (int)tileCap; p.workCapacity = (int)(tileCap / 4 + 2 * numBlocks + 64); p.maskCapacity = (int)tileCap;
            if (!predicted || attempt > 0) HIPCHECK(hipEventRecord(evRebuild[0], stream));
            launchNeighborBuild<Real>(p, stream);
            HIPCHECK(hipEventRecord(evRebuild[1], stream));
            if (sideBuilding) {      // build beside the steps: publish the totals and leave; finishSideBuild reads them when the rebuild falls due
                sideSeq = ++nbPubSeq;
                launchNeighborPublish(dCounters.p, dNbPub, sideSeq, stream);
                return true;
            }
            int h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            // the totals come back through mapped host memory and a sequence number the host spins on (a sleeping hipStreamSynchronize wakes
            // up 30-45 us after the kernel has ended, with the GPU idle); the synchronisation behind it then returns at once
            static const bool noSpin = getenv("SNB_NB_NO_SPIN") != nullptr;
            if (hNbPub && dNbPub && !noSpin) {
                const int seq = ++nbPubSeq;
                launchNeighborPublish(dCounters.p, dNbPub, seq, stream);
                waitForTotals(seq, h);
                HIPCHECK(hipStreamSynchronize(stream));      // (the events below are read next; nothing is running any more)
            } else {
                HIPCHECK(hipMemcpyAsync(h, dCounters.p, sizeof(h), hipMemcpyDeviceToHost, stream));
                HIPCHECK(hipStreamSynchronize(stream));
            }
            // (insurance for the above: a padded count that has doubled since the last accepted build, out of a replayed graph, is built once more
            // from plain launches before anything is sized by it)
            if (!sortGraphSuspect && stats.n_rebuilds > 0 && h[7] > 2 * std::max(Npad, N) && !sideBuilding) {
                sortGraphSuspect = true;
                { static const bool verbose = getenv("SNB_VERBOSE") != nullptr; if (verbose) fprintf(stderr, "[snb] rebuild: padded count %d after %d: phase A once more without its graph\n", h[7], Npad); }
                return gpuRebuild();
            }
            if (predicted && (h[7] > Npad || h[7] < N || (h[7] & 31))) {      // the prediction was too small (or the count is inconsistent): once more, waiting for the exact count
                if (h[7] < N || (h[7] & 31)) throw HipError{"neighbour build: inconsistent padded atom count"};
                npadPredict = 0; padMispredictions++;
                { static const bool verbose = getenv("SNB_VERBOSE") != nullptr; if (verbose) fprintf(stderr, "[snb] rebuild: padded count %d exceeded the predicted %d, repeating with the exact count (%lld so far)\n", h[7], Npad, (long long)padMispredictions); }
                return gpuRebuild();
            }
            if (nbTrace && h[3] == 0) {
                std::vector<long long> tr((size_t)4 * numBlocks);
                HIPCHECK(hipMemcpy(tr.data(), dNbTrace.p, sizeof(long long) * tr.size(), hipMemcpyDeviceToHost));
                long long t0min = tr[0], t1max = tr[1]; std::vector<long long> dur(numBlocks);
                tr.resize((size_t)4 * numBlocks);
                for (int b = 0; b < numBlocks; b++) { t0min = std::min(t0min, tr[2 * b]); t1max = std::max(t1max, tr[2 * b + 1]); dur[b] = tr[2 * b + 1] - tr[2 * b]; }
                std::vector<long long> sorted = dur; std::sort(sorted.begin(), sorted.end());
                fprintf(stderr, "[snb] nb trace: span %.1f us; per-block duration median %.1f us, p90 %.1f, p99 %.1f, max %.1f us; last start at %.1f us\n", (t1max - t0min) / 100.0, sorted[numBlocks / 2] / 100.0,
                        sorted[numBlocks * 9 / 10] / 100.0, sorted[numBlocks * 99 / 100] / 100.0, sorted.back() / 100.0, (t1max - t0min) / 100.0);
                { std::vector<long long> a(numBlocks), b(numBlocks), c(numBlocks);
                  for (int k = 0; k < numBlocks; k++) { a[k] = tr[2 * numBlocks + 2 * k] - tr[2 * k]; b[k] = tr[2 * numBlocks + 2 * k + 1] - tr[2 * numBlocks + 2 * k]; c[k] = tr[2 * k + 1] - tr[2 * numBlocks + 2 * k + 1]; }
                  std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end()); std::sort(c.begin(), c.end());
                  fprintf(stderr, "[snb] nb trace: median prologue %.1f us, gather %.1f us, final flush %.1f us\n", a[numBlocks / 2] / 100.0, b[numBlocks / 2] / 100.0, c[numBlocks / 2] / 100.0); }
                long long lastStart = 0; for (int b = 0; b < numBlocks; b++) lastStart = std::max(lastStart, tr[2 * b] - t0min);
                fprintf(stderr, "[snb] nb trace: last block start %.1f us after the first\n", lastStart / 100.0);
            }
            if (h[3] == 0) { acceptBuild(h, sortMs); gpuBuilt = true; needRebuild = false; paramsDirty = false; stepsSinceRebuild = 0; (void)t0; return true; }
            if (getenv("SNB_VERBOSE")) fprintf(stderr, "[snb] gpu neighbour build attempt %d: tiles %d work %d masks %d overflow %d partial %d maxPartTiles %d maxPartWork %d (cap %zu, region %zu / %zu)\n", attempt, h[0], h[1], h[2], h[3], h[4], h[5], h[6], tileCap, tileCap / 64, (tileCap / 4 + 2 * numBlocks + 64) / 64);
            // capacity: a partition (1/64 of the arrays) ran out of tiles, masks or work items -> grow and retry
            if ((size_t)h[5] > tileCap / 64 || (size_t)h[6] > (tileCap / 4 + 2 * numBlocks + 64) / 64) { tileCap = std::max((size_t)h[5] * 64 * 5 / 4, tileCap * 3 / 2) + 4096; continue; }
            return false;   // a block gathered more than its LDS list holds, or a block is too extended for tile images: host path
        }
        return false;
    }

    // ------------------------------------------------------------------------------------------
    void fillPme(PmeParams<Real>& p, PmePlan<Real>& plan, bool wantEnergy) {
        p.d = plan.d; p.nsub = nGrids; p.natoms = Npad; p.posq = posq.p; p.sigeps = sigeps.p; p.atomSubset = atomSubset.p; p.atomGrid = atomGrid.p;
        p.cells = pmeCells.p;
        { static const bool tr = getenv("SNB_PME_TRACE") != nullptr; if (tr) { if (!dPmeTrace.p) { dPmeTrace.resize(8); HIPCHECK(hipMemset(dPmeTrace.p, 0, 64)); } p.trace = dPmeTrace.p; } }
        p.stepTrace = (traceThisStep && plan.dispersion == (cfg.method == SNB_LJPME)) ? dStepTrace.p : nullptr;      // (the step's LAST mesh: its inverse z transform is what the second launch follows)
        p.cellsReady = (!plan.dispersion && cellsFromGather) ? 1 : 0;
        p.fixDev = dFixScale.p ? dFixScale.p + (plan.dispersion ? 2 : 0) : nullptr;      // (k_fixScale keeps it in step with the parameters)
        p.gridReal = plan.gridReal.p; p.gridCplx = plan.gridCplx.p; p.planeB = plan.gridCplxB.p; p.planeEterm = plan.planeEtermReady ? plan.planeEterm.p : nullptr; p.twx = plan.twx.p; p.twy = plan.twy.p; p.twz = plan.twz.p;
        p.modx = plan.modx.p; p.mody = plan.mody.p; p.modz = plan.modz.p;
        const double det = box[0] * box[4] * box[8], sc = 1.0 / det;
        const double r[9] = {box[4] * box[8] * sc, 0, 0, -box[3] * box[8] * sc, box[0] * box[8] * sc, 0,
                             (box[3] * box[7] - box[4] * box[6]) * sc, -box[0] * box[7] * sc, box[0] * box[4] * sc};
        for (int i = 0; i < 9; i++) { p.recip[i] = (Real)r[i]; p.recipLo[i] = (Real)(r[i] - (double)p.recip[i]); }
        p.alpha = (Real)plan.alpha; p.volume = (Real)det; p.dispersion = plan.dispersion ? 1 : 0;
        p.lambdas = dLambdas.p; p.sliceNeed = energySelective ? dSliceNeedSel.p : dSliceNeedAll.p; p.gridSubset = gridSubset.p; p.nsubTotal = nsub; p.mix = cfg.shard_count == 1 ? 1 : 0;
        { static const bool m16 = getenv("SNB_MIX_16X16") != nullptr; p.mix16 = m16 ? 1 : 0; }
        p.sliceE = sliceE.p; p.fpx = fpx.p; p.fpy = fpy.p; p.fpz = fpz.p; p.wantEnergy = wantEnergy ? 1 : 0;
        // brick kernels: the sort columns were cut for the Coulomb mesh; any mesh whose cells tile those columns can use them,
        // with bricks of `group` columns when one column is narrower than 5 cells (stencil 4 + 1 cell of drift)
        p.sortNcx = p.sortNcy = 0; p.groupX = p.groupY = 1; p.colRange = nullptr;
        p.zSlabs = 1;
        // measured on c3: f64 accumulation 1 slab 110 us, 2 slabs 105 us, 4 slabs 145 us; fixed-point (single precision, even nz) 1 slab 61 us, 2 slabs 67 us
        if (plan.d.nz % 2 == 0 && plan.d.nz >= 32 && !(sizeof(Real) == 4)) p.zSlabs = 2;
        if (const char* zs = getenv("SNB_ZSLABS")) { const int k = atoi(zs); if (k >= 1 && plan.d.nz % k == 0) p.zSlabs = k; }
        int ncx, ncy, gx, gy;
        if (brickGeometry(plan, ncx, ncy, gx, gy)) {
            p.sortNcx = ncx; p.sortNcy = ncy; p.groupX = gx; p.groupY = gy; p.colRange = colRange.p;
            p.ownSlabs = plan.ownSlabs; p.ownMargin = plan.ownMargin; p.ownPartial = plan.ownPartial.p; p.ownBusy = plan.ownBusy.p;
            p.strays = plan.strays.p; p.strayCount = dStrayCount.p ? dStrayCount.p + (plan.dispersion ? 1 : 0) : nullptr;
            if (!p.ownPartial || !p.ownBusy || !p.strays || !p.strayCount) p.ownSlabs = 0;
        }
    }
    // brick kernels: do this mesh's cells tile the sort columns (cut for the Coulomb mesh)?  Bricks of `group` columns when one column is
    // narrower than 5 cells (stencil 4 + 1 cell of drift).
    bool brickGeometry(const PmePlan<Real>& plan, int& ncx, int& ncy, int& gx, int& gy) const {
        if (colCells[0] <= 0) return false;
        ncx = pme.d.nx / colCells[0]; ncy = pme.d.ny / colCells[1];
        static const int gMin = getenv("SNB_BRICK_GROUP") ? atoi(getenv("SNB_BRICK_GROUP")) : 1;
        auto group = [](int n, int ncols) { if (n % ncols) return 0; const int cpc = n / ncols; for (int g = gMin; g <= 4; g++) if (g * cpc >= 5 && ncols % g == 0) return g; return 0; };
        gx = group(plan.d.nx, ncx); gy = group(plan.d.ny, ncy);
        const bool packable = plan.d.nx < 1024 && plan.d.ny < 1024 && plan.d.nz < 1024;   // k_pmeCells packs 10 bits per axis
        return packable && gx > 0 && gy > 0 && sizeof(double) * (size_t)(gx * plan.d.nx / ncx) * (gy * plan.d.ny / ncy) * plan.d.nz <= 100 * 1024;
    }
    // Geometry and buffers of the own-atoms spreader for one mesh (called from rebuild(): nothing may allocate inside a graph capture).
    // Margin: the cells an atom can drift across its column's border during a list's life (skin / 2), at least one; slabs: the fewest that
    // bring a work-group's LDS region under 40 KB (four work-groups per CU), at least two.
    // Plane path (pme.hip k_planeXY): its table of reciprocal-space kernel values follows the box and alpha, both fixed between rebuilds.
    void planPlaneTable(PmePlan<Real>& plan) {
        if (!plan.planeEterm.p || !plan.gridCplxB.p || nGrids <= 0) { plan.planeEtermReady = false; return; }
        // (the table depends on the box, alpha and the mesh only: a rebuild with the same box keeps it -- 11 us of kernel + a launch per rebuild)
        double key[10]; for (int i = 0; i < 9; i++) key[i] = box[i]; key[9] = plan.alpha;
        if (plan.planeEtermReady && plan.planeEtermFilled && std::memcmp(key, plan.planeEtermKey, sizeof(key)) == 0) return;
        std::memcpy(plan.planeEtermKey, key, sizeof(key));
        plan.planeEtermReady = false;
        PmeParams<Real> pp; std::memset(&pp, 0, sizeof(pp));
        fillPme(pp, plan, false);
        plan.planeEtermFilled = launchPlaneEterm<Real>(pp, plan.planeEterm.p, stream);
        plan.planeEtermReady = true;
    }
    void planOwnSpread(PmePlan<Real>& plan) {
        plan.ownSlabs = 0;
        int ncx, ncy, gx, gy;
        if (nGrids <= 0 || !brickGeometry(plan, ncx, ncy, gx, gy)) return;
        const double det = box[0] * box[4] * box[8], sc = 1.0 / det;
        const double r[9] = {box[4] * box[8] * sc, 0, 0, -box[3] * box[8] * sc, box[0] * box[8] * sc, 0,
                             (box[3] * box[7] - box[4] * box[6]) * sc, -box[0] * box[7] * sc, box[0] * box[4] * sc};
        const double perNmX = plan.d.nx * std::sqrt(r[0] * r[0] + r[3] * r[3] + r[6] * r[6]), perNmY = plan.d.ny * std::sqrt(r[1] * r[1] + r[4] * r[4] + r[7] * r[7]);      // mesh cells per nm of displacement
        int M = std::max(1, (int)std::ceil(0.5 * std::max(cfg.neighbor_padding, 0.0) * std::max(perNmX, perNmY) + 0.01));
        if (const char* e = getenv("SNB_SPREAD_MARGIN")) M = std::max(0, atoi(e));      // test switch (0: every border crossing becomes a stray)
        static const bool noFixed = getenv("SNB_NO_FIXED_SPREAD") != nullptr;
        const bool fixed = sizeof(Real) == 4 && !noFixed;
        const size_t accBytes = fixed ? 4 : 8;
        const int cx = gx * (plan.d.nx / ncx), cy = gy * (plan.d.ny / ncy), nz = plan.d.nz;
        const int RX = cx + 4 + 2 * M, RY = cy + 4 + 2 * M;
        if (RX > plan.d.nx || RY > plan.d.ny || gx * gy > 16 || nz > 256) return;
        int best = 0;
        static const int forced = getenv("SNB_OWN_SLABS") ? atoi(getenv("SNB_OWN_SLABS")) : 0;
        for (int pass = 0; pass < 2 && !best; pass++)
            for (int k = 2; k <= 32; k++) {      // (at least two: a single slab's region, nz + 4 planes, would wrap onto itself)
                if (forced > 0 && k != forced) continue;
                if (nz % k) continue;
                const int sz = nz / k;
                if (sz < 4 || (sz & 1)) continue;      // (8- or 16-byte copies of the regions)
                if (accBytes * (size_t)RX * RY * (sz + 4) <= (size_t)(pass == 0 ? 40 : 64) * 1024) { best = k; break; }
            }
        if (!best) return;
        // few, wide bricks (coarse mesh, 2 x 2 columns per brick): more slabs until a grid has ~400 work-groups (c3l's 60^3 dispersion mesh: 100 bricks)
        if (forced <= 0) for (int k = best + 1; k <= 16 && (ncx / gx) * (ncy / gy) * best < 400; k++) if (nz % k == 0 && (nz / k) >= 8 && !((nz / k) & 1)) best = k;
        const size_t nreg = (size_t)nGrids * (ncx / gx) * (ncy / gy) * best;
        plan.ownPartial.resize(nreg * RX * RY * (nz / best + 4) * accBytes);
        plan.ownBusy.resize(nreg); plan.strays.resize(std::max(Npad, 1));
        plan.ownSlabs = best; plan.ownMargin = M;
    }

    void execute(int includeForces, int includeEnergy, int includeDirect, int includeRecip, double* energyOut) override {
        (void)includeForces;
        if (includeEnergy == 2 && energyOut) { err = "snb_execute: include_energy == 2 (derivative-only step) delivers no total energy; pass energy = NULL"; throw (int)SNB_ERR_INVALID_ARGUMENT; }
        if (!haveParticles) throw HipError{"snb_execute: particles were not set"};
        if (!havePositions) throw HipError{"snb_execute: positions were not set"};
        if (isPeriodic()) {
            if (!haveBox) throw HipError{"snb_execute: box was not set"};
            const double minAllowed = 1.999999 * cfg.cutoff;
            if (box[0] < minAllowed || box[4] < minAllowed || box[8] < minAllowed) { err = "The periodic box size has decreased to less than twice the nonbonded cutoff."; throw (int)SNB_ERR_BOX_TOO_SMALL; }
        }
        if (cfg.method == SNB_Ewald && includeRecip) {
            if (cfg.kmax[0] < 1 || cfg.kmax[1] < 1 || cfg.kmax[2] < 1) { err = "Ewald: kmax must be given explicitly (snb_config.kmax)"; throw (int)SNB_ERR_INVALID_ARGUMENT; }
            if (box[3] != 0 || box[6] != 0 || box[7] != 0) { err = "SlicedNonbondedForce: Ewald is not supported with non-rectangular boxes.  Use PME instead."; throw (int)SNB_ERR_UNSUPPORTED; }
        }
        if (dLambdas.p == nullptr) setLambdas(lambdas.data());
        // rebuild_interval < 0: automatic -- rebuild when the displacement watch of the position-gather pass has seen an atom move
        // 0.8 * skin/2 since the last rebuild (the flag lives in mapped host memory: no synchronisation; it is read one execute late,
        // hence the margin), and after -rebuild_interval executes at the latest
        const bool autoMode = cfg.rebuild_interval < 0;
        // ... stream-ordered: the flag is read only after the execute before the previous one has COMPLETED on the GPU (an event recorded at
        // the end of every execute, two slots), so the host can never run far ahead of the watch -- a forces-only step is an asynchronous
        // graph launch.  The decision therefore sees every displacement up to two executes ago; the 0.8 factor leaves 0.2 * skin/2 for
        // those two steps.
        if (autoMode && cfg.neighbor_padding > 0) {
            hipEvent_t& ev = evStepDone[stepCounter & 1];
            if (ev) HIPCHECK(hipEventSynchronize(ev));
        }
        if (flagsResetPending && hipEventQuery(evFlagsReset) == hipSuccess) flagsResetPending = false;
        const bool due = autoMode ? ((hDispFlags[0] != 0 && !flagsResetPending) || stepsSinceRebuild >= -cfg.rebuild_interval) : (cfg.rebuild_interval <= 1 || stepsSinceRebuild >= cfg.rebuild_interval);
        bool rebuilding = needRebuild || paramsDirty || due || cfg.neighbor_padding <= 0;
        if (sidePending) {
            // anything but the positions changed since the side build started: its list is of no use
            if (needRebuild || paramsDirty || staticDirty || valuesDirty || excValuesDirty || cfg.neighbor_padding <= 0) cancelSideBuild();
            else if (rebuilding && finishSideBuild()) rebuilding = false;
        }
        if ((valuesDirty || excValuesDirty) && !staticDirty && !rebuilding) refreshValues();
        if (rebuilding) { if (valuesDirty || excValuesDirty) { staticDirty = true; valuesDirty = excValuesDirty = false; } rebuild(); }
        stepsSinceRebuild++;
        outputWritten = outPtr != nullptr;
        const bool energy = includeEnergy != 0;
        energySelective = includeEnergy == 2;      // derivative-only step: only the slices named by snb_set_energy_slices
        lastRecip = includeRecip && (isPme() || cfg.method == SNB_Ewald);
        // Forces-only steps replay a captured hipGraph (the ~14 small launches of a step are host-launch-bound otherwise:
        // 7-8 us of idle GPU between kernels).  Every 32nd step -- and every energy step -- is enqueued eagerly with HIP events
        // around the pair kernel and the reciprocal pipeline; those samples feed snb_stats' kernel timers.
        static const bool noStepGraph = getenv("SNB_NO_STEP_GRAPH") != nullptr;      // measurement aid: every step as plain launches
        // (the step right after a rebuild goes out as plain launches: the GPU is idle at that point -- the rebuild ended with a host
        // read-back -- and capturing + instantiating the step graph first would keep it idle for another ~50 us; the capture then happens
        // at the next step, while this one is executing)
        // Energy steps (per-slice energies, the step of every force with energy-parameter derivatives, Q4) are graph steps like any other:
        // they end with the device-side sum of the slice-energy partitions and leave the result there until it is asked for.
        // (round 4: a rebuild step replays its graph too when one exists for these buffers -- refreshing it costs the host ~20 us of capture +
        // hipGraphExecUpdate, against ~40 us of launch gaps of an eager step, and an overlapped step is 37 us shorter than a serial one)
        const GraphKey stepKey{devUserPos, posIsDouble, posStride4, includeDirect != 0, includeRecip != 0, energy ? (energySelective ? 2 : 1) : 0, outPtr, outIsDouble, outAccumulate};
        bool haveGraph = false;
        for (auto& g : graphs) if (g.key == stepKey && g.exec) haveGraph = true;
        static const bool eagerRebuildSteps = getenv("SNB_EAGER_REBUILD_STEP") != nullptr;      // test switch: the rebuild step as plain launches (rounds 1-3)
        const bool eager = cfg.disable_graph || noStepGraph || (rebuilding && (!haveGraph || eagerRebuildSteps)) || (timingInterval > 0 && !sidePending && execCount++ % timingInterval == 0);      // (no timed step while a list is being built beside it: the kernel timers are for kernels running alone)
        if (eager) {
            EvSet& ev = ring[ringPos]; ringPos = (ringPos + 1) % RING;
            if (ev.pending) harvest(ev);
            enqueueStep(energy, includeDirect != 0, includeRecip != 0, &ev);
            ev.pending = true;
        } else {
            const GraphKey& key = stepKey;
            hipGraphExec_t graphExec = nullptr;
            CachedGraph* cached = nullptr;
            for (auto& g : graphs) if (g.key == key) { cached = &g; if (!g.stale) graphExec = g.exec; break; }
            if (!graphExec) {
                if (!stream2 && (concurrentPme || overlapMode)) {   // created outside the capture
                    int lo = 0, hi = 0;
                    HIPCHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
                    HIPCHECK(hipStreamCreateWithPriority(&stream2, hipStreamNonBlocking, getenv("SNB_PME_PRIO_LOW") ? lo : (getenv("SNB_PME_PRIO_HIGH") ? hi : (lo + hi) / 2)));
                    HIPCHECK(hipEventCreateWithFlags(&evFork, hipEventDisableTiming)); HIPCHECK(hipEventCreateWithFlags(&evJoin, hipEventDisableTiming)); HIPCHECK(hipEventCreateWithFlags(&evPairA, hipEventDisableTiming));
                }
                hipGraph_t graph = nullptr;
                const auto tc0 = std::chrono::steady_clock::now();
                HIPCHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
                try { enqueueStep(energy, includeDirect != 0, includeRecip != 0, nullptr); }
                catch (...) { (void)hipStreamEndCapture(stream, &graph); if (graph) (void)hipGraphDestroy(graph); throw; }
                HIPCHECK(hipStreamEndCapture(stream, &graph));
                const auto tc1 = std::chrono::steady_clock::now();
                bool updated = false;
                static const bool noUpdate = getenv("SNB_NO_GRAPH_UPDATE") != nullptr;      // test switch: a new instantiation after every rebuild, as before round 4
                if (cached && cached->exec && !noUpdate) {
                    hipGraphNode_t errNode = nullptr; hipGraphExecUpdateResult res;
                    if (hipGraphExecUpdate(cached->exec, graph, &errNode, &res) == hipSuccess) { graphExec = cached->exec; cached->stale = false; updated = true; }
                    else (void)hipGetLastError();
                }
                if (!updated) {
                    HIPCHECK(hipGraphInstantiate(&graphExec, graph, nullptr, nullptr, 0));
                    if (cached) { if (cached->exec) (void)hipGraphExecDestroy(cached->exec); cached->exec = graphExec; cached->stale = false; }
                    else if (graphs.size() < MAX_GRAPHS) graphs.push_back({key, graphExec, false});
                    else { (void)hipGraphExecDestroy(graphs[graphVictim].exec); graphs[graphVictim] = {key, graphExec, false}; graphVictim = (graphVictim + 1) % MAX_GRAPHS; }
                }
                HIPCHECK(hipGraphDestroy(graph));
                { static const bool verbose = getenv("SNB_VERBOSE") != nullptr;
                  if (verbose) fprintf(stderr, "[snb] step graph: capture %.0f us, %s %.0f us (host)\n", std::chrono::duration<double, std::micro>(tc1 - tc0).count(), updated ? "update" : "instantiate",
                                       std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tc1).count()); }
            }
            { static const bool verbose = getenv("SNB_VERBOSE") != nullptr;
              if (verbose && rebuilding) {
                  const auto tl0 = std::chrono::steady_clock::now();
                  HIPCHECK(hipGraphLaunch(graphExec, stream));
                  const auto tl1 = std::chrono::steady_clock::now();
                  HIPCHECK(hipStreamSynchronize(stream));
                  fprintf(stderr, "[snb] rebuild step: graph launch %.0f us (host), step done after %.0f us\n", std::chrono::duration<double, std::micro>(tl1 - tl0).count(),
                          std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tl0).count());
              } else HIPCHECK(hipGraphLaunch(graphExec, stream)); }
            if (overlapMode && dOverlap.p) { static int dbg = getenv("SNB_OVERLAP_DEBUG") ? 3 : 0; if (dbg > 0) { dbg--; dumpOverlapTable(); } }
        }
        if (autoMode && cfg.neighbor_padding > 0) {
            hipEvent_t& ev = evStepDone[stepCounter & 1];
            if (!ev) HIPCHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            HIPCHECK(hipEventRecord(ev, stream));
        }
        stepCounter++;
        // the rebuild that falls due sideLead executes from now starts here, beside the steps, from the positions of the step just enqueued
        if (!autoMode && stepsSinceRebuild == cfg.rebuild_interval - sideLead && sideBuildPossible()) startSideBuild();
        if (energy) {
            energyPending = true;
            if (energyOut) { fetchSliceEnergies(); double e = 0; for (int i = 0; i < 2 * S; i++) e += lambdas[i] * hostSliceE[i]; *energyOut = e; }      // (synchronises)
        } else if (energyOut) *energyOut = 0.0;
    }

    // SNB_OVERLAP_DEBUG: how the limited launch of the last overlapped step spread over the physical CUs (synchronises)
    void dumpOverlapTable() {
        std::vector<int> h(SNB_OVERLAP_INTS);
        HIPCHECK(hipStreamSynchronize(stream));
        HIPCHECK(hipMemcpy(h.data(), dOverlap.p, sizeof(int) * h.size(), hipMemcpyDeviceToHost));
        int keys = 0, stayed = 0, arrived = 0, hist[8] = {0}; unsigned orKey = 0;
        const int* cu = h.data() + SNB_WORK_SHARDS * 32; int claims = 0; for (int k = 0; k < SNB_WORK_SHARDS; k++) claims += h[k * 32];
        int stayHist[8] = {0};
        for (int k = 0; k < SNB_CU_SLOTS; k++) if (cu[k] > 0) { const int arr = cu[k] & 0xFFFF, st = cu[k] >> 16; keys++; arrived += arr; stayed += st; hist[std::min(arr, 7)]++; stayHist[std::min(st, 7)]++; orKey |= (unsigned)k; }
        fprintf(stderr, "[snb] overlap: items claimed %d of %d; %d CU keys seen (or of keys 0x%x), %d work-groups arrived, %d stayed; arrivals per key 1..7+:", claims, numWorkItems, keys, orKey, arrived, stayed);
        for (int k = 1; k < 8; k++) fprintf(stderr, " %d", hist[k]);
        fprintf(stderr, "; CUs with 0..4 resident:");
        for (int k = 0; k < 5; k++) fprintf(stderr, " %d", stayHist[k]);
        fprintf(stderr, "\n");
        if (dOverlapTrace.p) {
            std::vector<int> t(SNB_CU_SLOTS * 8);
            HIPCHECK(hipMemcpy(t.data(), dOverlapTrace.p, sizeof(int) * t.size(), hipMemcpyDeviceToHost));
            int shown = 0;
            for (int k = 0; k < SNB_CU_SLOTS && shown < 12; k++) if (cu[k] > 0) {
                fprintf(stderr, "[snb] overlap: CU key 0x%03x arrivals %d, stayed %d: (GPR_ALLOC, LDS_ALLOC) of the first four:", k, cu[k] & 0xFFFF, cu[k] >> 16);
                for (int a = 0; a < 4; a++) fprintf(stderr, " (%08x, %08x)", (unsigned)t[(k * 4 + a) * 2], (unsigned)t[(k * 4 + a) * 2 + 1]);
                fprintf(stderr, "\n"); shown++;
            }
        }
    }

    // The raw slice energies of the last energy step, read back on demand (the only synchronisation of an energy / derivative step)
    void fetchSliceEnergies() {
        if (!energyPending) return;
        std::vector<double> dev((size_t)S * 2);
        HIPCHECK(hipMemcpyAsync(dev.data(), sliceTotal.p, sizeof(double) * dev.size(), hipMemcpyDeviceToHost, stream));
        HIPCHECK(hipStreamSynchronize(stream));
        hostSliceE = dev;
        energyPending = false;
    }

    void enqueueStep(bool energy, bool includeDirect, bool includeRecip, EvSet* ev) {
        struct StampScope { StampScope(KernelStamps* k) { g_stamps = k; } ~StampScope() { g_stamps = nullptr; } };
        if (ev) for (int k = 0; k < 16; k++) ev->ks.used[k] = false;
        static const bool noStamps = getenv("SNB_NO_KERNEL_STAMPS") != nullptr;      // measurement aid: only the pair-kernel / pipeline timers
        // (a stamped launch completes a signal of its own: ~8 us per kernel, 70 us per step with every PME kernel stamped -- measured: 20-step
        // region 0.484 ms per step against 0.467 without them.  Every third eager step carries the per-kernel stamps, starting with the
        // first one after snb_reset_timers; the pair-kernel and pipeline timers keep every eager step.)
        const bool fullStamps = ev && !noStamps && (stampCounter++ % 3 == 0);
        StampScope stampScope(fullStamps ? &ev->ks : nullptr);
        if (ev) HIPCHECK(hipEventRecord(ev->e[0], stream));
        // one pass: sorted positions, cleared force arrays, and (PME on the brick path) the packed Coulomb-mesh cell of every atom
        GatherCells<Real> gc;
        std::memset(&gc, 0, sizeof(gc));
        cellsFromGather = false;
        if (includeRecip && isPme() && nGrids > 0) {
            PmeParams<Real> pp;
            std::memset(&pp, 0, sizeof(pp));
            fillPme(pp, pme, false);
            if (pp.sortNcx > 0 && pp.colRange != nullptr) {
                for (int i = 0; i < 9; i++) { gc.recip[i] = pp.recip[i]; gc.recipLo[i] = pp.recipLo[i]; }
                gc.nx = pp.d.nx; gc.ny = pp.d.ny; gc.nz = pp.d.nz; gc.cells = pp.cells; gc.atomGrid = pp.atomGrid;
                cellsFromGather = true;
            }
        }
        if (cfg.neighbor_padding > 0 && posRef.p) {
            gc.posRef = posRef.p; gc.flags = dDispFlags;
            const double half = 0.5 * cfg.neighbor_padding;
            gc.fail2 = (Real)(half * half); gc.warn2 = (Real)(0.64 * half * half);      // rebuild request at 80 % of skin/2: the flag is read one step late
        }
        if (energy && Npad > 0) { gc.clearE = sliceE.p; gc.nClearE = S * 2 * SNB_SLICE_E_PARTS; }
        if (includeRecip && isPme() && dStrayCount.p) { gc.zeroInts = dStrayCount.p; gc.nZeroInts = 2; }
        if (!ev && overlapMode && dOverlap.p) { gc.zeroInts2 = dOverlap.p; gc.nZeroInts2 = SNB_OVERLAP_INTS; }
        traceThisStep = !ev && dStepTrace.p != nullptr;      // (SNB_STEP_TRACE; replayed steps only: the stamps of the last one are printed when the engine is destroyed)
        gc.stepTrace = traceThisStep ? dStepTrace.p : nullptr;
        launchGatherPositions<Real>(devUserPos, posIsDouble, posStride4, dSortedToUser.p, imageOffset.p, posq.p, Npad, forceBuf.p, forceArrays(), gc, stream);
        if (energy && Npad <= 0) launchZeroFill(sliceE.p, sizeof(double) * S * 2 * SNB_SLICE_E_PARTS, stream);      // (inside the step graph: a kernel, not a memset node)
        const bool ew = cfg.method >= SNB_Ewald;
        // Opt-in (SNB_CONCURRENT_PME=1): forces-only graph steps run the reciprocal pipeline on a second stream beside the pair
        // kernel (disjoint force arrays fx.. / fpx..).  Timed (eager) steps stay serial so the per-kernel event timers stay clean.
        // overlapped step (see overlapMode above): any graph step with both halves; needs the GPU-built work list (static item order is irrelevant)
        const bool overlap = !ev && overlapMode && includeDirect && includeRecip && isPme() && nGrids > 0 && stream2 && dOverlap.p && numWorkItems > 0 && shardTiles >= overlapMinTiles;
        const bool fork = overlap || (!ev && !energy && includeDirect && includeRecip && isPme() && nGrids > 0 && concurrentPme && stream2);
        hipStream_t pmeStream = stream;
        if (fork) {
            HIPCHECK(hipEventRecord(evFork, stream));
            HIPCHECK(hipStreamWaitEvent(stream2, evFork, 0));
            pmeStream = stream2;
        }
        PairListParams<Real> q;
        std::memset(&q, 0, sizeof(q));
        // O(N) pair lists: one rank only when sharded -- the LAST one, which carries no PME grid once there are more ranks than grids
        const bool haveLists = includeDirect && cfg.shard_rank == cfg.shard_count - 1;
        if (haveLists) {
            q.posq = posq.p; q.fx = fx.p; q.fy = fy.p; q.fz = fz.p; q.fs = fstride; q.fixed = fixedForces(); q.sliceE = sliceE.p; q.lambdas = dLambdas.p; q.sliceNeed = energySelective ? dSliceNeedSel.p : dSliceNeedAll.p;
            const bool exPeriodic = (cfg.method == SNB_NoCutoff || cfg.method == SNB_CutoffNonPeriodic) ? false : cfg.exceptions_periodic != 0;
            q.periodic = exPeriodic ? 1 : 0; q.imageOffset = imageOffset.p;
            q.sigeps = sigeps.p; q.blockSubset = blockSubset.p; q.exclStart = exclStart.p; q.exclList = exclList.p; q.nSlices = S; q.sortedToUser = dSortedToUser.p; q.userToSorted = dUserToSorted.p;
            for (int i = 0; i < 9; i++) q.box[i] = (Real)box[i];
            q.alpha = (Real)cfg.alpha; q.alpha64 = cfg.alpha; q.alphaD = (Real)cfg.alpha_d; q.ljpme = cfg.method == SNB_LJPME;
            q.pairs = pairs14.p; q.params = params14.p; q.n = n14;
            q.nExclAtoms = (ew && nExcl > 0) ? Npad : 0;
        }
        bool listsDone = !haveLists, kernelTimed = false, finished = false, energyFinished = false;
        DirectParams<Real> directB; int directMc = 0;      // overlapped step: the second launch of the tile kernel
        std::memset(&directB, 0, sizeof(directB));
        if (includeDirect) {
            DirectParams<Real> p;
            std::memset(&p, 0, sizeof(p));
            p.posq = posq.p; p.sigeps = sigeps.p; p.blockSubset = blockSubset.p; p.workItems = workItems.p;
            p.tileJ = tileJ.p; p.tileInfo = tileInfo.p; p.masks = masks.p; p.fx = fx.p; p.fy = fy.p; p.fz = fz.p; p.fs = fstride; p.fixed = fixedForces(); p.sliceE = sliceE.p; p.lambdas = dLambdas.p; p.sliceNeed = energySelective ? dSliceNeedSel.p : dSliceNeedAll.p;
            const int r = cfg.shard_rank, c = cfg.shard_count;
            (void)r; (void)c;
            p.workStart = 0; p.workStride = 1; p.numWork = numWorkItems;      // the lists hold only the i-blocks this engine owns (block % shard_count == shard_rank)
            p.nsub = nsub;
            p.cutoff2 = (Real)(cfg.cutoff * cfg.cutoff);
            p.krf = (Real)(std::pow(cfg.cutoff, -3.0) * (cfg.rf_dielectric - 1.0) / (2.0 * cfg.rf_dielectric + 1.0));
            p.crf = (Real)((1.0 / cfg.cutoff) * (3.0 * cfg.rf_dielectric) / (2.0 * cfg.rf_dielectric + 1.0));
            p.alpha = (Real)cfg.alpha; p.alphaD = (Real)cfg.alpha_d; p.k4pe = (Real)SNB_ONE_4PI_EPS0;
            p.alpha2l2e = (Real)(cfg.alpha * cfg.alpha * 1.4426950408889634);
            for (int i = 0; i <= EW_DEG; i++) p.ewPoly[i] = (Real)ewPoly[i];
            for (int i = 0; i < 14; i++) p.ewPolyE[i] = (Real)ewPolyE[i];
            for (int i = 0; i < 21; i++) p.dispPoly[i] = (Real)dispPoly[i];
            p.ewScale = (Real)(2.0 / ewR2Max);
            { static const bool noPoly = getenv("SNB_EWALD_ERFC") != nullptr; p.ewUsePoly = noPoly ? 0 : 1; }
            const double ic2 = 1.0 / (cfg.cutoff * cfg.cutoff), ic6 = ic2 * ic2 * ic2;
            const double dar2 = cfg.alpha_d * cfg.alpha_d * cfg.cutoff * cfg.cutoff;
            p.invCut6 = (Real)ic6; p.multShift6 = (Real)(ic6 * (1.0 - std::exp(-dar2) * (1.0 + dar2 + 0.5 * dar2 * dar2)));
            const bool sw = cfg.use_switch && cfg.method != SNB_NoCutoff && cfg.method != SNB_LJPME;
            p.useSwitch = sw ? 1 : 0; p.switchDist = (Real)cfg.switch_distance;
            p.invSwitchWidth = (Real)(sw ? 1.0 / (cfg.cutoff - cfg.switch_distance) : 0.0);
            for (int i = 0; i < 9; i++) p.box[i] = (Real)(gpuBuilt ? tileCell[i] : box[i]);
            if (isPeriodic()) { p.invBoxDiag[0] = (Real)(1.0 / box[0]); p.invBoxDiag[1] = (Real)(1.0 / box[4]); p.invBoxDiag[2] = (Real)(1.0 / box[8]); }
            p.boxDiag[0] = (Real)box[0]; p.boxDiag[1] = (Real)box[4]; p.boxDiag[2] = (Real)box[8];
            int mc = MC_NOCUTOFF;
            if (cfg.method == SNB_CutoffNonPeriodic || cfg.method == SNB_CutoffPeriodic) mc = MC_RF;
            else if (cfg.method == SNB_Ewald || cfg.method == SNB_PME) mc = MC_EWALD;
            else if (cfg.method == SNB_LJPME) mc = MC_LJPME;
            static const bool noFuse = getenv("SNB_NO_FUSED_LISTS") != nullptr;
            p.stepTrace = traceThisStep ? dStepTrace.p : nullptr; p.traceSlot = 2;
            if (overlap) {      // first launch: resident beside the reciprocal pipeline, at most overlapCuLimit work-groups per CU
                p.workCounter = dOverlap.p; p.cuSlots = dOverlap.p + SNB_WORK_SHARDS * 32; p.cuLimit = overlapCuLimit;
                { static const bool byCount = getenv("SNB_OVERLAP_BY_COUNT") != nullptr; p.cuBaseMax = byCount ? -1 : 0x7fffffff; }      // (0x7fffffff: the launcher fills in the kernel's own allocation)
                p.gridCap = overlapGridA > 0 ? overlapGridA : 6 * numCUs; p.listsLast = 1;
                p.cuTrace = dOverlapTrace.p;      // (SNB_OVERLAP_DEBUG; null otherwise)
                directB = p; directMc = mc;
            }
            if (launchDirect<Real>(p, mc, wrapMode, energy, (haveLists && !noFuse) ? &q : nullptr, stream, ev ? ev->e[1] : nullptr, ev ? ev->e[2] : nullptr, &kernelTimed)) listsDone = true;
        }
        if (ev && !kernelTimed) { HIPCHECK(hipEventRecord(ev->e[1], stream)); HIPCHECK(hipEventRecord(ev->e[2], stream)); }   // no tile kernel this step
        if (!listsDone) launchPairLists<Real>(q, energy, stream);
        if (overlap) HIPCHECK(hipEventRecord(evPairA, stream));      // the first launch of the tile kernel (and the pair lists) are done
        if (ev) HIPCHECK(hipEventRecord(ev->e[3], stream));
        if (includeRecip && isPme()) {
            if (nGrids > 0) {
                PmeParams<Real> pp;
                std::memset(&pp, 0, sizeof(pp));
                // the interpolation of the step's last mesh also writes the user-order force (no k_finishForces launch): unsharded, brick path,
                // reciprocal work on the step's own stream (the pair kernel's accumulators are complete by then)
                static const bool noFuse = getenv("SNB_NO_FUSED_FINISH") != nullptr;
                // (an overlapped step keeps the fused finish: its last interpolation waits for both launches of the tile kernel)
                const bool canFinish = outPtr && (!fork || overlap) && !noFuse && cfg.shard_count == 1;
                static const bool noFusedE = getenv("SNB_NO_FUSED_ENERGY_FINISH") != nullptr;      // test switch: k_finishSliceEnergies as a kernel of its own
                auto withOutput = [&](PmeParams<Real>& q, bool last) {
                    q.outForces = (canFinish && last) ? outPtr : nullptr; q.outIsDouble = outIsDouble; q.outAccumulate = outAccumulate;
                    q.finOut = nullptr;
                    if (canFinish && last && energy && q.mix && !noFusedE) { q.finParts = sliceE.p; q.finOut = sliceTotal.p; q.finN = 2 * S; q.fin = makeSliceFinish(includeDirect, includeRecip); }
                    q.dfx = fx.p; q.dfy = fy.p; q.dfz = fz.p; q.dfs = fstride; q.dfixed = fixedForces(); q.sortedToUser = dSortedToUser.p;
                };
                // Overlapped step: everything up to the last mesh's inverse transform runs beside the resident first launch of the tile kernel;
                // then the second launch (unlimited: it takes what the first has not claimed, and the first keeps claiming) fills the chip,
                // and the last interpolation -- which also delivers the step's forces -- follows both.
                auto beforeLastInterpolation = [&]() {
                    if (!overlap) return;
                    directB.cuSlots = nullptr; directB.cuLimit = 0; directB.listsLast = 0; directB.traceSlot = 4; directB.gridCap = overlapGridB > 0 ? overlapGridB : 4 * numCUs;
                    bool t = false;
                    launchDirect<Real>(directB, directMc, wrapMode, energy, nullptr, stream2, nullptr, nullptr, &t);
                    HIPCHECK(hipStreamWaitEvent(stream2, evPairA, 0));
                };
                fillPme(pp, pme, energy); withOutput(pp, cfg.method != SNB_LJPME);
                runPmeFront(pp, pmeStream);
                if (cfg.method != SNB_LJPME) beforeLastInterpolation();
                finished = launchPmeInterpolate<Real>(pp, pmeStream);
                energyFinished = finished && pp.finOut != nullptr;
                if (cfg.method == SNB_LJPME) {
                    fillPme(pp, dpme, energy); withOutput(pp, true);
                    runPmeFront(pp, pmeStream);
                    beforeLastInterpolation();
                    finished = launchPmeInterpolate<Real>(pp, pmeStream);
                    energyFinished = finished && pp.finOut != nullptr;
                }
            }
        }
        if (fork) { HIPCHECK(hipEventRecord(evJoin, stream2)); HIPCHECK(hipStreamWaitEvent(stream, evJoin, 0)); }
        if (includeRecip && cfg.method == SNB_Ewald && cfg.shard_rank == 0) runEwald(energy);
        if (outPtr && !finished) {   // the step's last kernel: user-order forces into the caller's buffer (part of the graph)
            const bool recipDone = includeRecip && (isPme() || cfg.method == SNB_Ewald);
            launchFinishForces<Real>(fx.p, fy.p, fz.p, fstride, fixedForces(), recipDone ? fpx.p : nullptr, fpy.p, fpz.p, dUserToSorted.p, N, outPtr, outIsDouble, outAccumulate, stream);
        }
        if (energy) {
            const SliceFinish f = makeSliceFinish(includeDirect, includeRecip);
            if (!(finished && energyFinished))
            launchFinishSliceEnergies(sliceE.p, sliceTotal.p, 2 * S, f, stream);
        }
        if (ev) HIPCHECK(hipEventRecord(ev->e[4], stream));
    }

    // closed-form terms of the slice energies, added on the device (rank 0 only when sharded), as the reference GPU path keeps them next to its
    // kernels (CommonNonbondedSlicingKernels.cpp:618-638, 1129-1139)
    SliceFinish makeSliceFinish(bool includeDirect, bool includeRecip) const {
        SliceFinish f; std::memset(&f, 0, sizeof(f));
        if (cfg.shard_rank == 0) {
            const double volume = box[0] * box[4] * box[8];
            if (includeRecip && cfg.method >= SNB_Ewald) {
                f.sums = dParamSums.p;
                f.selfCoulomb = -SNB_ONE_4PI_EPS0 * cfg.alpha / std::sqrt(SNB_PI);
                f.selfDispersion = cfg.method == SNB_LJPME ? std::pow(cfg.alpha_d, 6.0) / 12.0 : 0.0;
                f.background = (-1.0 / (4 * cfg.alpha * cfg.alpha)) / (2 * SNB_EPSILON0 * volume);
            }
            if (includeDirect && (cfg.method == SNB_CutoffPeriodic || cfg.method == SNB_Ewald || cfg.method == SNB_PME)) { f.dispCoef = dDispCoef.p; f.invVolume = 1.0 / volume; }
        }
        return f;
    }

    // classic Ewald: half-space k-vectors in the reference's enumeration order (ReferenceSlicedLJCoulombIxn.cpp:288-355)
    void runEwald(bool energy) {
        if (hKvec.empty()) {
            int lowry = 0, lowrz = 1;
            for (int rx = 0; rx < cfg.kmax[0]; rx++) {
                for (int ry = lowry; ry < cfg.kmax[1]; ry++) {
                    for (int rz = lowrz; rz < cfg.kmax[2]; rz++) { hKvec.push_back(make_int3(rx, ry, rz)); lowrz = 1 - cfg.kmax[2]; }
                    lowry = 1 - cfg.kmax[1];
                }
            }
            dKvec.upload(hKvec, stream);
            dCosSin.resize(hKvec.size() * 2 * nsub);
        }
        EwaldParams<Real> q;
        std::memset(&q, 0, sizeof(q));
        q.natoms = Npad; q.nsub = nsub; q.nk = (int)hKvec.size(); q.posq = posq.p; q.atomSubset = atomSubset.p; q.kvec = dKvec.p; q.cosSin = dCosSin.p;
        q.recipBox[0] = (Real)(2 * SNB_PI / box[0]); q.recipBox[1] = (Real)(2 * SNB_PI / box[4]); q.recipBox[2] = (Real)(2 * SNB_PI / box[8]);
        q.factorEwald = -1 / (4 * cfg.alpha * cfg.alpha); q.recipCoeff = SNB_ONE_4PI_EPS0 * 4 * SNB_PI / (box[0] * box[4] * box[8]);
        q.lambdas = dLambdas.p; q.sliceE = sliceE.p; q.wantEnergy = energy ? 1 : 0; q.fpx = fpx.p; q.fpy = fpy.p; q.fpz = fpz.p;
        launchEwald<Real>(q, stream);
    }

    void harvest(EvSet& ev) {
        HIPCHECK(hipEventSynchronize(ev.e[4]));
        float a = 0, b = 0, c = 0;
        HIPCHECK(hipEventElapsedTime(&a, ev.e[1], ev.e[2]));
        HIPCHECK(hipEventElapsedTime(&b, ev.e[3], ev.e[4]));
        HIPCHECK(hipEventElapsedTime(&c, ev.e[0], ev.e[4]));
        stats.last_direct_ms = a; stats.last_recip_ms = b; stats.last_total_ms = c;
        stats.sum_direct_ms += a; stats.sum_recip_ms += b; stats.sum_total_ms += c; stats.n_timed++;
        for (int k = 0; k < 16; k++) if (ev.ks.used[k]) {
            float t = 0;
            if (hipEventElapsedTime(&t, ev.ks.start[k], ev.ks.stop[k]) == hipSuccess) { stats.sum_kernel_ms[k] += t; stats.n_kernel_timed[k]++; }
            ev.ks.used[k] = false;
        }
        ev.pending = false;
    }
    // (also restarts the eager-step cadence: the first step after a reset is a timed one, so even a short measured region has a sample)
    void resetTimers() override { for (auto& r : ring) if (r.pending) harvest(r); stats.sum_direct_ms = stats.sum_recip_ms = stats.sum_total_ms = 0; stats.n_timed = 0; for (int k = 0; k < 16; k++) { stats.sum_kernel_ms[k] = 0; stats.n_kernel_timed[k] = 0; } execCount = 0; stampCounter = 0; }

    void runPmeFront(PmeParams<Real>& pp, hipStream_t st) {      // one mesh up to its potentials in real space; launchPmeInterpolate follows
        const int zDone = launchPmeSpread<Real>(pp, st);
        if (zDone == 2) launchPmePlanePath<Real>(pp, st);      // per-plane x / y transforms + convolution in LDS, then mix + inverse z
        else {
            launchPmeForwardFFT<Real>(pp, st, zDone == 1);
            launchPmeConvolution<Real>(pp, st);
            launchPmeInverseFFT<Real>(pp, st);
        }
    }

    void setForceOutput(void* out, int isDouble, int accumulate) override { outPtr = out; outIsDouble = isDouble; outAccumulate = accumulate; outputWritten = false; }
    void getForces(void* out, int isDevice, int isDouble, int accumulate) override {
        if (isDevice && out == outPtr && isDouble == outIsDouble && outputWritten) return;   // the last execute already delivered them there
        const size_t bytes = (size_t)N * 3 * (isDouble ? 8 : 4);
        const Real* px = lastRecip ? fpx.p : nullptr;
        if (isDevice) { launchFinishForces<Real>(fx.p, fy.p, fz.p, fstride, fixedForces(), px, fpy.p, fpz.p, dUserToSorted.p, N, out, isDouble, accumulate, stream); return; }
        DevBuf<unsigned char> tmp; tmp.resize(bytes);
        if (accumulate) HIPCHECK(hipMemcpyAsync(tmp.p, out, bytes, hipMemcpyHostToDevice, stream));
        launchFinishForces<Real>(fx.p, fy.p, fz.p, fstride, fixedForces(), px, fpy.p, fpz.p, dUserToSorted.p, N, tmp.p, isDouble, accumulate, stream);
        HIPCHECK(hipMemcpyAsync(out, tmp.p, bytes, hipMemcpyDeviceToHost, stream));
        HIPCHECK(hipStreamSynchronize(stream));
    }
    const double* sliceEnergiesDevice() override { return sliceTotal.p; }
    void getSliceEnergies(double* out) override { fetchSliceEnergies(); std::memcpy(out, hostSliceE.data(), sizeof(double) * S * 2); }
    void getStats(snb_stats* o) override {
        HIPCHECK(hipStreamSynchronize(stream));
        stats.n_tiles = 0;
        // tiles processed by this shard
        stats.n_list_overruns = listOverruns + (hDispFlags ? hDispFlags[4] + (hDispFlags[1] ? 1 : 0) : 0);
        stats.n_tiles = shardTiles; stats.n_blocks = numBlocks; stats.n_padded_atoms = Npad; stats.n_exclusion_tiles = numMaskTiles;
        for (int d = 0; d < 3; d++) { stats.grid[d] = isPme() ? pme.d.nx * (d == 0) + pme.d.ny * (d == 1) + pme.d.nz * (d == 2) : 0; stats.dgrid[d] = cfg.method == SNB_LJPME ? dpme.d.nx * (d == 0) + dpme.d.ny * (d == 1) + dpme.d.nz * (d == 2) : 0; }
        for (int k = 0; k < RING; k++) { EvSet& r = ring[(ringPos + k) % RING]; if (r.pending) harvest(r); }
        stats.n_spread_strays = 0;
        if (dStrayCount.p) { int h[2] = {0, 0}; HIPCHECK(hipMemcpy(h, dStrayCount.p, sizeof(h), hipMemcpyDeviceToHost)); stats.n_spread_strays = (int64_t)h[0] + h[1]; }
        *o = stats;
    }
    void getPme(double* alpha, int32_t* g, bool dispersion) override {
        if (dispersion) { *alpha = cfg.alpha_d; g[0] = dpme.d.nx; g[1] = dpme.d.ny; g[2] = dpme.d.nz; }
        else { *alpha = cfg.alpha; g[0] = pme.d.nx; g[1] = pme.d.ny; g[2] = pme.d.nz; }
    }
};

template <typename Real> static void testFFT(int device, int batch, int nx, int ny, int nz, const double* in, double* spectrum, double* roundtrip) {
    HIPCHECK(hipSetDevice(device));
    hipStream_t s; HIPCHECK(hipStreamCreate(&s));
    {
        PmePlan<Real> plan; int g[3] = {nx, ny, nz};
        plan.init(g, batch, s);
        PmeParams<Real> p; std::memset(&p, 0, sizeof(p));
        p.d = plan.d; p.nsub = batch; p.gridReal = plan.gridReal.p; p.gridCplx = plan.gridCplx.p; p.twx = plan.twx.p; p.twy = plan.twy.p; p.twz = plan.twz.p;
        const size_t nr = (size_t)batch * nx * ny * nz, ncx = (size_t)batch * nx * ny * (nz / 2 + 1);
        std::vector<Real> h(nr);
        for (size_t i = 0; i < nr; i++) h[i] = (Real)in[i];
        HIPCHECK(hipMemcpyAsync(plan.gridReal.p, h.data(), sizeof(Real) * nr, hipMemcpyHostToDevice, s));
        launchPmeForwardFFT<Real>(p, s, false);
        launchPmeFFTX<Real>(p, -1, s);
        std::vector<typename Vec<Real>::T2> hc(ncx);
        HIPCHECK(hipMemcpyAsync(hc.data(), plan.gridCplx.p, sizeof(typename Vec<Real>::T2) * ncx, hipMemcpyDeviceToHost, s));
        HIPCHECK(hipStreamSynchronize(s));
        for (size_t i = 0; i < ncx; i++) { spectrum[2 * i] = hc[i].x; spectrum[2 * i + 1] = hc[i].y; }
        launchPmeFFTX<Real>(p, +1, s);
        launchPmeInverseFFT<Real>(p, s);
        HIPCHECK(hipMemcpyAsync(h.data(), plan.gridReal.p, sizeof(Real) * nr, hipMemcpyDeviceToHost, s));
        HIPCHECK(hipStreamSynchronize(s));
        for (size_t i = 0; i < nr; i++) roundtrip[i] = h[i];
    }
    HIPCHECK(hipStreamDestroy(s));
}

}  // namespace snb

// =====================================================================================================
// C ABI
// =====================================================================================================
using namespace snb;

struct snb_engine { EngineBase* impl; };

// Every entry point runs on the engine's own device and leaves the caller's current device as it found it (a process that drives
// several GPUs -- torch, OpenMM with several contexts -- switches the current device between calls).
struct DeviceScope {
    int prev = -1; bool switched = false;
    explicit DeviceScope(int dev) { if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess; }
    ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
};

template <typename F> static snb_status guard(snb_handle h, F&& f) {
    if (!h || !h->impl) return SNB_ERR_INVALID_ARGUMENT;
    DeviceScope scope(h->impl->cfg.device);
    try { f(); return SNB_OK; }
    catch (HipError& e) { h->impl->err = e.msg; return e.msg.find("hip") == 0 ? SNB_ERR_HIP : SNB_ERR_INVALID_ARGUMENT; }
    catch (int code) { return (snb_status)code; }
    catch (std::exception& e) { h->impl->err = e.what(); return SNB_ERR_INVALID_ARGUMENT; }
}

extern "C" {

int32_t snb_abi_version(void) { return SNB_ABI_VERSION; }
int32_t snb_legal_grid_size(int32_t n) { return legalGridSize(n); }

snb_status snb_create(const snb_config* cfg, snb_handle* out) {
    if (!cfg || !out) { g_createError = "null argument"; return SNB_ERR_INVALID_ARGUMENT; }
    *out = nullptr;
    if (cfg->abi_version != SNB_ABI_VERSION) { g_createError = "snb_config.abi_version mismatch"; return SNB_ERR_INVALID_ARGUMENT; }
    if (cfg->n_atoms < 0 || cfg->n_subsets < 1 || cfg->method < 0 || cfg->method > 5 || (cfg->precision != SNB_SINGLE && cfg->precision != SNB_DOUBLE && cfg->precision != SNB_MIXED)) {
        g_createError = "invalid n_atoms / n_subsets / method / precision"; return SNB_ERR_INVALID_ARGUMENT;
    }
    if (cfg->method != SNB_NoCutoff && !(cfg->cutoff > 0)) { g_createError = "cutoff must be positive"; return SNB_ERR_INVALID_ARGUMENT; }
    if (cfg->method >= SNB_Ewald && !(cfg->alpha > 0)) { g_createError = "alpha must be given explicitly for Ewald/PME/LJPME"; return SNB_ERR_INVALID_ARGUMENT; }
    if ((cfg->method == SNB_PME || cfg->method == SNB_LJPME) && (cfg->grid[0] < 1 || cfg->grid[1] < 1 || cfg->grid[2] < 1)) { g_createError = "PME grid must be given explicitly"; return SNB_ERR_INVALID_ARGUMENT; }
    if (cfg->method == SNB_LJPME && (!(cfg->alpha_d > 0) || cfg->dgrid[0] < 1 || cfg->dgrid[1] < 1 || cfg->dgrid[2] < 1)) { g_createError = "LJPME dispersion alpha/grid must be given explicitly"; return SNB_ERR_INVALID_ARGUMENT; }
    if (cfg->shard_count > 1 && (cfg->shard_rank < 0 || cfg->shard_rank >= cfg->shard_count)) { g_createError = "invalid shard_rank"; return SNB_ERR_INVALID_ARGUMENT; }
    // every shard must re-sort and rebuild on the SAME execute (sorted order, padded count and block ownership are per-rebuild facts all ranks share);
    // the displacement-triggered mode decides from a flag each rank sees at its own time, so sharded engines take a fixed interval (or
    // the caller agrees on the step itself and calls snb_rebuild_neighbors on every rank)
    if (cfg->shard_count > 1 && cfg->rebuild_interval < 0) { g_createError = "rebuild_interval < 0 (displacement-triggered rebuilds) is not available with shard_count > 1"; return SNB_ERR_UNSUPPORTED; }
    try {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_createError = "hip: no HIP device available (the engine has no CPU fallback)"; return SNB_ERR_HIP; }
        if (cfg->device < 0 || cfg->device >= ndev) { g_createError = "invalid device ordinal"; return SNB_ERR_INVALID_ARGUMENT; }
        DeviceScope scope(cfg->device);
        EngineBase* e = cfg->precision == SNB_DOUBLE ? (EngineBase*)new Engine<double>(*cfg) : (EngineBase*)new Engine<float>(*cfg);
        *out = new snb_engine{e};
        return SNB_OK;
    } catch (HipError& e) { g_createError = e.msg; return SNB_ERR_HIP; }
    catch (std::exception& e) { g_createError = e.what(); return SNB_ERR_INVALID_ARGUMENT; }
}

void snb_destroy(snb_handle h) { if (h) { if (h->impl) { DeviceScope scope(h->impl->cfg.device); delete h->impl; } delete h; } }
const char* snb_last_error(snb_handle h) { return h && h->impl ? h->impl->err.c_str() : g_createError.c_str(); }

snb_status snb_set_particles(snb_handle h, const double* q, const double* s, const double* e, const int32_t* sub) {
    if (!q || !s || !e || !sub) return SNB_ERR_INVALID_ARGUMENT;
    return guard(h, [&] { h->impl->setParticles(q, s, e, sub); });
}
snb_status snb_set_exceptions(snb_handle h, int32_t m, const int32_t* pairs, const double* qq, const double* s, const double* e, const int32_t* f14) {
    if (m < 0 || (m > 0 && (!pairs || !qq || !s || !e))) return SNB_ERR_INVALID_ARGUMENT;
    return guard(h, [&] { h->impl->setExceptions(m, pairs, qq, s, e, f14); });
}
snb_status snb_set_parameter_offsets(snb_handle h, int32_t nGlobals, int32_t nP, const int32_t* particle, const int32_t* pGlobal, const double* pDelta,
                                     int32_t nE, const int32_t* exception, const int32_t* eGlobal, const double* eDelta) {
    if ((nP > 0 && (!particle || !pGlobal || !pDelta)) || (nE > 0 && (!exception || !eGlobal || !eDelta))) return SNB_ERR_INVALID_ARGUMENT;
    return guard(h, [&] { h->impl->setParameterOffsets(nGlobals, nP, particle, pGlobal, pDelta, nE, exception, eGlobal, eDelta); });
}
snb_status snb_set_global_parameters(snb_handle h, int32_t n, const double* values) {
    if (n < 0 || (n > 0 && !values)) return SNB_ERR_INVALID_ARGUMENT;
    return guard(h, [&] { h->impl->setGlobalParameters(n, values); });
}
snb_status snb_set_energy_slices(snb_handle h, const int32_t* mask) { if (!mask) return SNB_ERR_INVALID_ARGUMENT; return guard(h, [&] { h->impl->setEnergySlices(mask); }); }
snb_status snb_set_lambdas(snb_handle h, const double* l) { if (!l) return SNB_ERR_INVALID_ARGUMENT; return guard(h, [&] { h->impl->setLambdas(l); }); }
snb_status snb_set_dispersion_coefficients(snb_handle h, const double* c) { return guard(h, [&] { h->impl->setDispersion(c); }); }
snb_status snb_compute_dispersion_coefficients(int32_t n, int32_t nsub, const double* sigma, const double* epsilon, const int32_t* subset, double cutoff,
                                               int32_t useSwitch, double switchDist, double* out) {
    if (n < 0 || nsub < 1 || !out || (n > 0 && (!sigma || !epsilon || !subset))) return SNB_ERR_INVALID_ARGUMENT;
    dispersionCoefficients(n, nsub, sigma, epsilon, subset, cutoff, useSwitch, switchDist, out);
    return SNB_OK;
}
snb_status snb_set_box(snb_handle h, const double* b) { if (!b) return SNB_ERR_INVALID_ARGUMENT; return guard(h, [&] { h->impl->setBox(b); }); }
snb_status snb_set_positions(snb_handle h, const void* pos, int32_t isDevice, int32_t isDouble, int32_t stride4) {
    if (!pos) return SNB_ERR_INVALID_ARGUMENT;
    return guard(h, [&] { h->impl->setPositions(pos, isDevice, isDouble, stride4); });
}
snb_status snb_rebuild_neighbors(snb_handle h) { return guard(h, [&] { h->impl->requestRebuild(); }); }
snb_status snb_execute(snb_handle h, int32_t f, int32_t e, int32_t d, int32_t r, double* energy) { return guard(h, [&] { h->impl->execute(f, e, d, r, energy); }); }
snb_status snb_get_forces(snb_handle h, void* out, int32_t isDevice, int32_t isDouble, int32_t acc) {
    if (!out) return SNB_ERR_INVALID_ARGUMENT;
    return guard(h, [&] { h->impl->getForces(out, isDevice, isDouble, acc); });
}
snb_status snb_set_shard_blocks(snb_handle h, int32_t begin, int32_t end, int32_t period) { return guard(h, [&] { h->impl->setShardBlocks(begin, end, period); }); }
snb_status snb_set_force_output(snb_handle h, void* out, int32_t isDouble, int32_t acc) { return guard(h, [&] { h->impl->setForceOutput(out, isDouble, acc); }); }
snb_status snb_get_slice_energies(snb_handle h, double* out) { if (!out) return SNB_ERR_INVALID_ARGUMENT; return guard(h, [&] { h->impl->getSliceEnergies(out); }); }
snb_status snb_slice_energies_device(snb_handle h, const double** out) { if (!out) return SNB_ERR_INVALID_ARGUMENT; return guard(h, [&] { *out = h->impl->sliceEnergiesDevice(); }); }
snb_status snb_synchronize(snb_handle h) { return guard(h, [&] { h->impl->sync(); }); }
snb_status snb_get_pme_parameters(snb_handle h, double* alpha, int32_t grid[3]) {
    if (!h || !h->impl || !alpha || !grid) return SNB_ERR_INVALID_ARGUMENT;
    if (h->impl->cfg.method != SNB_PME && h->impl->cfg.method != SNB_LJPME) { h->impl->err = "getPMEParametersInContext: This Context is not using PME or LJPME"; return SNB_ERR_NOT_PME; }
    return guard(h, [&] { h->impl->getPme(alpha, grid, false); });
}
snb_status snb_get_ljpme_parameters(snb_handle h, double* alpha, int32_t grid[3]) {
    if (!h || !h->impl || !alpha || !grid) return SNB_ERR_INVALID_ARGUMENT;
    if (h->impl->cfg.method != SNB_LJPME) { h->impl->err = "getPMEParametersInContext: This Context is not using LJPME"; return SNB_ERR_NOT_PME; }
    return guard(h, [&] { h->impl->getPme(alpha, grid, true); });
}
snb_status snb_reset_timers(snb_handle h) { return guard(h, [&] { h->impl->resetTimers(); }); }
snb_status snb_set_timing_interval(snb_handle h, int32_t n) { return guard(h, [&] { h->impl->setTimingInterval(n); }); }
snb_status snb_get_stats(snb_handle h, snb_stats* out) { if (!out) return SNB_ERR_INVALID_ARGUMENT; return guard(h, [&] { h->impl->getStats(out); }); }

snb_status snb_test_fft3d(int32_t precision, int32_t device, int32_t batch, int32_t nx, int32_t ny, int32_t nz, const double* in, double* spectrum, double* roundtrip) {
    if (!in || !spectrum || !roundtrip || batch < 1) return SNB_ERR_INVALID_ARGUMENT;
    if (legalGridSize(nx) != nx || legalGridSize(ny) != ny || legalGridSize(nz) != nz) return SNB_ERR_UNSUPPORTED;
    try {
        if (precision == SNB_DOUBLE) testFFT<double>(device, batch, nx, ny, nz, in, spectrum, roundtrip);
        else testFFT<float>(device, batch, nx, ny, nz, in, spectrum, roundtrip);
        return SNB_OK;
    } catch (HipError& e) { g_createError = e.msg; return SNB_ERR_HIP; }
}

}  // extern "C"
