// HipNonbondedSlicingKernels.cpp -- OpenMM HIP-platform adapter for libsnb_hip.so.
//
// STATUS: written against the OpenMM 8.2+ HIP platform headers (openmm/hip/HipContext.h) and the reference's own plugin interface
// (openmmapi/include/NonbondedSlicingKernels.h:27-85), but NOT compiled in this repository: the build image has no OpenMM.  It
// contains no arithmetic of the hot path -- it maps ContextImpl state onto the C ABI of include/snb.h, exactly as
// openmm-nonbonded-slicing_amd/context.py does for the Python tests (that mirror IS tested, tests/test_gpu_parity.py), and is the
// file a maintainer drops into <reference>/platforms/hip/src/.  Build line and registration: INTEGRATION.md.
//
// Shape follows platforms/cuda/src/CudaNonbondedSlicingKernelFactory.cpp:19-54 (registration) and
// platforms/reference/src/ReferenceNonbondedSlicingKernels.cpp:58-268, 339-391 (what initialize/execute must do).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "NonbondedSlicingKernels.h"                    // NonbondedSlicing::CalcSlicedNonbondedForceKernel
#include "SlicedNonbondedForce.h"
#include "internal/SlicedNonbondedForceImpl.h"          // calcPMEParameters, calcDispersionCorrections
#include "openmm/OpenMMException.h"
#include "openmm/internal/ContextImpl.h"
#include "openmm/hip/HipContext.h"
#include "openmm/hip/HipPlatform.h"
#include "snb.h"

using namespace OpenMM;

namespace NonbondedSlicing {

// forces[N][3] (user = OpenMM atom-index order, float or double) -> OpenMM's 64-bit fixed-point force buffer, which is laid out
// [3][paddedNumAtoms] in the CONTEXT's (reordered) atom order; atomIndex[contextSlot] = user index (HipContext::getAtomIndexArray()).
template <typename Real>
__global__ void addForcesToContext(const Real* __restrict__ forces, const int* __restrict__ atomIndex, unsigned long long* __restrict__ forceBuffers,
                                   int numAtoms, int paddedNumAtoms) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= numAtoms) return;
    const int user = atomIndex[slot];
    const double scale = 4294967296.0;      // 0x100000000: OpenMM's fixed-point force scale
    for (int d = 0; d < 3; d++)
        atomicAdd(&forceBuffers[slot + d * (size_t)paddedNumAtoms], (unsigned long long)(long long)((double)forces[3 * (size_t)user + d] * scale));
}

// posq of the context (float4 or double4 per context slot; mixed precision adds posqCorrection, ignored here as the reference's own
// single-precision kernels do) -> positions[N][4] in user order
// dE/dlambda on the device: raw slice energies (double[S][2], snb_slice_energies_device) -> OpenMM's energy-parameter-derivative buffer
// (one slot per derivative in the first thread's row).  binding[k] = derivative slot of (slice, term) k, or -1.
template <typename Mixed>
__global__ void addDerivativesToContext(const double* __restrict__ sliceEnergies, const int* __restrict__ binding, int n, Mixed* __restrict__ derivBuffer) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n && binding[k] >= 0) atomicAdd(&derivBuffer[binding[k]], (Mixed)sliceEnergies[k]);
}

template <typename Real4>
__global__ void gatherUserPositions(const Real4* __restrict__ posq, const int* __restrict__ atomIndex, Real4* __restrict__ userPos, int numAtoms) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= numAtoms) return;
    userPos[atomIndex[slot]] = posq[slot];
}

class HipCalcSlicedNonbondedForceKernel : public CalcSlicedNonbondedForceKernel {
public:
    HipCalcSlicedNonbondedForceKernel(std::string name, const Platform& platform, HipContext& cu, const System& system)
        : CalcSlicedNonbondedForceKernel(name, platform), cu(cu), engine(nullptr), userPos(nullptr), userForces(nullptr), derivBinding(nullptr) {}
    ~HipCalcSlicedNonbondedForceKernel() {
        snb_destroy(engine);
        if (userPos) (void)hipFree(userPos);
        if (userForces) (void)hipFree(userForces);
        if (derivBinding) (void)hipFree(derivBinding);
    }

    void initialize(const System& system, const SlicedNonbondedForce& force) override {
        cu.setAsCurrent();
        numParticles = force.getNumParticles();
        numSubsets = force.getNumSubsets();
        numSlices = force.getNumSlices();
        useDouble = cu.getUseDoublePrecision();
        method = (int) force.getNonbondedMethod();              // enum values are identical (NonbondedSlicingKernels.h:29-36)

        snb_config cfg = {};
        cfg.abi_version = SNB_ABI_VERSION;
        cfg.n_atoms = numParticles;  cfg.n_subsets = numSubsets;  cfg.method = method;
        // Precision property of the platform: single / mixed / double.  mixed: float arithmetic, 64-bit fixed-point force sums (snb.h SNB_MIXED)
        cfg.precision = useDouble ? SNB_DOUBLE : (cu.getUseMixedPrecision() ? SNB_MIXED : SNB_SINGLE);
        cfg.use_switch = force.getUseSwitchingFunction();  cfg.switch_distance = force.getSwitchingDistance();
        cfg.cutoff = force.getCutoffDistance();  cfg.rf_dielectric = force.getReactionFieldDielectric();
        cfg.exceptions_periodic = force.getExceptionsUsePeriodicBoundaryConditions();
        cfg.device = cu.getDeviceIndex();  cfg.stream = cu.getCurrentStream();
        // Neighbour list: a skin and displacement-triggered rebuilds (rebuild_interval < 0: the engine watches every atom's displacement since
        // the last rebuild on the device and rebuilds when one has moved 0.8 * skin / 2, stream-ordered; at the latest after that many
        // steps) -- like OpenMM's own padded list, it never loses a pair however fast the atoms move.  SNB_NEIGHBOR_PADDING (nm) and
        // SNB_REBUILD_INTERVAL override (a positive interval fixes the cadence; snb_stats.n_list_overruns then reports skins outrun).
        cfg.shard_count = 1;
        cfg.neighbor_padding = getenv("SNB_NEIGHBOR_PADDING") ? atof(getenv("SNB_NEIGHBOR_PADDING")) : 0.1;
        cfg.rebuild_interval = getenv("SNB_REBUILD_INTERVAL") ? atoi(getenv("SNB_REBUILD_INTERVAL")) : -100;
        if (method == SlicedNonbondedForce::PME || method == SlicedNonbondedForce::LJPME) {
            int nx, ny, nz;
            SlicedNonbondedForceImpl::calcPMEParameters(system, force, cfg.alpha, nx, ny, nz, false);
            cfg.grid[0] = nx; cfg.grid[1] = ny; cfg.grid[2] = nz;   // the engine rounds up to its FFT's legal sizes
        }
        if (method == SlicedNonbondedForce::LJPME) {
            int nx, ny, nz;
            SlicedNonbondedForceImpl::calcPMEParameters(system, force, cfg.alpha_d, nx, ny, nz, true);
            cfg.dgrid[0] = nx; cfg.dgrid[1] = ny; cfg.dgrid[2] = nz;
        }
        if (method == SlicedNonbondedForce::Ewald) {
            int kx, ky, kz;
            SlicedNonbondedForceImpl::calcEwaldParameters(system, force, cfg.alpha, kx, ky, kz);
            cfg.kmax[0] = kx; cfg.kmax[1] = ky; cfg.kmax[2] = kz;
        }
        check(snb_create(&cfg, &engine), nullptr);

        // scaling parameters -> (slice, term) bindings with derivative flags (ReferenceNonbondedSlicingKernels.cpp:58-87)
        std::vector<std::string> derivs;
        for (int i = 0; i < force.getNumEnergyParameterDerivatives(); i++) derivs.push_back(force.getEnergyParameterDerivativeName(i));
        bindings.assign(2 * (size_t) numSlices, Binding());
        for (int k = 0; k < force.getNumScalingParameters(); k++) {
            std::string name;  int s1, s2;  bool coul, lj;
            force.getScalingParameter(k, name, s1, s2, coul, lj);
            const int i = std::max(s1, s2), j = std::min(s1, s2), slice = i * (i + 1) / 2 + j;
            const bool hasDeriv = std::find(derivs.begin(), derivs.end(), name) != derivs.end();
            if (coul) bindings[2 * slice] = Binding{name, hasDeriv};
            if (lj) bindings[2 * slice + 1] = Binding{name, hasDeriv};
            if (hasDeriv) hasDerivatives = true;
        }
        for (auto& d : derivs) cu.addEnergyParameterDerivative(d);
        {   // device-side derivative accumulation: (slice, term) -> slot of the derivative in the context's buffer
            const std::vector<std::string>& all = cu.getEnergyParamDerivNames();
            std::vector<int> slot(bindings.size(), -1);
            for (size_t k = 0; k < bindings.size(); k++)
                if (bindings[k].hasDerivative) slot[k] = (int) (std::find(all.begin(), all.end(), bindings[k].name) - all.begin());
            check(hipMalloc(&derivBinding, sizeof(int) * std::max<size_t>(slot.size(), 1)) == hipSuccess ? SNB_OK : SNB_ERR_HIP, nullptr);
            check(hipMemcpy(derivBinding, slot.data(), sizeof(int) * slot.size(), hipMemcpyHostToDevice) == hipSuccess ? SNB_OK : SNB_ERR_HIP, nullptr);
        }

        readDefinition(system, force);
        check(hipMalloc(&userPos, (size_t) numParticles * 4 * (useDouble ? 8 : 4)) == hipSuccess ? SNB_OK : SNB_ERR_HIP, nullptr);
        check(hipMalloc(&userForces, (size_t) numParticles * 3 * (useDouble ? 8 : 4)) == hipSuccess ? SNB_OK : SNB_ERR_HIP, nullptr);
        check(snb_set_force_output(engine, userForces, useDouble, 0), engine);
        paramsDirty = true;
    }

    double execute(ContextImpl& context, bool includeForces, bool includeEnergy, bool includeDirect, bool includeReciprocal) override {
        cu.setAsCurrent();
        pushParameters(context);
        Vec3 a, b, c;  context.getPeriodicBoxVectors(a, b, c);
        const double box[9] = {a[0], a[1], a[2], b[0], b[1], b[2], c[0], c[1], c[2]};
        check(snb_set_box(engine, box), engine);
        // positions in user order (the engine keeps its own sorted order; OpenMM's reordering is invisible to it)
        hipStream_t stream = cu.getCurrentStream();
        const int blocks = (numParticles + 255) / 256;
        const int* atomIndex = (const int*) cu.getAtomIndexArray().getDevicePointer();
        if (useDouble) hipLaunchKernelGGL(gatherUserPositions<double4>, dim3(blocks), dim3(256), 0, stream, (const double4*) cu.getPosq().getDevicePointer(), atomIndex, (double4*) userPos, numParticles);
        else hipLaunchKernelGGL(gatherUserPositions<float4>, dim3(blocks), dim3(256), 0, stream, (const float4*) cu.getPosq().getDevicePointer(), atomIndex, (float4*) userPos, numParticles);
        check(snb_set_positions(engine, userPos, /*is_device=*/1, useDouble, /*stride4=*/1), engine);

        double energy = 0;
        const bool wantE = includeEnergy || hasDerivatives;          // Q4: derivatives accumulate whether or not the energy is requested
        // energy == NULL: the step stays asynchronous (a replayed graph); the slice energies are summed on the device
        check(snb_execute(engine, includeForces, wantE, includeDirect, includeReciprocal, includeEnergy ? &energy : nullptr), engine);
        if (includeForces) {
            unsigned long long* forceBuffers = (unsigned long long*) cu.getLongForceBuffer().getDevicePointer();
            if (useDouble) hipLaunchKernelGGL(addForcesToContext<double>, dim3(blocks), dim3(256), 0, stream, (const double*) userForces, atomIndex, forceBuffers, numParticles, cu.getPaddedNumAtoms());
            else hipLaunchKernelGGL(addForcesToContext<float>, dim3(blocks), dim3(256), 0, stream, (const float*) userForces, atomIndex, forceBuffers, numParticles, cu.getPaddedNumAtoms());
        }
        if (hasDerivatives) {      // dE/dlambda_k += E_raw[slice][term], added on the device: no read-back on the MD path
            const double* sliceE = nullptr;
            check(snb_slice_energies_device(engine, &sliceE), engine);
            const int n = 2 * numSlices;
            if (cu.getUseDoublePrecision() || cu.getUseMixedPrecision())
                hipLaunchKernelGGL(addDerivativesToContext<double>, dim3((n + 63) / 64), dim3(64), 0, stream, sliceE, derivBinding, n, (double*) cu.getEnergyParamDerivBuffer().getDevicePointer());
            else
                hipLaunchKernelGGL(addDerivativesToContext<float>, dim3((n + 63) / 64), dim3(64), 0, stream, sliceE, derivBinding, n, (float*) cu.getEnergyParamDerivBuffer().getDevicePointer());
        }
        return includeEnergy ? energy : 0.0;
    }

    void copyParametersToContext(ContextImpl& context, const SlicedNonbondedForce& force) override {
        if (force.getNumParticles() != numParticles) throw OpenMMException("updateParametersInContext: The number of particles has changed");
        const int old14 = (int) exceptionIs14Count;
        readDefinition(context.getSystem(), force);
        if ((int) exceptionIs14Count != old14) throw OpenMMException("updateParametersInContext: The number of non-excluded exceptions has changed");
        paramsDirty = true;
        cu.invalidateMolecules();
    }

    void getPMEParameters(double& alpha, int& nx, int& ny, int& nz) const override { int g[3]; check(snb_get_pme_parameters(engine, &alpha, g), engine); nx = g[0]; ny = g[1]; nz = g[2]; }
    void getLJPMEParameters(double& alpha, int& nx, int& ny, int& nz) const override { int g[3]; check(snb_get_ljpme_parameters(engine, &alpha, g), engine); nx = g[0]; ny = g[1]; nz = g[2]; }

private:
    struct Binding { std::string name; bool hasDerivative = false; };
    struct Offset { std::string param; int index; double dq, dsigma, deps; };

    static void check(snb_status s, snb_handle h) { if (s != SNB_OK) throw OpenMMException(snb_last_error(h)); }

    // base parameters, offsets, subsets, exceptions and dispersion coefficients of the force (what the reference keeps in
    // baseParticleParams / particleParamOffsets / baseExceptionParams, ReferenceNonbondedSlicingKernels.cpp:89-160)
    void readDefinition(const System& system, const SlicedNonbondedForce& force) {
        baseQ.resize(numParticles); baseSigma.resize(numParticles); baseEps.resize(numParticles); subsets.resize(numParticles);
        for (int i = 0; i < numParticles; i++) { force.getParticleParameters(i, baseQ[i], baseSigma[i], baseEps[i]); subsets[i] = force.getParticleSubset(i); }
        const int m = force.getNumExceptions();
        excPairs.resize(2 * (size_t) m); excQQ.resize(m); excSigma.resize(m); excEps.resize(m); excForce14.assign(m, 0);
        for (int k = 0; k < m; k++) { int p1, p2; force.getExceptionParameters(k, p1, p2, excQQ[k], excSigma[k], excEps[k]); excPairs[2 * k] = p1; excPairs[2 * k + 1] = p2; }
        particleOffsets.clear(); exceptionOffsets.clear();
        for (int k = 0; k < force.getNumParticleParameterOffsets(); k++) { Offset o; force.getParticleParameterOffset(k, o.param, o.index, o.dq, o.dsigma, o.deps); particleOffsets.push_back(o); }
        for (int k = 0; k < force.getNumExceptionParameterOffsets(); k++) {
            Offset o; force.getExceptionParameterOffset(k, o.param, o.index, o.dq, o.dsigma, o.deps); exceptionOffsets.push_back(o);
            excForce14[o.index] = 1;                                 // Q6: an exception with an offset is always a 1-4 interaction
        }
        exceptionIs14Count = 0;
        for (int k = 0; k < m; k++) if (excQQ[k] != 0.0 || excEps[k] != 0.0 || excForce14[k]) exceptionIs14Count++;
        if (force.getUseDispersionCorrection() && method != SlicedNonbondedForce::LJPME && method >= SlicedNonbondedForce::CutoffPeriodic) {
            std::vector<double> coef = SlicedNonbondedForceImpl::calcDispersionCorrections(system, force);
            check(snb_set_dispersion_coefficients(engine, coef.data()), engine);
        } else
            check(snb_set_dispersion_coefficients(engine, nullptr), engine);
    }

    // global parameters -> lambdas and the values the parameter offsets refer to.  The effective particle / exception parameters are
    // formed ON THE DEVICE (snb_set_parameter_offsets + snb_set_global_parameters; the reference: nonbondedParameters.cc:4-179): a changed
    // global parameter costs two small kernels ahead of the step -- no re-sort, no tile rebuild, no graph re-capture, no synchronisation.
    void pushParameters(ContextImpl& context) {
        std::vector<double> lambdas(2 * (size_t) numSlices, 1.0);
        for (size_t k = 0; k < bindings.size(); k++) if (!bindings[k].name.empty()) lambdas[k] = context.getParameter(bindings[k].name);
        if (lambdas != lastLambdas) { check(snb_set_lambdas(engine, lambdas.data()), engine); lastLambdas = lambdas; }
        if (paramsDirty) {      // base values, exceptions and offsets (initialize / copyParametersToContext)
            check(snb_set_particles(engine, baseQ.data(), baseSigma.data(), baseEps.data(), subsets.data()), engine);
            check(snb_set_exceptions(engine, (int32_t) excQQ.size(), excPairs.data(), excQQ.data(), excSigma.data(), excEps.data(), excForce14.data()), engine);
            offsetGlobals.clear();
            auto slotOf = [&](const std::string& name) {
                auto it = std::find(offsetGlobals.begin(), offsetGlobals.end(), name);
                if (it == offsetGlobals.end()) { offsetGlobals.push_back(name); return (int32_t) offsetGlobals.size() - 1; }
                return (int32_t) (it - offsetGlobals.begin());
            };
            std::vector<int32_t> pi, pg, ei, eg; std::vector<double> pd, ed;
            for (auto& o : particleOffsets) { pi.push_back(o.index); pg.push_back(slotOf(o.param)); pd.push_back(o.dq); pd.push_back(o.dsigma); pd.push_back(o.deps); }
            for (auto& o : exceptionOffsets) { ei.push_back(o.index); eg.push_back(slotOf(o.param)); ed.push_back(o.dq); ed.push_back(o.dsigma); ed.push_back(o.deps); }
            check(snb_set_parameter_offsets(engine, (int32_t) offsetGlobals.size(), (int32_t) pi.size(), pi.data(), pg.data(), pd.data(),
                                            (int32_t) ei.size(), ei.data(), eg.data(), ed.data()), engine);
            lastOffsetValues.clear();
            paramsDirty = false;
        }
        std::vector<double> values;
        for (auto& name : offsetGlobals) values.push_back(context.getParameter(name));
        if (values != lastOffsetValues) { check(snb_set_global_parameters(engine, (int32_t) values.size(), values.data()), engine); lastOffsetValues = values; }
    }

    HipContext& cu;
    snb_handle engine;
    void* userPos;  void* userForces;  int* derivBinding;
    std::vector<std::string> offsetGlobals;
    int numParticles = 0, numSubsets = 0, numSlices = 0, method = 0;
    bool useDouble = false, hasDerivatives = false, paramsDirty = true;
    size_t exceptionIs14Count = 0;
    std::vector<Binding> bindings;
    std::vector<double> baseQ, baseSigma, baseEps, excQQ, excSigma, excEps, lastLambdas, lastOffsetValues;
    std::vector<int32_t> subsets, excPairs, excForce14;
    std::vector<Offset> particleOffsets, exceptionOffsets;
};

class HipNonbondedSlicingKernelFactory : public KernelFactory {
public:
    KernelImpl* createKernelImpl(std::string name, const Platform& platform, ContextImpl& context) const override {
        HipPlatform::PlatformData& data = *static_cast<HipPlatform::PlatformData*>(context.getPlatformData());
        if (data.contexts.size() > 1)
            throw OpenMMException("SlicedNonbondedForce (MI355X engine): OpenMM's in-process multi-device contexts are not used; "
                                  "multi-GPU runs shard subset grids across processes (snb_config.shard_rank/shard_count)");
        if (name == CalcSlicedNonbondedForceKernel::Name())
            return new HipCalcSlicedNonbondedForceKernel(name, platform, *data.contexts[0], context.getSystem());
        throw OpenMMException((std::string("Tried to create kernel with illegal kernel name '") + name + "'").c_str());
    }
};

}  // namespace NonbondedSlicing

extern "C" void registerPlatforms() {}

extern "C" void registerKernelFactories() {                          // same shape as CudaNonbondedSlicingKernelFactory.cpp:19-31
    try {
        Platform& platform = Platform::getPlatformByName("HIP");
        platform.registerKernelFactory(NonbondedSlicing::CalcSlicedNonbondedForceKernel::Name(), new NonbondedSlicing::HipNonbondedSlicingKernelFactory());
    } catch (std::exception&) {
        // HIP platform not present: nothing to register
    }
}

extern "C" void registerNonbondedSlicingHipKernelFactories() {
    try { Platform::getPlatformByName("HIP"); } catch (...) { Platform::registerPlatform(new HipPlatform()); }
    registerKernelFactories();
}
