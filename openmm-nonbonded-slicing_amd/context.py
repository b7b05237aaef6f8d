"""Host-side mirror of the reference's kernel boundary for the hot path, over the C ABI (``include/snb.h``).

``HipCalcSlicedNonbondedForceKernel`` has the interface of ``NonbondedSlicing::CalcSlicedNonbondedForceKernel``
(openmmapi/include/NonbondedSlicingKernels.h:27-85): ``initialize / execute / copyParametersToContext /
getPMEParameters / getLJPMEParameters``.  ``SlicedNonbondedForceImpl`` restates the validation and the force-group ->
(includeDirect, includeReciprocal) mapping of openmmapi/src/SlicedNonbondedForceImpl.cpp:33-142.
``System`` / ``Context`` / ``State`` are the minimum of OpenMM's driver needed so that tests read like the
reference's own (tests/TestSlicedNonbondedForce.h): they hold positions, box vectors and global parameters, nothing else.

All arithmetic happens in ``libsnb_hip.so``; this module only moves parameters (scaling parameters -> lambdas,
parameter offsets -> effective (q, sigma, epsilon), raw slice energies -> energy and dE/dlambda).
"""
from __future__ import annotations

import ctypes
import math

import numpy as np

from . import _capi
from .force import OpenMMException, SlicedNonbondedForce, sliceIndex


def _dp(a): return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
def _ip(a): return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def calcPMEParameters(force, boxVectors, lj=False):
    """OpenMM ``NonbondedForceImpl::calcPMEParameters`` (third-party; formula restated from OpenMM 8.x and
    NOT pinned by any fixture of the reference -- parity runs always pass explicit PME parameters)."""
    alpha, nx, ny, nz = force.getLJPMEParameters() if lj else force.getPMEParameters()
    if alpha == 0.0:
        tol = force.getEwaldErrorTolerance()
        alpha = math.sqrt(-math.log(2 * tol)) / force.getCutoffDistance()
        dims = []
        for d in range(3):
            L = boxVectors[d][d]
            n = (alpha * L / (3 * tol ** 0.2)) if lj else (2 * alpha * L / (3 * tol ** 0.2))
            dims.append(max(int(math.ceil(n)), 6))
        nx, ny, nz = dims
    return alpha, nx, ny, nz


def calcEwaldParameters(force, boxVectors):
    """OpenMM ``NonbondedForceImpl::calcEwaldParameters`` (third-party; restated from OpenMM 8.x, NOT pinned by a
    reference fixture).  A mesh-less explicit override for parity runs: ``force.ewaldKmax = (kx, ky, kz)`` with the alpha of
    ``setPMEParameters``."""
    alpha = force.getPMEParameters()[0]
    tol = force.getEwaldErrorTolerance()
    if alpha == 0.0:
        alpha = math.sqrt(-math.log(2 * tol)) / force.getCutoffDistance()
    explicit = getattr(force, "ewaldKmax", None)
    if explicit is not None:
        return (alpha,) + tuple(int(k) for k in explicit)

    def find(width):
        f = lambda k: tol - 0.05 * math.sqrt(width * alpha) * k * math.exp(-(k * math.pi / (width * alpha)) ** 2)
        k = 10
        v = f(k)
        if v > 0.0:
            while v > 0.0 and k > 0:
                k -= 1; v = f(k)
            k += 1
        else:
            while v < 0.0:
                k += 1; v = f(k)
        return k if k % 2 == 1 else k + 1
    return alpha, find(boxVectors[0][0]), find(boxVectors[1][1]), find(boxVectors[2][2])


class HipCalcSlicedNonbondedForceKernel:
    """MI355X kernel object behind the reference's ``CalcSlicedNonbondedForceKernel`` interface."""

    @staticmethod
    def Name():
        return "CalcSlicedNonbondedForce"

    def __init__(self, precision="single", device=0, neighbor_padding=0.0, rebuild_interval=1, shard_rank=0, shard_count=1, stream=None):
        self._lib = _capi.lib()
        self._h = ctypes.c_void_p()
        self.precision = precision
        self.device = device
        self.neighbor_padding = neighbor_padding
        self.rebuild_interval = rebuild_interval
        self.shard_rank, self.shard_count = shard_rank, shard_count
        self.stream = stream
        self.force = None

    def __del__(self):
        try:
            if self._h:
                self._lib.snb_destroy(self._h); self._h = ctypes.c_void_p()
        except Exception:
            pass

    # -- error plumbing: status codes -> the exceptions the reference throws -----------------------
    def _check(self, status):
        if status != _capi.SNB_OK:
            msg = self._lib.snb_last_error(self._h)
            raise OpenMMException((msg or b"").decode() or "snb error %d" % status)

    # -- CalcSlicedNonbondedForceKernel::initialize (NonbondedSlicingKernels.h:48) ------------------
    def initialize(self, system, force):
        self.force = force
        self.numParticles = force.getNumParticles()
        self.numSubsets = force.getNumSubsets()
        self.numSlices = force.getNumSlices()
        cfg = _capi.SnbConfig()
        cfg.abi_version = _capi.SNB_ABI_VERSION
        cfg.n_atoms = self.numParticles; cfg.n_subsets = self.numSubsets
        method = force.getNonbondedMethod()
        cfg.method = method
        cfg.precision = {"single": 0, "double": 1, "mixed": 2}[self.precision]
        cfg.use_switch = int(force.getUseSwitchingFunction() and method != SlicedNonbondedForce.NoCutoff)
        cfg.exceptions_periodic = int(force.getExceptionsUsePeriodicBoundaryConditions())
        cfg.device = self.device
        cfg.cutoff = force.getCutoffDistance(); cfg.switch_distance = force.getSwitchingDistance()
        cfg.rf_dielectric = force.getReactionFieldDielectric()
        box = system.getDefaultPeriodicBoxVectors()
        if method in (SlicedNonbondedForce.PME, SlicedNonbondedForce.LJPME):
            a, nx, ny, nz = calcPMEParameters(force, box, False)
            cfg.alpha = a; cfg.grid[0], cfg.grid[1], cfg.grid[2] = nx, ny, nz
        if method == SlicedNonbondedForce.LJPME:
            a, nx, ny, nz = calcPMEParameters(force, box, True)
            cfg.alpha_d = a; cfg.dgrid[0], cfg.dgrid[1], cfg.dgrid[2] = nx, ny, nz
        if method == SlicedNonbondedForce.Ewald:
            a, kx, ky, kz = calcEwaldParameters(force, box)
            cfg.alpha = a; cfg.kmax[0], cfg.kmax[1], cfg.kmax[2] = kx, ky, kz
        cfg.neighbor_padding = self.neighbor_padding; cfg.rebuild_interval = self.rebuild_interval
        cfg.shard_rank = self.shard_rank; cfg.shard_count = self.shard_count
        cfg.stream = self.stream
        status = self._lib.snb_create(ctypes.byref(cfg), ctypes.byref(self._h))
        if status != _capi.SNB_OK:
            raise OpenMMException((self._lib.snb_last_error(None) or b"").decode() or "snb_create failed (%d)" % status)
        self._upload_definition(force)

    def _upload_definition(self, force):
        n = self.numParticles
        self._base = np.array([force.getParticleParameters(i) for i in range(n)], dtype=np.float64).reshape(n, 3)
        self._subset = np.array([force.getParticleSubset(i) for i in range(n)], dtype=np.int32)
        m = force.getNumExceptions()
        exc = [force.getExceptionParameters(k) for k in range(m)]
        self._excPairs = np.array([[e[0], e[1]] for e in exc], dtype=np.int32).reshape(m, 2)
        self._excBase = np.array([[e[2], e[3], e[4]] for e in exc], dtype=np.float64).reshape(m, 3)
        self._particleOffsets = [force.getParticleParameterOffset(k) for k in range(force.getNumParticleParameterOffsets())]
        self._exceptionOffsets = [force.getExceptionParameterOffset(k) for k in range(force.getNumExceptionParameterOffsets())]
        self._force14 = np.zeros(max(m, 1), dtype=np.int32)
        for (_, idx, _, _, _) in self._exceptionOffsets:
            self._force14[idx] = 1   # Q6: an exception with an offset is a 1-4 even when its base values are zero
        # scaling parameters -> (slice, term) bindings (ReferenceNonbondedSlicingKernels.cpp:58-87)
        self._binding = {}
        derivs = set(force.getEnergyParameterDerivativeName(i) for i in range(force.getNumEnergyParameterDerivatives()))
        for k in range(force.getNumScalingParameters()):
            name, s1, s2, incC, incLJ = force.getScalingParameter(k)
            s = sliceIndex(s1, s2)
            if incC: self._binding[(s, 0)] = (name, name in derivs)
            if incLJ: self._binding[(s, 1)] = (name, name in derivs)
        self._derivNames = derivs
        self._lastLambdas = None
        # dispersion-correction coefficients at DEFAULT parameter values (SlicedNonbondedForceImpl.cpp:281-291)
        defaults = {force.getGlobalParameterName(i): force.getGlobalParameterDefaultValue(i) for i in range(force.getNumGlobalParameters())}
        self._set_dispersion(force, defaults)
        self._push_definition()
        # slices whose raw energy a derivative-only step must produce: those bound to a parameter with a requested derivative
        mask = np.zeros(self.numSlices, dtype=np.int32)
        for (sl, _t), (_name, hasDeriv) in self._binding.items():
            if hasDeriv:
                mask[sl] = 1
        self._check(self._lib.snb_set_energy_slices(self._h, _ip(mask)))

    def _effective(self, params):
        p = self._base.copy()
        for (name, idx, dq, ds, de) in self._particleOffsets:
            v = params[name]
            p[idx, 0] += v * dq; p[idx, 1] += v * ds; p[idx, 2] += v * de
        e = self._excBase.copy()
        for (name, idx, da, db, dc) in self._exceptionOffsets:
            v = params[name]
            e[idx, 0] += v * da; e[idx, 1] += v * db; e[idx, 2] += v * dc
        return p, e

    def _set_dispersion(self, force, params):
        method = force.getNonbondedMethod()
        coef = np.zeros(self.numSlices)
        if force.getUseDispersionCorrection() and method in (SlicedNonbondedForce.CutoffPeriodic, SlicedNonbondedForce.Ewald, SlicedNonbondedForce.PME, SlicedNonbondedForce.LJPME):
            p, _ = self._effective(params)
            sig = np.ascontiguousarray(p[:, 1]); eps = np.ascontiguousarray(p[:, 2])
            useSwitch = int(force.getUseSwitchingFunction())
            self._check(self._lib.snb_compute_dispersion_coefficients(self.numParticles, self.numSubsets, _dp(sig), _dp(eps), _ip(self._subset),
                                                                      force.getCutoffDistance(), useSwitch, force.getSwitchingDistance(), _dp(coef)))
        self._dispCoef = coef
        self._check(self._lib.snb_set_dispersion_coefficients(self._h, _dp(coef)))

    def _push_definition(self):
        """Base parameters, exceptions and parameter offsets to the engine (once per initialize / copyParametersToContext): the
        effective values base + sum global * offset are formed ON THE DEVICE (include/snb.h, snb_set_parameter_offsets), as the
        reference does in platforms/common/src/kernels/nonbondedParameters.cc:4-179."""
        p = self._base
        q = np.ascontiguousarray(p[:, 0]); sg = np.ascontiguousarray(p[:, 1]); ep = np.ascontiguousarray(p[:, 2])
        self._check(self._lib.snb_set_particles(self._h, _dp(q), _dp(sg), _dp(ep), _ip(self._subset)))
        m = self._excPairs.shape[0]
        if m > 0:
            e = self._excBase
            qq = np.ascontiguousarray(e[:, 0]); es = np.ascontiguousarray(e[:, 1]); ee = np.ascontiguousarray(e[:, 2])
            self._check(self._lib.snb_set_exceptions(self._h, m, _ip(self._excPairs), _dp(qq), _dp(es), _dp(ee), _ip(self._force14)))
        else:
            self._check(self._lib.snb_set_exceptions(self._h, 0, None, None, None, None, None))
        names = sorted(set(o[0] for o in self._particleOffsets) | set(o[0] for o in self._exceptionOffsets))
        self._offsetGlobals = names
        index = {n: k for k, n in enumerate(names)}
        pi = np.array([o[1] for o in self._particleOffsets], dtype=np.int32); pg = np.array([index[o[0]] for o in self._particleOffsets], dtype=np.int32)
        pd = np.ascontiguousarray(np.array([[o[2], o[3], o[4]] for o in self._particleOffsets], dtype=np.float64).reshape(-1, 3))
        ei = np.array([o[1] for o in self._exceptionOffsets], dtype=np.int32); eg = np.array([index[o[0]] for o in self._exceptionOffsets], dtype=np.int32)
        ed = np.ascontiguousarray(np.array([[o[2], o[3], o[4]] for o in self._exceptionOffsets], dtype=np.float64).reshape(-1, 3))
        self._check(self._lib.snb_set_parameter_offsets(self._h, len(names), len(pi), _ip(pi) if len(pi) else None, _ip(pg) if len(pi) else None, _dp(pd) if len(pi) else None,
                                                        len(ei), _ip(ei) if len(ei) else None, _ip(eg) if len(ei) else None, _dp(ed) if len(ei) else None))
        self._lastGlobals = None

    def _push_parameters(self, params):
        if self._offsetGlobals:
            vals = np.array([params[n] for n in self._offsetGlobals], dtype=np.float64)
            if self._lastGlobals is None or not np.array_equal(vals, self._lastGlobals):
                self._check(self._lib.snb_set_global_parameters(self._h, len(vals), _dp(vals)))
                self._lastGlobals = vals
        lam = np.ones((self.numSlices, 2))
        for (s, t), (name, _) in self._binding.items():
            lam[s, t] = params[name]
        if self._lastLambdas is None or not np.array_equal(lam, self._lastLambdas):
            self._check(self._lib.snb_set_lambdas(self._h, _dp(np.ascontiguousarray(lam))))
            self._lastLambdas = lam

    # -- CalcSlicedNonbondedForceKernel::execute (NonbondedSlicingKernels.h:59) ---------------------
    def execute(self, context, includeForces, includeEnergy, includeDirect, includeReciprocal):
        params = context.getParameters()
        self._push_parameters(params)
        box = np.ascontiguousarray(context.getPeriodicBoxVectors(), dtype=np.float64).reshape(9)
        self._check(self._lib.snb_set_box(self._h, _dp(box)))
        pos = context._positions
        self._check(self._lib.snb_set_positions(self._h, pos.ctypes.data_as(ctypes.c_void_p), 0, 1, 0))
        energy = ctypes.c_double(0.0)
        wantE = bool(includeEnergy) or bool(self._derivNames)
        # Q4: derivatives accumulate whether or not the energy is requested -- then only the slices they are bound to are evaluated (mode 2)
        mode = 1 if includeEnergy else (2 if self._derivNames else 0)
        self._check(self._lib.snb_execute(self._h, int(includeForces), mode, int(includeDirect), int(includeReciprocal), ctypes.byref(energy) if mode == 1 else None))
        if includeForces:
            self._check(self._lib.snb_get_forces(self._h, context._forces.ctypes.data_as(ctypes.c_void_p), 0, 1, 1))
        if wantE:
            sl = np.zeros((self.numSlices, 2))
            self._check(self._lib.snb_get_slice_energies(self._h, _dp(sl)))
            self.lastSliceEnergies = sl
            # Q4: derivatives accumulate whether or not the energy was requested (ReferenceNonbondedSlicingKernels.cpp:259-265)
            for (s, t), (name, hasDeriv) in self._binding.items():
                if hasDeriv:
                    context._energyParamDerivs[name] = context._energyParamDerivs.get(name, 0.0) + sl[s, t]
        return energy.value if includeEnergy else 0.0

    # -- CalcSlicedNonbondedForceKernel::copyParametersToContext (NonbondedSlicingKernels.h:66) ------
    def copyParametersToContext(self, context, force):
        if force.getNumParticles() != self.numParticles:
            raise OpenMMException("updateParametersInContext: The number of particles has changed")
        old14 = int(((self._excBase[:, 0] != 0) | (self._excBase[:, 2] != 0) | (self._force14[:len(self._excBase)] != 0)).sum()) if len(self._excBase) else 0
        m = force.getNumExceptions()
        if m != self._excPairs.shape[0]:
            raise OpenMMException("updateParametersInContext: The number of non-excluded exceptions has changed")
        self._upload_definition(force)
        new14 = int(((self._excBase[:, 0] != 0) | (self._excBase[:, 2] != 0) | (self._force14[:len(self._excBase)] != 0)).sum()) if len(self._excBase) else 0
        if new14 != old14:
            raise OpenMMException("updateParametersInContext: The number of non-excluded exceptions has changed")

    def getPMEParameters(self):
        a = ctypes.c_double(); g = (ctypes.c_int32 * 3)()
        self._check(self._lib.snb_get_pme_parameters(self._h, ctypes.byref(a), g))
        return a.value, g[0], g[1], g[2]

    def getLJPMEParameters(self):
        a = ctypes.c_double(); g = (ctypes.c_int32 * 3)()
        self._check(self._lib.snb_get_ljpme_parameters(self._h, ctypes.byref(a), g))
        return a.value, g[0], g[1], g[2]

    def getStats(self):
        st = _capi.SnbStats()
        self._check(self._lib.snb_get_stats(self._h, ctypes.byref(st)))
        return st


class SlicedNonbondedForceImpl:
    """openmmapi/src/SlicedNonbondedForceImpl.cpp:33-142."""

    def __init__(self, owner, kernelFactory):
        self.owner = owner
        self.kernel = kernelFactory()

    def initialize(self, context):
        owner = self.owner; system = context.getSystem()
        if owner.getNumParticles() != system.getNumParticles():
            raise OpenMMException("SlicedNonbondedForce must have exactly as many particles as the System it belongs to.")
        if owner.getUseSwitchingFunction():
            if owner.getSwitchingDistance() < 0 or owner.getSwitchingDistance() >= owner.getCutoffDistance():
                raise OpenMMException("SlicedNonbondedForce: Switching distance must satisfy 0 <= r_switch < r_cutoff")
        for i in range(owner.getNumParticles()):
            _, sigma, epsilon = owner.getParticleParameters(i)
            if sigma < 0: raise OpenMMException("SlicedNonbondedForce: sigma for a particle cannot be negative")
            if epsilon < 0: raise OpenMMException("SlicedNonbondedForce: epsilon for a particle cannot be negative")
        seen = set()
        for i in range(owner.getNumExceptions()):
            p1, p2, _, sigma, epsilon = owner.getExceptionParameters(i)
            for p in (p1, p2):
                if p < 0 or p >= owner.getNumParticles():
                    raise OpenMMException("SlicedNonbondedForce: Illegal particle index for an exception: %d" % p)
            if (min(p1, p2), max(p1, p2)) in seen:
                raise OpenMMException("SlicedNonbondedForce: Multiple exceptions are specified for particles %d and %d" % (p1, p2))
            seen.add((min(p1, p2), max(p1, p2)))
            if sigma < 0: raise OpenMMException("SlicedNonbondedForce: sigma for an exception cannot be negative")
            if epsilon < 0: raise OpenMMException("SlicedNonbondedForce: epsilon for an exception cannot be negative")
        for i in range(owner.getNumParticleParameterOffsets()):
            idx = owner.getParticleParameterOffset(i)[1]
            if idx < 0 or idx >= owner.getNumParticles():
                raise OpenMMException("SlicedNonbondedForce: Illegal particle index for a particle parameter offset: %d" % idx)
        for i in range(owner.getNumExceptionParameterOffsets()):
            idx = owner.getExceptionParameterOffset(i)[1]
            if idx < 0 or idx >= owner.getNumExceptions():
                raise OpenMMException("SlicedNonbondedForce: Illegal exception index for an exception parameter offset: %d" % idx)
        method = owner.getNonbondedMethod()
        if method not in (SlicedNonbondedForce.NoCutoff, SlicedNonbondedForce.CutoffNonPeriodic):
            box = system.getDefaultPeriodicBoxVectors()
            cutoff = owner.getCutoffDistance()
            if cutoff > 0.5 * box[0][0] or cutoff > 0.5 * box[1][1] or cutoff > 0.5 * box[2][2]:
                raise OpenMMException("SlicedNonbondedForce: The cutoff distance cannot be greater than half the periodic box size.")
            if method == SlicedNonbondedForce.Ewald and (box[1][0] != 0.0 or box[2][0] != 0.0 or box[2][1] != 0):
                raise OpenMMException("SlicedNonbondedForce: Ewald is not supported with non-rectangular boxes.  Use PME instead.")
        offsetParams = set(owner.getParticleParameterOffset(i)[0] for i in range(owner.getNumParticleParameterOffsets()))
        offsetParams |= set(owner.getExceptionParameterOffset(i)[0] for i in range(owner.getNumExceptionParameterOffsets()))
        for i in range(owner.getNumScalingParameters()):
            if owner.getScalingParameter(i)[0] in offsetParams:
                raise OpenMMException("SlicedNonbondedForce: Cannot use a global parameter for both slice energy scaling and parameter offset.")
        self.kernel.initialize(system, owner)

    def calcForcesAndEnergy(self, context, includeForces, includeEnergy, groups):
        owner = self.owner
        includeDirect = owner.getIncludeDirectSpace() and (groups & (1 << owner.getForceGroup())) != 0
        reciprocalGroup = owner.getReciprocalSpaceForceGroup()
        if reciprocalGroup < 0:
            reciprocalGroup = owner.getForceGroup()
        includeReciprocal = (groups & (1 << reciprocalGroup)) != 0
        return self.kernel.execute(context, includeForces, includeEnergy, includeDirect, includeReciprocal)

    def getDefaultParameters(self):
        return {self.owner.getGlobalParameterName(i): self.owner.getGlobalParameterDefaultValue(i) for i in range(self.owner.getNumGlobalParameters())}

    def updateParametersInContext(self, context):
        self.kernel.copyParametersToContext(context, self.owner)


class System:
    def __init__(self):
        self._masses = []
        self._box = np.diag([2.0, 2.0, 2.0]).astype(float)
        self._forces = []

    def addParticle(self, mass): self._masses.append(float(mass)); return len(self._masses) - 1
    def getNumParticles(self): return len(self._masses)
    def setDefaultPeriodicBoxVectors(self, a, b, c): self._box = np.array([a, b, c], dtype=float)
    def getDefaultPeriodicBoxVectors(self): return self._box.copy()
    def addForce(self, force): self._forces.append(force); return len(self._forces) - 1
    def getNumForces(self): return len(self._forces)
    def getForce(self, i): return self._forces[i]
    def usesPeriodicBoundaryConditions(self): return any(f.usesPeriodicBoundaryConditions() for f in self._forces)


class State:
    def __init__(self, energy, forces, derivs, params):
        self._e, self._f, self._d, self._p = energy, forces, derivs, params

    def getPotentialEnergy(self): return self._e
    def getForces(self): return self._f
    def getEnergyParameterDerivatives(self): return dict(self._d)
    def getParameters(self): return dict(self._p)


class Context:
    """The slice of OpenMM's Context/ContextImpl the kernel boundary talks to."""

    def __init__(self, system, precision="single", device=0, **kernelOptions):
        self._system = system
        self._kernelFactory = lambda: HipCalcSlicedNonbondedForceKernel(precision=precision, device=device, **kernelOptions)
        self._box = system.getDefaultPeriodicBoxVectors()
        self._positions = None
        self._parameters = {}
        self._energyParamDerivs = {}
        self._build()

    def _build(self):
        self._impls = []
        for f in self._system._forces:
            impl = SlicedNonbondedForceImpl(f, self._kernelFactory)
            for k, v in impl.getDefaultParameters().items():
                self._parameters.setdefault(k, v)
            self._impls.append(impl)
        for impl in self._impls:
            impl.initialize(self)

    def getSystem(self): return self._system
    def getParameters(self): return dict(self._parameters)
    def getParameter(self, name): return self._parameters[name]
    def setParameter(self, name, value):
        if name not in self._parameters:
            raise OpenMMException("Called setParameter() with invalid parameter name: " + name)
        self._parameters[name] = float(value)
    def setPositions(self, positions):
        p = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 3)
        if p.shape[0] != self._system.getNumParticles():
            raise OpenMMException("Called setPositions() on a Context with the wrong number of positions")
        self._positions = p
    def setPeriodicBoxVectors(self, a, b, c): self._box = np.array([a, b, c], dtype=float)
    def getPeriodicBoxVectors(self): return self._box.copy()

    def reinitialize(self, preserveState=False):
        pos, params = self._positions, dict(self._parameters)
        self._parameters = {}
        self._box = self._system.getDefaultPeriodicBoxVectors() if not preserveState else self._box
        self._build()
        if preserveState:
            self._positions = pos
            for k, v in params.items():
                if k in self._parameters:
                    self._parameters[k] = v
        else:
            self._positions = None

    def getState(self, getEnergy=False, getForces=False, getParameterDerivatives=False, groups=0xFFFFFFFF):
        if self._positions is None:
            raise OpenMMException("Particle positions have not been set")
        n = self._system.getNumParticles()
        self._forces = np.zeros((n, 3))
        self._energyParamDerivs = {}
        energy = 0.0
        for impl in self._impls:
            energy += impl.calcForcesAndEnergy(self, getForces, getEnergy or getParameterDerivatives, groups)
        return State(energy, self._forces.copy(), self._energyParamDerivs, self._parameters)

    def _updateParametersInContext(self, force):
        for impl in self._impls:
            if impl.owner is force:
                impl.updateParametersInContext(self)

    def _kernelFor(self, force):
        for impl in self._impls:
            if impl.owner is force:
                return impl.kernel
        raise OpenMMException("force not in this context")
