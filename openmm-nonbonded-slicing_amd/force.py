"""Host-side mirror of the reference's ``SlicedNonbondedForce`` data container.

Same method names, argument meaning and error behaviour as
``openmmapi/include/SlicedNonbondedForce.h:26-96`` / ``openmmapi/src/SlicedNonbondedForce.cpp:31-194``
(plus the inherited ``OpenMM::NonbondedForce`` accessors the hot path reads), so that parity tests
read like the reference's own tests.  It holds no arithmetic: everything numeric happens behind the
C-ABI in ``csrc/`` (``include/snb.h``).
"""
from __future__ import annotations

import math


class OpenMMException(Exception):
    """Stands in for ``OpenMM::OpenMMException`` (surfaced to Python as ``Exception`` by the reference's
    SWIG layer, ``python/nonbondedslicing.i:42-49``)."""


def sliceIndex(i: int, j: int) -> int:
    """``openmmapi/include/SlicedNonbondedForce.h:22``."""
    return i * (i + 1) // 2 + j if i > j else j * (j + 1) // 2 + i


class _ScalingParameterInfo:
    """``SlicedNonbondedForce::ScalingParameterInfo`` (SlicedNonbondedForce.h:75-96)."""

    def __init__(self, globalParamIndex, subset1, subset2, includeCoulomb, includeLJ):
        if not (includeCoulomb or includeLJ):
            raise OpenMMException("Keywords 'includeCoulomb' and 'includeLJ' cannot be both false")
        self.globalParamIndex = globalParamIndex
        self.subset1, self.subset2 = subset1, subset2
        self.includeCoulomb, self.includeLJ = bool(includeCoulomb), bool(includeLJ)

    def getSlice(self):
        return sliceIndex(self.subset1, self.subset2)

    def clashesWith(self, info):
        return self.getSlice() == info.getSlice() and (
            (self.includeCoulomb and info.includeCoulomb) or (self.includeLJ and info.includeLJ))


def _value(x):
    """Plain float of a number or of an openmm.unit.Quantity (md unit system: nm, kJ/mol, e)."""
    if hasattr(x, "value_in_unit_system"):
        import openmm.unit as _u   # only reached when the source force came from OpenMM itself
        return float(x.value_in_unit_system(_u.md_unit_system))
    return float(x)


class SlicedNonbondedForce:
    # NonbondedSlicingKernels.h:29-36
    NoCutoff, CutoffNonPeriodic, CutoffPeriodic, Ewald, PME, LJPME = range(6)

    def __init__(self, numSubsets, numSubsetsForCopy=None):
        """SlicedNonbondedForce(numSubsets), or SlicedNonbondedForce(force, numSubsets): a copy of `force` -- any object with
        OpenMM's NonbondedForce getters, e.g. an openmm.NonbondedForce or another SlicedNonbondedForce -- with every particle in
        subset 0 (openmmapi/src/SlicedNonbondedForce.cpp:34-82)."""
        source = None
        if numSubsetsForCopy is not None:
            source, numSubsets = numSubsets, numSubsetsForCopy
        self.numSubsets = int(numSubsets)
        self._particles = []          # [charge, sigma, epsilon]
        self._exceptions = []         # [p1, p2, chargeProd, sigma, epsilon]
        self._exceptionMap = {}
        self._subsets = {}
        self._globalParams = []       # [name, default]
        self._particleOffsets = []    # (param index, particle, dq, dsigma, deps)
        self._exceptionOffsets = []   # (param index, exception, dqq, dsigma, deps)
        self._scalingParameters = []
        self._energyParameterDerivatives = []
        self._method = self.NoCutoff
        self._cutoff = 1.0
        self._useSwitch = False
        self._switchDistance = -1.0
        self._rfDielectric = 78.3
        self._ewaldTol = 5e-4
        self._pme = (0.0, 0, 0, 0)
        self._ljpme = (0.0, 0, 0, 0)
        self._useDispersionCorrection = True
        self._exceptionsPeriodic = False
        self._includeDirect = True
        self._forceGroup = 0
        self._recipForceGroup = -1
        self._name = "SlicedNonbondedForce"      # OpenMM Force::getName default: the class name
        self.useCuFFT = True  # kept for surface compatibility; meaningless here
        if source is not None:
            self._copyFrom(source)

    # ---- NonbondedForce surface -------------------------------------------------------------
    def _copyFrom(self, force):
        # same order of calls as the reference's converting constructor (SlicedNonbondedForce.cpp:37-81)
        self.setForceGroup(force.getForceGroup())
        self.setNonbondedMethod(int(force.getNonbondedMethod()))
        self.setCutoffDistance(_value(force.getCutoffDistance()))
        self.setUseSwitchingFunction(force.getUseSwitchingFunction())
        self.setSwitchingDistance(_value(force.getSwitchingDistance()))
        self.setEwaldErrorTolerance(force.getEwaldErrorTolerance())
        self.setReactionFieldDielectric(force.getReactionFieldDielectric())
        self.setUseDispersionCorrection(force.getUseDispersionCorrection())
        self.setIncludeDirectSpace(force.getIncludeDirectSpace())
        a, nx, ny, nz = force.getPMEParameters()
        self.setPMEParameters(_value(a), nx, ny, nz)
        a, nx, ny, nz = force.getLJPMEParameters()
        self.setLJPMEParameters(_value(a), nx, ny, nz)
        self.setReciprocalSpaceForceGroup(force.getReciprocalSpaceForceGroup())
        for i in range(force.getNumParticles()):
            q, sg, ep = force.getParticleParameters(i)
            self.addParticle(_value(q), _value(sg), _value(ep))
        for i in range(force.getNumExceptions()):
            p1, p2, qq, sg, ep = force.getExceptionParameters(i)
            self.addException(p1, p2, _value(qq), _value(sg), _value(ep))
        self.setExceptionsUsePeriodicBoundaryConditions(force.getExceptionsUsePeriodicBoundaryConditions())
        for i in range(force.getNumGlobalParameters()):
            self.addGlobalParameter(force.getGlobalParameterName(i), force.getGlobalParameterDefaultValue(i))
        for i in range(force.getNumParticleParameterOffsets()):
            name, idx, dq, dsg, dep = force.getParticleParameterOffset(i)
            self.addParticleParameterOffset(name, idx, dq, dsg, dep)
        for i in range(force.getNumExceptionParameterOffsets()):
            name, idx, dqq, dsg, dep = force.getExceptionParameterOffset(i)
            self.addExceptionParameterOffset(name, idx, dqq, dsg, dep)

    def getNumParticles(self): return len(self._particles)
    def getNumExceptions(self): return len(self._exceptions)
    def getNumGlobalParameters(self): return len(self._globalParams)
    def getNumParticleParameterOffsets(self): return len(self._particleOffsets)
    def getNumExceptionParameterOffsets(self): return len(self._exceptionOffsets)
    def getNonbondedMethod(self): return self._method
    def setNonbondedMethod(self, method):
        if method not in range(6):
            raise OpenMMException("NonbondedForce: Illegal value for nonbonded method")
        self._method = int(method)
    def getCutoffDistance(self): return self._cutoff
    def setCutoffDistance(self, d): self._cutoff = float(d)
    def getUseSwitchingFunction(self): return self._useSwitch
    def setUseSwitchingFunction(self, use): self._useSwitch = bool(use)
    def getSwitchingDistance(self): return self._switchDistance
    def setSwitchingDistance(self, d): self._switchDistance = float(d)
    def getReactionFieldDielectric(self): return self._rfDielectric
    def setReactionFieldDielectric(self, d): self._rfDielectric = float(d)
    def getEwaldErrorTolerance(self): return self._ewaldTol
    def setEwaldErrorTolerance(self, tol): self._ewaldTol = float(tol)
    def getPMEParameters(self): return self._pme
    def setPMEParameters(self, alpha, nx, ny, nz): self._pme = (float(alpha), int(nx), int(ny), int(nz))
    def getLJPMEParameters(self): return self._ljpme
    def setLJPMEParameters(self, alpha, nx, ny, nz): self._ljpme = (float(alpha), int(nx), int(ny), int(nz))
    def getUseDispersionCorrection(self): return self._useDispersionCorrection
    def setUseDispersionCorrection(self, use): self._useDispersionCorrection = bool(use)
    def getExceptionsUsePeriodicBoundaryConditions(self): return self._exceptionsPeriodic
    def setExceptionsUsePeriodicBoundaryConditions(self, p): self._exceptionsPeriodic = bool(p)
    def getIncludeDirectSpace(self): return self._includeDirect
    def setIncludeDirectSpace(self, inc): self._includeDirect = bool(inc)
    def getName(self): return self._name
    def setName(self, name): self._name = str(name)
    def getForceGroup(self): return self._forceGroup
    def setForceGroup(self, g): self._forceGroup = int(g)
    def getReciprocalSpaceForceGroup(self): return self._recipForceGroup
    def setReciprocalSpaceForceGroup(self, g): self._recipForceGroup = int(g)
    def usesPeriodicBoundaryConditions(self):
        return self._method in (self.CutoffPeriodic, self.Ewald, self.PME, self.LJPME)

    def addParticle(self, charge, sigma, epsilon):
        self._particles.append([float(charge), float(sigma), float(epsilon)])
        return len(self._particles) - 1
    def getParticleParameters(self, index):
        self._check("Index", index, len(self._particles))
        return tuple(self._particles[index])
    def setParticleParameters(self, index, charge, sigma, epsilon):
        self._check("Index", index, len(self._particles))
        self._particles[index] = [float(charge), float(sigma), float(epsilon)]

    def addException(self, particle1, particle2, chargeProd, sigma, epsilon, replace=False):
        key = (min(particle1, particle2), max(particle1, particle2))
        if key in self._exceptionMap:
            if not replace:
                raise OpenMMException("NonbondedForce: There is already an exception for particles %d and %d" % (particle1, particle2))
            self._exceptions[self._exceptionMap[key]] = [int(particle1), int(particle2), float(chargeProd), float(sigma), float(epsilon)]
            return self._exceptionMap[key]
        self._exceptions.append([int(particle1), int(particle2), float(chargeProd), float(sigma), float(epsilon)])
        self._exceptionMap[key] = len(self._exceptions) - 1
        return len(self._exceptions) - 1
    def getExceptionParameters(self, index):
        self._check("Index", index, len(self._exceptions))
        return tuple(self._exceptions[index])
    def setExceptionParameters(self, index, particle1, particle2, chargeProd, sigma, epsilon):
        self._check("Index", index, len(self._exceptions))
        old = self._exceptions[index]
        self._exceptionMap.pop((min(old[0], old[1]), max(old[0], old[1])), None)
        self._exceptions[index] = [int(particle1), int(particle2), float(chargeProd), float(sigma), float(epsilon)]
        self._exceptionMap[(min(particle1, particle2), max(particle1, particle2))] = index

    def createExceptionsFromBonds(self, bonds, coulomb14Scale, lj14Scale):
        """OpenMM ``NonbondedForce::createExceptionsFromBonds`` (third-party behaviour: 1-2 and 1-3 pairs are
        excluded, 1-4 pairs get scaled Lorentz-Berthelot parameters)."""
        n = self.getNumParticles()
        bonded12 = [set() for _ in range(n)]
        for a, b in bonds:
            bonded12[a].add(b); bonded12[b].add(a)
        exclusions = [set() for _ in range(n)]
        for i in range(n):
            self._addExclusionsToSet(bonded12, exclusions[i], i, i, 2)
        for i in range(n):
            bonded13 = set()
            self._addExclusionsToSet(bonded12, bonded13, i, i, 1)
            for j in sorted(exclusions[i]):
                if j < i:
                    if j not in bonded13:
                        qi, si, ei = self._particles[j]
                        qj, sj, ej = self._particles[i]
                        self.addException(j, i, coulomb14Scale * qi * qj, 0.5 * (si + sj), lj14Scale * math.sqrt(ei * ej))
                    else:
                        self.addException(j, i, 0.0, 1.0, 0.0)

    def _addExclusionsToSet(self, bonded12, exclusions, baseParticle, fromParticle, currentLevel):
        for i in bonded12[fromParticle]:
            if i != baseParticle:
                exclusions.add(i)
            if currentLevel > 0:
                self._addExclusionsToSet(bonded12, exclusions, baseParticle, i, currentLevel - 1)

    def addGlobalParameter(self, name, defaultValue):
        self._globalParams.append([str(name), float(defaultValue)])
        return len(self._globalParams) - 1
    def getGlobalParameterName(self, index): return self._globalParams[index][0]
    def getGlobalParameterDefaultValue(self, index): return self._globalParams[index][1]
    def setGlobalParameterDefaultValue(self, index, v): self._globalParams[index][1] = float(v)
    def addParticleParameterOffset(self, parameter, particleIndex, chargeScale, sigmaScale, epsilonScale):
        self._particleOffsets.append((self._getGlobalParameterIndex(parameter), int(particleIndex), float(chargeScale), float(sigmaScale), float(epsilonScale)))
        return len(self._particleOffsets) - 1
    def getParticleParameterOffset(self, index):
        p, i, a, b, c = self._particleOffsets[index]
        return (self._globalParams[p][0], i, a, b, c)
    def addExceptionParameterOffset(self, parameter, exceptionIndex, chargeProdScale, sigmaScale, epsilonScale):
        self._exceptionOffsets.append((self._getGlobalParameterIndex(parameter), int(exceptionIndex), float(chargeProdScale), float(sigmaScale), float(epsilonScale)))
        return len(self._exceptionOffsets) - 1
    def getExceptionParameterOffset(self, index):
        p, i, a, b, c = self._exceptionOffsets[index]
        return (self._globalParams[p][0], i, a, b, c)

    # ---- SlicedNonbondedForce surface (SlicedNonbondedForce.cpp:84-178) -----------------------
    def getNumSubsets(self): return self.numSubsets
    def getNumSlices(self): return self.numSubsets * (self.numSubsets + 1) // 2
    def getNumScalingParameters(self): return len(self._scalingParameters)
    def getNumEnergyParameterDerivatives(self): return len(self._energyParameterDerivatives)
    def getUseCuFFT(self): return self.useCuFFT
    def setUseCuFFT(self, use): self.useCuFFT = bool(use)

    def getNonbondedMethodName(self):
        return ["NoCutoff", "CutoffNonPeriodic", "CutoffPeriodic", "Ewald", "PME", "LJPME"][self._method]

    def setParticleSubset(self, index, subset):
        self._check("Index", index, self.getNumParticles())
        self._check("Subset", subset, self.numSubsets)
        self._subsets[int(index)] = int(subset)

    def getParticleSubset(self, index):
        self._check("Index", index, self.getNumParticles())
        return self._subsets.get(int(index), 0)

    def addScalingParameter(self, parameter, subset1, subset2, includeCoulomb, includeLJ):
        self._check("Subset", subset1, self.numSubsets)
        self._check("Subset", subset2, self.numSubsets)
        info = _ScalingParameterInfo(self._getGlobalParameterIndex(parameter), subset1, subset2, includeCoulomb, includeLJ)
        for param in self._scalingParameters:
            if param.clashesWith(info):
                raise OpenMMException("Clash detected between scaling parameters")
        self._scalingParameters.append(info)
        return len(self._scalingParameters) - 1

    def getScalingParameter(self, index):
        self._check("Index", index, len(self._scalingParameters))
        info = self._scalingParameters[index]
        return (self._globalParams[info.globalParamIndex][0], info.subset1, info.subset2, info.includeCoulomb, info.includeLJ)

    def setScalingParameter(self, index, parameter, subset1, subset2, includeCoulomb, includeLJ):
        self._check("Index", index, len(self._scalingParameters))
        self._check("Subset", subset1, self.numSubsets)
        self._check("Subset", subset2, self.numSubsets)
        info = _ScalingParameterInfo(self._getGlobalParameterIndex(parameter), subset1, subset2, includeCoulomb, includeLJ)
        old = self._scalingParameters[index]
        if not old.clashesWith(info):
            for param in self._scalingParameters:
                if param.clashesWith(info):
                    raise OpenMMException("A scaling parameter has already been defined for this slice & contribution(s)")
        self._scalingParameters[index] = info

    def addEnergyParameterDerivative(self, parameter):
        idx = self._getScalingParameterIndex(parameter)
        if idx in self._energyParameterDerivatives:
            raise OpenMMException("This scaling parameter derivative has already been requested")
        self._energyParameterDerivatives.append(idx)
        return len(self._energyParameterDerivatives) - 1

    def getEnergyParameterDerivativeName(self, index):
        self._check("Index", index, len(self._energyParameterDerivatives))
        return self._globalParams[self._scalingParameters[self._energyParameterDerivatives[index]].globalParamIndex][0]

    # ---- helpers -----------------------------------------------------------------------------
    def _getGlobalParameterIndex(self, parameter):
        for i, (name, _) in enumerate(self._globalParams):
            if name == parameter:
                return i
        raise OpenMMException("There is no global parameter called '" + parameter + "'")

    def _getScalingParameterIndex(self, parameter):
        for i, info in enumerate(self._scalingParameters):
            if self._globalParams[info.globalParamIndex][0] == parameter:
                return i
        raise OpenMMException("There is no scaling parameter called '" + parameter + "'")

    @staticmethod
    def _check(name, value, upper):
        if value < 0 or value >= upper:
            raise OpenMMException("%s out of range" % name)

    # Context-facing conveniences (SlicedNonbondedForce.cpp:184-194)
    def updateParametersInContext(self, context):
        context._updateParametersInContext(self)

    def getPMEParametersInContext(self, context):
        return context._kernelFor(self).getPMEParameters()

    def getLJPMEParametersInContext(self, context):
        return context._kernelFor(self).getLJPMEParameters()
