"""Known-answer tests restated from the reference's own test-suite
(``/root/reference/tests/TestSlicedNonbondedForce.h``; line ranges cited per case).

Each case takes an *evaluator*  ``ev(force, positions, box=None, parameters=None,
include_direct=True, include_reciprocal=True) -> dict(energy, forces, slice_energies, derivatives)``
so the very same cases pin (a) the CPU oracle (``-m "not gpu"``) and (b) the HIP engine through the
C-ABI (``-m gpu``).  Assertion helpers follow ``openmmapi/include/internal/AssertionUtilities.h:7-44``.
"""
import math

import numpy as np

ONE_4PI_EPS0 = 138.93545764438198
SQRT_TWO = math.sqrt(2.0)
TOL = 1e-4  # tests/TestSlicedNonbondedForce.h:27


def assertEqualTo(expected, found, tol):
    scale = max(abs(expected), 1.0)
    assert abs(expected - found) / scale <= tol, "expected %r found %r (tol %g)" % (expected, found, tol)


def assertEqualVec(expected, found, tol):
    expected = np.asarray(expected, dtype=float); found = np.asarray(found, dtype=float)
    norm = max(math.sqrt(float(expected @ expected)), 1.0)
    diff = math.sqrt(float((expected - found) @ (expected - found)))
    assert diff / norm <= tol, "expected %r found %r (tol %g)" % (expected, found, tol)


def assertForces(f1, f2, tol):
    for a, b in zip(np.asarray(f1), np.asarray(f2)):
        assertEqualVec(a, b, tol)


def cubic(L):
    return np.diag([L, L, L]).astype(float)


# --- :87-109 ------------------------------------------------------------------------------------
def testCoulomb(ev, F, tol=TOL):
    ff = F(1)
    ff.addParticle(0.5, 1, 0)
    ff.addParticle(-1.5, 1, 0)
    assert not ff.usesPeriodicBoundaryConditions()
    r = ev(ff, [[0, 0, 0], [2, 0, 0]])
    force = ONE_4PI_EPS0 * (-0.75) / 4.0
    assertEqualVec([-force, 0, 0], r["forces"][0], tol)
    assertEqualVec([force, 0, 0], r["forces"][1], tol)
    assertEqualTo(ONE_4PI_EPS0 * (-0.75) / 2.0, r["energy"], tol)


# --- :111-135 -----------------------------------------------------------------------------------
def testLJ(ev, F, tol=TOL):
    ff = F(1)
    ff.addParticle(0, 1.2, 1)
    ff.addParticle(0, 1.4, 2)
    r = ev(ff, [[0, 0, 0], [2, 0, 0]])
    x = 1.3 / 2.0
    eps = SQRT_TWO
    force = 4.0 * eps * (12 * x ** 12 - 6 * x ** 6) / 2.0
    assertEqualVec([-force, 0, 0], r["forces"][0], tol)
    assertEqualVec([force, 0, 0], r["forces"][1], tol)
    assertEqualTo(4.0 * eps * (x ** 12 - x ** 6), r["energy"], tol)


def _chain5(F, method=None, cutoff=None, rf=None):
    sliced = F(1)
    if method is not None:
        sliced.setNonbondedMethod(method)
    for _ in range(5):
        sliced.addParticle(0, 1.5, 0)
    if cutoff is not None:
        sliced.setCutoffDistance(cutoff)
    if rf is not None:
        sliced.setReactionFieldDielectric(rf)
    sliced.createExceptionsFromBonds([(0, 1), (1, 2), (2, 3), (3, 4)], 0.0, 0.0)
    first14 = second14 = None
    for i in range(sliced.getNumExceptions()):
        p1, p2, *_ = sliced.getExceptionParameters(i)
        if {p1, p2} == {0, 3}:
            first14 = i
        if {p1, p2} == {1, 4}:
            second14 = i
    assert first14 is not None and second14 is not None
    return sliced, first14, second14


# --- :137-222 -----------------------------------------------------------------------------------
def testExclusionsAnd14(ev, F, tol=TOL):
    sliced, first14, second14 = _chain5(F)
    for i in range(1, 5):
        r_ = 1.0
        positions = [[0, j, 0] for j in range(5)]
        for j in range(5):
            sliced.setParticleParameters(j, 0, 1.5, 0)
        sliced.setParticleParameters(0, 0, 1.5, 1)
        sliced.setParticleParameters(i, 0, 1.5, 1)
        sliced.setExceptionParameters(first14, 0, 3, 0, 1.5, 0.5 if i == 3 else 0.0)
        sliced.setExceptionParameters(second14, 1, 4, 0, 1.5, 0.0)
        positions[i] = [r_, 0, 0]
        res = ev(sliced, positions)
        x = 1.5 / r_
        force = 4.0 * (12 * x ** 12 - 6 * x ** 6) / r_
        energy = 4.0 * (x ** 12 - x ** 6)
        if i == 3:
            force *= 0.5; energy *= 0.5
        if i < 3:
            force = 0; energy = 0
        assertEqualVec([-force, 0, 0], res["forces"][0], tol)
        assertEqualVec([force, 0, 0], res["forces"][i], tol)
        assertEqualTo(energy, res["energy"], tol)
        # Coulomb
        sliced.setParticleParameters(0, 2, 1.5, 0)
        sliced.setParticleParameters(i, 2, 1.5, 0)
        sliced.setExceptionParameters(first14, 0, 3, 4 / 1.2 if i == 3 else 0, 1.5, 0)
        sliced.setExceptionParameters(second14, 1, 4, 0, 1.5, 0)
        res = ev(sliced, positions)
        force = ONE_4PI_EPS0 * 4 / (r_ * r_)
        energy = ONE_4PI_EPS0 * 4 / r_
        if i == 3:
            force /= 1.2; energy /= 1.2
        if i < 3:
            force = 0; energy = 0
        assertEqualVec([-force, 0, 0], res["forces"][0], tol)
        assertEqualVec([force, 0, 0], res["forces"][i], tol)
        assertEqualTo(energy, res["energy"], tol)


# --- :224-260 -----------------------------------------------------------------------------------
def testCutoff(ev, F, tol=TOL):
    ff = F(1)
    for _ in range(3):
        ff.addParticle(1.0, 1, 0)
    ff.setNonbondedMethod(ff.CutoffNonPeriodic)
    cutoff = 2.9
    ff.setCutoffDistance(cutoff)
    eps = 50.0
    ff.setReactionFieldDielectric(eps)
    r = ev(ff, [[0, 0, 0], [0, 2, 0], [0, 3, 0]])
    krf = (1.0 / cutoff ** 3) * (eps - 1.0) / (2.0 * eps + 1.0)
    crf = (1.0 / cutoff) * (3.0 * eps) / (2.0 * eps + 1.0)
    force1 = ONE_4PI_EPS0 * (0.25 - 2.0 * krf * 2.0)
    force2 = ONE_4PI_EPS0 * (1.0 - 2.0 * krf * 1.0)
    assertEqualVec([0, -force1, 0], r["forces"][0], tol)
    assertEqualVec([0, force1 - force2, 0], r["forces"][1], tol)
    assertEqualVec([0, force2, 0], r["forces"][2], tol)
    energy1 = ONE_4PI_EPS0 * (0.5 + krf * 4.0 - crf)
    energy2 = ONE_4PI_EPS0 * (1.0 + krf * 1.0 - crf)
    assertEqualTo(energy1 + energy2, r["energy"], tol)


# --- :262-356 -----------------------------------------------------------------------------------
def testCutoff14(ev, F, tol=TOL):
    cutoff = 3.5
    sliced, first14, second14 = _chain5(F, method=1, cutoff=cutoff, rf=30.0)
    positions = [[i, 0, 0] for i in range(5)]
    for i in range(1, 5):
        sliced.setParticleParameters(0, 0, 1.5, 1)
        for j in range(1, 5):
            sliced.setParticleParameters(j, 0, 1.5, 0)
        sliced.setParticleParameters(i, 0, 1.5, 1)
        sliced.setExceptionParameters(first14, 0, 3, 0, 1.5, 0.5 if i == 3 else 0.0)
        sliced.setExceptionParameters(second14, 1, 4, 0, 1.5, 0.0)
        res = ev(sliced, positions)
        r_ = positions[i][0]
        x = 1.5 / r_
        force = 4.0 * (12 * x ** 12 - 6 * x ** 6) / r_
        energy = 4.0 * (x ** 12 - x ** 6)
        if i == 3:
            force *= 0.5; energy *= 0.5
        if i < 3 or r_ > cutoff:
            force = 0; energy = 0
        assertEqualVec([-force, 0, 0], res["forces"][0], tol)
        assertEqualVec([force, 0, 0], res["forces"][i], tol)
        assertEqualTo(energy, res["energy"], tol)
        q = 0.7
        sliced.setParticleParameters(0, q, 1.5, 0)
        sliced.setParticleParameters(i, q, 1.5, 0)
        sliced.setExceptionParameters(first14, 0, 3, q * q / 1.2 if i == 3 else 0, 1.5, 0)
        sliced.setExceptionParameters(second14, 1, 4, 0, 1.5, 0)
        res = ev(sliced, positions)
        force = ONE_4PI_EPS0 * q * q / (r_ * r_)
        energy = ONE_4PI_EPS0 * q * q / r_
        if i == 3:
            force /= 1.2; energy /= 1.2
        if i < 3 or r_ > cutoff:
            force = 0; energy = 0
        assertEqualVec([-force, 0, 0], res["forces"][0], tol)
        assertEqualVec([force, 0, 0], res["forces"][i], tol)
        assertEqualTo(energy, res["energy"], tol)


# --- :358-392 -----------------------------------------------------------------------------------
def testPeriodic(ev, F, tol=TOL):
    sliced = F(1)
    for _ in range(3):
        sliced.addParticle(1.0, 1, 0)
    sliced.addException(0, 1, 0.0, 1.0, 0.0)
    sliced.setNonbondedMethod(sliced.CutoffPeriodic)
    cutoff = 2.0
    sliced.setCutoffDistance(cutoff)
    assert sliced.usesPeriodicBoundaryConditions()
    r = ev(sliced, [[0, 0, 0], [2, 0, 0], [3, 0, 0]], cubic(4))
    eps = 78.3
    krf = (1.0 / cutoff ** 3) * (eps - 1.0) / (2.0 * eps + 1.0)
    crf = (1.0 / cutoff) * (3.0 * eps) / (2.0 * eps + 1.0)
    force = ONE_4PI_EPS0 * (1.0 - 2.0 * krf * 1.0)
    assertEqualVec([force, 0, 0], r["forces"][0], tol)
    assertEqualVec([-force, 0, 0], r["forces"][1], tol)
    assertEqualVec([0, 0, 0], r["forces"][2], tol)
    assertEqualTo(2 * ONE_4PI_EPS0 * (1.0 + krf * 1.0 - crf), r["energy"], tol)


# --- :394-430 -----------------------------------------------------------------------------------
def testPeriodicExceptions(ev, F, tol=TOL):
    sliced = F(1)
    sliced.addParticle(1.0, 1, 0)
    sliced.addParticle(1.0, 1, 0)
    sliced.addException(0, 1, 1.0, 1.0, 0.0)
    sliced.setNonbondedMethod(sliced.CutoffPeriodic)
    sliced.setCutoffDistance(2.0)
    pos = [[0, 0, 0], [3, 0, 0]]
    r = ev(sliced, pos, cubic(4))
    force = ONE_4PI_EPS0 / (3 * 3)
    assertEqualVec([-force, 0, 0], r["forces"][0], tol)
    assertEqualVec([force, 0, 0], r["forces"][1], tol)
    assertEqualTo(ONE_4PI_EPS0 / 3, r["energy"], tol)
    sliced.setExceptionsUsePeriodicBoundaryConditions(True)
    r = ev(sliced, pos, cubic(4))
    force = ONE_4PI_EPS0 / (1 * 1)
    assertEqualVec([force, 0, 0], r["forces"][0], tol)
    assertEqualVec([-force, 0, 0], r["forces"][1], tol)
    assertEqualTo(ONE_4PI_EPS0 / 1, r["energy"], tol)


# --- :432-492 (SFMT stream replaced by a seeded NumPy generator) ----------------------------------
def testTriclinic(ev, F, tol=1e-4, iterations=50):
    a = np.array([3.1, 0, 0]); b = np.array([0.4, 3.5, 0]); c = np.array([-0.1, -0.5, 4.0])
    box = np.array([a, b, c])
    sliced = F(1)
    sliced.addParticle(1.0, 1, 0)
    sliced.addParticle(1.0, 1, 0)
    sliced.setNonbondedMethod(sliced.CutoffPeriodic)
    cutoff = 1.5
    sliced.setCutoffDistance(cutoff)
    rng = np.random.default_rng(0)
    eps = 78.3
    krf = (1.0 / cutoff ** 3) * (eps - 1.0) / (2.0 * eps + 1.0)
    crf = (1.0 / cutoff) * (3.0 * eps) / (2.0 * eps + 1.0)
    for _ in range(iterations):
        u = rng.random(6)
        p0 = a * u[0] + b * u[1] + c * u[2]
        p1 = a * u[3] + b * u[4] + c * u[5]
        delta = None; d2 = 100.0
        for i in (-1, 0, 1):
            for j in (-1, 0, 1):
                for k in (-1, 0, 1):
                    d = p1 - p0 + a * i + b * j + c * k
                    if d @ d < d2:
                        delta = d; d2 = float(d @ d)
        dist = math.sqrt(d2)
        r = ev(sliced, [p0, p1], box)
        if dist >= cutoff:
            assert r["energy"] == 0.0
            assertEqualVec([0, 0, 0], r["forces"][0], 0)
            assertEqualVec([0, 0, 0], r["forces"][1], 0)
        else:
            force = delta * ONE_4PI_EPS0 * (-1.0 / dist ** 3 + 2.0 * krf)
            assertEqualTo(ONE_4PI_EPS0 * (1.0 / dist + krf * dist * dist - crf), r["energy"], tol)
            assertEqualVec(force, r["forces"][0], tol)
            assertEqualVec(-force, r["forces"][1], tol)


# --- :614-681 -----------------------------------------------------------------------------------
def testDispersionCorrection(ev, F, tol=TOL):
    gridSize = 5
    numParticles = gridSize ** 3
    boxSize = gridSize * 0.7
    cutoff = boxSize / 3
    sliced = F(1)
    positions = []
    for i in range(gridSize):
        for j in range(gridSize):
            for k in range(gridSize):
                sliced.addParticle(0, 1.1, 0.5)
                positions.append([i * boxSize / gridSize, j * boxSize / gridSize, k * boxSize / gridSize])
    sliced.setNonbondedMethod(sliced.CutoffPeriodic)
    sliced.setCutoffDistance(cutoff)
    box = cubic(boxSize)
    energy1 = ev(sliced, positions, box)["energy"]
    sliced.setUseDispersionCorrection(False)
    energy2 = ev(sliced, positions, box)["energy"]
    term1 = (0.5 * 1.1 ** 12 / cutoff ** 9) / 9
    term2 = (0.5 * 1.1 ** 6 / cutoff ** 3) / 3
    expected = 8 * math.pi * numParticles * numParticles * (term1 - term2) / boxSize ** 3
    assertEqualTo(expected, energy1 - energy2, tol)
    numType2 = 0
    for i in range(0, numParticles, 2):
        sliced.setParticleParameters(i, 0, 1, 1)
        numType2 += 1
    numType1 = numParticles - numType2
    energy2 = ev(sliced, positions, box)["energy"]
    sliced.setUseDispersionCorrection(True)
    energy1 = ev(sliced, positions, box)["energy"]
    term1 = ((numType1 * (numType1 + 1)) // 2) * (0.5 * 1.1 ** 12 / cutoff ** 9) / 9
    term2 = ((numType1 * (numType1 + 1)) // 2) * (0.5 * 1.1 ** 6 / cutoff ** 3) / 3
    term1 += ((numType2 * (numType2 + 1)) // 2) * (1 * 1.0 ** 12 / cutoff ** 9) / 9
    term2 += ((numType2 * (numType2 + 1)) // 2) * (1 * 1.0 ** 6 / cutoff ** 3) / 3
    combinedSigma = 0.5 * (1 + 1.1)
    combinedEpsilon = math.sqrt(1 * 0.5)
    term1 += (numType1 * numType2) * (combinedEpsilon * combinedSigma ** 12 / cutoff ** 9) / 9
    term2 += (numType1 * numType2) * (combinedEpsilon * combinedSigma ** 6 / cutoff ** 3) / 3
    term1 /= (numParticles * (numParticles + 1)) // 2
    term2 /= (numParticles * (numParticles + 1)) // 2
    expected = 8 * math.pi * numParticles * numParticles * (term1 - term2) / boxSize ** 3
    assertEqualTo(expected, energy1 - energy2, tol)


# --- :760-813 -----------------------------------------------------------------------------------
def testSwitchingFunction(ev, F, method, pme=None, tol=TOL, fd_tol=1e-3):
    sliced = F(1)
    sliced.addParticle(0, 1.2, 1)
    sliced.addParticle(0, 1.4, 2)
    sliced.setNonbondedMethod(method)
    sliced.setCutoffDistance(2.0)
    sliced.setUseSwitchingFunction(True)
    sliced.setSwitchingDistance(1.5)
    sliced.setUseDispersionCorrection(False)
    if pme is not None:
        sliced.setPMEParameters(*pme)
    box = cubic(6)
    eps = SQRT_TWO
    r_ = 1.0
    while r_ < 2.5:
        res = ev(sliced, [[0, 0, 0], [r_, 0, 0]], box)
        x = 1.3 / r_
        expectedEnergy = 4.0 * eps * (x ** 12 - x ** 6)
        if r_ <= 1.5:
            sw = 1
        elif r_ >= 2.0:
            sw = 0
        else:
            t = (r_ - 1.5) / 0.5
            sw = 1 + t * t * t * (-10 + t * (15 - t * 6))
        assertEqualTo(sw * expectedEnergy, res["energy"], tol)
        delta = 1e-3
        e1 = ev(sliced, [[0, 0, 0], [r_ - delta, 0, 0]], box)["energy"]
        e2 = ev(sliced, [[0, 0, 0], [r_ + delta, 0, 0]], box)["energy"]
        assertEqualTo((e2 - e1) / (2 * delta), res["forces"][0][0], fd_tol)
        r_ += 0.1


# --- :815-871 (each force evaluated separately: force groups are the caller's concern) -------------
def testTwoForces(ev, F, tol=TOL):
    nb1 = F(1); nb1.addParticle(-1.5, 1, 1.2); nb1.addParticle(0.5, 1, 1.0)
    nb2 = F(1); nb2.addParticle(0.4, 1.4, 0.5); nb2.addParticle(0.3, 1.8, 1.0)
    pos = [[0, 0, 0], [1.5, 0, 0]]
    assertEqualTo(ONE_4PI_EPS0 * (-1.5 * 0.5) / 1.5 + 4.0 * math.sqrt(1.2 * 1.0) * ((1.0 / 1.5) ** 12 - (1.0 / 1.5) ** 6), ev(nb1, pos)["energy"], tol)
    assertEqualTo(ONE_4PI_EPS0 * (0.4 * 0.3) / 1.5 + 4.0 * math.sqrt(0.5 * 1.0) * ((1.6 / 1.5) ** 12 - (1.6 / 1.5) ** 6), ev(nb2, pos)["energy"], tol)
    nb1.setParticleParameters(0, -1.2, 1.1, 1.4)
    nb2.setParticleParameters(0, 0.5, 1.6, 0.6)
    assertEqualTo(ONE_4PI_EPS0 * (-1.2 * 0.5) / 1.5 + 4.0 * math.sqrt(1.4 * 1.0) * ((1.05 / 1.5) ** 12 - (1.05 / 1.5) ** 6), ev(nb1, pos)["energy"], tol)
    assertEqualTo(ONE_4PI_EPS0 * (0.5 * 0.3) / 1.5 + 4.0 * math.sqrt(0.6 * 1.0) * ((1.7 / 1.5) ** 12 - (1.7 / 1.5) ** 6), ev(nb2, pos)["energy"], tol)


# --- :883-945 -----------------------------------------------------------------------------------
def testParameterOffsets(ev, F, tol=1e-4):
    force = F(1)
    force.addParticle(0.0, 1.0, 0.5)
    force.addParticle(1.0, 0.5, 0.6)
    force.addParticle(-1.0, 2.0, 0.7)
    force.addParticle(0.5, 2.0, 0.8)
    force.addException(0, 3, 0.0, 1.0, 0.0)
    force.addException(2, 3, 0.5, 1.0, 1.5)
    force.addException(0, 1, 1.0, 1.5, 1.0)
    force.addGlobalParameter("p1", 0.0)
    force.addGlobalParameter("p2", 1.0)
    force.addParticleParameterOffset("p1", 0, 3.0, 0.5, 0.5)
    force.addParticleParameterOffset("p2", 1, 1.0, 1.0, 2.0)
    force.addExceptionParameterOffset("p1", 1, 0.5, 0.5, 1.5)
    positions = [[i, 0, 0] for i in range(4)]
    params = {"p1": 0.5, "p2": 1.5}
    particleCharge = [0.0 + 3.0 * 0.5, 1.0 + 1.0 * 1.5, -1.0, 0.5]
    particleSigma = [1.0 + 0.5 * 0.5, 0.5 + 1.0 * 1.5, 2.0, 2.0]
    particleEpsilon = [0.5 + 0.5 * 0.5, 0.6 + 2.0 * 1.5, 0.7, 0.8]
    qq = {}; sg = {}; ep = {}
    for i in range(4):
        for j in range(i + 1, 4):
            qq[i, j] = particleCharge[i] * particleCharge[j]
            sg[i, j] = 0.5 * (particleSigma[i] + particleSigma[j])
            ep[i, j] = math.sqrt(particleEpsilon[i] * particleEpsilon[j])
    qq[0, 3] = 0.0; sg[0, 3] = 1.0; ep[0, 3] = 0.0
    qq[2, 3] = 0.5 + 0.5 * 0.5; sg[2, 3] = 1.0 + 0.5 * 0.5; ep[2, 3] = 1.5 + 1.5 * 0.5
    qq[0, 1] = 1.0; sg[0, 1] = 1.5; ep[0, 1] = 1.0
    energy = 0.0
    for i in range(4):
        for j in range(i + 1, 4):
            dist = j - i
            x = sg[i, j] / dist
            energy += ONE_4PI_EPS0 * qq[i, j] / dist + 4.0 * ep[i, j] * (x ** 12 - x ** 6)
    assertEqualTo(energy, ev(force, positions, None, params)["energy"], tol)


_POS4 = [[0, 0, 0], [1.5, 0, 0], [0, 0.5, 0.5], [0.2, 1.3, 0]]


# --- :947-985 (explicit PME parameters: auto-selection is OpenMM's, a13) ---------------------------
def testEwaldExceptions(ev, F, pme=(3.0, 24, 24, 24), ljpme=(3.0, 24, 24, 24), tol=1e-4):
    force = F(1)
    force.setNonbondedMethod(force.LJPME)
    force.setCutoffDistance(1.0)
    force.setPMEParameters(*pme)
    force.setLJPMEParameters(*ljpme)
    force.addParticle(1.0, 0.5, 1.0)
    force.addParticle(1.0, 0.5, 1.0)
    force.addParticle(-1.0, 0.5, 1.0)
    force.addParticle(-1.0, 0.5, 1.0)
    box = cubic(2)
    e1 = ev(force, _POS4, box)["energy"]
    force.addException(0, 1, 0.2, 0.8, 2.0)
    force.setExceptionsUsePeriodicBoundaryConditions(True)
    e2 = ev(force, _POS4, box)["energy"]
    r = 0.5
    expectedChange = ONE_4PI_EPS0 * (0.2 - 1.0) / r + 4 * 2.0 * ((0.8 / r) ** 12 - (0.8 / r) ** 6) - 4 * 1.0 * ((0.5 / r) ** 12 - (0.5 / r) ** 6)
    assertEqualTo(expectedChange, e2 - e1, tol)


# --- :987-1029 ----------------------------------------------------------------------------------
def testDirectAndReciprocal(ev, F, pme=(3.0, 24, 24, 24), tol=1e-4):
    force = F(1)
    force.setNonbondedMethod(force.PME)
    force.setCutoffDistance(1.0)
    force.setPMEParameters(*pme)
    force.addParticle(1.0, 0.5, 1.0)
    force.addParticle(1.0, 0.5, 1.0)
    force.addParticle(-1.0, 0.5, 1.0)
    force.addParticle(-1.0, 0.5, 1.0)
    force.addException(0, 2, -2.0, 0.5, 3.0)
    box = cubic(2)
    e1 = ev(force, _POS4, box)["energy"]
    e2 = ev(force, _POS4, box, None, True, False)["energy"]
    e3 = ev(force, _POS4, box, None, False, True)["energy"]
    assertEqualTo(e1, e2 + e3, tol)
    assert e2 != 0 and e3 != 0
    force.setIncludeDirectSpace(False)
    e4 = ev(force, _POS4, box)["energy"]
    assertEqualTo(e3, e4, tol)


# --- construction of testNonbondedSlicing (:1031-1318): lambda emulated in an n=1 force by scaling q and eps
def dimer_lattice(numMolecules, L):
    """Lattice generator shared by testLargeSystem/testNonbondedSlicing (:1059-1076)."""
    M = int(numMolecules ** (1.0 / 3.0))
    if M * M * M < numMolecules:
        M += 1
    pos = []
    for k in range(numMolecules):
        iz = k // (M * M); iy = (k - iz * M * M) // M; ix = k - M * (iy + iz * M)
        center = np.array([ix + 0.5, iy + 0.5, iz + 0.5]) * L / M
        delta = np.array([0.5 - ix % 2, 0.5 - iy % 2, 0.5 - iz % 2]) / 2
        pos.append(center + delta); pos.append(center - delta)
    return np.array(pos)


def testNonbondedSlicing(ev, F, method, exceptions, lj, tol=TOL, pme=None, ljpme=None, seed=0):
    includeLJ = lj; includeCoulomb = not lj
    numMolecules = 100; numParticles = 200
    cutoff = 3.5
    L = 7.0 if exceptions else 10.0
    box = cubic(L)
    positions = dimer_lattice(numMolecules, L)
    rng = np.random.default_rng(seed)
    subset = rng.integers(0, 2, numParticles)
    q = lambda k: 1 - 2 * (k % 2)
    eps = 1.0

    def make(n):
        f = F(n)
        f.setNonbondedMethod(method)
        f.setCutoffDistance(cutoff)
        f.setUseDispersionCorrection(True)
        if pme is not None:
            f.setPMEParameters(*pme)
        if ljpme is not None:
            f.setLJPMEParameters(*ljpme)
        return f

    def fill(f, lam, sliced):
        # plain force: subset-1 atoms carry q*lam (Coulomb run) or eps*lam^2 (LJ run); sliced: lambdas do it
        for k in range(numParticles):
            s = int(subset[k])
            qs = (lam if (s == 1 and includeCoulomb) else 1.0) if not sliced else 1.0
            es = (lam * lam if (s == 1 and includeLJ) else 1.0) if not sliced else 1.0
            f.addParticle(q(k) * qs, 1, eps * es)
        if exceptions:
            for m in range(numMolecules):
                i, j = 2 * m, 2 * m + 1
                sc = 1.0
                if not sliced:
                    n1 = int(subset[i]) + int(subset[j])
                    sc = lam ** n1
                f.addException(i, j, q(i) * q(j) * (sc if includeCoulomb else 1.0), 1, eps * (sc if includeLJ else 1.0))

    results = {}
    for lam in (1.0, 0.0, 0.5):
        plain = make(1); fill(plain, lam, False)
        sliced = make(2); fill(sliced, lam, True)
        for k in range(numParticles):
            sliced.setParticleSubset(k, int(subset[k]))
        sliced.addGlobalParameter("lambda", lam)
        sliced.addGlobalParameter("lambdaSq", lam * lam)
        sliced.addScalingParameter("lambda", 0, 1, includeCoulomb, includeLJ)
        sliced.addScalingParameter("lambdaSq", 1, 1, includeCoulomb, includeLJ)
        sliced.addEnergyParameterDerivative("lambda")
        sliced.addEnergyParameterDerivative("lambdaSq")
        for dirflag, recflag in ((True, False), (False, True), (True, True)):
            r1 = ev(plain, positions, box, None, dirflag, recflag)
            r2 = ev(sliced, positions, box, None, dirflag, recflag)
            assertEqualTo(r1["energy"], r2["energy"], tol)
            assertForces(r1["forces"], r2["forces"], tol)
        results[lam] = (r1, r2)
    # E(lambda=1) - E(lambda=0) = sum of dE/dlambda (:1281-1286); slice energies are lambda-independent
    e1 = results[1.0][1]["energy"]; e0 = results[0.0][1]["energy"]
    d = results[0.0][1]["derivatives"]
    assertEqualTo(e1 - e0, d["lambda"] + d["lambdaSq"], tol)
    # E = sum lambda dE/dlambda + (slice 00) (:1310-1317, :1419-1423)
    r = results[0.5][1]
    total = float((r["lambdas"] * r["slice_energies"]).sum())
    assertEqualTo(r["energy"], total, tol)
    # sum of all slices at lambda=1 equals the unsliced energy
    assertEqualTo(results[1.0][0]["energy"], float(results[1.0][1]["slice_energies"].sum()), tol)


# --- :29-85  testInstantiateFromNonbondedForce ----------------------------------------------------------------------------------
# The reference builds an OpenMM NonbondedForce, converts it with SlicedNonbondedForce(force, 1) and requires both to agree
# (direct and reciprocal groups, before and after context.setParameter("p1", 1)).  OpenMM's NonbondedForce is not available here,
# so the source object is a plain 1-subset force filled through the same setters (the converting constructor only uses
# NonbondedForce's getters), and `ev` evaluates source and copy.
def testInstantiateFromNonbondedForce(ev, F, method, pme=None, tol=TOL):
    force = F(1)
    force.setCutoffDistance(2.0)
    force.setNonbondedMethod(method)
    force.addParticle(0.0, 1.0, 0.5)
    force.addParticle(1.0, 0.5, 0.6)
    force.addParticle(-1.0, 2.0, 0.7)
    force.addParticle(0.5, 2.0, 0.8)
    force.addParticle(-0.5, 2.0, 0.8)
    force.addException(0, 3, 0.0, 1.0, 0.0)
    force.addException(2, 3, 0.5, 1.0, 1.5)
    force.addException(0, 1, 1.0, 1.5, 1.0)
    force.addGlobalParameter("p1", 0.5)
    force.addGlobalParameter("p2", 1.0)
    force.addParticleParameterOffset("p1", 0, -2.0, 0.5, 0.5)
    force.addParticleParameterOffset("p2", 1, 1.0, 1.0, 2.0)
    force.addExceptionParameterOffset("p1", 1, 0.5, 0.5, 1.5)
    if pme is not None:
        force.setPMEParameters(*pme)
        force.setLJPMEParameters(*pme)
    sliced = F(force, 1)
    assert sliced.getNumSubsets() == 1 and sliced.getNumParticles() == 5 and sliced.getNumExceptions() == 3
    assert sliced.getNumParticleParameterOffsets() == 2 and sliced.getNumExceptionParameterOffsets() == 1
    assert sliced.getNonbondedMethod() == method and sliced.getCutoffDistance() == 2.0
    N = 5
    box = cubic(float(N))
    positions = [[i, 0, 0] for i in range(N)]
    periodic = method >= 2
    for params in (None, {"p1": 1.0}):
        for dirflag, recflag in ((True, False), (False, True)) if method >= 3 else ((True, True),):
            r1 = ev(force, positions, box if periodic else None, params, dirflag, recflag)
            r2 = ev(sliced, positions, box if periodic else None, params, dirflag, recflag)
            assertEqualTo(r1["energy"], r2["energy"], tol)
            assertForces(r1["forces"], r2["forces"], tol)


# --- :494-555 testLargeSystem: geometry generator (the comparison partner is the oracle, see the test files) -------------------------
def largeSystem(F, method, seed=0):
    numMolecules = 600; cutoff = 2.0; boxSize = 20.0
    rng = np.random.default_rng(seed)
    f = F(1)
    positions = np.zeros((2 * numMolecules, 3))
    for i in range(numMolecules):
        eps = 0.1 if i < numMolecules // 2 else 0.2
        f.addParticle(-1.0, 0.2, eps); f.addParticle(1.0, 0.1, eps)
        positions[2 * i] = boxSize * rng.random(3)
        positions[2 * i + 1] = positions[2 * i] + [1.0, 0.0, 0.0]
        f.addException(2 * i, 2 * i + 1, 0.0, 0.15, 0.0)
    f.setNonbondedMethod(method)
    f.setCutoffDistance(cutoff)
    return f, positions, cubic(boxSize)


# --- :683-758 testChangingParameters: dimer lattice generator; the parameter change applied to every fifth particle ----------------------
def changingParametersSystem(F):
    numMolecules = 600; cutoff = 2.0; boxSize = 20.0
    f = F(1)
    positions = np.zeros((2 * numMolecules, 3))
    M = int(numMolecules ** (1.0 / 3.0))
    if M * M * M < numMolecules:
        M += 1
    for k in range(numMolecules):
        iz = k // (M * M); iy = (k - iz * M * M) // M; ix = k - M * (iy + iz * M)
        x, y, z = (ix + 0.5) * boxSize / M, (iy + 0.5) * boxSize / M, (iz + 0.5) * boxSize / M
        dx, dy, dz = (0.5 - ix % 2) / 2, (0.5 - iy % 2) / 2, (0.5 - iz % 2) / 2
        eps = 0.1 if k < numMolecules // 2 else 0.2
        f.addParticle(-1.0, 0.2, eps); f.addParticle(1.0, 0.1, eps)
        positions[2 * k] = [x + dx, y + dy, z + dz]
        positions[2 * k + 1] = [x - dx, y - dy, z - dz]
        f.addException(2 * k, 2 * k + 1, 0.0, 0.15, 0.0)
    f.setNonbondedMethod(4)
    f.setCutoffDistance(cutoff)
    return f, positions, cubic(boxSize)


def changeEveryFifthParticle(f):
    for i in range(0, f.getNumParticles(), 5):      # :746-752
        charge, sigma, epsilon = f.getParticleParameters(i)
        f.setParticleParameters(i, 1.5 * charge, 1.1 * sigma, 1.7 * epsilon)


# --- :557-612 testHugeSystem: energy change along the force direction (size-independent property; gridSize 150 = 3.4 M particles) -------
def testHugeSystem(evEnergy, evForces, F, gridSize=150, tol=1e-4, seed=0, scaledDown=False):
    spacing = 0.3; boxSize = gridSize * spacing
    force = F(1)
    force.setNonbondedMethod(F.CutoffPeriodic)
    force.setCutoffDistance(1.0)
    force.setUseSwitchingFunction(True)
    force.setSwitchingDistance(0.9)
    rng = np.random.default_rng(seed)
    g = np.stack(np.meshgrid(np.arange(gridSize), np.arange(gridSize), np.arange(gridSize), indexing="ij"), -1).reshape(-1, 3)
    positions = g * spacing + rng.random(g.shape) * 0.1
    force.addParticles(np.zeros(len(g)), np.full(len(g), 0.1), np.ones(len(g))) if hasattr(force, "addParticles") else [force.addParticle(0.0, 0.1, 1.0) for _ in range(len(g))]
    box = cubic(boxSize)
    f = np.asarray(evForces(force, positions, box))
    norm = math.sqrt(float((f * f).sum()))
    delta = 0.3
    if scaledDown:      # smaller grids: keep the per-atom displacement of the 150^3 original (|F| grows like sqrt(N)) ...
        delta *= math.sqrt(len(g) / 150.0 ** 3)
    step = 0.5 * delta / norm
    e2 = evEnergy(force, positions - f * step, box)
    e3 = evEnergy(force, positions + f * step, box)
    if scaledDown:      # ... and test the energy DIFFERENCE itself, which is then small against |E|
        assert abs((e2 - e3) - norm * delta) <= 20 * tol * norm * delta, (e2 - e3, norm * delta)
    else:
        assertEqualTo(e2, e3 + norm * delta, tol)


# --- :1320-1457 testScalingParameterSeparation -------------------------------------------------------------------------------------
def testScalingParameterSeparation(ev, F, method, exceptions, pme=None, ljpme=None, tol=1e-4, seed=0):
    numMolecules = 100; numParticles = 200; cutoff = 3.5
    L = 7.0 if exceptions else 10.0
    box = cubic(L)
    positions = dimer_lattice(numMolecules, L)
    q = lambda k: 1 - 2 * (k % 2)

    def base(n):
        f = F(n)
        f.setNonbondedMethod(method); f.setCutoffDistance(cutoff); f.setUseDispersionCorrection(True)
        if pme is not None:
            f.setPMEParameters(*pme)
        if ljpme is not None:
            f.setLJPMEParameters(*ljpme)
        for k in range(numMolecules):
            i, j = 2 * k, 2 * k + 1
            f.addParticle(q(i), 1, 1); f.addParticle(q(j), 1, 1)
            if exceptions:
                f.addException(i, j, q(i) * q(j), 1, 1)
        return f

    sliced1, sliced2 = base(2), base(2)
    rng = np.random.default_rng(seed)
    for k in range(numParticles):
        if rng.random() < 0.5:
            sliced1.setParticleSubset(k, 1); sliced2.setParticleSubset(k, 1)
    lam, value = 0.5, 0.3
    sliced1.addGlobalParameter("lambda", lam)
    sliced1.addScalingParameter("lambda", 0, 1, True, True)
    sliced1.addEnergyParameterDerivative("lambda")
    sliced2.addGlobalParameter("lambdaCoulomb", lam)
    sliced2.addGlobalParameter("lambdaLJ", lam)
    sliced2.addScalingParameter("lambdaCoulomb", 0, 1, True, False)
    sliced2.addScalingParameter("lambdaLJ", 0, 1, False, True)
    sliced2.addEnergyParameterDerivative("lambdaCoulomb")
    sliced2.addEnergyParameterDerivative("lambdaLJ")
    sliced1.addGlobalParameter("alpha", value)
    sliced1.addScalingParameter("alpha", 0, 0, True, True)
    sliced1.addEnergyParameterDerivative("alpha")
    sliced1.addGlobalParameter("beta", value)
    sliced1.addScalingParameter("beta", 1, 1, True, True)
    sliced1.addEnergyParameterDerivative("beta")
    sliced2.addGlobalParameter("gamma", value)
    sliced2.addScalingParameter("gamma", 0, 0, True, True)
    sliced2.addScalingParameter("gamma", 1, 1, True, True)
    sliced2.addEnergyParameterDerivative("gamma")
    periodic = method >= 2
    groups = ((True, True), (True, False), (False, True)) if method >= 3 else ((True, True),)
    for dirflag, recflag in groups:       # overall, direct space, reciprocal space
        r1 = ev(sliced1, positions, box if periodic else None, None, dirflag, recflag)
        r2 = ev(sliced2, positions, box if periodic else None, None, dirflag, recflag)
        d1, d2 = r1["derivatives"], r2["derivatives"]
        assertEqualTo(r1["energy"], r2["energy"], tol)
        assertForces(r1["forces"], r2["forces"], tol)
        assertEqualTo(d1["lambda"], d2["lambdaCoulomb"] + d2["lambdaLJ"], tol)
        assertEqualTo(r1["energy"], lam * d1["lambda"] + value * (d1["alpha"] + d1["beta"]), tol)
        assertEqualTo(d1["alpha"] + d1["beta"], d2["gamma"], tol)


# --- independent absolute pin of the Ewald / PME half (not in the reference's suite) -----------------------------
# Rock salt: simple-cubic sites of spacing r0 with alternating charges +-1.  Published constants:
#   E / ion pair            = -M k / r0,            M = 1.7475645946331822  (Madelung constant of NaCl)
#   like-charge sub-lattice = fcc one-component plasma in its neutralising background: E / ion = -0.8958736152 k / r_s
#                             (r_s = Wigner-Seitz radius of the fcc lattice; Fuchs 1935, e.g. Baus & Hansen, Phys. Rep. 59 (1980) table 1)
# With cations and anions as two subsets the diagonal slices carry the fcc sums (which exercises the per-slice background term, Q3) and
# the cross slice the rest.  The paths pinned: ReferencePME.cpp:400-496, 598-702 and ReferenceSlicedLJCoulombIxn.cpp:203-222, 256-358.
MADELUNG_NACL = 1.7475645946331822
MADELUNG_FCC_OCP = 0.8958736152


def rockSalt(F, nsub, method, cells=4, r0=0.25, cutoff=0.9, alpha=4.8, grid=96, shift=(0.013, 0.007, 0.021)):
    m = 2 * cells
    L = m * r0
    idx = np.stack(np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij"), -1).reshape(-1, 3)
    sign = np.where(idx.sum(1) % 2 == 0, 1.0, -1.0)
    pos = idx * r0 + np.asarray(shift)
    ff = F(nsub)
    ff.setNonbondedMethod(method); ff.setCutoffDistance(cutoff); ff.setUseDispersionCorrection(False)
    ff.setPMEParameters(alpha, grid, grid, grid)
    if method == 5:
        ff.setLJPMEParameters(alpha, grid // 2, grid // 2, grid // 2)
    if method == 3:
        ff.ewaldKmax = (24, 24, 24)      # explicit k-vector bounds for the engine's host mirror (context.py calcEwaldParameters)
    for i, s in enumerate(sign):
        ff.addParticle(float(s), 0.2, 0.0)
        if nsub == 2:
            ff.setParticleSubset(i, 0 if s > 0 else 1)
    return ff, pos, cubic(L), len(sign)


def testMadelung(ev, F, method, nsub, tol=1e-5, force_tol=None, cells=4, grid=96, **kw):
    ff, pos, box, n = rockSalt(F, nsub, method, cells=cells, grid=grid)
    r0 = 0.25
    r = ev(ff, pos, box, **kw)
    total = -(n / 2) * MADELUNG_NACL * ONE_4PI_EPS0 / r0
    assertEqualTo(total, r["energy"], tol)
    # every ion sits on an inversion centre: the forces vanish (scale: the nearest-neighbour pair force k / r0^2)
    fscale = ONE_4PI_EPS0 / r0 ** 2
    assert np.abs(r["forces"]).max() <= (force_tol if force_tol is not None else tol) * fscale, np.abs(r["forces"]).max() / fscale
    se = np.asarray(r["slice_energies"])
    if nsub == 1:
        assertEqualTo(total, se[0, 0], tol)
    else:
        rs = (3.0 / (4.0 * math.pi) * (2 * r0) ** 3 / 4.0) ** (1.0 / 3.0)
        like = -(n / 2) * MADELUNG_FCC_OCP * ONE_4PI_EPS0 / rs
        assertEqualTo(like, se[0, 0], tol)              # cation-cation
        assertEqualTo(like, se[2, 0], tol)              # anion-anion
        assertEqualTo(total - 2 * like, se[1, 0], tol)  # cross slice
    assert np.abs(se[:, 1]).max() <= tol * abs(total)   # no Lennard-Jones / dispersion energy (eps = 0)
