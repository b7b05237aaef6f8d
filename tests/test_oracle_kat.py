"""Pins the CPU oracle (oracle/snb_oracle.c) against the closed-form known-answer tests of the
reference's own test-suite and the reference's slicing identities.  CPU only."""
import functools
import importlib

import numpy as np
import pytest

import kat_cases as K

PME = (3.2, 32, 32, 32)


@pytest.fixture(scope="module")
def F(snb):
    return snb.SlicedNonbondedForce


@pytest.fixture(scope="module")
def ev(oracle):
    def _ev(force, positions, box=None, parameters=None, include_direct=True, include_reciprocal=True, **kw):
        return oracle.evaluate(force, np.asarray(positions, dtype=float), box, parameters, include_direct, include_reciprocal, **kw)
    return _ev


@pytest.mark.parametrize("case", ["testCoulomb", "testLJ", "testExclusionsAnd14", "testCutoff", "testCutoff14", "testPeriodic",
                                  "testPeriodicExceptions", "testTriclinic", "testDispersionCorrection", "testTwoForces",
                                  "testParameterOffsets", "testEwaldExceptions", "testDirectAndReciprocal"])
def test_reference_kat(case, ev, F):
    getattr(K, case)(ev, F)


@pytest.mark.parametrize("method", [3, 4, 5])
@pytest.mark.parametrize("nsub", [1, 2])
def test_madelung_constants(method, nsub, ev, F):
    """Independent absolute pin of the oracle's Ewald k-sum, PME and LJPME paths: the Madelung constant of rock salt and, with cations and
    anions as two subsets, the fcc one-component-plasma constant on the diagonal slices (per-slice background term included)."""
    kw = dict(kmax=(24, 24, 24)) if method == 3 else {}
    K.testMadelung(ev, F, method, nsub, tol=1e-6, **kw)
    if method == 4 and nsub == 2:
        K.testMadelung(ev, F, method, nsub, tol=1e-6, force_tol=5e-6, cells=12, grid=160)      # the 13 824-ion lattice the GPU test runs on the brick kernels (coarser mesh: 1.3e-6 of k/r0^2 on the forces)


def test_switching_function(ev, F):
    K.testSwitchingFunction(ev, F, 1)
    K.testSwitchingFunction(ev, F, 4, pme=(2.0, 30, 30, 30))


@pytest.mark.parametrize("method", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("exceptions", [False, True])
@pytest.mark.parametrize("lj", [False, True])
def test_nonbonded_slicing(method, exceptions, lj, ev, F):
    L = 7.0 if exceptions else 10.0
    n = 28 if exceptions else 40
    kw = {}
    if method == 3:
        kw = dict(pme=(1.0, 0, 0, 0), kmax=(8, 8, 8))
    ev2 = functools.partial(ev, **kw) if kw else ev
    K.testNonbondedSlicing(ev2, F, method, exceptions, lj,
                           pme=(1.0, n, n, n) if method in (4, 5) else ((1.0, 0, 0, 0) if method == 3 else None),
                           ljpme=(1.0, n, n, n) if method == 5 else None)


@pytest.mark.parametrize("method", [0, 1, 2, 4, 5])
def test_instantiate_from_nonbonded_force(method, ev, F):
    K.testInstantiateFromNonbondedForce(ev, F, method, pme=(1.0, 20, 20, 20) if method >= 4 else None)


@pytest.mark.parametrize("method", [2, 4, 5])
@pytest.mark.parametrize("exceptions", [False, True])
def test_scaling_parameter_separation(method, exceptions, ev, F):
    n = 28 if exceptions else 40
    K.testScalingParameterSeparation(ev, F, method, exceptions, pme=(1.0, n, n, n) if method >= 4 else None, ljpme=(1.0, n, n, n) if method == 5 else None)


def test_huge_system_property_small(ev, F):
    """testHugeSystem's finite-difference property on a 12^3 grid (the 150^3 original runs on the GPU)."""
    K.testHugeSystem(lambda f, x, b: ev(f, x, b)["energy"], lambda f, x, b: ev(f, x, b)["forces"], F, gridSize=12, scaledDown=True)


def test_changing_parameters_system(ev, F):
    """testChangingParameters' system (TestSlicedNonbondedForce.h:683-758) on the oracle: the direct and the reciprocal force group add up to
    the whole, before and after every fifth particle is changed, and the forces stay the gradient of the energy (the GPU test compares
    the engine's updateParametersInContext with these numbers)."""
    force, pos, box = K.changingParametersSystem(F)
    force.setPMEParameters(1.5, 48, 48, 48)
    energies = []
    for _ in range(2):
        d, r, t = ev(force, pos, box, None, True, False), ev(force, pos, box, None, False, True), ev(force, pos, box)
        K.assertEqualTo(t["energy"], d["energy"] + r["energy"], 1e-10)
        K.assertForces(t["forces"], d["forces"] + r["forces"], 1e-10)
        step = 1e-4 * t["forces"] / np.linalg.norm(t["forces"])
        e2 = ev(force, pos - step, box)["energy"]; e1 = ev(force, pos + step, box)["energy"]
        K.assertEqualTo(np.linalg.norm(t["forces"]), (e2 - e1) / 2e-4, 1e-4)
        energies.append(t["energy"])
        K.changeEveryFifthParticle(force)
    assert abs(energies[0] - energies[1]) > 1.0


def test_fft_against_numpy(oracle):
    rng = np.random.default_rng(1)
    for shape in [(28, 25, 30), (21, 25, 27), (8, 6, 10), (7, 11, 13)]:
        a = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
        np.testing.assert_allclose(oracle.fft3d(a, -1), np.fft.fftn(a), rtol=0, atol=1e-10 * a.size ** 0.5)
        np.testing.assert_allclose(oracle.fft3d(a, +1), np.fft.ifftn(a) * a.size, rtol=0, atol=1e-10 * a.size ** 0.5)


def test_pme_converges_to_ewald(ev, F, oracle):
    """PME (fine grid) must agree with the classic Ewald k-sum restated from
    ReferenceSlicedLJCoulombIxn.cpp:256-358 -- two independent reciprocal paths of the reference."""
    rng = np.random.default_rng(5)
    n = 60; L = 3.0
    pos = rng.random((n, 3)) * L
    q = rng.uniform(-1, 1, n); q -= q.mean()
    f = {}
    for method in (3, 4):
        ff = F(3)
        ff.setNonbondedMethod(method); ff.setCutoffDistance(1.2); ff.setUseDispersionCorrection(False)
        ff.setPMEParameters(3.0, 60, 60, 60)
        for i in range(n):
            ff.addParticle(q[i], 0.2, 0.3); ff.setParticleSubset(i, i % 3)
        ff.addGlobalParameter("l", 0.6); ff.addScalingParameter("l", 0, 2, True, False)
        f[method] = ev(ff, pos, K.cubic(L), kmax=(15, 15, 15))
    K.assertEqualTo(f[3]["energy"], f[4]["energy"], 1e-5)
    np.testing.assert_allclose(f[3]["slice_energies"], f[4]["slice_energies"], rtol=0, atol=2e-3)
    K.assertForces(f[3]["forces"], f[4]["forces"], 1e-4)


def test_force_is_energy_gradient(ev, F):
    """Forces of every method are -dE/dx of the lambda-weighted energy (finite differences)."""
    rng = np.random.default_rng(7)
    n = 24; L = 2.6
    pos = rng.random((n, 3)) * L
    for method in (2, 4, 5):
        ff = F(2)
        ff.setNonbondedMethod(method); ff.setCutoffDistance(1.1)
        ff.setPMEParameters(3.0, 36, 36, 36); ff.setLJPMEParameters(3.0, 36, 36, 36)
        for i in range(n):
            ff.addParticle((-1) ** i * 0.5, 0.25, 0.4); ff.setParticleSubset(i, i % 2)
        ff.addException(0, 1, 0.1, 0.2, 0.3); ff.addException(2, 5, 0.0, 0.2, 0.0)
        ff.addGlobalParameter("a", 0.7); ff.addGlobalParameter("b", 0.4)
        ff.addScalingParameter("a", 0, 1, True, False); ff.addScalingParameter("b", 0, 1, False, True)
        r0 = ev(ff, pos, K.cubic(L))
        for (i, d) in [(0, 0), (3, 1), (5, 2), (10, 0)]:
            h = 1e-5
            p1 = pos.copy(); p1[i, d] += h
            p2 = pos.copy(); p2[i, d] -= h
            g = (ev(ff, p1, K.cubic(L))["energy"] - ev(ff, p2, K.cubic(L))["energy"]) / (2 * h)
            # cutoff methods are discontinuous at rc; tolerance reflects the LJPME/PME mesh smoothness
            K.assertEqualTo(-g, r0["forces"][i, d], 2e-3 if method != 2 else 1e-4)
