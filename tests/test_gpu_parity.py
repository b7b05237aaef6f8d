"""Parity tests proper (-m gpu): the HIP engine, driven through the C ABI (include/snb.h) by the host-side mirror
of the reference interface, against (a) the reference's closed-form known answers and (b) the CPU oracle on
identical seeded inputs.  Tolerances: 1e-3 single / 1e-5 double relative with the reference's max(|x|,1) scaling
(BASELINE.json north_star; openmmapi/include/internal/AssertionUtilities.h:7-26)."""
import functools
import os

import numpy as np
import pytest

import kat_cases as K
import systems

pytestmark = pytest.mark.gpu

TOLS = {"single": 1e-3, "double": 1e-5, "mixed": 1e-3}      # mixed: single-precision arithmetic, 64-bit fixed-point force sums (snb.h SNB_MIXED)


def make_ev(snb, precision, **opts):
    def ev(force, positions, box=None, parameters=None, include_direct=True, include_reciprocal=True):
        positions = np.asarray(positions, dtype=float).reshape(-1, 3)
        system = snb.System()
        for _ in range(force.getNumParticles()):
            system.addParticle(1.0)
        if box is not None:
            system.setDefaultPeriodicBoxVectors(*np.asarray(box, dtype=float))
        system.addForce(force)
        force.setForceGroup(0); force.setReciprocalSpaceForceGroup(1)
        ctx = snb.Context(system, precision=precision, **opts)
        if parameters:
            for k, v in parameters.items():
                ctx.setParameter(k, v)
        ctx.setPositions(positions)
        groups = (1 if include_direct else 0) | (2 if include_reciprocal else 0)
        st = ctx.getState(getEnergy=True, getForces=True, getParameterDerivatives=True, groups=groups)
        kern = ctx._kernelFor(force)
        return dict(energy=st.getPotentialEnergy(), forces=st.getForces(), derivatives=st.getEnergyParameterDerivatives(),
                    slice_energies=kern.lastSliceEnergies.copy(), lambdas=kern._lastLambdas.copy(), stats=kern.getStats())
    return ev


@pytest.fixture(scope="module")
def F(snb):
    return snb.SlicedNonbondedForce


@pytest.fixture(scope="module", params=["single", "double", "mixed"])
def prec(request):
    return request.param


def test_native_library_loaded(snb):
    L = snb.capi.lib()
    assert L.snb_abi_version() == snb.capi.SNB_ABI_VERSION


@pytest.mark.parametrize("case", ["testCoulomb", "testLJ", "testExclusionsAnd14", "testCutoff", "testCutoff14", "testPeriodic",
                                  "testPeriodicExceptions", "testTriclinic", "testDispersionCorrection", "testTwoForces",
                                  "testParameterOffsets", "testEwaldExceptions", "testDirectAndReciprocal"])
def test_reference_kat(case, snb, F, prec):
    """The reference's closed-form known answers at the reference's own tolerance, TOL = 1e-4 on every platform and precision
    (tests/TestSlicedNonbondedForce.h:27, 106-108, 132-134, 254-259, 388-391, 487-489); testEwaldExceptions is 1e-3 on a
    single-precision GPU platform there too (:620-622)."""
    tol = K.TOL
    if case == "testEwaldExceptions" and prec != "double":
        tol = 1e-3
    kw = dict(tol=tol)
    if case == "testTriclinic":
        kw["iterations"] = 12
    getattr(K, case)(make_ev(snb, prec), F, **kw)


@pytest.mark.parametrize("method", [3, 4, 5])
@pytest.mark.parametrize("nsub", [1, 2])
def test_madelung_constants(method, nsub, snb, F, prec):
    """The engine against two published lattice constants (no oracle involved): rock salt's Madelung constant for the total energy and
    the cross slice, the fcc one-component-plasma constant (neutralising background) for the like-charge slices; forces vanish by
    symmetry.  512 ions exercise the small-box paths (host tile lists, per-pair wrap, atomic spreader), 13 824 ions the GPU builder and
    the brick kernels.  Classic Ewald, PME and LJPME (eps = 0)."""
    tol = 1e-5 if prec == "double" else 1e-3
    ev = make_ev(snb, prec)
    K.testMadelung(ev, F, method, nsub, tol=tol)
    if method != 3:
        K.testMadelung(ev, F, method, nsub, tol=tol, cells=12, grid=160)


def test_switching_function(snb, F, prec):
    tol = K.TOL                                       # TestSlicedNonbondedForce.h:800 (TOL) and :811 (finite difference, 1e-3), all platforms
    K.testSwitchingFunction(make_ev(snb, prec), F, 1, tol=tol, fd_tol=1e-3)
    if prec == "double":
        K.testSwitchingFunction(make_ev(snb, prec), F, 4, pme=(2.0, 30, 30, 30), tol=tol)


@pytest.mark.parametrize("method", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("exceptions", [False, True])
@pytest.mark.parametrize("lj", [False, True])
def test_nonbonded_slicing(method, exceptions, lj, snb, F, prec):
    n = 28 if exceptions else 40
    Fm = F
    if method == 3:   # classic Ewald: explicit alpha and kmax (auto-selection is OpenMM's calcEwaldParameters, unpinned)
        def Fm(nsub):
            f = F(nsub); f.ewaldKmax = (8, 8, 8); return f
    K.testNonbondedSlicing(make_ev(snb, prec), Fm, method, exceptions, lj, tol=1e-3 if prec != "double" else K.TOL,      # TestSlicedNonbondedForce.h:1039
                           pme=(1.0, n, n, n) if method in (4, 5) else ((1.0, 0, 0, 0) if method == 3 else None), ljpme=(1.0, n, n, n) if method == 5 else None)


def test_ewald_vs_oracle(snb, F, oev, prec):
    """Classic Ewald k-sum kernels (ewald.hip) against the oracle's restatement of ReferenceSlicedLJCoulombIxn.cpp:256-358."""
    force, pos, box = systems.random_box(F, 1500, 3, 3, 2.6, 1.0, pme=(2.6283, 0, 0, 0))
    force.ewaldKmax = (11, 11, 11)
    _compare(make_ev(snb, prec), oev, force, pos, box, TOLS[prec], kmax=(11, 11, 11))


def _compare(ev, oev, force, pos, box, tol, **okw):
    r = ev(force, pos, box)
    o = oev(force, pos, box, **okw)
    K.assertEqualTo(o["energy"], r["energy"], tol)
    S = o["slice_energies"].shape[0]
    for s in range(S):
        for t in range(2):
            K.assertEqualTo(o["slice_energies"][s, t], r["slice_energies"][s, t], tol)
    fo, fr = o["forces"], r["forces"]
    scale = np.maximum(np.linalg.norm(fo, axis=1), 1.0)
    err = np.linalg.norm(fo - fr, axis=1) / scale
    assert err.max() <= tol, "max force error %g at atom %d" % (err.max(), int(err.argmax()))
    for k, v in o["derivatives"].items():
        K.assertEqualTo(v, r["derivatives"][k], tol)
    return r, o


@pytest.fixture(scope="module")
def oev(oracle):
    def _o(force, positions, box=None, parameters=None, include_direct=True, include_reciprocal=True, **kw):
        return oracle.evaluate(force, np.asarray(positions, dtype=float), box, parameters, include_direct, include_reciprocal, **kw)
    return _o


CASES = [
    # name, n, nsub, method, L, cutoff, pme, ljpme, switch
    ("C1_nocutoff_1000_n2", 1000, 2, 0, 2.154, 1.0, None, None, False),
    ("rf_periodic_3000_n3", 3000, 3, 2, 3.2, 1.0, None, None, True),
    ("cutoff_nonperiodic_2000_n2", 2000, 2, 1, 2.8, 1.0, None, None, False),
    ("pme_4096_n2", 4096, 2, 4, 3.5, 1.0, (2.6283, 32, 32, 32), None, False),
    ("pme_6000_n4", 6000, 4, 4, 4.0, 1.0, (2.6283, 36, 36, 36), None, True),
    ("ljpme_3000_n4", 3000, 4, 5, 3.2, 1.0, (2.6283, 28, 28, 28), (2.6283, 20, 20, 20), False),
    ("ljpme_brickgroup2_6000_n3", 6000, 3, 5, 4.0, 1.0, (2.6283, 36, 36, 36), (2.6283, 18, 18, 18), False),
    ("pme_smallbox_wrap_600_n2", 600, 2, 4, 2.05, 1.0, (2.6283, 20, 20, 20), None, False),
    ("pme_mesh_factors_11_13_4096_n2", 4096, 2, 4, 3.5, 1.0, (2.6283, 33, 39, 44), None, False),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_random_system_vs_oracle(case, snb, F, oev, prec):
    name, n, nsub, method, L, cutoff, pme, ljpme, switch = case
    force, pos, box = systems.random_box(F, n, nsub, method, L, cutoff, pme=pme, ljpme=ljpme, switch=switch)
    r, o = _compare(make_ev(snb, prec), oev, force, pos, box, TOLS[prec])
    assert r["stats"].n_tiles > 0


def test_sparse_subsets_stay_on_gpu_builder(snb, F, oev, prec):
    """Subsets that are NOT compact -- scattered single atoms ('ions'), a hollow spherical shell, a sphere centred on the box corner
    (it straddles the periodic boundary in all three directions) -- must neither lose pairs nor push the engine off its GPU
    neighbour builder: blocks are segmented at jumps of the sorted order and re-imaged compactly (neighbor.hip, k_nbJumpFlags)."""
    n, L = 13824, 6.0
    force, pos, box = systems.random_box(F, n, 4, 4, L, 1.0, pme=(2.6283, 48, 48, 48))
    sub = np.zeros(n, dtype=int)
    d = pos - 0.5 * L
    r = np.linalg.norm(d, axis=1)
    sub[(r > 2.2) & (r < 2.6)] = 2                                  # hollow shell: a column crosses it twice, far apart in z
    dc = pos - L * np.round(pos / L)                                # minimum image to the corner (0,0,0)
    sub[np.linalg.norm(dc, axis=1) < 1.3] = 3                       # eight octants of one sphere, one per box corner
    sub[np.arange(n) % 97 == 5] = 1                                 # scattered single atoms
    for i in range(n):
        force.setParticleSubset(i, int(sub[i]))
    assert all((sub == k).sum() > 32 for k in range(4))
    r, o = _compare(make_ev(snb, prec), oev, force, pos, box, TOLS[prec])
    assert r["stats"].n_tiles > 0 and r["stats"].n_host_rebuilds == 0


BIG_CASES = [
    # boxes large enough for the GPU neighbour builder and the PME brick kernels (L >= ~5.5 nm at this density)
    # name, n, nsub, method, L, pme, ljpme
    ("pme_oddgrid45_13824_n3", 13824, 3, 4, 6.0, (2.6283, 45, 45, 45), None),                       # odd nz: f64 LDS accumulation, odd-length z pairs
    ("ljpme_grids48_24_13824_n3", 13824, 3, 5, 6.0, (2.6283, 48, 48, 48), (2.6283, 24, 24, 24)),    # dispersion mesh: bricks of column groups
    ("pme_grid54_13824_n4", 13824, 4, 4, 6.0, (2.6283, 54, 54, 54), None),                          # 54 = 6 x 9 two-pass FFT split
    ("pme_grid54x54x50_13824_n3", 13824, 3, 4, 6.0, (2.6283, 54, 54, 50), None),                    # plane path (square 54 x 54 planes) with a z length that has no two-pass split: staged inverse z FFT behind the mix
    ("pme_grid54x60x54_13824_n2", 13824, 2, 4, 6.0, (2.6283, 54, 60, 54), None),                    # non-square planes (round 4: plane path with run-time splits 6 x 9 and 6 x 10; nx = nz, Q1)
    ("pme_grid48x56x48_13824_n3", 13824, 3, 4, 6.0, (2.6283, 48, 56, 48), None),                    # non-square planes, 6 x 8 and 7 x 8: neither axis has a kernel of its own
]


@pytest.mark.parametrize("case", BIG_CASES, ids=[c[0] for c in BIG_CASES])
def test_gpu_builder_and_brick_kernels_vs_oracle(case, snb, F, oev, prec):
    name, n, nsub, method, L, pme, ljpme = case
    force, pos, box = systems.random_box(F, n, nsub, method, L, 1.0, pme=pme, ljpme=ljpme)
    r, o = _compare(make_ev(snb, prec), oev, force, pos, box, TOLS[prec])
    assert r["stats"].n_tiles > 0 and r["stats"].n_host_rebuilds == 0
    if "x" in name.split("_")[1] and prec != "double" and name.split("_")[1].startswith(("grid54x60", "grid48x56")) and not any(k in os.environ for k in ("SNB_NO_FUSED_Z", "SNB_NO_OWN_SPREAD", "SNB_OWN_SLABS", "SNB_NO_PLANE_FFT", "SNB_NO_RECT_PLANES")):
        t = [int(x) for x in r["stats"].n_kernel_timed]      # stamp slots of the first (eager) step: 4 = plane kernel / fused x kernel, 3 and 5 = the y passes of the three-pass pipeline
        assert t[4] > 0 and t[3] == 0 and t[5] == 0, ("rectangular planes must run the plane path", t)


FORCE_ONLY_CASES = [
    # name, n, nsub, method, L, pme, ljpme, switch
    ("pme_13824_n4", 13824, 4, 4, 6.0, (2.6283, 48, 48, 48), None, False),        # packed pair kernel (polynomial Ewald), brick PME, graph replay
    ("rf_13824_n3", 13824, 3, 2, 6.0, None, None, False),                         # packed pair kernel, reaction field
    ("rf_switch_13824_n2", 13824, 2, 2, 6.0, None, None, True),                   # switching function: scalar forces-only kernel
    ("ljpme_13824_n3", 13824, 3, 5, 6.0, (2.6283, 48, 48, 48), (2.6283, 24, 24, 24), False),
    ("pme_smallbox_4096_n2", 4096, 2, 4, 3.5, (2.6283, 32, 32, 32), None, False),  # host-built lists, atomic spreader
]


@pytest.mark.parametrize("case", FORCE_ONLY_CASES, ids=[c[0] for c in FORCE_ONLY_CASES])
def test_forces_only_steps_vs_oracle(case, snb, F, oev, prec):
    """The production path: forces-only evaluations (no energy, no parameter derivatives) run the packed single-precision pair
    kernel and replay a captured hipGraph from the second step on; every step's forces must match the oracle's."""
    name, n, nsub, method, L, pme, ljpme, switch = case
    force, pos, box = systems.random_box(F, n, nsub, method, L, 1.0, pme=pme, ljpme=ljpme, switch=switch, derivatives=False)
    system = snb.System()
    for _ in range(n):
        system.addParticle(1.0)
    system.setDefaultPeriodicBoxVectors(*box)
    system.addForce(force)
    ctx = snb.Context(system, precision=prec, neighbor_padding=0.1, rebuild_interval=10)
    rng = np.random.default_rng(7)
    tol = TOLS[prec]
    for step in range(4):
        ctx.setPositions(pos)
        fr = ctx.getState(getForces=True).getForces()
        fo = oev(force, pos, box)["forces"]
        err = np.linalg.norm(fo - fr, axis=1) / np.maximum(np.linalg.norm(fo, axis=1), 1.0)
        assert err.max() <= tol, "step %d: max force error %g at atom %d" % (step, err.max(), int(err.argmax()))
        pos = pos + rng.normal(0.0, 0.004, pos.shape)      # the list (skin 0.1 nm) is reused, the graph replayed


def test_alternating_position_buffers_keep_their_step_graphs(snb):
    """A caller that alternates between several device position buffers (double buffering against its integrator) must get, from every
    buffer, what a fresh single-buffer engine gives for those coordinates: the engine keeps one captured step graph per buffer (a handful),
    drops them all at a rebuild and evicts the oldest beyond its capacity (six buffers here, capacity four)."""
    import torch
    import bench
    w = bench.build_workload(24000, 6.2145, 4, np.random.default_rng(bench.SEED))
    n = len(w["q"])
    base = torch.tensor(w["pos"], dtype=torch.float32, device="cuda")
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    bufs = [base + torch.randn(base.shape, generator=g, device="cuda") * 0.002 for _ in range(6)]
    eng = bench.Engine(snb, w, 4, 54, 0, "single", 0, 0, 1, 0.1, 12)
    out = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    got = {}
    for step in range(40):                 # several rebuilds (every 12 steps), eager (timed) and replayed steps, evictions
        k = (step * 5 + step // 7) % 6
        eng.set_positions_device(bufs[k].data_ptr(), False); eng.execute(False); eng.forces_to(out.data_ptr(), False); eng.sync()
        f = out.double().cpu().numpy()
        assert np.isfinite(f).all()
        if k in got:
            err = np.linalg.norm(f - got[k], axis=1) / np.maximum(np.linalg.norm(got[k], axis=1), 1.0)
            assert err.max() < 5e-4, (step, k, err.max())      # same coordinates; the sorted order, the list age and the float summation order differ
        else:
            got[k] = f
    eng.close()
    ref = bench.Engine(snb, w, 4, 54, 0, "single", 0, 0, 1, 0.1, 1 << 30)
    for k in (0, 3, 5):
        ref.set_positions_device(bufs[k].data_ptr(), False); ref.execute(False); ref.forces_to(out.data_ptr(), False); ref.sync()
        f = out.double().cpu().numpy()
        err = np.linalg.norm(f - got[k], axis=1) / np.maximum(np.linalg.norm(f, axis=1), 1.0)
        assert err.max() < 5e-4, (k, err.max())
    ref.close()


def test_force_output_inside_the_step_graph(snb):
    """snb_set_force_output: the user-order force write becomes the last kernel of the (graph-replayed) step; the buffer must hold
    what snb_get_forces would have delivered, on eager and on replayed steps alike."""
    import torch
    import bench
    w = bench.build_workload(24000, 6.2145, 4, np.random.default_rng(bench.SEED))
    n = len(w["q"])
    pos = torch.tensor(w["pos"], dtype=torch.float32, device="cuda")
    ref = bench.Engine(snb, w, 4, 54, 0, "single", 0, 0, 1, 0.1, 10)
    eng = bench.Engine(snb, w, 4, 54, 0, "single", 0, 0, 1, 0.1, 10)
    out = torch.full((n, 3), float("nan"), dtype=torch.float32, device="cuda")
    eng.set_force_output(out.data_ptr(), False)
    fr = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    for step in range(5):
        ref.set_positions_device(pos.data_ptr(), False); ref.execute(False); ref.forces_to(fr.data_ptr(), False); ref.sync()
        eng.set_positions_device(pos.data_ptr(), False); eng.execute(False); eng.forces_to(out.data_ptr(), False); eng.sync()
        a, b = fr.double().cpu().numpy(), out.double().cpu().numpy()
        assert np.isfinite(b).all()
        err = np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(a, axis=1), 1.0)
        assert err.max() < 2e-4, (step, err.max())      # same kernels; only the atomic summation order differs
        pos = pos + torch.randn(pos.shape, generator=g, device="cuda") * 0.003
    ref.close(); eng.close()


@pytest.mark.parametrize("method", [0, 1, 2, 4, 5])
def test_instantiate_from_nonbonded_force(method, snb, F, prec):
    K.testInstantiateFromNonbondedForce(make_ev(snb, prec), F, method, pme=(1.0, 20, 20, 20) if method >= 4 else None, tol=K.TOL)      # TestSlicedNonbondedForce.h:76-84


@pytest.mark.parametrize("method", [2, 4, 5])
@pytest.mark.parametrize("exceptions", [False, True])
def test_scaling_parameter_separation(method, exceptions, snb, F, prec):
    n = 28 if exceptions else 40
    K.testScalingParameterSeparation(make_ev(snb, prec), F, method, exceptions, pme=(1.0, n, n, n) if method >= 4 else None,
                                     ljpme=(1.0, n, n, n) if method == 5 else None, tol=1e-3 if prec != "double" else 1e-4)


@pytest.mark.parametrize("method", [0, 1, 2])
def test_large_system_vs_oracle(method, snb, F, oev, prec):
    """testLargeSystem (TestSlicedNonbondedForce.h:494-555): 600 bonded dimers at random positions in a 20 nm box, cutoff 2 nm; the
    reference compares its platform with OpenMM's NonbondedForce, here the comparison partner is the oracle."""
    force, pos, box = K.largeSystem(F, method)
    _compare(make_ev(snb, prec), oev, force, pos, box if method == 2 else None, TOLS[prec])


def test_changing_parameters(snb, F, oev, prec):
    """testChangingParameters (TestSlicedNonbondedForce.h:683-758): 600 dimers on the reference's staggered lattice, PME with a 2 nm
    cutoff in a 20 nm box, direct and reciprocal space in separate force groups; then every fifth particle gets 1.5 q, 1.1 sigma,
    1.7 epsilon through updateParametersInContext.  The reference compares with OpenMM's NonbondedForce (tol 2e-3); the partner here
    is the oracle, before and after the update, group by group."""
    force, pos, box = K.changingParametersSystem(F)
    force.setPMEParameters(1.5, 48, 48, 48)
    force.setReciprocalSpaceForceGroup(3); force.setForceGroup(1)
    system = snb.System()
    for _ in range(force.getNumParticles()):
        system.addParticle(1.0)
    system.setDefaultPeriodicBoxVectors(*box)
    system.addForce(force)
    ctx = snb.Context(system, precision=prec)
    ctx.setPositions(pos)
    tol = 2e-3 if prec != "double" else 1e-5

    def check():
        for groups, direct, recip in ((1 << 1, True, False), (1 << 3, False, True), (-1, True, True)):
            st = ctx.getState(getEnergy=True, getForces=True, groups=groups)
            o = oev(force, pos, box, None, direct, recip)
            K.assertEqualTo(o["energy"], st.getPotentialEnergy(), tol)
            fo, fr = o["forces"], st.getForces()
            err = np.linalg.norm(fo - fr, axis=1) / np.maximum(np.linalg.norm(fo, axis=1), 1.0)
            assert err.max() <= tol, (groups, err.max())

    check()
    K.changeEveryFifthParticle(force)
    force.updateParametersInContext(ctx)
    check()


def test_box_change_between_rebuilds(snb, F, oev, prec):
    """A barostat-like sequence: the periodic box (and the coordinates with it) is rescaled between evaluations.  Every rebuild must
    pick the new cell up -- the replayed graph of the rebuild's sort phase is keyed on it -- and the answers must stay the oracle's."""
    n, L = 13824, 6.0
    force, pos, box = systems.random_box(F, n, 3, 4, L, 1.0, pme=(2.6283, 48, 48, 48))
    system = snb.System()
    for _ in range(n):
        system.addParticle(1.0)
    system.setDefaultPeriodicBoxVectors(*box)
    system.addForce(force)
    ctx = snb.Context(system, precision=prec, neighbor_padding=0.05, rebuild_interval=1)
    tol = TOLS[prec]
    pos = np.asarray(pos, dtype=float)
    for scale in (1.0, 1.01, 1.01, 0.995, 1.0):      # (the repeated value replays the captured sort phase)
        b = [[v * scale for v in row] for row in box]
        ctx.setPeriodicBoxVectors(*b)
        ctx.setPositions(pos * scale)
        st = ctx.getState(getEnergy=True, getForces=True)
        o = oev(force, pos * scale, b)
        K.assertEqualTo(o["energy"], st.getPotentialEnergy(), tol)
        fo, fr = o["forces"], st.getForces()
        err = np.linalg.norm(fo - fr, axis=1) / np.maximum(np.linalg.norm(fo, axis=1), 1.0)
        assert err.max() <= tol, (scale, err.max())


def test_huge_system_energy_follows_forces(snb, F, prec):
    """testHugeSystem (TestSlicedNonbondedForce.h:557-612) at its full size: 150^3 = 3 375 000 particles, CutoffPeriodic with a
    switching function; stepping along the force direction must change the energy by |F| * delta."""
    import torch
    state = {}

    def ctx_for(force, positions, box):
        if "ctx" not in state:
            system = snb.System()
            system.addParticles(force.getNumParticles()) if hasattr(system, "addParticles") else [system.addParticle(1.0) for _ in range(force.getNumParticles())]
            system.setDefaultPeriodicBoxVectors(*box)
            system.addForce(force)
            state["ctx"] = snb.Context(system, precision=prec)
        state["ctx"].setPositions(positions)
        return state["ctx"]

    def energy(force, positions, box):
        return ctx_for(force, positions, box).getState(getEnergy=True).getPotentialEnergy()

    def forces(force, positions, box):
        return ctx_for(force, positions, box).getState(getForces=True).getForces()

    K.testHugeSystem(energy, forces, F, gridSize=150, tol=1e-4)
    torch.cuda.empty_cache()


TRICLINIC = np.array([[6.0, 0.0, 0.0], [1.5, 6.0, 0.0], [-1.2, 2.0, 6.0]])


@pytest.mark.parametrize("method", [2, 4, 5, 55], ids=["2", "4", "5", "5_both_meshes_54"])
def test_triclinic_cell_on_the_gpu_builder(method, snb, F, oev, prec):
    """A triclinic cell (OpenMM's reduced form) large enough for the GPU neighbour builder and the PME brick kernels: fractional sort
    columns, lattice-vector tile images, sheared candidate search.  Energies, forces and derivatives against the oracle; then two
    forces-only steps (packed kernel, graph replay) after small moves.  The last case is the regression asked for after round 3's
    unexplained fault: the 54^3 = (6 x 9)^3 mesh on the plane path for the Coulomb AND the dispersion kernel table
    (erfc form, Nyquist planes averaged) in a triclinic cell."""
    n, L = 13824, 6.0
    both54 = method == 55
    method = 5 if both54 else method
    pme = ((2.6283, 54, 54, 54) if (method == 4 or both54) else (2.6283, 48, 48, 48)) if method >= 4 else None      # (54: the plane path's kernel table in a triclinic cell; 48: the three-pass pipeline)
    ljpme = ((2.6283, 54, 54, 54) if both54 else (2.6283, 24, 24, 24)) if method == 5 else None
    force, pos, _ = systems.random_box(F, n, 3, method, L, 1.0, pme=pme, ljpme=ljpme)
    pos = (pos / L) @ TRICLINIC                           # the jittered lattice, sheared with the cell
    box = TRICLINIC.copy()
    r, o = _compare(make_ev(snb, prec), oev, force, pos, box, TOLS[prec])
    assert r["stats"].n_tiles > 0 and r["stats"].n_host_rebuilds == 0
    force2, _, _ = systems.random_box(F, n, 3, method, L, 1.0, pme=pme, ljpme=ljpme, derivatives=False)
    system = snb.System()
    for _ in range(n):
        system.addParticle(1.0)
    system.setDefaultPeriodicBoxVectors(*box)
    system.addForce(force2)
    ctx = snb.Context(system, precision=prec, neighbor_padding=0.1, rebuild_interval=10)
    rng = np.random.default_rng(3)
    for step in range(3):
        ctx.setPositions(pos)
        fr = ctx.getState(getForces=True).getForces()
        fo = oev(force2, pos, box)["forces"]
        err = np.linalg.norm(fo - fr, axis=1) / np.maximum(np.linalg.norm(fo, axis=1), 1.0)
        assert err.max() <= TOLS[prec], "step %d: max force error %g" % (step, err.max())
        pos = pos + rng.normal(0.0, 0.004, pos.shape)


def test_nonperiodic_cutoff_on_the_gpu_builder(snb, F, oev, prec):
    """CutoffNonPeriodic at a size where the GPU neighbour builder applies: it runs inside an enclosing cell with more than a list
    radius of empty margin, so no image is ever in reach; the answer must be the oracle's non-periodic one and no list may be
    built on the host."""
    n, L = 13824, 6.0
    force, pos, box = systems.random_box(F, n, 3, 1, L, 1.0)
    pos = pos + np.array([-7.3, 4.1, 12.0])               # away from the origin (kept moderate: single precision carries absolute coordinates)
    r, o = _compare(make_ev(snb, prec), oev, force, pos, None, TOLS[prec])
    assert r["stats"].n_tiles > 0 and r["stats"].n_host_rebuilds == 0


def test_lambda_changes_between_replayed_steps(snb, F, oev, prec):
    """The alchemical inner loop: context.setParameter on scaling parameters between forces-only steps.  The lambdas live in a device
    table the captured step graph reads, so a replayed step must follow the new values without a rebuild or a re-capture."""
    n, L = 13824, 6.0
    force, pos, box = systems.random_box(F, n, 3, 4, L, 1.0, pme=(2.6283, 48, 48, 48), derivatives=False)
    system = snb.System()
    for _ in range(n):
        system.addParticle(1.0)
    system.setDefaultPeriodicBoxVectors(*box)
    system.addForce(force)
    ctx = snb.Context(system, precision=prec, neighbor_padding=0.1, rebuild_interval=50)
    ctx.setPositions(pos)
    kern = ctx._kernelFor(force)
    for step, (le, lv, l12) in enumerate([(0.7, 0.9, 0.45), (0.2, 0.9, 0.45), (0.2, 0.35, 1.0), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)]):
        params = {"lam_elec_01": le, "lam_vdw_01": lv, "lam_12": l12}
        for k, v in params.items():
            ctx.setParameter(k, v)
        fr = ctx.getState(getForces=True).getForces()
        fo = oev(force, pos, box, dict(ctx.getParameters()))["forces"]
        err = np.linalg.norm(fo - fr, axis=1) / np.maximum(np.linalg.norm(fo, axis=1), 1.0)
        assert err.max() <= TOLS[prec], "step %d: max force error %g" % (step, err.max())
    assert kern.getStats().n_rebuilds == 1


def test_parameter_update_without_rebuild(snb, F, oev, prec):
    """updateParametersInContext with new charges / sigmas / epsilons / exception parameters but the same subsets and exception
    pairs (the alchemical use of the reference's copyParametersToContext, CommonNonbondedSlicingKernels.cpp:1404-1568) must give the
    oracle's answer for the new parameters WITHOUT re-sorting atoms or rebuilding tiles."""
    n, L = 13824, 6.0
    force, pos, box = systems.random_box(F, n, 3, 4, L, 1.0, pme=(2.6283, 48, 48, 48))
    system = snb.System()
    for _ in range(n):
        system.addParticle(1.0)
    system.setDefaultPeriodicBoxVectors(*box)
    system.addForce(force)
    ctx = snb.Context(system, precision=prec, neighbor_padding=0.1, rebuild_interval=1000)
    ctx.setPositions(pos)
    ctx.getState(getEnergy=True, getForces=True)
    kern = ctx._kernelFor(force)
    rebuilds = kern.getStats().n_rebuilds
    rng = np.random.default_rng(11)
    for i in range(0, n, 3):
        q, sg, ep = force.getParticleParameters(i)
        force.setParticleParameters(i, q * 0.5 + 0.01, sg * 1.05, ep * 0.7)
    for k in range(0, force.getNumExceptions(), 5):
        a, b, qq, sg, ep = force.getExceptionParameters(k)
        if qq != 0.0 or ep != 0.0:
            force.setExceptionParameters(k, a, b, qq * 0.3, sg, ep * 1.5)
    force.updateParametersInContext(ctx)
    st = ctx.getState(getEnergy=True, getForces=True, getParameterDerivatives=True)
    o = oev(force, pos, box)
    tol = TOLS[prec]
    K.assertEqualTo(o["energy"], st.getPotentialEnergy(), tol)
    fo, fr = o["forces"], st.getForces()
    err = np.linalg.norm(fo - fr, axis=1) / np.maximum(np.linalg.norm(fo, axis=1), 1.0)
    assert err.max() <= tol, err.max()
    for name, v in o["derivatives"].items():
        K.assertEqualTo(v, st.getEnergyParameterDerivatives()[name], tol)
    assert kern.getStats().n_rebuilds == rebuilds, "parameter values alone must not trigger a neighbour rebuild"


def test_automatic_rebuild_follows_displacements(snb):
    """rebuild_interval < 0: the position-gather pass watches displacements since the last rebuild and the engine rebuilds when an
    atom has moved 0.8 * skin/2 (the reference relies on OpenMM's padded neighbour list doing the same).  A random walk must trigger
    rebuilds by itself, keep matching a rebuild-every-step engine, and never overrun the list; a fixed interval that is too long for
    the same walk must be reported through snb_stats.n_list_overruns."""
    import torch
    import bench
    w = bench.build_workload(24000, 6.2145, 4, np.random.default_rng(bench.SEED))
    n = len(w["q"])
    pos = torch.tensor(w["pos"], dtype=torch.float32, device="cuda")
    auto = bench.Engine(snb, w, 4, 54, 0, "single", 0, 0, 1, 0.1, -200)
    fixed = bench.Engine(snb, w, 4, 54, 0, "single", 0, 0, 1, 0.1, 1000)
    ref = bench.Engine(snb, w, 4, 54, 0, "single", 0, 0, 1, 0.1, 1)
    fa = torch.zeros((n, 3), dtype=torch.float32, device="cuda"); fr = torch.zeros_like(fa); ff = torch.zeros_like(fa)
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    worst = 0.0
    for step in range(60):
        for eng, out in ((auto, fa), (fixed, ff), (ref, fr)):
            eng.set_positions_device(pos.data_ptr(), False); eng.execute(False); eng.forces_to(out.data_ptr(), False); eng.sync()
        a, b = fr.double().cpu().numpy(), fa.double().cpu().numpy()
        worst = max(worst, float((np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(a, axis=1), 1.0)).max()))
        pos = pos + torch.randn(pos.shape, generator=g, device="cuda") * 0.004          # ~0.03 nm rms after 60 steps, tails beyond skin/2
    sa, sf = auto.stats(), fixed.stats()
    assert worst < 2e-3, worst      # summation-order noise of two different tile lists; a missed pair shows up as O(0.1-1)
    assert 2 <= sa.n_rebuilds < 30 and sa.n_list_overruns == 0, (sa.n_rebuilds, sa.n_list_overruns)
    assert sf.n_rebuilds == 1 and sf.n_list_overruns >= 1, (sf.n_rebuilds, sf.n_list_overruns)
    for e in (auto, fixed, ref):
        e.close()


def test_padding_and_rebuild_interval(snb, F, oev):
    """Tiles built with a skin and reused across steps must give the same answer as a fresh list."""
    force, pos, box = systems.random_box(F, 4096, 2, 4, 3.5, 1.0, pme=(2.6283, 32, 32, 32))
    ev = make_ev(snb, "double", neighbor_padding=0.15, rebuild_interval=10)
    _compare(ev, oev, force, pos, box, 1e-5)


def test_fft_against_numpy(snb):
    import ctypes
    L = snb.capi.lib()
    rng = np.random.default_rng(3)
    dp = ctypes.POINTER(ctypes.c_double)
    # sizes of the reference's FFT tests (platforms/cuda/tests/TestCudaCuFFT3D.cpp:36-141) plus the bench grids
    for prec, tol in ((1, 1e-10), (0, 2e-4)):
        for (nx, ny, nz), batch in [((28, 25, 25), 1), ((25, 28, 25), 2), ((25, 25, 28), 3), ((21, 25, 27), 2), ((28, 25, 30), 1), ((80, 80, 80), 2), ((120, 120, 120), 1),
                                  ((22, 26, 33), 2), ((52, 44, 39), 1), ((143, 26, 22), 1)]:      # factors 11 and 13 (legal on the reference's GPU path: FFT3DFactory.h:45-47)
            a = rng.standard_normal((batch, nx, ny, nz))
            spec = np.zeros((batch, nx, ny, nz // 2 + 1, 2)); rt = np.zeros_like(a)
            st = L.snb_test_fft3d(prec, 0, batch, nx, ny, nz, a.ctypes.data_as(dp), spec.ctypes.data_as(dp), rt.ctypes.data_as(dp))
            assert st == 0
            ref = np.fft.rfftn(a, axes=(1, 2, 3))
            got = spec[..., 0] + 1j * spec[..., 1]
            scale = np.abs(ref).max()
            assert np.abs(got - ref).max() / scale < tol, (prec, nx, ny, nz)
            assert np.abs(rt / (nx * ny * nz) - a).max() < tol * 10, (prec, nx, ny, nz)


def test_bench_workload_24k_vs_oracle(snb):
    """The bench generator's water + solute-blob box at 24k atoms (all 27 periodic image codes occur; molecules straddle the
    box faces, exceptions are non-periodic) against the oracle, with the neighbour skin the bench uses."""
    import ctypes
    import torch
    import bench
    w = bench.build_workload(24000, 6.2145, 4, np.random.default_rng(bench.SEED))
    fo, so, _, _ = bench.oracle_eval(w, 4, 54, 0)
    n = len(w["q"])
    for prec, tol in (("double", 1e-5), ("single", 1e-3)):
        eng = bench.Engine(snb, w, 4, 54, 0, prec, 0, 0, 1, 0.1, 1 << 30)
        dt = torch.float64 if prec == "double" else torch.float32
        pos = torch.tensor(w["pos"], dtype=dt, device="cuda"); forces = torch.zeros((n, 3), dtype=dt, device="cuda")
        eng.set_positions_device(pos.data_ptr(), prec == "double")
        eng.execute(True); eng.forces_to(forces.data_ptr(), prec == "double"); eng.sync()
        f = forces.double().cpu().numpy(); se = eng.slice_energies(so.shape[0])
        ferr = np.max(np.linalg.norm(f - fo, axis=1) / np.maximum(np.linalg.norm(fo, axis=1), 1.0))
        eerr = np.max(np.abs(se - so) / np.maximum(np.abs(so), 1.0))
        assert ferr < tol and eerr < tol, (prec, ferr, eerr)



def test_uneven_shard_block_ranges_sum_to_unsharded(snb, oev):
    """snb_set_shard_blocks (include/snb.h): three engines with uneven i-block ranges, one of them empty, one re-ranged after its
    first evaluation, still add up to the full result (the load-balanced decomposition bench.py --gpus N uses)."""
    import torch
    import bench
    w = bench.build_workload(12000, 4.932, 4, np.random.default_rng(bench.SEED))
    fo, so, _, _ = bench.oracle_eval(w, 4, 42, 0)
    n = len(w["q"])
    pos = torch.tensor(w["pos"], dtype=torch.float64, device="cuda")
    ftot = np.zeros((n, 3)); etot = np.zeros_like(so)
    ranges = [(0, 0), (0, 5), (5, 16)]
    for rank in range(3):
        eng = bench.Engine(snb, w, 4, 42, 0, "double", 0, rank, 3, 0.05, 1 << 30)
        forces = torch.zeros((n, 3), dtype=torch.float64, device="cuda")
        eng.set_positions_device(pos.data_ptr(), True)
        if rank == 2:
            eng.execute(True); eng.sync()          # first with the default ownership, then re-ranged: the lists must be rebuilt
        eng.set_shard_blocks(ranges[rank][0], ranges[rank][1], 16)
        eng.execute(True); eng.forces_to(forces.data_ptr(), True); eng.sync()
        ftot += forces.cpu().numpy(); etot += eng.slice_energies(so.shape[0])
        if rank == 0:
            assert eng.stats().n_tiles == 0
        eng.close()
    ferr = np.max(np.linalg.norm(ftot - fo, axis=1) / np.maximum(np.linalg.norm(fo, axis=1), 1.0))
    eerr = np.max(np.abs(etot - so) / np.maximum(np.abs(so), 1.0))
    assert ferr < 1e-5 and eerr < 1e-5, (ferr, eerr)
    eng = bench.Engine(snb, w, 4, 42, 0, "single", 0, 0, 2, 0.05, 1 << 30)
    with pytest.raises(RuntimeError):
        eng.set_shard_blocks(3, 2, 4)
    eng.close()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_engines_sum_to_unsharded(world, snb, oev):
    """The multi-GPU decomposition rehearsed on ONE GPU: `world` engines with shard_rank 0..world-1 evaluate the
    same 8-subset system; their partial forces and raw slice energies must add up to the oracle's full result."""
    import torch
    import bench
    w = bench.build_workload(12000, 4.932, 8, np.random.default_rng(bench.SEED))
    fo, so, _, _ = bench.oracle_eval(w, 4, 42, 0)
    n = len(w["q"])
    for prec, tol in (("double", 1e-5), ("single", 1e-3)):
        dt = torch.float64 if prec == "double" else torch.float32
        pos = torch.tensor(w["pos"], dtype=dt, device="cuda")
        ftot = np.zeros((n, 3)); etot = np.zeros_like(so)
        for rank in range(world):
            eng = bench.Engine(snb, w, 4, 42, 0, prec, 0, rank, world, 0.05, 1 << 30)
            forces = torch.zeros((n, 3), dtype=dt, device="cuda")
            eng.set_positions_device(pos.data_ptr(), prec == "double")
            eng.execute(True); eng.forces_to(forces.data_ptr(), prec == "double"); eng.sync()
            ftot += forces.double().cpu().numpy(); etot += eng.slice_energies(so.shape[0])
            eng.close()
        ferr = np.max(np.linalg.norm(ftot - fo, axis=1) / np.maximum(np.linalg.norm(fo, axis=1), 1.0))
        eerr = np.max(np.abs(etot - so) / np.maximum(np.abs(so), 1.0))
        # (sharded energies come from real-space interpolation of the potentials, not the k-space Gram sum: same bar)
        print("sharded x%d %s: force err %.2e, slice-energy err %.2e" % (world, prec, ferr, eerr))
        assert ferr < tol and eerr < tol, (prec, world, ferr, eerr)


def test_parameter_offsets_follow_global_parameters_on_the_device(snb, F, oev, prec):
    """Parameter offsets (SlicedNonbondedForce::addParticleParameterOffset / addExceptionParameterOffset) are applied on the device
    (snb_set_parameter_offsets + snb_set_global_parameters; the reference: platforms/common/src/kernels/nonbondedParameters.cc:4-179):
    Context::setParameter between evaluations -- energy steps and replayed forces-only steps -- must give the oracle's answer for the
    new effective charges / sigmas / epsilons / 1-4 parameters WITHOUT a neighbour rebuild.  An exception whose base parameters are all
    zero but which carries an offset is a 1-4 interaction (Q6)."""
    n, L = 13824, 6.0
    force, pos, box = systems.random_box(F, n, 3, 4, L, 1.0, pme=(2.6283, 48, 48, 48))
    force.addGlobalParameter("dq", 0.0); force.addGlobalParameter("dlj", 0.0)
    rng = np.random.default_rng(5)
    for i in rng.choice(n, 900, replace=False):
        force.addParticleParameterOffset("dq", int(i), float(rng.uniform(-0.4, 0.4)), 0.0, 0.0)
    for i in rng.choice(n, 700, replace=False):
        force.addParticleParameterOffset("dlj", int(i), 0.0, float(rng.uniform(-0.03, 0.03)), float(rng.uniform(0.0, 0.5)))
    zero_base = None
    for k in range(0, force.getNumExceptions(), 7):
        a, b, qq, sg, ep = force.getExceptionParameters(k)
        if qq == 0.0 and ep == 0.0 and zero_base is None:
            zero_base = k
            force.addExceptionParameterOffset("dq", k, 0.05, 0.3, 0.2)          # excluded pair that becomes a 1-4 when dq != 0
        elif qq != 0.0:
            force.addExceptionParameterOffset("dlj", k, 0.3 * qq, 0.01, 0.1)
    assert zero_base is not None
    system = snb.System()
    for _ in range(n):
        system.addParticle(1.0)
    system.setDefaultPeriodicBoxVectors(*box)
    system.addForce(force)
    ctx = snb.Context(system, precision=prec, neighbor_padding=0.1, rebuild_interval=1000)
    ctx.setPositions(pos)
    kern = ctx._kernelFor(force)
    tol = TOLS[prec]
    for step, (dq, dlj) in enumerate([(0.0, 0.0), (1.0, 0.0), (1.0, 1.0), (-0.5, 0.4), (0.25, 0.4)]):
        ctx.setParameter("dq", dq); ctx.setParameter("dlj", dlj)
        o = oev(force, pos, box, dict(ctx.getParameters()))
        if step % 2 == 0:
            st = ctx.getState(getEnergy=True, getForces=True, getParameterDerivatives=True)
            K.assertEqualTo(o["energy"], st.getPotentialEnergy(), tol)
            for name, v in o["derivatives"].items():
                K.assertEqualTo(v, st.getEnergyParameterDerivatives()[name], tol)
        else:
            st = ctx.getState(getForces=True)
        fo, fr = o["forces"], st.getForces()
        err = np.linalg.norm(fo - fr, axis=1) / np.maximum(np.linalg.norm(fo, axis=1), 1.0)
        if prec == "mixed":
            # 64-bit fixed point at 2^32 per kJ/mol/nm holds |component| < 2^31 = 2.1e9, in the reference as here; the excluded pair that
            # this test turns into a 1-4 sits at bonded distance with sigma = 0.3 nm and pushes its two atoms with 1.6e12
            err[np.abs(fo).max(axis=1) > 2.0e9] = 0.0
        assert err.max() <= tol, "step %d: max force error %g" % (step, err.max())
    assert kern.getStats().n_rebuilds == 1, "global-parameter changes must not rebuild the neighbour structure"


def test_changing_subsets_through_update_parameters(snb, F, oev, prec):
    """updateParametersInContext after setParticleSubset: atoms move between subsets, so every per-slice quantity changes -- block
    layout, tile slices, the slice of every 1-4 exception.  (The reference's GPU path attributes every 1-4 exception to slice 0 after
    an update, CommonNonbondedSlicingKernels.cpp:1511 vs :741-743 -- SURVEY Appendix D2, invisible in its own tests, which update
    with n = 1; here the slices are recomputed from the current subsets and the partner is the oracle on the updated force.)"""
    n, L = 13824, 6.0
    force, pos, box = systems.random_box(F, n, 3, 4, L, 1.0, pme=(2.6283, 48, 48, 48))
    system = snb.System()
    for _ in range(n):
        system.addParticle(1.0)
    system.setDefaultPeriodicBoxVectors(*box)
    system.addForce(force)
    ctx = snb.Context(system, precision=prec, neighbor_padding=0.1, rebuild_interval=1000)
    ctx.setPositions(pos)
    tol = TOLS[prec]

    def check():
        st = ctx.getState(getEnergy=True, getForces=True, getParameterDerivatives=True)
        o = oev(force, pos, box, dict(ctx.getParameters()))
        K.assertEqualTo(o["energy"], st.getPotentialEnergy(), tol)
        kern = ctx._kernelFor(force)
        for s in range(o["slice_energies"].shape[0]):
            for t in range(2):
                K.assertEqualTo(o["slice_energies"][s, t], kern.lastSliceEnergies[s, t], tol)
        fo, fr = o["forces"], st.getForces()
        err = np.linalg.norm(fo - fr, axis=1) / np.maximum(np.linalg.norm(fo, axis=1), 1.0)
        assert err.max() <= tol, err.max()

    check()
    rng = np.random.default_rng(17)
    d = np.asarray(pos) - 0.5 * L
    inner = np.linalg.norm(d, axis=1) < 1.6                      # a sphere in the middle of the slab layout goes to subset 2 ...
    for i in np.where(inner)[0]:
        force.setParticleSubset(int(i), 2)
    for i in rng.choice(n, 300, replace=False):                 # ... and scattered atoms (whole exception chains cross slices) to subset 1
        force.setParticleSubset(int(i), 1)
    force.updateParametersInContext(ctx)
    check()


@pytest.mark.parametrize("mesh", [48, 54], ids=["three_pass_48", "plane_path_54"])
def test_derivative_only_steps_evaluate_only_the_bound_slices(mesh, snb, F, oev, prec):
    """A force with energy-parameter derivatives accumulates dE/dlambda on every evaluation, energy requested or not (Q4,
    ReferenceNonbondedSlicingKernels.cpp:259-265).  Such derivative-only steps run with include_energy == 2: only the slices bound to a
    derivative-requested parameter are evaluated (snb_set_energy_slices), the pair kernel runs forces-only arithmetic on every other
    tile.  Forces and derivatives must still be the oracle's, step after step (graph replay included).  Both reciprocal pipelines: on the 54^3
    mesh the wanted slices' energies come from the plane kernel's Parseval sums, on the 48^3 mesh from k_convolveX's Gram sums."""
    n, L = 13824, 6.0
    force, pos, box = systems.random_box(F, n, 4, 4, L, 1.0, pme=(2.6283, mesh, mesh, mesh))
    system = snb.System()
    for _ in range(n):
        system.addParticle(1.0)
    system.setDefaultPeriodicBoxVectors(*box)
    system.addForce(force)
    ctx = snb.Context(system, precision=prec, neighbor_padding=0.1, rebuild_interval=10)
    rng = np.random.default_rng(3)
    tol = TOLS[prec]
    for step in range(4):
        ctx.setPositions(pos)
        st = ctx.getState(getForces=True, getParameterDerivatives=True)          # no energy: derivative-only step
        o = oev(force, pos, box, dict(ctx.getParameters()))
        err = np.linalg.norm(o["forces"] - st.getForces(), axis=1) / np.maximum(np.linalg.norm(o["forces"], axis=1), 1.0)
        assert err.max() <= tol, "step %d: max force error %g" % (step, err.max())
        assert o["derivatives"], "the test system must request derivatives"
        for name, v in o["derivatives"].items():
            K.assertEqualTo(v, st.getEnergyParameterDerivatives()[name], tol)
        pos = pos + rng.normal(0.0, 0.004, pos.shape)
    # and a full energy evaluation afterwards still produces every slice
    ctx.setPositions(pos)
    st = ctx.getState(getEnergy=True)
    o = oev(force, pos, box, dict(ctx.getParameters()))
    K.assertEqualTo(o["energy"], st.getPotentialEnergy(), tol)


def test_mixed_precision_forces_are_reproducible_bit_for_bit(snb):
    """SNB_MIXED accumulates the direct-space forces in 64-bit fixed point, as the reference's GPU platforms do (pme.cc:381-389): integer
    sums do not depend on the order in which waves arrive.  Two engines on the same input -- each with its own neighbour-list build, whose
    tile order is decided by atomics -- must deliver IDENTICAL forces, step after step; plain single precision (float atomics) need not."""
    import torch
    import bench
    w = bench.build_workload(24000, 6.3, 4, np.random.default_rng(3))
    n = len(w["q"])
    rng = np.random.default_rng(11)
    walk = [torch.tensor(w["pos"] + rng.normal(0.0, 0.003 * k, w["pos"].shape), dtype=torch.float32, device="cuda") for k in range(4)]
    outs = []
    for trial in range(2):
        eng = bench.Engine(snb, w, 4, 54, 0, "mixed", 0, 0, 1, 0.1, 2)      # a rebuild every second step
        got = []
        for k, pos in enumerate(walk):
            f = torch.zeros((n, 3), dtype=torch.float64, device="cuda")
            eng.set_positions_device(pos.data_ptr(), False)
            eng.execute(k == 0); eng.forces_to(f.data_ptr(), True); eng.sync()
            got.append(f.cpu().numpy())
        eng.close()
        outs.append(got)
    for a, b in zip(*outs):
        assert np.abs(a).max() > 1.0
        assert np.array_equal(a, b), "max difference %g" % np.abs(a - b).max()


def test_coarse_mesh_keeps_the_fixed_point_spreader_in_range(snb, F, oev):
    """The single-precision spreader accumulates 32-bit fixed point with a headroom of max(16, 8 x atoms per mesh cell) times the largest
    per-atom value (misc.hip, k_fixScale): a mesh point collects the weights of every atom within its stencil, and on a COARSE mesh they add
    up to many atoms' worth.  36 atoms per cell, every charge +0.4, i.e. 14.4 units of charge per mesh point: a fixed headroom of 16 (31 bits at a
    scale of 2^30 / (16 max|q|): 12.8 units) wraps around on most points -- SNB_FIX_HEADROOM=16 restores that rule and fails this test; the
    energies and forces must match the oracle on the same mesh."""
    n, L = 36000, 6.8
    force, pos, box = systems.random_box(F, n, 2, 4, L, 1.0, pme=(2.6283, 10, 10, 10))
    for i in range(n):
        q, sg, ep = force.getParticleParameters(i)
        force.setParticleParameters(i, 0.4, sg, ep)
    for prec in ("single", "mixed"):
        r, o = _compare(make_ev(snb, prec), oev, force, pos, box, TOLS[prec])
        assert r["stats"].n_host_rebuilds == 0


def test_exception_becoming_nonzero_between_replayed_steps(snb):
    """snb_set_exceptions with the SAME pairs but a different set of non-zero exceptions changes the 1-4 list's length and
    buffers without a neighbour rebuild; captured step graphs baked the old count and pointers in.  Flip a third of the zero exceptions
    to non-zero (and some 1-4s to zero) between graph-replayed steps and compare with the oracle for the new definition."""
    import ctypes
    import torch
    import bench
    w = bench.build_workload(24000, 6.2145, 4, np.random.default_rng(bench.SEED))
    n = len(w["q"])
    eng = bench.Engine(snb, w, 4, 54, 0, "double", 0, 0, 1, 0.1, 1 << 30)
    pos = torch.tensor(w["pos"], dtype=torch.float64, device="cuda"); forces = torch.zeros((n, 3), dtype=torch.float64, device="cuda")
    eng.set_positions_device(pos.data_ptr(), True)
    for _ in range(4):                     # rebuild step, capture, replays
        eng.execute(False)
    eng.sync()
    n14_before = eng.stats().n_14
    rebuilds = eng.stats().n_rebuilds
    w2 = dict(w)
    qq = w["exc_qq"].copy(); ee = w["exc_eps"].copy()
    zero = np.flatnonzero((qq == 0.0) & (ee == 0.0)); nonzero = np.flatnonzero((qq != 0.0) | (ee != 0.0))
    assert len(zero) > 10 and len(nonzero) > 10
    for k in zero[::3]:
        a, b = w["exc_pairs"][k]
        qq[k] = 0.5 * w["q"][a] * w["q"][b]      # (Coulomb only: these are bonded neighbours at 0.1 nm)
    qq[nonzero[::4]] = 0.0; ee[nonzero[::4]] = 0.0
    w2["exc_qq"] = qq; w2["exc_eps"] = ee
    dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)); ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    eng.ok(eng.L.snb_set_exceptions(eng.h, len(qq), ip(w["exc_pairs"]), dp(qq), dp(w["exc_sigma"]), dp(ee), None))
    fo, so, _, _ = bench.oracle_eval(w2, 4, 54, 0)
    for step in range(3):                  # replayed (re-captured) steps with the new 1-4 list
        eng.execute(False); eng.forces_to(forces.data_ptr(), True); eng.sync()
        f = forces.cpu().numpy()
        ferr = np.max(np.linalg.norm(f - fo, axis=1) / np.maximum(np.linalg.norm(fo, axis=1), 1.0))
        assert ferr < 1e-5, (step, ferr)
    st = eng.stats()
    assert st.n_14 != n14_before and st.n_rebuilds == rebuilds, "same pairs: the 1-4 membership changes without a neighbour rebuild"
    eng.execute(True); eng.sync()
    se = eng.slice_energies(so.shape[0])
    assert np.max(np.abs(se - so) / np.maximum(np.abs(so), 1.0)) < 1e-5
    eng.close()


@pytest.mark.parametrize("precision", ["single", "mixed", "double"])
@pytest.mark.parametrize("method,dgrid", [(4, 0), (5, 27)])
@pytest.mark.parametrize("accumulate", [0, 1])
def test_fused_force_output_and_get_forces_elsewhere(precision, method, dgrid, accumulate, snb):
    """The interpolation kernel of the step's last mesh delivers the user-order force into the buffer named by snb_set_force_output
    (the path bench.py times).  All precisions (the SNB_MIXED branch reads 64-bit fixed-point accumulators), PME and LJPME (two meshes,
    atoms with eps == 0 take the q == 0 hand-over on the dispersion mesh), accumulate on and off; and snb_get_forces into ANOTHER
    buffer after such a step must still return the complete force (the fused path used to leave fpx.. without the last
    mesh's part)."""
    import torch
    import bench
    w = bench.build_workload(24000, 6.2145, 4, np.random.default_rng(bench.SEED))
    n = len(w["q"])
    isd = precision == "double"
    dt = torch.float64 if isd else torch.float32
    pos = torch.tensor(w["pos"], dtype=dt, device="cuda")
    ref = bench.Engine(snb, w, method, 54, dgrid, precision, 0, 0, 1, 0.1, 1 << 30)
    eng = bench.Engine(snb, w, method, 54, dgrid, precision, 0, 0, 1, 0.1, 1 << 30)
    fr = torch.zeros((n, 3), dtype=dt, device="cuda")
    out = torch.full((n, 3), 1.5 if accumulate else float("nan"), dtype=dt, device="cuda")
    other = torch.zeros((n, 3), dtype=dt, device="cuda")
    eng.set_force_output(out.data_ptr(), isd, accumulate)
    tol = 1e-9 if isd else 2e-4            # same kernels; in float only the atomic summation order differs
    for step in range(3):                  # rebuild step (eager), captured step, replayed step
        ref.set_positions_device(pos.data_ptr(), isd); ref.execute(False); ref.forces_to(fr.data_ptr(), isd); ref.sync()
        if accumulate:
            out.fill_(1.5)
        eng.set_positions_device(pos.data_ptr(), isd); eng.execute(False); eng.sync()
        a = fr.double().cpu().numpy(); b = out.double().cpu().numpy() - (1.5 if accumulate else 0.0)
        assert np.isfinite(b).all()
        err = np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(a, axis=1), 1.0)
        assert err.max() < max(tol, 2e-6 if accumulate else 0), (step, err.max())
        eng.forces_to(other.data_ptr(), isd); eng.sync()          # a different target: rebuilt from the accumulators + fpx..
        c = other.double().cpu().numpy()
        err = np.linalg.norm(a - c, axis=1) / np.maximum(np.linalg.norm(a, axis=1), 1.0)
        assert err.max() < tol, ("get_forces elsewhere", step, err.max())
        host = np.zeros((n, 3), dtype=np.float64 if isd else np.float32)
        eng.ok(eng.L.snb_get_forces(eng.h, host.ctypes.data_as(__import__("ctypes").c_void_p), 0, int(isd), 0))
        err = np.linalg.norm(a - host.astype(float), axis=1) / np.maximum(np.linalg.norm(a, axis=1), 1.0)
        assert err.max() < tol, ("get_forces to the host", step, err.max())
    assert eng.stats().n_host_rebuilds == 0
    ref.close(); eng.close()


def test_derivative_only_step_rejects_an_energy_pointer(snb):
    """include/snb.h: include_energy == 2 produces the selected raw slice energies only; asking it for the total energy is an error
    (it used to return a lambda-weighted sum over unspecified slices)."""
    import ctypes
    import torch
    import bench
    w = bench.build_workload(6000, 3.915, 2, np.random.default_rng(bench.SEED))
    eng = bench.Engine(snb, w, 4, 36, 0, "single", 0, 0, 1, 0.1, 1 << 30)
    pos = torch.tensor(w["pos"], dtype=torch.float32, device="cuda")
    eng.set_positions_device(pos.data_ptr(), False)
    e = ctypes.c_double(0.0)
    assert eng.L.snb_execute(eng.h, 1, 2, 1, 1, ctypes.byref(e)) == eng.capi.SNB_ERR_INVALID_ARGUMENT
    assert eng.L.snb_execute(eng.h, 1, 2, 1, 1, None) == 0
    # and a parameter offset must not outlive the exception it points at
    m = len(w["exc_qq"])
    ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)); dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    tgt = np.array([m - 1], dtype=np.int32); glob = np.array([0], dtype=np.int32); delta = np.array([0.1, 0.0, 0.0])
    eng.ok(eng.L.snb_set_parameter_offsets(eng.h, 1, 0, None, None, None, 1, ip(tgt), ip(glob), dp(delta)))
    st = eng.L.snb_set_exceptions(eng.h, m - 1, ip(w["exc_pairs"]), dp(w["exc_qq"]), dp(w["exc_sigma"]), dp(w["exc_eps"]), None)
    assert st == eng.capi.SNB_ERR_INVALID_ARGUMENT, st
    eng.close()


_DRIFT_SCRIPT = r'''
import sys, json
import numpy as np, torch, importlib
sys.path[:0] = [ROOT, ROOT + "/tests", ROOT + "/oracle"]
import bench
snb = importlib.import_module("openmm-nonbonded-slicing_amd")
method, dgrid, prec = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
w = bench.build_workload(24000, 6.2145, 4, np.random.default_rng(bench.SEED))
n = len(w["q"]); isd = prec == "double"; dt = torch.float64 if isd else torch.float32
eng = bench.Engine(snb, w, method, 54, dgrid, prec, 0, 0, 1, 0.2, 1 << 30)      # skin 0.2 nm, never re-sorted after the first step
rng = np.random.default_rng(3)
out = {}
pos = w["pos"].copy()
forces = torch.zeros((n, 3), dtype=dt, device="cuda")
for step in range(3):
    if step:
        # a jitter that keeps the molecules' geometry sane plus a common translation: 0.09 nm of drift in two steps, inside skin / 2,
        # across mesh-cell and column borders for a third of the atoms
        pos = pos + rng.uniform(-0.008, 0.008, pos.shape) + np.array([0.04, 0.035, 0.03])
    pt = torch.tensor(pos, dtype=dt, device="cuda")
    eng.set_positions_device(pt.data_ptr(), isd); eng.execute(True); eng.forces_to(forces.data_ptr(), isd); eng.sync()
    w2 = dict(w); w2["pos"] = np.ascontiguousarray(pt.double().cpu().numpy())
    fo, so, _, _ = bench.oracle_eval(w2, method, 54, dgrid)
    f = forces.double().cpu().numpy(); se = eng.slice_energies(so.shape[0])
    st = eng.stats()
    out[step] = dict(ferr=float(np.max(np.linalg.norm(f - fo, axis=1) / np.maximum(np.linalg.norm(fo, axis=1), 1.0))),
                     eerr=float(np.max(np.abs(se - so) / np.maximum(np.abs(so), 1.0))), strays=int(st.n_spread_strays), rebuilds=int(st.n_rebuilds))
print("RESULT " + json.dumps(out))
'''


@pytest.mark.parametrize("env,expect_strays", [({}, False), ({"SNB_SPREAD_MARGIN": "0"}, True), ({"SNB_SPREAD_MARGIN": "0", "SNB_NO_FUSED_Z": "1"}, True),
                                               ({"SNB_NO_OWN_SPREAD": "1"}, False), ({"SNB_OWN_SLABS": "3"}, False)])
@pytest.mark.parametrize("method,dgrid,prec", [(4, 0, "single"), (5, 27, "double")])
def test_spreading_follows_drifting_atoms(env, expect_strays, method, dgrid, prec, snb):
    """The own-atoms spreader (pme.hip k_spreadOwn / k_spreadMerge) lets a work-group handle only the atoms sorted into its columns, in an
    LDS region with a drift margin; an atom that has left the region is a 'stray' and is added, exactly, by the merge kernel.  Atoms drift
    for two steps without a re-sort; with the margin forced to zero (the engine reads its switches once per process: a child process) every
    border crossing is a stray -- forces and slice energies must match the oracle either way, as they must with the scanning spreader
    (SNB_NO_OWN_SPREAD), another slab count and the unfused merge."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % root + _DRIFT_SCRIPT, str(method), str(dgrid), prec], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    tol = 1e-5 if prec == "double" else 1e-3
    for step, rec in res.items():
        assert rec["ferr"] < tol and rec["eerr"] < tol, (env, step, rec)
        assert rec["rebuilds"] == 1
    if expect_strays:
        assert res["2"]["strays"] > 0, res
    elif not env:
        assert res["2"]["strays"] == 0, res


_PLANE_SCRIPT = r'''
import sys, json
import numpy as np, torch, importlib
sys.path[:0] = [ROOT, ROOT + "/tests", ROOT + "/oracle"]
import bench
snb = importlib.import_module("openmm-nonbonded-slicing_amd")
method, grid, dgrid, nsub = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
w = bench.build_workload(24000, 6.2145, nsub, np.random.default_rng(bench.SEED))
n = len(w["q"])
eng = bench.Engine(snb, w, method, grid, dgrid, "single", 0, 0, 1, 0.1, 1 << 30)
eng.set_timing_interval(1)                       # every step eager with per-kernel stamps: the stamp slots tell which pipeline ran
pt = torch.tensor(w["pos"], dtype=torch.float32, device="cuda")
forces = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
eng.set_positions_device(pt.data_ptr(), False)
eng.execute(True); eng.forces_to(forces.data_ptr(), False); eng.sync()
fe = forces.double().cpu().numpy(); se = eng.slice_energies(nsub * (nsub + 1) // 2)
for _ in range(6):
    eng.execute(False)                           # forces-only steps
eng.forces_to(forces.data_ptr(), False); eng.sync()
ff = forces.double().cpu().numpy()
w2 = dict(w); w2["pos"] = np.ascontiguousarray(pt.double().cpu().numpy())
fo, so, _, _ = bench.oracle_eval(w2, method, grid, dgrid)
st = eng.stats()
rel = lambda f: float(np.max(np.linalg.norm(f - fo, axis=1) / np.maximum(np.linalg.norm(fo, axis=1), 1.0)))
print("RESULT " + json.dumps(dict(ferr_energy_step=rel(fe), ferr_forces_step=rel(ff), eerr=float(np.max(np.abs(se - so) / np.maximum(np.abs(so), 1.0))),
                                  timed=[int(x) for x in st.n_kernel_timed], host_rebuilds=int(st.n_host_rebuilds))))
'''


@pytest.mark.parametrize("method,grid,dgrid,nsub", [(4, 54, 0, 4), (4, 42, 0, 3), (4, 64, 0, 2), (4, 54, 0, 1), (4, 54, 0, 5), (4, 48, 0, 8), (5, 54, 54, 4)],
                         ids=["pme54_n4", "pme42_n3", "pme64_n2", "pme54_n1", "pme54_n5", "pme48x_n8", "ljpme54_54_n4"])
def test_plane_path_and_three_pass_pipeline_agree_with_the_oracle(method, grid, dgrid, nsub, snb):
    """Round 3: on square single-precision meshes whose (subset, kz) plane fits LDS the reciprocal pipeline runs k_planeXY (FFT_y, FFT_x,
    kernel value, inverse FFT_x / FFT_y of one plane in LDS, slice energies by Parseval over the plane) and k_fftZInvMix (lambda mix on the
    matrix cores + inverse z FFT) instead of the y pass, k_convolveX and the inverse y / z passes.  Both pipelines (SNB_NO_PLANE_FFT=1
    selects the old one; switches are read once per process: child processes) must meet the oracle on forces and slice energies, for an
    odd number of subsets (one line of a pair empty), for more than four (two groups of matrix-core output rows), for one subset, for the
    dispersion mesh of LJPME, and the stamp slots must show which pipeline ran (slots 3 / 5 are the y passes).  The 48^3 mesh (6 x 8 is not an
    instantiated split) runs the plane kernel with run-time splits (round 4)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for tag, env in (("plane", {}), ("three_pass", {"SNB_NO_PLANE_FFT": "1"})):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % root + _PLANE_SCRIPT, str(method), str(grid), str(dgrid), str(nsub)], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
        got[tag] = res
        assert res["host_rebuilds"] == 0
        assert res["ferr_energy_step"] < 1e-3 and res["ferr_forces_step"] < 1e-3 and res["eerr"] < 1e-3, (tag, res)
    if any(k in os.environ for k in ("SNB_NO_FUSED_Z", "SNB_NO_OWN_SPREAD", "SNB_OWN_SLABS", "SNB_FFT_TWOPASS", "SNB_NO_PLANE_FFT")):
        return      # (tools/switch_matrix.sh: these switches take the plane path's front end away -- parity above is all there is to check)
    plane_expected = grid in (42, 54, 64, 48)      # (round 4: 48 = 6 x 8 has no kernel of its own and runs the kernel with run-time splits)
    t = got["plane"]["timed"]
    assert t[4] > 0 and t[6] > 0, t
    assert (t[3] == 0 and t[5] == 0) == plane_expected, ("y-pass stamps", t)
    if method == 5 and plane_expected:
        assert (t[8 + 3] == 0 and t[8 + 5] == 0) == (dgrid in (42, 54, 64)), ("dispersion mesh", t)
    t = got["three_pass"]["timed"]
    assert t[3] > 0 and t[5] > 0, t


_OVERLAP_SCRIPT = r'''
import sys, json
import numpy as np, torch, importlib
sys.path[:0] = [ROOT, ROOT + "/tests", ROOT + "/oracle"]
import bench
snb = importlib.import_module("openmm-nonbonded-slicing_amd")
method, dgrid, prec = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
w = bench.build_workload(24000, 6.2145, 4, np.random.default_rng(bench.SEED))
n = len(w["q"]); isd = prec == "double"; dt = torch.float64 if isd else torch.float32
S = 10
eng = bench.Engine(snb, w, method, 54, dgrid, prec, 0, 0, 1, 0.2, 1 << 30)
eng.set_timing_interval(0)                      # no eager (serial) steps in between: every step after the first is a replayed graph
eng.set_energy_slices(np.ones(S, dtype=np.int32))
rng = np.random.default_rng(5)
pos = w["pos"].copy()
forces = torch.zeros((n, 3), dtype=dt, device="cuda")
out = {}
for step in range(6):
    if step:
        pos = pos + rng.uniform(-0.004, 0.004, pos.shape)
    pt = torch.tensor(pos, dtype=dt, device="cuda")
    eng.set_positions_device(pt.data_ptr(), isd)
    derivative = step % 2 == 1                  # alternate forces-only steps and derivative steps (include_energy = 2)
    if derivative:
        eng.execute(2, fetch=False)
    else:
        eng.execute(False)
    eng.forces_to(forces.data_ptr(), isd); eng.sync()
    w2 = dict(w); w2["pos"] = np.ascontiguousarray(pt.double().cpu().numpy())
    fo, so, _, _ = bench.oracle_eval(w2, method, 54, dgrid)
    f = forces.double().cpu().numpy()
    rec = dict(ferr=float(np.max(np.linalg.norm(f - fo, axis=1) / np.maximum(np.linalg.norm(fo, axis=1), 1.0))), derivative=derivative)
    if derivative:
        se = eng.slice_energies(S)
        rec["eerr"] = float(np.max(np.abs(se - so) / np.maximum(np.abs(so), 1.0)))
    out[step] = rec
# the same step twice: 64-bit fixed-point force sums do not depend on which launch took which work item
pt = torch.tensor(pos, dtype=dt, device="cuda"); eng.set_positions_device(pt.data_ptr(), isd)
eng.execute(False); eng.forces_to(forces.data_ptr(), isd); eng.sync(); a = forces.clone()
eng.execute(False); eng.forces_to(forces.data_ptr(), isd); eng.sync()
out["bitwise_equal"] = bool(torch.equal(a, forces))
print("RESULT " + json.dumps(out))
'''


@pytest.mark.parametrize("method,dgrid,prec", [(4, 0, "single"), (4, 0, "mixed"), (5, 27, "double")])
def test_overlapped_steps_match_the_oracle(method, dgrid, prec, snb):
    """SNB_OVERLAP=1 (engine.hip, overlapMode): graph steps run the reciprocal pipeline beside a CU-limited resident launch of the tile
    kernel and finish the pair work with a second launch; both claim their work items from device counters.  Forces-only and derivative
    steps, replayed, against the oracle; in mixed precision the force of a step must also be bit-for-bit reproducible (integer sums do
    not depend on which launch took which item).  The reference's counterpart is its separate PME queue
    (platforms/common/src/CommonNonbondedSlicingKernels.cpp:520-530, 1176-1179, 1377-1380).  A child process: the switch is read once."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ); e["SNB_OVERLAP"] = "1"; e["SNB_OVERLAP_MIN_TILES"] = "0"; e.pop("SNB_NO_STEP_GRAPH", None)      # (the default since round 4; stated here so that a changed default keeps the test meaningful)
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % root + _OVERLAP_SCRIPT, str(method), str(dgrid), prec], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    tol = 1e-5 if prec == "double" else 1e-3
    for step in range(6):
        rec = res[str(step)]
        assert rec["ferr"] < tol, (step, rec)
        if rec["derivative"]:
            assert rec["eerr"] < tol, (step, rec)
    if prec == "mixed":
        assert res["bitwise_equal"], res


_PREDICT_SCRIPT = r'''
import sys, json, hashlib
import numpy as np, torch, importlib
sys.path[:0] = [ROOT, ROOT + "/tests", ROOT + "/oracle"]
import bench
snb = importlib.import_module("openmm-nonbonded-slicing_amd")
w = bench.build_workload(24000, 6.2145, 4, np.random.default_rng(bench.SEED))
n = len(w["q"])
eng = bench.Engine(snb, w, 4, 54, 0, "mixed", 0, 0, 1, 0.1, 3)       # a rebuild every third step
rng = np.random.default_rng(11)
pos = w["pos"].copy()
forces = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
h = hashlib.sha1(); worst = 0.0
for step in range(8):
    pos = pos + rng.uniform(-0.01, 0.01, pos.shape)
    pt = torch.tensor(pos, dtype=torch.float32, device="cuda")
    eng.set_positions_device(pt.data_ptr(), False); eng.execute(False); eng.forces_to(forces.data_ptr(), False); eng.sync()
    h.update(forces.cpu().numpy().tobytes())
    if step in (0, 7):
        w2 = dict(w); w2["pos"] = np.ascontiguousarray(pt.double().cpu().numpy())
        fo, so, _, _ = bench.oracle_eval(w2, 4, 54, 0)
        f = forces.double().cpu().numpy()
        worst = max(worst, float(np.max(np.linalg.norm(f - fo, axis=1) / np.maximum(np.linalg.norm(fo, axis=1), 1.0))))
st = eng.stats()
print("RESULT " + json.dumps(dict(sha=h.hexdigest(), tiles=int(st.n_tiles), padded=int(st.n_padded_atoms), rebuilds=int(st.n_rebuilds), host_rebuilds=int(st.n_host_rebuilds), ferr=worst)))
'''


_SIDE_SCRIPT = r'''
import sys, json
import numpy as np, torch, importlib
sys.path[:0] = [ROOT, ROOT + "/tests", ROOT + "/oracle"]
import bench
snb = importlib.import_module("openmm-nonbonded-slicing_amd")
w = bench.build_workload(24000, 6.2145, 4, np.random.default_rng(bench.SEED))
n = len(w["q"])
eng = bench.Engine(snb, w, 4, 54, 0, "mixed", 0, 0, 1, 0.1, 6)       # a rebuild every sixth step
eng.set_timing_interval(0)
rng = np.random.default_rng(11)
pos = w["pos"].copy()
forces = torch.full((n, 3), 7.0, dtype=torch.float32, device="cuda")
pt = torch.tensor(pos, dtype=torch.float32, device="cuda")
eng.set_force_output(forces.data_ptr(), False); eng.set_positions_device(pt.data_ptr(), False)
worst = 0.0; worstE = 0.0
for step in range(20):
    pos = pos + rng.uniform(-0.004, 0.004, pos.shape)
    pt.copy_(torch.tensor(pos, dtype=torch.float32))      # (the same buffer every step: the side build must have copied the positions it started from)
    if step in (12, 19):
        eng.execute(True, fetch=False)
    else:
        eng.execute(False)
    eng.sync()
    if step in (11, 12, 13, 18, 19):      # 12, 18: the steps at which a rebuild falls due; 11: the last step on an old list while the next is being built
        w2 = dict(w); w2["pos"] = np.ascontiguousarray(pt.double().cpu().numpy())
        fo, so, _, _ = bench.oracle_eval(w2, 4, 54, 0)
        f = forces.double().cpu().numpy()
        worst = max(worst, float(np.max(np.linalg.norm(f - fo, axis=1) / np.maximum(np.linalg.norm(fo, axis=1), 1.0))))
        if step in (12, 19):
            se = eng.slice_energies(10)
            worstE = max(worstE, float(np.max(np.abs(se - so)) / np.max(np.abs(so))))
st = eng.stats()
print("RESULT " + json.dumps(dict(tiles=int(st.n_tiles), padded=int(st.n_padded_atoms), rebuilds=int(st.n_rebuilds), host_rebuilds=int(st.n_host_rebuilds), overruns=int(st.n_list_overruns), ferr=worst, eerr=worstE)))
eng.close()
'''


def test_rebuild_beside_the_steps(snb):
    """Round 4: with a fixed rebuild interval the list for the next interval is built BESIDE the steps -- positions copied
    aside three steps before the rebuild falls due, the whole GPU build on a stream of its own into a second set of list buffers, the sets
    exchanged when it falls due (engine.hip startSideBuild / finishSideBuild; the reference rebuilds in line, in
    `CommonCalcSlicedNonbondedForceKernel::execute` -> OpenMM `NonbondedUtilities::prepareInteractions`).  Child processes on the 24k workload,
    a rebuild every sixth step, positions rewritten in the SAME device buffer every step: in line (SNB_SIDE_REBUILD=0), beside (default), beside
    with a lead of one step, beside with every side build discarded (the in-line repeat), and beside under the overlapped step.  Forces against
    the oracle on the last step of an old list, on the steps where the sets change places and on the steps after; slice energies on two
    (one of them a step at which the sets change places)."""
    import json
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    cases = (("inline", {"SNB_SIDE_REBUILD": "0"}), ("beside", {}), ("lead1", {"SNB_SIDE_LEAD": "1"}), ("rejected", {"SNB_SIDE_REJECT": "1"}),
             ("overlapped", {"SNB_OVERLAP": "1", "SNB_OVERLAP_MIN_TILES": "0"}))
    for tag, env in cases:
        e = dict(os.environ); e.update(env); e["SNB_VERBOSE"] = "1"
        r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % root + _SIDE_SCRIPT], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got[tag] = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
        m = re.search(r"rebuilds: (\d+), of them (\d+) built beside the steps; (\d+) side builds discarded", r.stderr)
        assert m, r.stderr[-1500:]
        got[tag]["side"], got[tag]["discarded"] = int(m.group(2)), int(m.group(3))
        assert got[tag]["host_rebuilds"] == 0 and got[tag]["overruns"] == 0 and got[tag]["ferr"] < 1e-3 and got[tag]["eerr"] < 1e-3, (tag, got[tag])
        assert got[tag]["rebuilds"] == 4, (tag, got[tag])      # steps 0, 6, 12, 18
    assert got["inline"]["side"] == 0 and got["inline"]["discarded"] == 0, got
    for tag in ("beside", "lead1", "overlapped"):
        assert got[tag]["side"] == 2 and got[tag]["discarded"] == 0, (tag, got[tag])      # (the first two rebuilds fix the padded size; 12 and 18 are built beside)
    assert got["rejected"]["side"] == 0 and got["rejected"]["discarded"] == 2, got


_SIDE_AUTO_SCRIPT = r'''
import sys, json
import numpy as np, torch, importlib
sys.path[:0] = [ROOT, ROOT + "/tests", ROOT + "/oracle"]
import bench
snb = importlib.import_module("openmm-nonbonded-slicing_amd")
w = bench.build_workload(24000, 6.2145, 4, np.random.default_rng(bench.SEED))
n = len(w["q"])
auto = bench.Engine(snb, w, 4, 54, 0, "single", 0, 0, 1, 0.1, -200)
ref = bench.Engine(snb, w, 4, 54, 0, "single", 0, 0, 1, 0.1, 1)
auto.set_timing_interval(0)
pos = torch.tensor(w["pos"], dtype=torch.float32, device="cuda")
fa = torch.zeros((n, 3), dtype=torch.float32, device="cuda"); fr = torch.zeros_like(fa)
auto.set_force_output(fa.data_ptr(), False); ref.set_force_output(fr.data_ptr(), False)
auto.set_positions_device(pos.data_ptr(), False); ref.set_positions_device(pos.data_ptr(), False)
g = torch.Generator(device="cuda"); g.manual_seed(9)
worst = 0.0
for step in range(160):
    auto.execute(False); ref.execute(False); auto.sync(); ref.sync()
    a, b = fr.double().cpu().numpy(), fa.double().cpu().numpy()
    worst = max(worst, float((np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(a, axis=1), 1.0)).max()))
    pos.add_(torch.randn(pos.shape, generator=g, device="cuda") * 0.002)      # (in place: the engines keep reading the same buffer)
st = auto.stats()
print("RESULT " + json.dumps(dict(rebuilds=int(st.n_rebuilds), host_rebuilds=int(st.n_host_rebuilds), overruns=int(st.n_list_overruns), ferr=worst, padded=int(st.n_padded_atoms), atoms=n)))
auto.close(); ref.close()
'''


def test_displacement_triggered_rebuilds_beside_a_side_building_engine(snb):
    """rebuild_interval < 0 (what the plugin adapter of INTEGRATION.md uses) rebuilds in line, at the pace of the displacement watch: side builds
    are for fixed intervals (a guess of the watch's next interval was built and measured slower, engine.hip sideBuildPossible).  A random walk
    of 160 steps against an engine that rebuilds every step, with side builds switched off and on (no difference for this engine: the switch
    must not matter), no overrun, and a sane padded count -- a replayed graph memset node once left every atom a block of its own (32 N slots;
    docs/MEASUREMENT_LOG.md round 4 section 5: zero fills inside captured graphs are kernels since)."""
    import json
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for tag, env in (("off", {"SNB_SIDE_REBUILD": "0"}), ("on", {})):
        e = dict(os.environ); e.update(env); e["SNB_VERBOSE"] = "1"
        r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % root + _SIDE_AUTO_SCRIPT], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got[tag] = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
        m = re.findall(r"rebuilds: (\d+), of them (\d+) built beside the steps; (\d+) side builds discarded", r.stderr)
        sides = [int(x[1]) for x in m if int(x[0]) < 150]      # (the reference engine rebuilds 160 times, in line)
        assert len(sides) == 1 and sides[0] == 0, r.stderr[-1500:]
        assert got[tag]["host_rebuilds"] == 0 and got[tag]["overruns"] == 0 and got[tag]["ferr"] < 2e-3, (tag, got[tag])
        assert got[tag]["padded"] < 2 * got[tag]["atoms"], (tag, got[tag])
        assert 4 <= got[tag]["rebuilds"] <= 40, (tag, got[tag])
    assert got["on"]["rebuilds"] == got["off"]["rebuilds"], got


def test_predicted_padded_count_and_its_repeat_path(snb):
    """Round 4: from the second rebuild on the GPU neighbour build sizes the padded arrays from the previous rebuild's count plus a margin and
    does not wait for this rebuild's count (engine.hip gpuRebuild; the reference's counterpart is OpenMM's findBlocksWithInteractions, which
    keeps its buffers and re-runs on overflow).  Three child processes: waiting for the count as before (SNB_NB_SYNC_PADDED=1), predicting
    (default), and predicting two blocks too FEW (SNB_NB_PREDICT_SHORT=2: every predicted rebuild overflows and is repeated with the exact
    count).  All three must meet the oracle; in mixed precision (integer force sums) the forces of all eight steps must be bit-for-bit
    identical between the three, spare padding blocks or not."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for tag, env in (("sync", {"SNB_NB_SYNC_PADDED": "1"}), ("predicted", {}), ("short", {"SNB_NB_PREDICT_SHORT": "2", "SNB_VERBOSE": "1"})):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % root + _PREDICT_SCRIPT], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        if tag == "short":
            assert "repeating with the exact count" in r.stderr, r.stderr[-1500:]      # the overflow was seen and the rebuild repeated
        got[tag] = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
        assert got[tag]["host_rebuilds"] == 0 and got[tag]["ferr"] < 1e-3, (tag, got[tag])
    assert got["sync"]["sha"] == got["predicted"]["sha"] == got["short"]["sha"], got
    assert got["sync"]["tiles"] == got["predicted"]["tiles"] == got["short"]["tiles"], got
    assert got["predicted"]["padded"] >= got["sync"]["padded"], got
