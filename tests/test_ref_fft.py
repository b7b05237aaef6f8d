"""The FFT of the reciprocal path pinned by the reference's own code.

The reference's Reference-platform PME transforms its grids with pocketfft::c2c from the header it vendors
(platforms/reference/src/ReferencePME.cpp:788-805), and its GPU FFT tests use the same call as their oracle
(platforms/cuda/tests/TestCudaCuFFT3D.cpp:97-103).  That header is the one piece of the reference that compiles without OpenMM:
oracle/_ref/libref_fft.so is built from it where it lies (oracle/Makefile, oracle/ref_fft_wrapper.cpp), and
tests/golden/ref_fft3d.npz holds its outputs for seeded inputs (tests/golden/make_fft_golden.py) for machines without the reference tree.

CPU tests: the oracle's FFT (orc_fft3d, which the oracle's PME runs) against the golden vectors, and against the reference library
itself wherever that file exists.  GPU test: the engine's FFT (snb_test_fft3d) against the same vectors."""
import ctypes
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_fft_golden", os.path.join(HERE, "golden", "make_fft_golden.py"))
G = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(G)
GOLD = np.load(os.path.join(HERE, "golden", "ref_fft3d.npz"))


def _cases():
    for k in range(int(GOLD["n_shapes"])):
        yield k, tuple(int(x) for x in GOLD["shape_%d" % k]), int(GOLD["seed_%d" % k])


def _check_against_golden(k, shape, seed, spec, tol):
    idx, w1, w2 = G.sample_plan(seed, shape)
    spec = np.asarray(spec).reshape(-1)
    scale = np.abs(GOLD["samples_%d" % k]).max()
    assert np.abs(spec[idx] - GOLD["samples_%d" % k]).max() <= tol * scale
    sums = np.array([np.dot(w1, spec), np.dot(w2, spec)])
    assert np.abs(sums - GOLD["checksums_%d" % k]).max() <= tol * scale * np.sqrt(spec.size) * 4


def test_golden_fixture_matches_the_reference_fft_where_it_is_built(oracle):
    """Guards the committed fixture: wherever oracle/_ref/libref_fft.so exists, the vectors regenerate exactly."""
    if oracle.ref_fft_lib() is None:
        pytest.skip("oracle/_ref/libref_fft.so not present (built only where /root/reference exists)")
    for k, shape, seed in _cases():
        spec = oracle.ref_fft3d(G.real_input(seed, shape)[0].astype(np.complex128))
        _check_against_golden(k, shape, seed, spec, 1e-15)
    assert np.array_equal(oracle.ref_fft3d(GOLD["tiny_in"], True), GOLD["tiny_forward"])


def test_oracle_fft_matches_the_reference_fft_vectors(oracle):
    """orc_fft3d (sign -1 = the reference's forward=true, +1 = forward=false, both unnormalised) against the reference's outputs."""
    for k, shape, seed in _cases():
        spec = oracle.fft3d(G.real_input(seed, shape)[0].astype(np.complex128), -1)
        _check_against_golden(k, shape, seed, spec, 1e-13)
    assert np.abs(oracle.fft3d(GOLD["tiny_in"], -1) - GOLD["tiny_forward"]).max() < 1e-12
    assert np.abs(oracle.fft3d(GOLD["tiny_in"], +1) - GOLD["tiny_backward"]).max() < 1e-12


def test_oracle_fft_matches_the_reference_library_on_pme_meshes(oracle):
    """Directly against the reference's code, both directions, on the mesh shapes the parity tests and the bench use."""
    if oracle.ref_fft_lib() is None:
        pytest.skip("oracle/_ref/libref_fft.so not present")
    rng = np.random.default_rng(12)
    for shape in [(20, 20, 20), (28, 28, 28), (32, 32, 32), (45, 45, 45), (54, 54, 54), (80, 80, 80), (120, 120, 120)]:
        a = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
        for forward, sign in ((True, -1), (False, +1)):
            ref = oracle.ref_fft3d(a, forward)
            got = oracle.fft3d(a, sign)
            assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max() * np.log2(a.size), (shape, forward)


@pytest.mark.gpu
def test_engine_fft_matches_the_reference_fft_vectors(snb):
    """The engine's batched real-to-complex 3D FFT (hand-written, pme.hip) against the reference's outputs: the half spectrum it keeps
    equals the corresponding entries of the reference's full complex transform of the same real input."""
    L = snb.capi.lib()
    dp = ctypes.POINTER(ctypes.c_double)
    for prec, tol in ((1, 1e-11), (0, 2e-4)):
        for k, shape, seed in _cases():
            nx, ny, nz = shape
            a = G.real_input(seed, shape)
            spec = np.zeros((1, nx, ny, nz // 2 + 1, 2)); rt = np.zeros_like(a)
            assert L.snb_test_fft3d(prec, 0, 1, nx, ny, nz, a.ctypes.data_as(dp), spec.ctypes.data_as(dp), rt.ctypes.data_as(dp)) == 0
            half = spec[0, ..., 0] + 1j * spec[0, ..., 1]
            full = np.zeros(shape, dtype=np.complex128)      # rebuild the full spectrum from the half by Hermitian symmetry
            full[:, :, :nz // 2 + 1] = half
            kx = (-np.arange(nx)) % nx; ky = (-np.arange(ny)) % ny
            for z in range(nz // 2 + 1, nz):
                full[:, :, z] = np.conj(half[kx][:, ky][:, :, nz - z])
            _check_against_golden(k, shape, seed, full, tol)
            assert np.abs(rt[0] / (nx * ny * nz) - a[0]).max() < tol * 10
