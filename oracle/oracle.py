"""ctypes front-end of the CPU ORACLE (``oracle/snb_oracle.c``).

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg.  Nothing under ``openmm-nonbonded-slicing_amd/`` imports this module.

``evaluate(force, positions, box, parameters)`` restates
``ReferenceCalcSlicedNonbondedForceKernel::execute`` + ``computeParameters``
(platforms/reference/src/ReferenceNonbondedSlicingKernels.cpp:187-268, 339-391) for a duck-typed
``SlicedNonbondedForce``-like object: scaling parameters -> lambdas, parameter offsets -> effective
(q, sigma, epsilon), then the C arithmetic, then  E = sum lambda*E_slice  and the dE/dlambda map.
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OrcConfig(ctypes.Structure):
    _fields_ = [
        ("n_atoms", ctypes.c_int), ("n_subsets", ctypes.c_int), ("method", ctypes.c_int),
        ("cutoff", ctypes.c_double), ("use_switch", ctypes.c_int), ("switch_distance", ctypes.c_double),
        ("rf_dielectric", ctypes.c_double), ("alpha", ctypes.c_double), ("grid", ctypes.c_int * 3),
        ("kmax", ctypes.c_int * 3), ("alpha_d", ctypes.c_double), ("dgrid", ctypes.c_int * 3),
        ("exceptions_periodic", ctypes.c_int), ("use_dispersion_correction", ctypes.c_int),
        ("include_direct", ctypes.c_int), ("include_reciprocal", ctypes.c_int),
        ("background_term", ctypes.c_int), ("correct_q1", ctypes.c_int),
    ]


def build(force_rebuild: bool = False) -> str:
    path = os.path.join(_HERE, "libsnb_oracle.so")
    src = os.path.join(_HERE, "snb_oracle.c")
    if force_rebuild or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libsnb_oracle.so"])
    return path


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        dp = ctypes.POINTER(ctypes.c_double); ip = ctypes.POINTER(ctypes.c_int)
        L.orc_evaluate.argtypes = [ctypes.POINTER(OrcConfig), dp, dp, dp, dp, dp, ip, ctypes.c_int, ip, dp, dp, dp, dp, dp, dp, dp]
        L.orc_evaluate.restype = ctypes.c_int
        L.orc_dispersion_coefficients.argtypes = [ctypes.c_int, ctypes.c_int, dp, dp, ip, ctypes.c_double, ctypes.c_int, ctypes.c_double, dp]
        L.orc_bspline_moduli.argtypes = [ctypes.c_int, ctypes.c_int, dp]
        L.orc_fft3d.argtypes = [dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.orc_last_pair_count.restype = ctypes.c_longlong
        L.orc_cutoff_band_pairs.argtypes = [ctypes.POINTER(OrcConfig), dp, dp, dp, dp, dp, ip, ctypes.c_int, ip, dp, ctypes.c_double, ctypes.c_longlong, ip, dp]
        L.orc_cutoff_band_pairs.restype = ctypes.c_longlong
        _LIB = L
    return _LIB


def _dp(a): return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
def _ip(a): return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


def slice_index(i, j):
    return i * (i + 1) // 2 + j if i > j else j * (j + 1) // 2 + i


def resolve(force, parameters):
    """computeParameters (ReferenceNonbondedSlicingKernels.cpp:339-391) for a SlicedNonbondedForce-like object."""
    n = force.getNumParticles()
    ns = force.getNumSubsets()
    S = ns * (ns + 1) // 2
    q = np.zeros(n); sig = np.zeros(n); eps = np.zeros(n); sub = np.zeros(n, dtype=np.int32)
    for i in range(n):
        q[i], sig[i], eps[i] = force.getParticleParameters(i)
        sub[i] = force.getParticleSubset(i)
    for k in range(force.getNumParticleParameterOffsets()):
        name, idx, dq, ds, de = force.getParticleParameterOffset(k)
        v = parameters[name]
        q[idx] += v * dq; sig[idx] += v * ds; eps[idx] += v * de
    m = force.getNumExceptions()
    pairs = np.zeros((max(m, 1), 2), dtype=np.int32); qq = np.zeros(max(m, 1)); esig = np.zeros(max(m, 1)); eeps = np.zeros(max(m, 1))
    for k in range(m):
        p1, p2, a, b, c = force.getExceptionParameters(k)
        pairs[k] = (p1, p2); qq[k] = a; esig[k] = b; eeps[k] = c
    for k in range(force.getNumExceptionParameterOffsets()):
        name, idx, da, db, dc = force.getExceptionParameterOffset(k)
        v = parameters[name]
        qq[idx] += v * da; esig[idx] += v * db; eeps[idx] += v * dc
    lam = np.ones((S, 2))
    binding = {}
    derivs = set(force.getEnergyParameterDerivativeName(i) for i in range(force.getNumEnergyParameterDerivatives()))
    for k in range(force.getNumScalingParameters()):
        name, s1, s2, incC, incLJ = force.getScalingParameter(k)
        s = slice_index(s1, s2)
        if incC:
            lam[s, 0] = parameters[name]; binding[(s, 0)] = name
        if incLJ:
            lam[s, 1] = parameters[name]; binding[(s, 1)] = name
    return dict(n=n, ns=ns, S=S, q=q, sigma=sig, epsilon=eps, subset=sub, m=m, pairs=pairs, qq=qq, esig=esig, eeps=eeps,
                lam=lam, binding=binding, derivs=derivs)


def default_parameters(force):
    return {force.getGlobalParameterName(i): force.getGlobalParameterDefaultValue(i) for i in range(force.getNumGlobalParameters())}


def evaluate(force, positions, box=None, parameters=None, include_direct=True, include_reciprocal=True,
             pme=None, ljpme=None, kmax=None, background=True, correct_q1=True):
    """Returns dict(energy, forces[N,3], slice_energies[S,2], derivatives{name: value}, pairs).

    ``pme`` = (alpha, nx, ny, nz) overrides the force's own PME parameters (which must be explicit:
    auto-selection belongs to OpenMM's NonbondedForceImpl::calcPMEParameters, third-party, a13)."""
    L = lib()
    # a Context starts from the force's default parameter values; `parameters` overrides some of them (Context::setParameter)
    parameters = dict(default_parameters(force), **(parameters or {}))
    r = resolve(force, parameters)
    pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 3)
    assert pos.shape[0] == r["n"]
    if box is None:
        box = np.diag([1e6, 1e6, 1e6]).astype(np.float64)
    box = np.ascontiguousarray(box, dtype=np.float64).reshape(9)
    cfg = OrcConfig()
    cfg.n_atoms = r["n"]; cfg.n_subsets = r["ns"]; cfg.method = force.getNonbondedMethod()
    cfg.cutoff = force.getCutoffDistance()
    cfg.use_switch = int(force.getUseSwitchingFunction()); cfg.switch_distance = force.getSwitchingDistance()
    cfg.rf_dielectric = force.getReactionFieldDielectric()
    a, nx, ny, nz = pme if pme is not None else force.getPMEParameters()
    cfg.alpha = a; cfg.grid[0], cfg.grid[1], cfg.grid[2] = nx, ny, nz
    ad, dx, dy, dz = ljpme if ljpme is not None else force.getLJPMEParameters()
    cfg.alpha_d = ad; cfg.dgrid[0], cfg.dgrid[1], cfg.dgrid[2] = dx, dy, dz
    if kmax is not None:
        cfg.kmax[0], cfg.kmax[1], cfg.kmax[2] = kmax
    method = force.getNonbondedMethod()
    if method in (4, 5) and (nx <= 0 or a <= 0):
        raise ValueError("oracle: explicit PME parameters required")
    if method == 5 and (dx <= 0 or ad <= 0):
        raise ValueError("oracle: explicit LJPME parameters required")
    if method == 3:
        if kmax is None:
            raise ValueError("oracle: explicit Ewald kmax required")
        if a <= 0:
            raise ValueError("oracle: explicit Ewald alpha required")
    cfg.exceptions_periodic = int(force.getExceptionsUsePeriodicBoundaryConditions())
    cfg.use_dispersion_correction = int(force.getUseDispersionCorrection())
    cfg.include_direct = int(include_direct and force.getIncludeDirectSpace())
    cfg.include_reciprocal = int(include_reciprocal)
    cfg.background_term = int(background); cfg.correct_q1 = int(correct_q1)
    forces = np.zeros((r["n"], 3)); sliceE = np.zeros((r["S"], 2))
    lam = np.ascontiguousarray(r["lam"])
    coef = dispersion_coefficients(force) if force.getUseDispersionCorrection() else np.zeros(r["S"])
    # The parity tests evaluate the same definition and coordinates once per engine precision: identical inputs (every array and scalar
    # that reaches orc_evaluate) return the stored result instead of a second evaluation.
    h = hashlib.sha1()
    for a in (pos, box, r["q"], r["sigma"], r["epsilon"], r["subset"], r["pairs"][:r["m"]], r["qq"][:r["m"]], r["esig"][:r["m"]], r["eeps"][:r["m"]], lam, coef):
        h.update(np.ascontiguousarray(a).tobytes()); h.update(b"|")
    h.update(bytes(cfg))
    key = h.hexdigest()
    if key in _memo:
        forces, sliceE, npairs = _memo[key]
        forces = forces.copy(); sliceE = sliceE.copy()
        rc = 0
    else:
        rc = _call(L, cfg, pos, box, r, lam, coef, forces, sliceE)
        npairs = int(L.orc_last_pair_count())
        if rc == 0 and forces.nbytes <= (8 << 20):      # (the test modules run precision by precision: keep a whole pass, up to 400 MB, oldest out first)
            global _memo_bytes
            while _memo and _memo_bytes + forces.nbytes > (400 << 20):
                _memo_bytes -= _memo.pop(next(iter(_memo)))[0].nbytes
            _memo[key] = (forces.copy(), sliceE.copy(), npairs)
            _memo_bytes += forces.nbytes
    return _finish(rc, r, lam, forces, sliceE, npairs)


_memo = {}
_memo_bytes = 0


def _call(L, cfg, pos, box, r, lam, coef, forces, sliceE):
    return L.orc_evaluate(ctypes.byref(cfg), _dp(pos), _dp(box), _dp(r["q"]), _dp(r["sigma"]), _dp(r["epsilon"]), _ip(r["subset"]),
                        r["m"], _ip(r["pairs"]), _dp(r["qq"]), _dp(r["esig"]), _dp(r["eeps"]), _dp(lam), _dp(coef), _dp(forces), _dp(sliceE))


def _finish(rc, r, lam, forces, sliceE, npairs):
    if rc == -1:
        raise RuntimeError("The periodic box size has decreased to less than twice the nonbonded cutoff.")
    if rc != 0:
        raise RuntimeError("oracle error %d" % rc)
    energy = float((lam * sliceE).sum())
    derivs = {name: 0.0 for name in r["derivs"]}
    for (s, t), name in r["binding"].items():
        if name in derivs:
            derivs[name] += sliceE[s, t]
    return dict(energy=energy, forces=forces, slice_energies=sliceE, derivatives=derivs, lambdas=lam, pairs=npairs)


def dispersion_coefficients(force, parameters=None):
    # a Context starts from the force's default parameter values; `parameters` overrides some of them (Context::setParameter)
    parameters = dict(default_parameters(force), **(parameters or {}))
    r = resolve(force, parameters)
    out = np.zeros(r["S"])
    lib().orc_dispersion_coefficients(r["n"], r["ns"], _dp(r["sigma"]), _dp(r["epsilon"]), _ip(r["subset"]), force.getCutoffDistance(),
                                      int(force.getUseSwitchingFunction()), force.getSwitchingDistance(), _dp(out))
    return out


def bspline_moduli(n, order=5):
    out = np.zeros(n)
    lib().orc_bspline_moduli(n, order, _dp(out))
    return out


_ref_fft = None


def ref_fft_lib():
    """oracle/_ref/libref_fft.so: the reference's OWN 3D FFT (its vendored pocketfft header compiled where it lies under /root/reference,
    oracle/ref_fft_wrapper.cpp) -- or None where neither the prebuilt file nor the reference tree exists."""
    global _ref_fft
    if _ref_fft is None:
        path = os.path.join(_HERE, "_ref", "libref_fft.so")
        if not os.path.exists(path) and os.path.exists("/root/reference/openmmapi/include/internal/pocketfft_hdronly.h"):
            subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])
        if not os.path.exists(path):
            return None
        L = ctypes.CDLL(path)
        L.ref_c2c_3d.argtypes = [ctypes.POINTER(ctypes.c_double)] + [ctypes.c_int] * 5
        L.ref_c2c_3d.restype = ctypes.c_int
        _ref_fft = L
    return _ref_fft


def ref_fft3d(a, forward=True):
    """Complex 3D transform of a [batch][nx][ny][nz] (or [nx][ny][nz]) array by the reference's pocketfft::c2c, called as
    ReferencePME.cpp:788-805 calls it (unnormalised)."""
    L = ref_fft_lib()
    if L is None:
        raise RuntimeError("oracle/_ref/libref_fft.so is not available")
    a = np.ascontiguousarray(a, dtype=np.complex128).copy()
    b = a.reshape((-1,) + a.shape[-3:])
    assert L.ref_c2c_3d(b.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), b.shape[0], b.shape[1], b.shape[2], b.shape[3], int(bool(forward))) == 0
    return a


def fft3d(a, sign=-1):
    a = np.ascontiguousarray(a, dtype=np.complex128).copy()
    nx, ny, nz = a.shape
    lib().orc_fft3d(a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), nx, ny, nz, sign)
    return a
