"""Helpers of the full-size parity tests (tests/test_gpu_fullsize.py) and of ``bench.py --check``: the bench workloads against the
CPU oracle on IDENTICAL inputs, with the one effect single precision cannot avoid named and bounded pair by pair.

Identical inputs.  A single-precision engine takes float coordinates (the reference's GPU platforms store ``posq`` as float4 in
single/mixed precision too), so the oracle is given the same float-representable coordinates, widened to double -- not the double
array they were rounded from.  Measured on c3 (300k atoms): against the unrounded doubles the worst atom is off by 2.2e-3, against
the coordinates the engine actually received by 3.7e-4 (tools/dbg_tail.py).

Truncation band.  The pair potential is cut at r < cutoff at full strength (ReferenceSlicedLJCoulombIxn.cpp:367: every pair of the
neighbour list, which holds exactly the pairs inside the cutoff), i.e. it is discontinuous there: with alpha = 2.6283/nm the real-space
Ewald force of a water O-H pair at 1.0 nm is 0.155 kJ/mol/nm, of an O-O pair 0.31.  A pair whose r^2 lies within rounding of
cutoff^2 falls on either side depending on how r^2 is rounded; with float coordinates of magnitude L the stored positions carry
L * 2^-24 of rounding each, so  |r^2/cutoff^2 - 1| < BAND_REL  (a few dozen to a thousand of the 6e7 pairs of c3) cannot be decided
in single precision.  ``orc_cutoff_band_pairs`` (oracle, diagnostic) lists exactly those pairs and what each contributes; the
comparison allows an atom the force of ITS band pairs on top of the relative tolerance (one pair, rarely two).  A slice energy sums
over all band pairs of the slice, each of which may or may not have flipped, with either sign: it is allowed three standard
deviations of that sum, 3 sqrt(sum E_k^2) (about 1 kJ/mol on the water-water slice of c3, |E| = 6.7e3).  Atoms without a band pair
(99 %) are held to the plain tolerance.  Double precision uses no band."""
import ctypes

import numpy as np

import bench

# relative half-width of the undecidable band in r^2.  Single precision, coordinates below 16 nm (ulp 9.5e-7 nm): each stored coordinate
# is off by <= 4.8e-7 nm (user + image offset, rounded once), a lattice-shifted j by as much again, so r^2 = 1 nm^2 is off by at most
# 2 * sqrt(3) * 1.4e-6 = 5e-6 (all roundings extreme and aligned), 1e-6 rms.
BAND_REL = {"single": 6e-6, "mixed": 6e-6, "double": 0.0}


def band_rel(w, precision):
    """BAND_REL for the workload's own coordinate range: the figure above is for coordinates below 16 nm; a float's spacing doubles at
    16 nm (the 1M-atom box of c5 is 21.5 nm wide), and the band with it."""
    if BAND_REL[precision] == 0.0:
        return 0.0
    ulp = float(np.spacing(np.float32(np.abs(w["pos"]).max())))
    return BAND_REL[precision] * max(1.0, ulp / 9.5367431640625e-07)


def float_positions(w):
    """The workload with its coordinates rounded to float32 and widened back: what a single-precision engine is given."""
    wf = dict(w)
    wf["pos"] = np.ascontiguousarray(w["pos"].astype(np.float32).astype(np.float64))
    return wf


def oracle_config(w, method, grid, dgrid):
    import oracle
    cfg = oracle.OrcConfig()
    cfg.n_atoms = len(w["q"]); cfg.n_subsets = w["nsub"]; cfg.method = method; cfg.cutoff = bench.CUTOFF; cfg.rf_dielectric = 78.3
    cfg.alpha = bench.ALPHA; cfg.grid[0] = cfg.grid[1] = cfg.grid[2] = grid
    cfg.alpha_d = bench.ALPHA; cfg.dgrid[0] = cfg.dgrid[1] = cfg.dgrid[2] = max(dgrid, 1)
    cfg.include_direct = 1; cfg.include_reciprocal = 1; cfg.background_term = 1; cfg.correct_q1 = 1
    return cfg


def band_allowance(w, method, grid, dgrid, rel_eps):
    """(force allowance per atom [N], energy allowance per slice [S][2], number of band pairs) for the workload's coordinates."""
    import oracle
    n = len(w["q"]); S = w["nsub"] * (w["nsub"] + 1) // 2
    fa = np.zeros(n); ea = np.zeros((S, 2))
    if rel_eps <= 0:
        return fa, ea, 0
    L = oracle.lib(); cfg = oracle_config(w, method, grid, dgrid)
    dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)); ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    cap = 1 << 16
    while True:
        ij = np.zeros((cap, 2), dtype=np.int32); vals = np.zeros((cap, 4))
        lam = np.ascontiguousarray(w["lam"]); box = bench.workload_box(w)
        cnt = L.orc_cutoff_band_pairs(ctypes.byref(cfg), dp(w["pos"]), dp(box), dp(w["q"]), dp(w["sigma"]), dp(w["epsilon"]), ip(w["subset"]),
                                      len(w["exc_qq"]), ip(w["exc_pairs"]), dp(lam), float(rel_eps), cap, ip(ij), dp(vals))
        assert cnt >= 0, cnt
        if cnt <= cap:
            break
        cap = int(cnt) + 16
    ij = ij[:cnt]; vals = vals[:cnt]
    np.add.at(fa, ij[:, 0], vals[:, 1]); np.add.at(fa, ij[:, 1], vals[:, 1])
    si, sj = w["subset"][ij[:, 0]], w["subset"][ij[:, 1]]
    hi, lo = np.maximum(si, sj), np.minimum(si, sj)
    sl = hi * (hi + 1) // 2 + lo
    np.add.at(ea[:, 0], sl, vals[:, 2] ** 2); np.add.at(ea[:, 1], sl, vals[:, 3] ** 2)
    return fa, 3.0 * np.sqrt(ea), int(cnt)


def compare(f, se, fo, so, tol, force_allow, energy_allow):
    """Relative errors with the reference's max(|x|, 1) scaling (AssertionUtilities.h:7-26) after the band allowance.  Returns a
    record; ``ok`` is the verdict."""
    fn = np.linalg.norm(fo, axis=1)
    err = np.linalg.norm(f - fo, axis=1)
    den = np.maximum(fn, 1.0)
    excess = np.maximum(err - force_allow, 0.0) / den            # what the band does not explain
    plain = err / den
    noband = force_allow == 0
    rec = {"max_force_rel_err_outside_band": float(plain[noband].max()) if noband.any() else 0.0,
           "max_force_rel_err_after_allowance": float(excess.max()), "max_force_rel_err_raw": float(plain.max()),
           "median_force_rel_err": float(np.median(plain)), "p999_force_rel_err": float(np.quantile(plain, 0.999)),
           "atoms_with_band_pair": int((~noband).sum()), "worst_atom": int(excess.argmax())}
    ok = excess.max() <= tol
    if se is not None:
        eerr = np.abs(se - so)
        eden = np.maximum(np.abs(so), 1.0)
        eex = np.maximum(eerr - energy_allow, 0.0) / eden
        rec["max_slice_energy_rel_err_raw"] = float((eerr / eden).max())
        rec["max_slice_energy_rel_err_after_allowance"] = float(eex.max())
        rec["worst_slice"] = [int(x) for x in np.unravel_index(eex.argmax(), eex.shape)]
        ok = ok and eex.max() <= tol
    rec["ok"] = bool(ok)
    return rec


# Session cache of the full-size cases (tests/test_gpu_fullsize.py): workload, oracle forces / slice energies and band allowance per
# (config, coordinates rounded to float or not).  c3 single / mixed and c4 single / c4 as eight ranks share their inputs, so the oracle
# evaluates every distinct input once per session (the GPU suite must stay inside the driver's limit).
_FULLSIZE = {}
_BANDS = {}


def fullsize_case(name, precision):
    """(workload, oracle forces, oracle slice energies, oracle seconds, pairs within the cutoff, force allowance, energy allowance, band pairs)
    of a bench config at full size, for the coordinates an engine of this precision receives."""
    n_target, Lbox, nsub, method, grid, dgrid, _ = bench.CONFIGS[name]
    rounded = precision != "double"
    key = (name, rounded)
    if key not in _FULLSIZE:
        w = bench.build_workload(n_target, Lbox, nsub, np.random.default_rng(bench.SEED))
        if rounded:
            w = float_positions(w)
        fo, so, seconds, pairs = bench.oracle_eval(w, method, grid, dgrid)
        _FULLSIZE[key] = (w, fo, so, seconds, pairs)
    w, fo, so, seconds, pairs = _FULLSIZE[key]
    rel = band_rel(w, precision)
    bkey = (name, rounded, rel)
    if bkey not in _BANDS:
        _BANDS[bkey] = band_allowance(w, method, grid, dgrid, rel)
    fa, ea, nband = _BANDS[bkey]
    return w, fo, so, seconds, pairs, fa, ea, nband
