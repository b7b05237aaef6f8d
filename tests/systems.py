"""Seeded synthetic systems shared by the GPU parity tests, smoke() and bench.py."""
import math

import numpy as np

SEED = 20251212


def jittered_lattice(n, L, rng, jitter=0.05):
    m = int(math.ceil(n ** (1.0 / 3.0)))
    g = np.stack(np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij"), -1).reshape(-1, 3)[:n]
    return (g + 0.5) * (L / m) + rng.uniform(-jitter, jitter, (n, 3))


def random_box(F, n, nsub, method, L, cutoff, seed=SEED, pme=None, ljpme=None, exclusions=True, lambdas=True, switch=False, density_jitter=0.05, derivatives=True):
    """Charged LJ particles on a jittered lattice, bonded in triplets (2 exclusions + one scaled 1-4 per triplet chain),
    subsets assigned by slabs along x so that blocks are compact."""
    rng = np.random.default_rng(seed)
    pos = jittered_lattice(n, L, rng, density_jitter)
    f = F(nsub)
    f.setNonbondedMethod(method)
    f.setCutoffDistance(cutoff)
    q = rng.uniform(0.2, 0.8, n) * rng.choice([-1.0, 1.0], n)
    q -= q.mean()
    sig = rng.uniform(0.25, 0.35, n)
    eps = rng.uniform(0.2, 1.0, n)
    sub = np.minimum((pos[:, 0] / L * nsub).astype(int), nsub - 1) if nsub > 1 else np.zeros(n, dtype=int)
    for i in range(n):
        f.addParticle(q[i], sig[i], eps[i])
        f.setParticleSubset(i, int(sub[i]))
    if exclusions:
        # lattice neighbours along z are adjacent indices: chains of 4 -> 1-2, 1-3 excluded, 1-4 scaled
        for a in range(0, n - 3, 4):
            f.addException(a, a + 1, 0.0, 1.0, 0.0)
            f.addException(a + 1, a + 2, 0.0, 1.0, 0.0)
            f.addException(a + 2, a + 3, 0.0, 1.0, 0.0)
            f.addException(a, a + 2, 0.0, 1.0, 0.0)
            f.addException(a + 1, a + 3, 0.0, 1.0, 0.0)
            f.addException(a, a + 3, 0.8333 * q[a] * q[a + 3], 0.5 * (sig[a] + sig[a + 3]), 0.5 * math.sqrt(eps[a] * eps[a + 3]))
    if pme is not None:
        f.setPMEParameters(*pme)
    if ljpme is not None:
        f.setLJPMEParameters(*ljpme)
    if switch:
        f.setUseSwitchingFunction(True)
        f.setSwitchingDistance(0.8 * cutoff)
    params = {}
    if lambdas and nsub > 1:
        vals = [0.7, 0.9, 0.5, 0.3, 0.6, 0.8, 0.4, 0.95]
        k = 0
        for s in range(1, nsub):
            ne, nv = "lam_elec_0%d" % s, "lam_vdw_0%d" % s
            f.addGlobalParameter(ne, vals[k % 8]); f.addGlobalParameter(nv, vals[(k + 1) % 8]); k += 2
            f.addScalingParameter(ne, 0, s, True, False); f.addScalingParameter(nv, 0, s, False, True)
            if derivatives:
                f.addEnergyParameterDerivative(ne); f.addEnergyParameterDerivative(nv)
        if nsub > 2:
            f.addGlobalParameter("lam_12", 0.45)
            f.addScalingParameter("lam_12", 1, 2, True, True)
            if derivatives:
                f.addEnergyParameterDerivative("lam_12")
    return f, pos, np.diag([L, L, L]).astype(float)
