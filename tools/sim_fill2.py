"""Exact sub-tile occupancy of the 32x32 tiles (round 2): i-blocks as the engine cuts them (column/z sort), the block's atoms
re-ordered into spatially compact groups, j-atoms gathered individually (within R of some i-atom) and ordered by which i-groups
they reach.  Reports the fraction of (i-group x j-group) sub-tiles that hold no pair within the list radius R."""
import numpy as np, sys
sys.path.insert(0, '/root/repo')
import bench
from scipy.spatial import cKDTree
rng = np.random.default_rng(1)
L = 6.2145
w = bench.build_workload(24000, L, 1, np.random.default_rng(bench.SEED))
pos = w["pos"] % L; N = len(pos)
R = float(sys.argv[1]) if len(sys.argv) > 1 else 1.1; rc = 1.0
a = (32 * L**3 / N) ** (1 / 3); nc = int(round(L / a)); cw = L / nc
cx = np.minimum((pos[:, 0] / cw).astype(int), nc - 1); cy = np.minimum((pos[:, 1] / cw).astype(int), nc - 1)
serp = cx * nc + np.where(cx % 2 == 1, nc - 1 - cy, cy)
zf = pos[:, 2] / L; zf = np.where(serp % 2 == 1, 1 - zf, zf)
order = np.lexsort((zf, serp)); P = pos[order]; nb = N // 32
tree = cKDTree(P, boxsize=L)
def split(p, ngroups):
    """recursive median bisection along the widest axis -> ngroups compact groups of equal size; returns permutation"""
    idx = [np.arange(len(p))]
    while len(idx) < ngroups:
        nxt = []
        for g in idx:
            ax = np.argmax(np.ptp(p[g], axis=0)); o = g[np.argsort(p[g, ax], kind='stable')]
            nxt += [o[:len(o) // 2], o[len(o) // 2:]]
        idx = nxt
    return np.concatenate(idx)
for gi, gj in ((8, 16),):
    ni = 32 // gi
    tot = 0; empty = 0; slots = 0; inrc = 0; inR = 0; ntiles = 0
    for I in rng.choice(nb, 120, replace=False):
        p = P[I * 32:(I + 1) * 32].copy(); p -= L * np.round((p - p[0]) / L)
        p = p[split(p, ni)] if ni > 1 else p
        cand = np.unique(np.concatenate(tree.query_ball_point(p % L, R)))
        cand = cand[(cand // 32) != I]
        cand = cand[::2]                         # ownership rule: half of the neighbours (statistically)
        q = P[cand]; q = q - L * np.round((q - p.mean(0)) / L)
        d = np.linalg.norm(p[:, None, :] - q[None, :, :], axis=2)      # [32][nj]
        # signature of a j-atom: which i-groups it reaches within R
        reach = np.stack([(d[g * gi:(g + 1) * gi] < R).any(0) for g in range(ni)], 0)      # [ni][nj]
        sig = (reach * (1 << np.arange(ni))[:, None]).sum(0)
        # order j by signature, then spatially (z) inside a signature
        mode = sys.argv[2] if len(sys.argv) > 2 else 'sig'
        colj = (cand // 1)  # sorted index: column-major already
        if mode == 'sig': oj = np.lexsort((q[:, 2], sig))
        elif mode.startswith('win'):
            W = int(mode[3:]); win = np.arange(len(cand)) // W      # windows of W consecutive candidates in memory order
            oj = np.lexsort((np.arange(len(cand)), sig, win))
        else: oj = np.arange(len(cand))
        d = d[:, oj]; reach = reach[:, oj]
        nj = len(oj); npad = (nj + 31) // 32 * 32
        ntiles += npad // 32
        for k in range(0, nj, gj):
            sub = d[:, k:k + gj]
            for g in range(ni):
                tot += 1
                if not (sub[g * gi:(g + 1) * gi] < R).any(): empty += 1
        tot += (npad - nj) // gj * ni; empty += (npad - nj) // gj * ni      # padding slots are empty sub-tiles too
        slots += 32 * npad; inrc += (d < rc).sum(); inR += (d < R).sum()
    print("R=%.2f  i-group %2d x j-group %2d: empty sub-tiles %.3f   (fill at rc %.3f, at R %.3f, tiles/block %.1f)" % (R, gi, gj, empty / tot, inrc / slots, inR / slots, ntiles / 120))
