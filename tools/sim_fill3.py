"""Sub-tile occupancy study for round 3: i-block of 32 (engine order) cut into compact groups, j gathered individually within R of some
i-atom (exact filter), j kept in memory (column) order or sorted; fraction of (i-group x j-group) sub-tiles with no pair inside the
CUTOFF rc (what a run-time skip test would see) and inside the list radius R (what a builder-time mask sees)."""
import numpy as np, sys
sys.path.insert(0, '/root/repo')
import bench
from scipy.spatial import cKDTree
rng = np.random.default_rng(1)
L = 6.2145
w = bench.build_workload(24000, L, 1, np.random.default_rng(bench.SEED))
pos = w["pos"] % L; N = len(pos)
R = 1.1; rc = 1.0
a = (32 * L**3 / N) ** (1 / 3); nc = int(round(L / a)); cw = L / nc
cx = np.minimum((pos[:, 0] / cw).astype(int), nc - 1); cy = np.minimum((pos[:, 1] / cw).astype(int), nc - 1)
serp = cx * nc + np.where(cx % 2 == 1, nc - 1 - cy, cy)
zf = pos[:, 2] / L; zf = np.where(serp % 2 == 1, 1 - zf, zf)
order = np.lexsort((zf, serp)); P = pos[order]; nb = N // 32
tree = cKDTree(P, boxsize=L)
def split(p, ngroups):
    idx = [np.arange(len(p))]
    while len(idx) < ngroups:
        nxt = []
        for g in idx:
            ax = np.argmax(np.ptp(p[g], axis=0)); o = g[np.argsort(p[g, ax], kind='stable')]
            nxt += [o[:len(o) // 2], o[len(o) // 2:]]
        idx = nxt
    return np.concatenate(idx)
blocks = rng.choice(nb, 100, replace=False)
for gi, gj, mode in ((8, 16, 'mem'), (8, 16, 'sig'), (8, 8, 'mem'), (8, 8, 'sig'), (16, 8, 'mem'), (4, 16, 'mem'), (4, 32, 'mem'), (8, 32, 'mem'), (16, 16, 'mem'), (4,8,'mem'), (8,16,'z'), (8,8,'z'), (8,8,'morton')):
    ni = 32 // gi
    tot = 0; emptyR = 0; emptyC = 0; slots = 0; inrc = 0
    for I in blocks:
        p = P[I * 32:(I + 1) * 32].copy(); p -= L * np.round((p - p[0]) / L)
        p = p[split(p, ni)] if ni > 1 else p
        cand = np.unique(np.concatenate(tree.query_ball_point(p % L, R)))
        cand = cand[(cand // 32) != I]
        # ownership by block parity rule
        J = cand // 32
        own = np.where(((I + J) & 1) == 1, I > J, I < J)
        cand = cand[own]
        q = P[cand]; q = q - L * np.round((q - p.mean(0)) / L)
        d = np.linalg.norm(p[:, None, :] - q[None, :, :], axis=2)
        reach = np.stack([(d[g * gi:(g + 1) * gi] < R).any(0) for g in range(ni)], 0)
        sig = (reach * (1 << np.arange(ni))[:, None]).sum(0)
        if mode == 'sig': oj = np.lexsort((q[:, 2], sig))
        elif mode == 'z': oj = np.argsort(q[:, 2], kind='stable')
        elif mode == 'morton':
            c = ((q - q.min(0)) / 0.35).astype(int)
            key = np.zeros(len(q), dtype=np.int64)
            for b in range(4):
                for ax in range(3): key |= ((c[:, ax] >> b) & 1) << (3 * b + ax)
            oj = np.argsort(key, kind='stable')
        else: oj = np.arange(len(cand))
        d = d[:, oj]
        nj = len(oj); npad = (nj + 31) // 32 * 32
        dp = np.full((32, npad), 1e9); dp[:, :nj] = d
        sub = dp.reshape(ni, gi, npad // gj, gj).min(axis=(1, 3))
        tot += sub.size; emptyR += (sub >= R).sum(); emptyC += (sub >= rc).sum()
        slots += 32 * npad; inrc += (d < rc).sum()
    print("i-group %2d x j-group %2d %-6s: empty at R %.3f, empty at rc %.3f   (fill at rc %.3f)" % (gi, gj, mode, emptyR / tot, emptyC / tot, inrc / slots))
# per-group j-lists (each i-group keeps only the j-atoms within R of one of ITS atoms): slots relative to the 32-atom block list
for gi in (16, 8, 4, 2, 1):
    ni = 32 // gi; tot = 0; kept = 0; inrc = 0
    for I in blocks:
        p = P[I * 32:(I + 1) * 32].copy(); p -= L * np.round((p - p[0]) / L)
        p = p[split(p, ni)] if ni > 1 else p
        cand = np.unique(np.concatenate(tree.query_ball_point(p % L, R)))
        cand = cand[(cand // 32) != I]
        J = cand // 32; own = np.where(((I + J) & 1) == 1, I > J, I < J); cand = cand[own]
        q = P[cand]; q = q - L * np.round((q - p.mean(0)) / L)
        d = np.linalg.norm(p[:, None, :] - q[None, :, :], axis=2)
        reach = np.stack([(d[g * gi:(g + 1) * gi] < R).any(0) for g in range(ni)], 0)
        tot += reach.size; kept += reach.sum(); inrc += (d < rc).sum() / gi
    print("per-group lists, i-group %2d: slots kept %.3f of the block list  -> fill at rc %.3f" % (gi, kept / tot, inrc / kept))
