"""Split-tile costing (round 3): i-blocks cut into their lower / upper 16 atoms; which share of a block's gathered j-atoms
reaches only one half, and how many tile steps a scheme that pairs the two one-half lists saves (exact distances, engine's block order)."""
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
blocks = rng.choice(nb, 150, replace=False)
tot_now=0; tot_split=0; both=0; o0=0; o1=0; nj=0
for mode in ("zhalf","median"):
  tot_now=0; tot_split=0; both=0; o0=0; o1=0; nj=0
  for I in blocks:
    p = P[I*32:(I+1)*32].copy(); p -= L*np.round((p-p[0])/L)
    if mode=="median":
        ax=np.argmax(np.ptp(p,axis=0)); p=p[np.argsort(p[:,ax],kind='stable')]
    cand = np.unique(np.concatenate(tree.query_ball_point(p % L, R)))
    cand = cand[(cand//32)!=I]
    J = cand//32; own = np.where(((I+J)&1)==1, I>J, I<J); cand=cand[own]
    q = P[cand]; q = q - L*np.round((q-p.mean(0))/L)
    d = np.linalg.norm(p[:,None,:]-q[None,:,:],axis=2)
    r0 = (d[:16]<R).any(0); r1=(d[16:]<R).any(0)
    b = (r0&r1).sum(); a0=(r0&~r1).sum(); a1=(~r0&r1).sum()
    both+=b; o0+=a0; o1+=a1; nj+=len(cand)
    tot_now += -(-len(cand)//32)
    tot_split += -(-b//32) + -(-max(a0,a1)//32)
  print(mode, "both %.3f only0 %.3f only1 %.3f | tiles now %.2f/block, split scheme %.2f/block  ratio %.3f" % (both/nj, o0/nj, o1/nj, tot_now/len(blocks), tot_split/len(blocks), tot_split/tot_now))
