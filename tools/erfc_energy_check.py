"""Systematic error of single-precision pair-energy formulas for qq*erfc(alpha r)/r summed over the pairs of a water box
(numpy float32 emulation, separate mul/add roundings): A&S 7.1.26 as in the kernel vs polynomial fits of Et(r^2) = erf(alpha r)/r."""
import numpy as np, sys, math
sys.path.insert(0, '/root/repo')
import bench
from scipy.spatial import cKDTree
from scipy.special import erfc, erf
from numpy.polynomial import chebyshev as C
L = 6.2145; alpha = bench.ALPHA; rc = 1.0
w = bench.build_workload(24000, L, 2, np.random.default_rng(bench.SEED))
pos = (w["pos"] % L).astype(np.float32).astype(np.float64)
t = cKDTree(pos, boxsize=L)
pairs = t.query_pairs(rc, output_type='ndarray')
ex = set(map(tuple, np.sort(w['exc_pairs'], axis=1)))
keep = np.array([(a, b) not in ex for a, b in pairs]); pairs = pairs[keep]
d = pos[pairs[:, 0]] - pos[pairs[:, 1]]; d -= L * np.round(d / L)
r2 = (d * d).sum(1); r = np.sqrt(r2)
q = w['q']; qq = 138.93545764438198 * q[pairs[:, 0]] * q[pairs[:, 1]]
si, sj = w['subset'][pairs[:, 0]], w['subset'][pairs[:, 1]]
ww = (si == 0) & (sj == 0)
exact = qq * erfc(alpha * r) / r
print("pairs", len(r), "water-water sum", exact[ww].sum(), "sum |E|", np.abs(exact[ww]).sum())
f32 = np.float32
def as_formula():
    r2f = r2.astype(f32); invR = (f32(1) / np.sqrt(r2f)).astype(f32); rr = (r2f * invR).astype(f32)
    ar = (rr * f32(alpha)).astype(f32)
    ex_ = np.exp2((r2f * f32(-alpha * alpha * 1.4426950408889634)).astype(f32)).astype(f32)
    den = (ar * f32(0.3275911) + f32(1)).astype(f32); tt = (f32(1) / den).astype(f32)
    p = (tt * f32(1.061405429) + f32(-1.453152027)).astype(f32)
    for c in (1.421413741, -0.284496736, 0.254829592):
        p = (p * tt + f32(c)).astype(f32)
    qqf = qq.astype(f32)
    return ((qqf * invR).astype(f32) * ((p * tt).astype(f32) * ex_).astype(f32)).astype(np.float64)
def poly(deg, comp=0, r2max=(1.0 + 0.1 + 0.02) ** 2):
    Et = lambda x: np.where(x < 1e-8, 2 * alpha / math.sqrt(math.pi), erf(alpha * np.sqrt(np.maximum(x, 1e-30))) / np.sqrt(np.maximum(x, 1e-30)))
    nodes = np.cos(np.pi * (np.arange(96) + 0.5) / 96)
    cheb = C.chebfit(nodes, Et(0.5 * (nodes + 1) * r2max), deg)
    mono = C.cheb2poly(cheb)
    r2f = r2.astype(f32); invR = (f32(1) / np.sqrt(r2f)).astype(f32)
    tt = (r2f * f32(2.0 / r2max) - f32(1)).astype(f32)
    hi = mono.astype(f32); lo = (mono - hi.astype(np.float64)).astype(f32)
    acc = np.full_like(tt, hi[deg])
    for k in range(deg - 1, -1, -1):
        acc = (acc * tt + hi[k]).astype(f32)
    if comp:      # residual polynomial of the low-order coefficients' rounding
        cacc = np.full_like(tt, lo[comp - 1])
        for k in range(comp - 2, -1, -1):
            cacc = (cacc * tt + lo[k]).astype(f32)
        acc = (acc + cacc).astype(f32)
    qqf = qq.astype(f32)
    return (qqf * (invR - acc).astype(f32)).astype(np.float64), np.abs(mono).max()
e = as_formula(); print("A&S            : water-water error %+.4f   all %+.4f" % ((e - exact)[ww].sum(), (e - exact).sum()))
for deg in (11, 13, 15, 17):
    for comp in (0, 2, 4, deg + 1):
        e, cm = poly(deg, comp)
        print("poly deg %2d comp %2d: water-water error %+.4f   all %+.4f   (max |coef| %.1f)" % (deg, comp, (e - exact)[ww].sum(), (e - exact).sum(), cm))
