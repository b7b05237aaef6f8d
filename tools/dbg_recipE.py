"""Reciprocal-only (or direct-only) raw slice energies of a bench config against the oracle; the oracle result is cached in /tmp so that
several engine variants (environment switches, precisions) can be compared in one gpurun call."""
import sys, os, ctypes, importlib, json, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'oracle'); sys.path.insert(0, 'tests')
import bench, torch, oracle, parity_tools as pt
pkg = importlib.import_module("openmm-nonbonded-slicing_amd")
name, prec, d, r = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
tag = sys.argv[5] if len(sys.argv) > 5 else ""
n_target, Lbox, nsub, method, grid, dgrid, _ = bench.CONFIGS[name]
w = pt.float_positions(bench.build_workload(n_target, Lbox, nsub, np.random.default_rng(bench.SEED)))
N = len(w['q']); S = nsub * (nsub + 1) // 2
cache = "/tmp/orc_%s_%d%d.npz" % (name, d, r)
if os.path.exists(cache):
    z = np.load(cache); fo, so = z['f'], z['s']
else:
    L = oracle.lib(); cfg = pt.oracle_config(w, method, grid, dgrid); cfg.include_direct = d; cfg.include_reciprocal = r
    fo = np.zeros((N, 3)); so = np.zeros((S, 2)); box = bench.workload_box(w)
    dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)); ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    rc = L.orc_evaluate(ctypes.byref(cfg), dp(w['pos']), dp(box), dp(w['q']), dp(w['sigma']), dp(w['epsilon']), ip(w['subset']), len(w['exc_qq']), ip(w['exc_pairs']),
                        dp(w['exc_qq']), dp(w['exc_sigma']), dp(w['exc_eps']), dp(np.ascontiguousarray(w['lam'])), None, dp(fo), dp(so))
    assert rc == 0
    np.savez(cache, f=fo, s=so)
isd = prec == "double"
dt = torch.float64 if isd else torch.float32
eng = bench.Engine(pkg, w, method, grid, dgrid, prec, 0, 0, 1, 0.1, 1 << 30)
pos = torch.tensor(w['pos'], dtype=dt, device='cuda'); forces = torch.zeros((N, 3), dtype=dt, device='cuda')
eng.set_positions_device(pos.data_ptr(), isd)
e = ctypes.c_double(); eng.ok(eng.L.snb_execute(eng.h, 1, 1, d, r, ctypes.byref(e)))
eng.forces_to(forces.data_ptr(), isd); eng.sync()
f = forces.double().cpu().numpy(); se = eng.slice_energies(S)
err = np.linalg.norm(f - fo, axis=1) / np.maximum(np.linalg.norm(fo, axis=1), 1)
rel = np.abs(se - so) / np.maximum(np.abs(so), 1)
print("== %s %s D%d R%d %s: max force err %.3e, max sliceE rel %.3e" % (name, prec, d, r, tag, err.max(), rel.max()))
for s in np.argsort(-rel.max(axis=1))[:8]:
    print("   slice %2d  C %14.4f vs %14.4f (abs %+.5f)   LJ %14.4f vs %14.4f (abs %+.5f)" % (s, se[s, 0], so[s, 0], se[s, 0] - so[s, 0], se[s, 1], so[s, 1], se[s, 1] - so[s, 1]))
