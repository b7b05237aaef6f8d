"""Where does the single-precision force tail at full size come from?  For a bench config: direct-only, reciprocal-only and full
evaluations (energy step and forces-only step) against the oracle fed (a) the double positions and (b) the float-rounded ones."""
import sys, os, ctypes, importlib, json, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'oracle'); sys.path.insert(0, 'tests')
import bench, torch, oracle
pkg = importlib.import_module("openmm-nonbonded-slicing_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
prec = sys.argv[2] if len(sys.argv) > 2 else "single"
watch = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else []      # atoms to report in every decomposition
n_target, Lbox, nsub, method, grid, dgrid, _ = bench.CONFIGS[name]
method = int(os.environ.get('SNB_DBG_METHOD', method))      # e.g. 4: the config's box with plain PME instead of LJPME
parts = sys.argv[4] if len(sys.argv) > 4 else 'D,R,DR'
w = bench.build_workload(n_target, Lbox, nsub, np.random.default_rng(bench.SEED))
N = len(w['q']); S = nsub * (nsub + 1) // 2
def orc(wv, d, r):
    L = oracle.lib(); cfg = oracle.OrcConfig()
    cfg.n_atoms = N; cfg.n_subsets = nsub; cfg.method = method; cfg.cutoff = 1.0; cfg.rf_dielectric = 78.3; cfg.alpha = bench.ALPHA
    cfg.grid[0] = cfg.grid[1] = cfg.grid[2] = grid; cfg.alpha_d = bench.ALPHA; cfg.dgrid[0] = cfg.dgrid[1] = cfg.dgrid[2] = max(dgrid, 1)
    cfg.include_direct = d; cfg.include_reciprocal = r; cfg.background_term = 1; cfg.correct_q1 = 1
    f = np.zeros((N, 3)); se = np.zeros((S, 2)); box = bench.workload_box(wv)
    dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)); ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    rc = L.orc_evaluate(ctypes.byref(cfg), dp(wv['pos']), dp(box), dp(wv['q']), dp(wv['sigma']), dp(wv['epsilon']), ip(wv['subset']), len(wv['exc_qq']), ip(wv['exc_pairs']),
                        dp(wv['exc_qq']), dp(wv['exc_sigma']), dp(wv['exc_eps']), dp(np.ascontiguousarray(wv['lam'])), None, dp(f), dp(se))
    assert rc == 0
    return f, se
wf = dict(w); wf['pos'] = np.ascontiguousarray(w['pos'].astype(np.float32).astype(np.float64))
isd = prec == "double"
dt = torch.float64 if isd else torch.float32
eng = bench.Engine(pkg, w, method, grid, dgrid, prec, 0, 0, 1, 0.1, 1 << 30)
pos = torch.tensor(w['pos'], dtype=dt, device='cuda'); forces = torch.zeros((N, 3), dtype=dt, device='cuda')
res = {}
for (d, r) in [x for x in ((1, 0), (0, 1), (1, 1)) if ('D' if x[0] else '') + ('R' if x[1] else '') in parts.split(',')]:
    fo, so = orc(w, d, r); fo2, so2 = (fo, so) if isd else orc(wf, d, r)
    for energy in (1, 0):
        eng.set_positions_device(pos.data_ptr(), isd)
        e = ctypes.c_double(); eng.ok(eng.L.snb_execute(eng.h, 1, energy, d, r, ctypes.byref(e)))
        eng.forces_to(forces.data_ptr(), isd); eng.sync()
        f = forces.double().cpu().numpy()
        for tag, ff, ss in (("dblpos", fo, so), ("fltpos", fo2, so2)):
            den = np.maximum(np.linalg.norm(ff, axis=1), 1)
            err = np.linalg.norm(f - ff, axis=1) / den
            top = np.argsort(-err)[:6]
            key = "%s%s %s %s" % ("D" if d else "", "R" if r else "", "energy" if energy else "forces", tag)
            rec = {"max": float(err.max()), "p999": float(np.quantile(err, 0.999)), "median": float(np.median(err)), "n>1e-3": int((err > 1e-3).sum()), "n>5e-4": int((err > 5e-4).sum()),
                   "top": [(int(i), float(err[i]), float(np.linalg.norm(ff[i])), float(np.linalg.norm(f[i] - ff[i])), int(w['subset'][i])) for i in top]}
            if energy:
                se = eng.slice_energies(S)
                rel = np.abs(se - ss) / np.maximum(np.abs(ss), 1)
                rec["sliceE_max_rel"] = float(rel.max()); rec["sliceE_arg"] = [int(x) for x in np.unravel_index(rel.argmax(), rel.shape)]
                rec["sliceE"] = se.tolist(); rec["sliceE_oracle"] = ss.tolist()
            res[key] = rec
            for i in watch:
                print("   atom %d  %s: |F_oracle| %.6g  |err| %.6g  f %s  oracle %s" % (i, key, np.linalg.norm(ff[i]), np.linalg.norm(f[i] - ff[i]), f[i].tolist(), ff[i].tolist()), flush=True)
            print(key, {k: v for k, v in rec.items() if not k.startswith("sliceE") or k in ("sliceE_max_rel", "sliceE_arg")}, flush=True)
json.dump(res, open("gpurun_out/dbg_tail_%s_%s.json" % (name, prec), "w"), indent=1)
