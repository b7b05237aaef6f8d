import sys, os, ctypes, importlib, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'oracle'); sys.path.insert(0,'tests')
import bench, torch
pkg = importlib.import_module("openmm-nonbonded-slicing_amd")
NT=int(sys.argv[1]); LB=float(sys.argv[2]); PAD=float(sys.argv[3]); w = bench.build_workload(NT, LB, 4, np.random.default_rng(bench.SEED))
N = len(w['q']); print("N", N, np.bincount(w['subset']))
grid = int(sys.argv[4])
import oracle
def orc(direct, recip):
    L = oracle.lib(); cfg = oracle.OrcConfig()
    cfg.n_atoms=N; cfg.n_subsets=4; cfg.method=4; cfg.cutoff=1.0; cfg.rf_dielectric=78.3; cfg.alpha=bench.ALPHA
    cfg.grid[0]=cfg.grid[1]=cfg.grid[2]=grid; cfg.include_direct=direct; cfg.include_reciprocal=recip; cfg.background_term=1; cfg.correct_q1=1
    f=np.zeros((N,3)); se=np.zeros((10,2)); box=np.diag([w['L']]*3).astype(float).reshape(9)
    dp=lambda a:a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)); ip=lambda a:a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    rc=L.orc_evaluate(ctypes.byref(cfg),dp(w['pos']),dp(box),dp(w['q']),dp(w['sigma']),dp(w['epsilon']),ip(w['subset']),len(w['exc_qq']),ip(w['exc_pairs']),dp(w['exc_qq']),dp(w['exc_sigma']),dp(w['exc_eps']),dp(np.ascontiguousarray(w['lam'])),None,dp(f),dp(se))
    assert rc==0; return f,se
for prec in ("double",):
    eng = bench.Engine(pkg, w, 4, grid, 0, prec, 0, 0, 1, PAD, 1<<30)
    dt = torch.float64 if prec=="double" else torch.float32
    pos = torch.tensor(w['pos'], dtype=dt, device='cuda'); forces = torch.zeros((N,3), dtype=dt, device='cuda')
    for (d,r) in ((1,0),(0,1),(1,1)):
        fo, so = orc(d,r)
        eng.set_positions_device(pos.data_ptr(), prec=="double")
        e = ctypes.c_double(); eng.ok(eng.L.snb_execute(eng.h,1,1,d,r,ctypes.byref(e)))
        eng.forces_to(forces.data_ptr(), prec=="double"); eng.sync()
        f = forces.double().cpu().numpy(); se = eng.slice_energies(10)
        err = np.linalg.norm(f-fo,axis=1)/np.maximum(np.linalg.norm(fo,axis=1),1)
        i = int(err.argmax())
        print(prec, "direct" if d else "", "recip" if r else "", "max ferr %.3e at %d (subset %d) |F|=%.3e" % (err.max(), i, w['subset'][i], np.linalg.norm(fo[i])), "n(err>1e-3)=", int((err>1e-3).sum()))
        print("   slice E err:", np.abs(se-so).max()); bad=np.where(err>1e-3)[0][:10]; print("   bad atoms", bad.tolist(), [int(w['subset'][b]) for b in bad], [w['pos'][b].round(3).tolist() for b in bad[:4]])
