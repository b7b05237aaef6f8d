import importlib, sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
snb = importlib.import_module("openmm-nonbonded-slicing_amd")
w = bench.build_workload(12000, 4.932, 8, np.random.default_rng(bench.SEED))
n = len(w["q"])
dt = torch.float64
pos = torch.tensor(w["pos"], dtype=dt, device="cuda")
res = {}
for world in (1, 2):
    for (d, r) in ((1, 0), (0, 1)):
        ftot = np.zeros((n, 3))
        for rank in range(world):
            eng = bench.Engine(snb, w, 4, 42, 0, "double", 0, rank, world, 0.05, 1 << 30)
            forces = torch.zeros((n, 3), dtype=dt, device="cuda")
            eng.set_positions_device(pos.data_ptr(), True)
            e = ctypes.c_double(0.0)
            eng.ok(eng.L.snb_execute(eng.h, 1, 1, d, r, ctypes.byref(e)))
            eng.forces_to(forces.data_ptr(), True); eng.sync()
            ftot += forces.double().cpu().numpy()
            eng.close()
        res[(world, d, r)] = ftot
for (d, r) in ((1, 0), (0, 1)):
    a, b = res[(1, d, r)], res[(2, d, r)]
    err = np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(a, axis=1), 1.0)
    print("direct" if d else "recip", "max err", err.max(), "n bad", (err > 1e-6).sum(), "ratio of norms", np.linalg.norm(b) / np.linalg.norm(a))
