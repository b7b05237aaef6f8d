"""(GPU box) the bench's displacement-triggered leg step by step: host time of every execute (diagnosis)."""
import sys, os, time, importlib
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
pkg = importlib.import_module("openmm-nonbonded-slicing_amd")
n_target, Lbox, nsub, method, grid, dgrid, precision = bench.CONFIGS["c3"]
w = bench.build_workload(n_target, Lbox, nsub, np.random.default_rng(bench.SEED))
torch.cuda.set_stream(torch.cuda.Stream())
N = len(w["q"])
walk_rng = np.random.default_rng(bench.SEED + 1)
walk = [torch.tensor(walk_rng.normal(0.0, 0.0015, (N, 3)), dtype=torch.float32, device="cuda") for _ in range(16)]
walk_sign = walk_rng.choice([-1.0, 1.0], size=1 << 16)
if os.environ.get("DBG_FIRST_ENGINE"):      # a fixed-interval engine first, left alive with a side build pending (what bench.py does)
    e1 = bench.Engine(pkg, w, method, grid, dgrid, precision, 0, 0, 1, 0.1, 20, stream=torch.cuda.current_stream().cuda_stream)
    p1 = torch.tensor(w["pos"], dtype=torch.float32, device="cuda"); f1 = torch.zeros((N, 3), dtype=torch.float32, device="cuda")
    e1.set_force_output(f1.data_ptr(), False); e1.set_positions_device(p1.data_ptr(), False); e1.set_timing_interval(0)
    for i in range(int(os.environ["DBG_FIRST_ENGINE"])):
        p1.add_(walk[i % 16], alpha=float(walk_sign[i % len(walk_sign)])); e1.execute(False)
    e1.sync()
e2 = bench.Engine(pkg, w, method, grid, dgrid, precision, 0, 0, 1, 0.1, -100, stream=torch.cuda.current_stream().cuda_stream)
p2 = torch.tensor(w["pos"], dtype=torch.float32, device="cuda"); f2 = torch.zeros((N, 3), dtype=torch.float32, device="cuda")
deriv = (np.abs(w["lam"] - 1.0).max(axis=1) > 0).astype(np.int32)
e2.set_force_output(f2.data_ptr(), False); e2.set_energy_slices(deriv); e2.set_positions_device(p2.data_ptr(), False); e2.set_timing_interval(0)
ts = []
for i in range(310):
    a = time.perf_counter()
    p2.add_(walk[i % 16], alpha=float(walk_sign[i % len(walk_sign)]))
    e2.execute(2, fetch=False)
    ts.append((time.perf_counter() - a) * 1e3)
e2.sync()
ts = np.array(ts)
print("host ms per execute: median %.3f, mean %.3f, max %.1f at step %d; steps over 2 ms: %s" % (np.median(ts), ts.mean(), ts.max(), int(ts.argmax()), [(int(i), round(float(t), 1)) for i, t in enumerate(ts) if t > 2.0][:30]))
print("rebuilds", int(e2.stats().n_rebuilds), "overruns", int(e2.stats().n_list_overruns))
